// Fused Baum-Welch E-step on the matrix cores (float32 data, S <= 16, D <= 48).
//
// One wave processes 16 utterances of ONE model at a time, step by step:
//   forward   emission contraction -> b~_t, then
//             alpha_t = (A^T alpha_{t-1}) * b~_t / c_t with the 16x16 mat-vec
//             as 4 MFMAs; alpha_t is spilled ([frames,16] floats).
//   backward  b~_t / c_t = alpha_t / (A^T alpha_{t-1}) from the spilled lattice (4 MFMAs and
//             four divisions: the factor the forward pass multiplied by, without contracting
//             the frame with the model again), beta_t = A (b~ beta / c)_{t+1}
//             as 4 MFMAs, gamma_t, and two accumulating products whose
//             contraction index is the utterance:
//                 G   += alpha_t   (b~ beta)_{t+1}^T      (16 x 16)     4 MFMAs
//                 obs += gamma_t   [x_t | x_t^2]          (16 x 2D)   4 NTX MFMAs
//             (xi sums are A o G, applied once at the end).
// All products run on exact-f32 MFMAs (v_mfma_f32_16x16x4_f32): re-estimated
// parameters have to hold 1e-5 relative, which a bf16-split emission misses.
// State placement.  MFMA row r of the 16-row tile carries state 4*(r&3)+(r>>2), so
// the accumulator register q of lane group g (rows 4g..4g+3) is state 4q+g --
// exactly the k index that lane group supplies as the B operand of k-step q.
// The lattice therefore feeds the next mat-vec straight from the accumulator
// registers, no lane movement.  Products that contract over utterances need the
// utterance on the k axis instead of the lane: alpha, gamma, b~beta and the
// frame go through small per-wave LDS tiles ([utterance][row]) for that.
// Memory.  The feature axis is cut into RQ blocks of 16 plus, if H, one of 8.
// Lane group g reads features 16q+4g..+3 of a block with one 16-byte raw buffer
// load (8q'+2g..+1 with an 8-byte load in the half block); loads are range-
// checked and rows are only 4-byte aligned.  Frames are
// requested PF steps ahead (backward pass: two, and the spilled alpha three) in a
// branch-free loop so the waits stay counted.
// Spilled lattice.  A batch owns as many lattice rows as its utterances have
// frames, starting at the exclusive prefix sum of the batch totals (a small scan
// kernel).  Inside, rows are time-major and dense: step t occupies the next
// nlive(t) rows, one per utterance still running, in lane order.  A wave's
// store is then one contiguous run of 64-byte rows (1 KiB when all 16 run) and
// every cache line is complete after at most two steps -- rows laid out per
// utterance leave half-written lines behind that L2 evicts and refills many times.
// Accumulation: f32 inside the MFMAs over one batch of 16 utterances, then
// added with f64 atomics to one of kReplicas global records of the batch's
// model; a last kernel sums the replicas in fixed order and applies A to G.
#include "srx_internal.h"

#include <math.h>

#include <algorithm>
#include <type_traits>

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

constexpr int kReplicas = 64;
constexpr int kWavesPerBlock = 4;
constexpr int PF = 3;
constexpr float kNegBig = -1.0e30f;
// The backward pass takes b~_t / c_t from the lattice (alpha_t / (A^T alpha_{t-1}), see the
// kernel) instead of contracting the frame with the model again.  (SRX_ESTEP_REEMIT: the
// kernel that recomputes the emission, for measurements.)
#ifdef SRX_ESTEP_REEMIT
constexpr bool kRatio = false;
#else
constexpr bool kRatio = true;
#endif
constexpr double kLn2 = 0.6931471805599453;

__host__ __device__ inline int sigma(int row) { return 4 * (row & 3) + (row >> 2); }

struct FusedTables {
    const float* W;      // [M][16 rows][KW = 8 RQ + 4 H][4 lane groups]
    const float* cst;    // [M][16 rows]
    const float* A;      // [M][16][16] zero padded, state order
    const float* pi;     // [M][16]
};

// Feature and kind (0: mean/var weight of x, 1: -1/(2 var) weight of x^2) that
// k-step kk of lane group g contracts.
__host__ __device__ inline void kstep_feature(int RQ, int kk, int g, int* d, int* kind) {
    if (kk < 8 * RQ) {
        *d = 16 * (kk >> 3) + 4 * g + (kk & 3);
        *kind = (kk >> 2) & 1;
    } else {
        const int e = kk - 8 * RQ;
        *d = 16 * RQ + 2 * g + (e & 1);
        *kind = e >> 1;
    }
}

__global__ void fused_prep_kernel(srx_models_t md, int RQ, int H, float* W, float* cst, float* A,
                                  float* pi) {
    const int M = md.n_models, S = md.n_states, D = md.n_features;
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    const int nth = gridDim.x * blockDim.x;
    const int KW = 8 * RQ + 4 * H;
    for (int i = tid; i < M * 16 * KW * 4; i += nth) {
        const int g = i & 3, kk = (i >> 2) % KW, row = ((i >> 2) / KW) % 16, m = (i >> 2) / KW / 16;
        const int s = sigma(row);
        int d, kind;
        kstep_feature(RQ, kk, g, &d, &kind);
        float v = 0.f;
        if (s < S && d < D) {
            const size_t k = ((size_t)m * S + s) * D + d;
            const double var = fmax(md.covars[k], 2.2250738585072014e-308);
            v = (float)(kind ? (-0.5 / var) * 1.4426950408889634
                             : (md.means[k] / var) * 1.4426950408889634);
        }
        W[i] = v;
    }
    // constants: one wave per row, lanes over features
    const int wave = tid >> 6, lane = tid & 63, n_wave = nth >> 6;
    for (int i = wave; i < M * 16; i += n_wave) {
        const int row = i % 16, m = i / 16, s = sigma(row);
        double acc = 0.0;
        if (s < S)
            for (int d = lane; d < D; d += 64) {
                const size_t k = ((size_t)m * S + s) * D + d;
                const double var = fmax(md.covars[k], 2.2250738585072014e-308);
                acc += log(var) + md.means[k] * md.means[k] / var;
            }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
        if (lane == 0) {
            cst[i] = s < S ? (float)(-0.5 * (acc + D * 1.8378770664093453) * 1.4426950408889634) : kNegBig;
            pi[i] = (row < S) ? (float)md.startprob[m * S + row] : 0.f;   // state order
        }
    }
    for (int i = tid; i < M * 256; i += nth) {
        const int j = i & 15, ii = (i >> 4) & 15, m = i >> 8;
        A[i] = (ii < S && j < S) ? (float)md.transmat[((size_t)m * S + ii) * S + j] : 0.f;
    }
}

__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Reductions over the 16 states of an utterance: lanes l, l^16, l^32, l^48,
// 4 registers each.  v_permlane16_swap / v_permlane32_swap of two copies of a
// value leave {v[l], v[partner]} in the two results, so commutative reductions
// need no LDS round trip.
__device__ __forceinline__ float sum16(f32x4 v) {
    float s = (v[0] + v[1]) + (v[2] + v[3]);
    unsigned u = __float_as_uint(s);
    auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    s = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    u = __float_as_uint(s);
    r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float max16(f32x4 v) {
    float s = fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3]));
    unsigned u = __float_as_uint(s);
    auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    s = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
    u = __float_as_uint(s);
    r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float sum_u(float v) {   // over the 16 utterance lanes of a group
    v += __shfl_xor(v, 1, 64);
    v += __shfl_xor(v, 2, 64);
    v += __shfl_xor(v, 4, 64);
    v += __shfl_xor(v, 8, 64);
    return v;
}

// Row stride of the frame tile in floats.
__host__ __device__ constexpr int frame_tile_stride(int dp) { return dp <= 16 ? 16 : 48; }

template <int RQ, int H>
struct Frame {
    u32x4 v[RQ ? RQ : 1];
    u32x2 h;
};

// POST = true is the posteriors variant of the same forward-backward pass: gamma of
// every frame goes to post_out[frame][state] and the utterance's log-probability to
// post_lp[utterance]; the accumulating products and the statistics are left out.
// OBS = false leaves out the sums over the features (post, obs, obs2 stay zero): the E-step
// of a caller that re-estimates start and transition probabilities only.  The backward pass
// then reads the lattice alone -- as it does in the posteriors variant, which needs the
// frames for nothing once b~ comes from the lattice.
template <int RQ, int H, int NTX, bool POST = false, bool OBS = true>
__global__ __launch_bounds__(256, 2) void estep_fused_kernel(
    const float* __restrict__ X, const int64_t* __restrict__ off, int64_t U, int64_t n_frames,
    const int32_t* __restrict__ utt_model, const int64_t* __restrict__ utt_order, int M, int S,
    int D, FusedTables tb, const int64_t* __restrict__ batch_base, float* __restrict__ lat,
    float* __restrict__ scale, double* __restrict__ replicas, int rec_size,
    float* __restrict__ post_out, double* __restrict__ post_lp, unsigned long long* __restrict__ queue) {
    constexpr int DP = 16 * RQ + 8 * H;   // padded feature count
    constexpr int KW = 8 * RQ + 4 * H;    // emission k-steps
    constexpr int XS = frame_tile_stride(DP);
    constexpr int TILE = 3 * 256 + 16 * XS;   // floats: gamma, alpha, b~beta tiles and the frame tile
    typedef Frame<RQ, H> frame_t;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63, g = lane >> 4, ul = lane & 15;
    float* tiles = (float*)smem_raw + (size_t)wv * 2 * TILE;   // two buffers per wave

    const srx_stats_layout L(S, D);
    const int64_t n_groups = (U + 15) / 16;
    const int64_t n_waves = (int64_t)gridDim.x * kWavesPerBlock;
    const int64_t wid = (int64_t)blockIdx.x * kWavesPerBlock + wv;
    const int64_t per = (n_groups + n_waves - 1) / n_waves;
    const int64_t k0 = wid * per, k1 = (k0 + per < n_groups) ? k0 + per : n_groups;
    double* my_rep = replicas + (size_t)(wid % kReplicas) * M * rec_size;
    const unsigned row_bytes = (unsigned)D * 4u;
    const unsigned below = (1u << ul) - 1u;       // lanes of lower utterances
    // Tile element (utt, r) sits at utt * stride + (r ^ 4 (utt >> 2)): the 16-byte
    // writes (utt = ul, r = 4g..) and the dword reads (utt = 4q + g, r = ul) are
    // then both free of bank conflicts (frame tile: stride 16 or 48 mod 64).
    const int wsw = 4 * g ^ 4 * (ul >> 2);
    const int hsw = 2 * g ^ 4 * (ul >> 2);

    float w[KW];
    f32x4 c0, pi4;
    float atop[4], aop[4];
#pragma unroll
    for (int q = 0; q < KW; ++q) w[q] = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) { atop[q] = aop[q] = 0.f; c0[q] = 0.f; pi4[q] = 0.f; }

    // With a `queue` (a zeroed counter) the batches are handed out on demand: the first to
    // every wave by its index, the following ones in the order the waves ask for them, so a
    // wave that runs slower, or draws longer utterances, takes fewer, and the grid ends
    // together.  Without one every wave takes a fixed run of batches.  Every wave leaves the
    // loop when the batches run out.
    const bool on_demand = queue != nullptr;
    int64_t k = on_demand ? wid : k0;
    const int64_t k_end = on_demand ? n_groups : k1;
    while (k < k_end) {
        // (the next batch is asked for now; the answer is needed when this one is done)
        unsigned long long ticket = 0;
        if (on_demand && lane == 0) ticket = atomicAdd(queue, 1ull);
        const int64_t e = k * 16 + ul;
        // (U counts the entries of the visiting order here; a negative entry is an empty lane)
        const int64_t u_raw = e < U ? (utt_order ? utt_order[e] : e) : -1;
        const bool valid = u_raw >= 0;
        const int64_t u = valid ? u_raw : 0;
        const int m_l = valid ? (utt_model ? utt_model[u] : 0) : -1;
        const int64_t f_start = valid ? off[u] : 0;
        const int T_l = valid ? (int)(off[u + 1] - f_start) : 0;
        bool pending = valid && T_l > 0;
        const int64_t row_base = batch_base[k];
        // (a batch of several models is walked once per model; the walks take the batch's
        // lattice rows one after the other)
        unsigned pass_base = 0;
        while (true) {
            const unsigned long long mask = __ballot(pending);
            if (mask == 0) break;
            const int first = __ffsll((long long)mask) - 1;
            const int m_cur = __builtin_amdgcn_readfirstlane(__shfl(m_l, first, 64));
            // One descriptor addresses 4 GiB: of this model's utterances take those
            // within reach of the lowest one now, the rest in a later pass.
            const bool cand = pending && m_l == m_cur;
            int64_t fmin = cand ? f_start : (int64_t)0x7FFFFFFFFFFFFFFFll;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const int64_t a = __shfl_xor(fmin, o, 64);
                fmin = a < fmin ? a : fmin;
            }
            const int64_t frame_bytes = D > 16 ? (int64_t)D * 4 : 64;   // widest of X, lattice rows
            const bool fits = cand && (f_start + T_l - fmin) * frame_bytes < 0xFFFFFF00ll &&
                              T_l < (1 << 22);   // 16 of them: lattice offsets stay 32-bit
            const bool span_ok = __ballot(fits) != 0;
            // (a single utterance beyond 4 GiB is not supported: its statistics are NaN)
            const bool active = span_ok ? fits : (cand && f_start == fmin);
            pending = pending && !active;
            // (read for every walk: kept across walks, the emission weights would hold their
            // registers through the backward pass, which has none to spare)
            {
                const float* Wm = tb.W + (size_t)m_cur * 16 * KW * 4;
#pragma unroll
                for (int q = 0; q < KW; ++q) w[q] = Wm[((size_t)ul * KW + q) * 4 + g];
                const float* Am = tb.A + (size_t)m_cur * 256;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    // forward: (A^T alpha)[row ul] with k = state 4q+g
                    atop[q] = Am[(4 * q + g) * 16 + sigma(ul)];
                    // backward: (A bb)[row ul]
                    aop[q] = Am[sigma(ul) * 16 + (4 * q + g)];
                    c0[q] = tb.cst[m_cur * 16 + 4 * g + q];
                    pi4[q] = tb.pi[m_cur * 16 + 4 * q + g];
                }
            }
            const int T = active ? T_l : 0;
            int Tmax = T;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) Tmax = max(Tmax, __shfl_xor(Tmax, o, 64));
            Tmax = __builtin_amdgcn_readfirstlane(Tmax);     // (wave-uniform values in scalar registers)

            // ---- descriptors: X from the lowest first row of the batch to its end,
            // the lattice from this batch's first row.  Offsets from kDrop up are out
            // of range for every one of them, which drops a store.
            constexpr unsigned kDrop = 0xFFFFFF00u;
            const auto make_rsrc = [&](const void* p, int64_t bytes) {
                const uintptr_t a = (uintptr_t)p;
                const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a);
                const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
                const unsigned n = (unsigned)(bytes < (int64_t)kDrop ? bytes : (int64_t)kDrop);
                return __builtin_amdgcn_make_buffer_rsrc((void*)(((uintptr_t)hi << 32) | lo), 0,
                                                         __builtin_amdgcn_readfirstlane(n), 0x00020000);
            };
            const __amdgpu_buffer_rsrc_t rsrc =
                make_rsrc(X + fmin * D, span_ok ? (n_frames - fmin) * (int64_t)D * 4 : 0);
            const __amdgpu_buffer_rsrc_t rsrc_lat = make_rsrc(lat + row_base * 16, (n_frames - row_base) * 64);
            // (scale factors: only the kernel that recomputes the emission keeps them)
            const __amdgpu_buffer_rsrc_t rsrc_sc =
                make_rsrc(kRatio ? nullptr : scale + row_base, kRatio ? 0 : (n_frames - row_base) * 4);
            const int64_t f0 = active ? f_start : fmin;
            // (posteriors: rows of S floats, within the same reach as the rows of X)
            const __amdgpu_buffer_rsrc_t rsrc_post =
                make_rsrc(POST ? post_out + fmin * S : nullptr,
                          POST && span_ok ? (n_frames - fmin) * (int64_t)S * 4 : 0);
            const unsigned post_row = (unsigned)((f0 - fmin) * S * 4);
            const unsigned lane_off = (unsigned)((f0 - fmin) * D * 4) + 16u * g;   // + 64 q for block q
            const unsigned half_off = (unsigned)((f0 - fmin) * D * 4) + 64u * RQ + 8u * g;
            auto loadx = [&](int t, frame_t& dst) {
                int tc = t < T ? t : T - 1;
                if (tc < 0) tc = 0;
                const unsigned o = (unsigned)tc * row_bytes;
#pragma unroll
                for (int q = 0; q < RQ; ++q)
                    dst.v[q] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane_off + o + 64u * q, 0, 0);
                if constexpr (H != 0) dst.h = __builtin_amdgcn_raw_buffer_load_b64(rsrc, half_off + o, 0, 0);
            };
            // b~ in log2 units for one frame; features past D meet zero weights.
            // (two accumulators: a dependent MFMA issues every 40 cycles, an
            // independent one every 32)
            auto emission = [&](const frame_t& xc) {
                f32x4 a4 = c0, a5 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int q = 0; q < RQ; ++q) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float x = __uint_as_float(xc.v[q][j]);
                        a4 = __builtin_amdgcn_mfma_f32_16x16x4f32(w[8 * q + j], x, a4, 0, 0, 0);
                        a5 = __builtin_amdgcn_mfma_f32_16x16x4f32(w[8 * q + 4 + j], x * x, a5, 0, 0, 0);
                    }
                }
                if constexpr (H != 0) {
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const float x = __uint_as_float(xc.h[j]);
                        a4 = __builtin_amdgcn_mfma_f32_16x16x4f32(w[8 * RQ + j], x, a4, 0, 0, 0);
                        a5 = __builtin_amdgcn_mfma_f32_16x16x4f32(w[8 * RQ + 2 + j], x * x, a5, 0, 0, 0);
                    }
                }
                return a4 + a5;
            };
            auto matvec = [&](const float (&aw)[4], const f32x4& v) {
                f32x4 p0 = {0.f, 0.f, 0.f, 0.f}, p1 = p0;
                p0 = __builtin_amdgcn_mfma_f32_16x16x4f32(aw[0], v[0], p0, 0, 0, 0);
                p1 = __builtin_amdgcn_mfma_f32_16x16x4f32(aw[1], v[1], p1, 0, 0, 0);
                p0 = __builtin_amdgcn_mfma_f32_16x16x4f32(aw[2], v[2], p0, 0, 0, 0);
                p1 = __builtin_amdgcn_mfma_f32_16x16x4f32(aw[3], v[3], p1, 0, 0, 0);
                return p0 + p1;
            };

            // (each pass in a scope of its own: what only one of them needs is dead in the other)
            // lattice rows of this walk: one per frame of its utterances
            unsigned pass_rows = (unsigned)T;
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) pass_rows += __shfl_xor(pass_rows, o, 64);
            pass_rows = __builtin_amdgcn_readfirstlane(pass_rows);
            double lp = 0.0;
            unsigned rows_done = pass_base;               // wave-uniform
            pass_base += span_ok ? pass_rows : 0u;
            frame_t xq[PF];
            {
            // ------------------------------------------------------ forward
            f32x4 alpha = {0.f, 0.f, 0.f, 0.f};
            // (scheduling fences: the queue must fill oldest slot first, or the
            // loop's first wait has to cover every load in flight)
#pragma unroll
            for (int i = 0; i < PF; ++i) {
                loadx(i, xq[i]);
                __builtin_amdgcn_sched_barrier(0);
            }
            auto fwd = [&](auto first, int t, const frame_t& xc) {
                const f32x4 ll = emission(xc);
                const float mx = max16(ll);
                f32x4 b, a4;
#pragma unroll
                for (int r = 0; r < 4; ++r) b[r] = __builtin_amdgcn_exp2f(ll[r] - mx);
                if constexpr (decltype(first)::value) {
                    a4 = pi4 * b;
                } else {
                    a4 = matvec(atop, alpha) * b;
                }
                const float c = sum16(a4);
                const float ci = __builtin_amdgcn_rcpf(c);
                const bool live = t < T;
                lp += live ? (double)(mx + __builtin_amdgcn_logf(c)) : 0.0;
                alpha = live ? a4 * ci : alpha;
                const unsigned lm = (unsigned)__ballot(live) & 0xFFFFu;
                const unsigned row = rows_done + __builtin_popcount(lm & below);
                rows_done += __builtin_popcount(lm);
                                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, alpha), rsrc_lat,
                                                       live ? row * 64u + 16u * g : kDrop, 0, 0);
                if constexpr (!kRatio)
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(ci), rsrc_sc,
                                                          live && g == 0 ? row * 4u : kDrop, 0, 0);
            };
            if (Tmax > 0) {
                fwd(std::true_type{}, 0, xq[0]);
                loadx(PF, xq[0]);
            }
            for (int t0 = 1; t0 < Tmax; t0 += PF) {
#pragma unroll
                for (int i = 0; i < PF; ++i) {
                    fwd(std::false_type{}, t0 + i, xq[(1 + i) % PF]);
                    loadx(t0 + i + PF, xq[(1 + i) % PF]);
                }
            }
            }

            {
            // ----------------------------------------------------- backward
            // Two steps per iteration.  Step t is produced (recursion, then its
            // gamma / alpha / b~beta / frame tiles written to buffer t & 1 ...) while
            // the tiles of step t + 1 in the other buffer are consumed by the
            // accumulating MFMAs: two independent instruction streams between
            // consecutive wave syncs.
            f32x4 bb = {0.f, 0.f, 0.f, 0.f}, post = bb, start = bb, G = bb;
            f32x4 O[NTX];
#pragma unroll
            for (int nt = 0; nt < NTX; ++nt) O[nt] = bb;
            // Step t needs alpha_t and alpha_{t-1}: the lattice rows are requested three steps
            // ahead into four slots (slot of step t = (Tmax - 1 - t) & 3), the frames two steps
            // ahead into two.  Lattice loads are issued in strictly descending t, so one
            // running row counter serves them.
            f32x4 alq[4];
            float ciq[4];
            auto loada = [&](int t, f32x4& ad, float& cd) {
                const unsigned lm = (unsigned)__ballot(t >= 0 && t < T) & 0xFFFFu;
                rows_done -= __builtin_popcount(lm);
                const unsigned row = rows_done + __builtin_popcount(lm & below);
                ad = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                                   rsrc_lat, row * 64u + 16u * g, 0, 0));
                if constexpr (!kRatio)
                    cd = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsrc_sc, row * 4u, 0, 0));
            };
            // (issued in the order the loop issues them, so that its waits stay counted)
            loada(Tmax - 1, alq[0], ciq[0]);
            __builtin_amdgcn_sched_barrier(0);
            constexpr bool kFrames = !POST && OBS;      // the backward pass reads X
            if constexpr (kFrames) loadx(Tmax - 1, xq[0]);
            __builtin_amdgcn_sched_barrier(0);
            loada(Tmax - 2, alq[1], ciq[1]);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (kFrames) loadx(Tmax - 2, xq[1]);
            __builtin_amdgcn_sched_barrier(0);
            loada(Tmax - 3, alq[2], ciq[2]);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (!POST) {
                // the first consume reads buffer 1: start it from zeros
                for (int i = lane; i < TILE / 4; i += 64) ((f32x4*)(tiles + TILE))[i] = f32x4{0.f, 0.f, 0.f, 0.f};
                wave_sync();
            }

            // (tile reads are issued as a group ahead of the MFMAs that use them)
            auto consume = [&](const float* tl) {
                const float* t_gam = tl;
                const float* t_alp = tl + 256;
                const float* t_bb = tl + 512;
                const float* t_x = tl + 768;
#pragma unroll
                for (int qh = 0; qh < 4; qh += 2) {
                    float ga[2], aa[2], ba[2], xa[2][NTX];
#pragma unroll
                    for (int qq = 0; qq < 2; ++qq) {
                        const int q = qh + qq;
                        const int row = (4 * q + g) * 16 + (ul ^ 4 * q);
                        aa[qq] = t_alp[row];
                        ba[qq] = t_bb[row];
                        if constexpr (OBS) {
                            ga[qq] = t_gam[row];
#pragma unroll
                            for (int nt = 0; nt < NTX; ++nt) {
                                // column c of [x (DP wide) | x^2 (DP wide)]
                                const int c = 16 * nt + ul;
                                const int d = c < DP ? c : c - DP;
                                xa[qq][nt] = t_x[(4 * q + g) * XS + (d ^ 4 * q)];
                            }
                        }
                    }
                    __builtin_amdgcn_sched_group_barrier(0x100, OBS ? 2 * (3 + NTX) : 4, 0);   // DS reads
#pragma unroll
                    for (int qq = 0; qq < 2; ++qq) {
                        G = __builtin_amdgcn_mfma_f32_16x16x4f32(aa[qq], ba[qq], G, 0, 0, 0);
                        if constexpr (OBS) {
#pragma unroll
                            for (int nt = 0; nt < NTX; ++nt) {
                                float xv = xa[qq][nt];
                                if (16 * nt + ul >= DP) xv *= xv;
                                O[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(ga[qq], xv, O[nt], 0, 0, 0);
                            }
                        }
                    }
                }
            };
            // al_prev: alpha of step t - 1 (kRatio; anything at t = 0, where b~ is not used)
            auto produce = [&](int t, float* tl, const frame_t& xc, const f32x4& al_in, float ci_in,
                               const f32x4& al_prev) {
                float* t_gam = tl;
                float* t_alp = tl + 256;
                float* t_bb = tl + 512;
                float* t_x = tl + 768;
                const bool live = t < T && t >= 0;
                const f32x4 al = live ? al_in : f32x4{0.f, 0.f, 0.f, 0.f};
                f32x4 b, beta;
                if constexpr (kRatio) {
                    // The forward pass formed alpha_t = (A^T alpha_{t-1}) b~_t / c_t, so the factor
                    // the backward recursion needs, b~_t / c_t, is alpha_t over that prediction:
                    // one 16 x 16 mat-vec and four divisions instead of the emission contraction
                    // and its exponentials.  A state the prediction gives no mass has alpha = 0
                    // and b~ beta = 0 either way.  Only a prediction below the smallest normal
                    // float counts as none (the hardware reciprocal of a subnormal is infinite,
                    // and infinity times a zero alpha or a zero of A would be a NaN two
                    // instructions later); the forward pass has no answer for such a step either
                    // (its own 1 / c_t is that reciprocal).  Everything above goes through: a
                    // state reached with probability 1e-33 whose emission beats the others by
                    // more than that carries alpha = 1, and 1 / pred <= 2^126 is finite; since
                    // alpha beta <= 1 per state, b~ beta <= 1 / pred and its row-stochastic
                    // combination stay finite too.  beta here is beta_t without its 1 / c_t.
                    const f32x4 pred = matvec(atop, al_prev);
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        b[r] = pred[r] >= 1.17549435e-38f ? al[r] * __builtin_amdgcn_rcpf(pred[r]) : 0.f;
                    beta = matvec(aop, bb);
                    if (t == T - 1) beta = f32x4{1.f, 1.f, 1.f, 1.f};
                } else {
                    const f32x4 ll = emission(xc);
                    const float mx = max16(ll);
#pragma unroll
                    for (int r = 0; r < 4; ++r) b[r] = __builtin_amdgcn_exp2f(ll[r] - mx);
                    const float ci = live ? ci_in : 0.f;
                    beta = matvec(aop, bb) * ci;
                    if (t == T - 1) beta = f32x4{ci, ci, ci, ci};
                }
                f32x4 gam = al * beta;
                float gs = sum16(gam);
                if (gs == 0.f) gs = 1.f;          // utils.normalize: zero rows divide by 1
                gam = gam * __builtin_amdgcn_rcpf(gs);
                if constexpr (POST) {
                    // register q of lane group g is state 4q + g
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        __builtin_amdgcn_raw_buffer_store_b32(
                            __float_as_uint(gam[q]), rsrc_post,
                            live && 4 * q + g < S ? post_row + (unsigned)t * (unsigned)(S * 4) + (unsigned)(4 * q + g) * 4u
                                                  : kDrop, 0, 0);
                } else {
                    *(f32x4*)(t_alp + ul * 16 + wsw) = al;
                    *(f32x4*)(t_bb + ul * 16 + wsw) = bb;
                    if constexpr (OBS) {
                        *(f32x4*)(t_gam + ul * 16 + wsw) = gam;
#pragma unroll
                        for (int q = 0; q < RQ; ++q) {
                            // (a finished utterance keeps its last frame here: its gamma is 0)
                            *(u32x4*)(t_x + ul * XS + 16 * q + wsw) = xc.v[q];
                        }
                        if constexpr (H != 0) {
                            *(u32x2*)(t_x + ul * XS + 16 * RQ + hsw) = xc.h;
                        }
                        post += gam;
                    }
                    if (t == 0) start = gam;
                }
                bb = live ? b * beta : f32x4{0.f, 0.f, 0.f, 0.f};
            };
            // Sums so far -> this wave's global float64 record, accumulators back to zero.
            // G and O are in accumulator layout: lane (g, ul) holds rows 4g+reg, column ul.
            // (Called at the end of a batch, and every kFlushSteps steps inside one: the float32
            // accumulators then never hold more than 16 * kFlushSteps terms, however long the
            // utterances are.)
            auto flush_sums = [&]() {
                double* dst = my_rep + (size_t)m_cur * rec_size;
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const int si = sigma(4 * g + reg), sj = sigma(ul);
                    if (si < S && sj < S) unsafeAtomicAdd(dst + L.trans + si * S + sj, (double)G[reg]);
                    G[reg] = 0.f;
                    if constexpr (OBS) {
#pragma unroll
                        for (int nt = 0; nt < NTX; ++nt) {
                            const int c = 16 * nt + ul;
                            const int d = c < DP ? c : c - DP;
                            if (si < S && d < D)
                                unsafeAtomicAdd(dst + (c < DP ? L.obs : L.obs2) + si * D + d, (double)O[nt][reg]);
                            O[nt][reg] = 0.f;
                        }
                    }
                }
                if constexpr (OBS) {
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) {
                        const float ps = sum_u(post[reg]);
                        const int s = 4 * reg + g;
                        if (ul == 0 && s < S) unsafeAtomicAdd(dst + L.post + s, (double)ps);
                        post[reg] = 0.f;
                    }
                }
            };
            // (the walk is cut into runs of kFlushSteps steps with one flush after each)
            constexpr int kFlushSteps = 512;                 // (a multiple of 4: the slots keep their steps)
            for (int hi = Tmax - 1; hi >= 0; hi -= kFlushSteps) {
                const int lo = hi >= kFlushSteps ? hi - kFlushSteps + 1 : 0;
                for (int t0 = hi; t0 >= lo; t0 -= 4) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int t = t0 - i;             // t < 0: a dead step, writes zero tiles
                        if constexpr (!POST) consume(tiles + TILE * ((i & 1) ^ 1));
                        produce(t, tiles + TILE * (i & 1), xq[i & 1], alq[i], ciq[i], alq[(i + 1) & 3]);
                        if constexpr (kFrames) loadx(t - 2, xq[i & 1]);
                        loada(t - 3, alq[(i + 3) & 3], ciq[(i + 3) & 3]);
                        if constexpr (!POST) wave_sync();
                    }
                }
                if constexpr (!POST) {
                    if (lo == 0) consume(tiles + TILE);   // the last step produced
                    flush_sums();
                }
            }

            // ---------------- what is left of the batch sums
            if constexpr (!POST) {
                double* dst = my_rep + (size_t)m_cur * rec_size;
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const float ss = sum_u(start[reg]);
                    const int s = 4 * reg + g;
                    if (ul == 0 && s < S) unsafeAtomicAdd(dst + L.start + s, (double)ss);
                }
            }
            }

            if constexpr (POST) {
                if (active && g == 0) post_lp[u] = span_ok ? lp * kLn2 : __builtin_nan("");
            } else {
                double* dst = my_rep + (size_t)m_cur * rec_size;
                double lps = span_ok ? lp : __builtin_nan("");
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) lps += __shfl_xor(lps, o, 64);
                const unsigned long long am = __ballot(active);
                if (lane == 0) {
                    unsafeAtomicAdd(dst + L.logprob, lps * kLn2);
                    unsafeAtomicAdd(dst + L.nobs, (double)__popcll(am & 0xFFFFull));
                }
            }
            wave_sync();
        }
        if (on_demand) {
            const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)ticket);
            const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(ticket >> 32));
            k = n_waves + (int64_t)(((unsigned long long)hi << 32) | lo);
        } else {
            ++k;
        }
    }
}

// Lattice row bases: the exclusive prefix sum of the frames per batch of 16 utterances (in
// processing order), in three short kernels with coalesced accesses:
//   fused_batch_total_kernel  frames of each batch, and the sum over each workgroup's 256 batches
//   fused_block_scan_kernel   exclusive prefix of those workgroup sums (one workgroup)
//   fused_batch_base_kernel   prefix inside the workgroup + the workgroup's prefix
__global__ __launch_bounds__(256) void fused_batch_total_kernel(const int64_t* __restrict__ off, int64_t U,
                                                                const int64_t* __restrict__ utt_order,
                                                                int64_t n_batches, int64_t* __restrict__ base,
                                                                int64_t* __restrict__ block_sum) {
    __shared__ int64_t part[4];
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t sum = 0;
    if (b < n_batches) {
        for (int j = 0; j < 16; ++j) {
            const int64_t e = b * 16 + j;
            if (e < U) {
                const int64_t u = utt_order ? utt_order[e] : e;
                if (u >= 0) sum += off[u + 1] - off[u];
            }
        }
        base[b] = sum;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = sum;
    __syncthreads();
    if (threadIdx.x == 0) block_sum[blockIdx.x] = (part[0] + part[1]) + (part[2] + part[3]);
}

__global__ __launch_bounds__(1024) void fused_block_scan_kernel(int64_t n, int64_t* __restrict__ v) {
    __shared__ int64_t part[1024];
    const int tid = threadIdx.x;
    const int64_t per = (n + 1023) / 1024;
    const int64_t b0 = tid * per, b1 = b0 + per < n ? b0 + per : n;
    int64_t sum = 0;
    for (int64_t b = b0; b < b1; ++b) sum += v[b];
    part[tid] = sum;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        const int64_t x = tid >= o ? part[tid - o] : 0;
        __syncthreads();
        part[tid] += x;
        __syncthreads();
    }
    int64_t run = part[tid] - sum;
    for (int64_t b = b0; b < b1; ++b) {
        const int64_t x = v[b];
        v[b] = run;
        run += x;
    }
}

__global__ __launch_bounds__(256) void fused_batch_base_kernel(int64_t n_batches, const int64_t* __restrict__ block_pre,
                                                               int64_t* __restrict__ base) {
    __shared__ int64_t part[256];
    const int tid = threadIdx.x;
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + tid;
    const int64_t mine = b < n_batches ? base[b] : 0;
    part[tid] = mine;
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1) {
        const int64_t x = tid >= o ? part[tid - o] : 0;
        __syncthreads();
        part[tid] += x;
        __syncthreads();
    }
    if (b < n_batches) base[b] = block_pre[blockIdx.x] + part[tid] - mine;
}

// Sum the replicas in fixed order, apply A to the xi accumulator.
__global__ void fused_finish_kernel(const double* __restrict__ replicas, srx_models_t md,
                                    int rec_size, double* __restrict__ out) {
    const int M = md.n_models, S = md.n_states, D = md.n_features;
    const srx_stats_layout L(S, D);
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    const int nth = gridDim.x * blockDim.x;
    for (int i = tid; i < M * rec_size; i += nth) {
        const int m = i / rec_size, r = i % rec_size;
        double s = 0.0;
        for (int k = 0; k < kReplicas; ++k) s += replicas[(size_t)k * M * rec_size + i];
        if (r >= L.trans && r < L.post) s *= md.transmat[(size_t)m * S * S + (r - L.trans)];
        out[i] = s;
    }
}

template <int RQ, int H, int NTX, bool POST = false, bool OBS = true>
int launch(srx_context* h, const float* X, const int64_t* off, int64_t U, int64_t n_frames,
           const int32_t* utt_model, const int64_t* utt_order, const srx_models_t* md,
           FusedTables tb, const int64_t* batch_base, float* lat, float* scale, double* replicas,
           int rec_size, hipStream_t st, float* post_out = nullptr, double* post_lp = nullptr,
           unsigned long long* queue = nullptr) {
    const size_t lds = POST ? 0 : sizeof(float) * 2 * (3 * 256 + 16 * frame_tile_stride(16 * RQ + 8 * H)) * kWavesPerBlock;
    auto kern = estep_fused_kernel<RQ, H, NTX, POST, OBS>;
    SRX_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds));
    const int64_t n_groups = (U + 15) / 16;
    int64_t grid = (n_groups + kWavesPerBlock - 1) / kWavesPerBlock;
    // (the posteriors variant needs fewer registers: three workgroups per CU)
    const int64_t cap = (int64_t)h->n_cu * (POST || !OBS ? 3 : 2);
    if (grid > cap) grid = cap;
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(64 * kWavesPerBlock), lds, st, X, off, U,
                       n_frames, utt_model, utt_order, md->n_models, md->n_states, md->n_features,
                       tb, batch_base, lat, scale, replicas, rec_size, post_out, post_lp, queue);
    SRX_HIP(hipGetLastError());
    return SRX_OK;
}

}  // namespace

// Visiting order that groups the utterances by model and, inside a model, by length in
// descending buckets (kOrderBuckets of equal width up to the longest utterance), each model's
// group padded to a multiple of 16 entries (-1 = empty lane): no batch of 16 mixes models,
// and the utterances of a batch have about the same length, so that few matrix columns idle
// while the longest one finishes (utterances of 60 to 240 frames in memory order fill 65 % of
// the columns, sorted ones 97 %); the longest batches come first, which is what handing them
// out on demand wants.  A counting sort in four kernels, without atomics whose result
// depends on their order, so the order -- ascending utterance index inside a model and
// bucket -- is the same on every run.
//   order_tmax_kernel     the longest utterance
//   order_hist_kernel     per workgroup of 256 utterances: counts per key (model, bucket)
//   order_offsets_kernel  per key (one block each): exclusive prefix of those counts over the
//                         workgroups (in place) and the key's total
//   order_bases_kernel    exclusive prefix of the totals over the keys, a model's total padded
//   order_scatter_kernel  position = key base + workgroup prefix + rank among the
//                         workgroup's lower-numbered utterances of that key
constexpr int kOrderKeys = 4096;         // LDS histogram entries
constexpr int kOrderBuckets = 32;
__device__ __forceinline__ int order_key(const int32_t* utt_model, const int64_t* off, int64_t u, int NB,
                                         unsigned tmax) {
    const int m = utt_model ? utt_model[u] : 0;
    const unsigned long long T = (unsigned long long)(off[u + 1] - off[u]);
    unsigned long long b = T * (unsigned)NB / ((unsigned long long)tmax + 1ull);
    if (b > (unsigned)(NB - 1)) b = NB - 1;
    return m * NB + (NB - 1 - (int)b);
}
__global__ __launch_bounds__(256) void order_tmax_kernel(const int64_t* __restrict__ off, int64_t U,
                                                         unsigned* __restrict__ tmax) {
    __shared__ unsigned part[4];
    const int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t T = u < U ? off[u + 1] - off[u] : 0;
    if (T > 0x7FFFFFFFll) T = 0x7FFFFFFFll;
    unsigned v = (unsigned)T;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = max(v, (unsigned)__shfl_xor((int)v, o, 64));
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        v = max(max(part[0], part[1]), max(part[2], part[3]));
        // one atomic per workgroup, and none when the value seen is already as large (a stale
        // read only costs an atomic; a maximum does not depend on the order)
        if (v > *(volatile unsigned*)tmax) atomicMax(tmax, v);
    }
}
__global__ __launch_bounds__(256) void order_hist_kernel(const int32_t* __restrict__ utt_model,
                                                         const int64_t* __restrict__ off, int64_t U, int K,
                                                         int NB, const unsigned* __restrict__ tmax,
                                                         unsigned* __restrict__ wg_count) {
    __shared__ unsigned hist[kOrderKeys];
    for (int k = threadIdx.x; k < K; k += blockDim.x) hist[k] = 0u;
    __syncthreads();
    const int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (u < U) atomicAdd(&hist[order_key(utt_model, off, u, NB, *tmax)], 1u);   // (LDS: counts only)
    __syncthreads();
    for (int k = threadIdx.x; k < K; k += blockDim.x) wg_count[(size_t)blockIdx.x * K + k] = hist[k];
}
__global__ __launch_bounds__(256) void order_offsets_kernel(unsigned* __restrict__ wg_count, int64_t n_wg,
                                                            int K, unsigned long long* __restrict__ total) {
    // one block per key: each thread owns a run of workgroups, a block scan joins the runs
    __shared__ unsigned long long part[256];
    const int k = blockIdx.x, tid = threadIdx.x;
    const int64_t per = (n_wg + 255) / 256;
    const int64_t w0 = tid * per, w1 = w0 + per < n_wg ? w0 + per : n_wg;
    unsigned long long sum = 0;
    for (int64_t w = w0; w < w1; ++w) sum += wg_count[(size_t)w * K + k];
    part[tid] = sum;
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1) {
        const unsigned long long v = tid >= o ? part[tid - o] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    unsigned long long run = part[tid] - sum;
    for (int64_t w = w0; w < w1; ++w) {
        const unsigned c = wg_count[(size_t)w * K + k];
        wg_count[(size_t)w * K + k] = (unsigned)run;
        run += c;
    }
    if (tid == 255) total[k] = part[255];
}
// One workgroup: a thread per model turns its buckets' totals into prefixes inside the model
// and pads the model's total to whole batches (kept in `mtot`), a scan over the models gives
// their bases, which are then added to the prefixes.
__global__ __launch_bounds__(1024) void order_bases_kernel(unsigned long long* __restrict__ total,
                                                           unsigned long long* __restrict__ mtot, int M,
                                                           int NB) {
    __shared__ unsigned long long part[1024];
    const int tid = threadIdx.x;
    const int per = (M + 1023) / 1024;
    const int m0 = tid * per, m1 = m0 + per < M ? m0 + per : M;
    unsigned long long sum = 0;
    for (int m = m0; m < m1; ++m) {
        unsigned long long run = 0;
        for (int b = 0; b < NB; ++b) {
            const unsigned long long c = total[(size_t)m * NB + b];
            total[(size_t)m * NB + b] = run;
            run += c;
        }
        mtot[m] = (run + 15ull) & ~15ull;
        sum += mtot[m];
    }
    part[tid] = sum;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        const unsigned long long v = tid >= o ? part[tid - o] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    unsigned long long base = part[tid] - sum;
    for (int m = m0; m < m1; ++m) {
        for (int b = 0; b < NB; ++b) total[(size_t)m * NB + b] += base;
        base += mtot[m];
    }
}
__global__ __launch_bounds__(256) void order_scatter_kernel(const int32_t* __restrict__ utt_model,
                                                            const int64_t* __restrict__ off, int64_t U, int K,
                                                            int NB, const unsigned* __restrict__ tmax,
                                                            const unsigned* __restrict__ wg_prefix,
                                                            const unsigned long long* __restrict__ base,
                                                            int64_t* __restrict__ order) {
    __shared__ int key_of[256];
    const int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int k_u = u < U ? order_key(utt_model, off, u, NB, *tmax) : -1;
    key_of[threadIdx.x] = k_u;
    __syncthreads();
    if (u >= U) return;
    unsigned rank = 0;
    for (int t = 0; t < (int)threadIdx.x; ++t) rank += key_of[t] == k_u;
    order[base[k_u] + wg_prefix[(size_t)blockIdx.x * K + k_u] + rank] = u;
}

// Builds that order in the context's back-pointer scratch (not otherwise used by the
// callers below); *order stays NULL when there is nothing to gain (one batch, or too many
// models for the histogram).  *entries is the length of the order: U, or an upper bound of
// the padded length (the host does not know the group sizes; what lies behind the last group
// is empty lanes).
static int build_model_order(srx_context* h, const int32_t* utt_model, const int64_t* off, int64_t U, int M,
                             hipStream_t st, const int64_t** order, int64_t* entries) {
    *order = nullptr;
    *entries = U;
    const int64_t n_wg = (U + 255) / 256;
    if (utt_model == nullptr) M = 1;
    if (U <= 16 || M > kOrderKeys) return SRX_OK;
    const int NB = kOrderBuckets * M <= kOrderKeys ? kOrderBuckets : kOrderKeys / M;
    const int K = M * NB;
    if (n_wg * K > (int64_t)(1 << 26)) return SRX_OK;
    const int64_t E = (U + 15 * (int64_t)M + 15) / 16 * 16;
    unsigned long long* total;
    int rc = srx_ws_ensure(h, WS_BACKPTR,
                           sizeof(int64_t) * ((size_t)E + (size_t)K + (size_t)M) + sizeof(unsigned) * ((size_t)n_wg * K + 64),
                           (void**)&total);
    if (rc) return rc;
    unsigned long long* mtot = total + K;
    int64_t* ord = (int64_t*)(mtot + M);
    unsigned* wg_count = (unsigned*)(ord + E);
    unsigned* tmax = wg_count + (size_t)n_wg * K;
    SRX_HIP(hipMemsetAsync(ord, 0xFF, sizeof(int64_t) * (size_t)E, st));
    SRX_HIP(hipMemsetAsync(tmax, 0, sizeof(unsigned), st));
    hipLaunchKernelGGL(order_tmax_kernel, dim3((unsigned)n_wg), dim3(256), 0, st, off, U, tmax);
    hipLaunchKernelGGL(order_hist_kernel, dim3((unsigned)n_wg), dim3(256), 0, st, utt_model, off, U, K, NB,
                       (const unsigned*)tmax, wg_count);
    hipLaunchKernelGGL(order_offsets_kernel, dim3((unsigned)K), dim3(256), 0, st, wg_count, n_wg, K, total);
    hipLaunchKernelGGL(order_bases_kernel, dim3(1), dim3(1024), 0, st, total, mtot, M, NB);
    hipLaunchKernelGGL(order_scatter_kernel, dim3((unsigned)n_wg), dim3(256), 0, st, utt_model, off, U, K, NB,
                       (const unsigned*)tmax, (const unsigned*)wg_count, (const unsigned long long*)total, ord);
    *order = ord;
    *entries = E;
    return SRX_OK;
}

int srx_fused_estep(srx_context* h, const void* X, const int64_t* off, int64_t U, int64_t n_frames,
                    const int32_t* utt_model, const int64_t* utt_order, const srx_models_t* md,
                    double* out_stats, hipStream_t st) {
    const int M = md->n_models, S = md->n_states, D = md->n_features;
    if (S > 16 || D > 48 || n_frames <= 0 || U <= 0) return SRX_EUNSUPPORTED;
    const int H = (D % 16 >= 1 && D % 16 <= 8) ? 1 : 0;
    const int RQ = H ? D / 16 : (D + 15) / 16;
    const int NTX = (2 * D + 15) / 16;
    const int KW = 8 * RQ + 4 * H;
    const int rec_size = (int)srx_stats_layout(S, D).size;
    const size_t n_w = (size_t)M * 16 * KW * 4;
    float* base;
    int rc = srx_ws_ensure(h, WS_PARAMS,
                           sizeof(float) * (n_w + (size_t)M * 16 * 2 + (size_t)M * 256),
                           (void**)&base);
    if (rc) return rc;
    FusedTables tb;
    float* cst = base + n_w;
    float* A = cst + (size_t)M * 16;
    float* pi = A + (size_t)M * 256;
    tb.W = base; tb.cst = cst; tb.A = A; tb.pi = pi;
    float *lat, *scale;
    double* replicas;
    if ((rc = srx_ws_ensure(h, WS_LATTICE, sizeof(float) * (size_t)n_frames * 16, (void**)&lat))) return rc;
    scale = nullptr;
    if (!kRatio && (rc = srx_ws_ensure(h, WS_SCALE, sizeof(float) * (size_t)n_frames, (void**)&scale))) return rc;
    // (the replica records and, behind them, the counter that hands out the batches)
    const size_t rep_bytes = sizeof(double) * (size_t)kReplicas * M * rec_size;
    if ((rc = srx_ws_ensure(h, WS_PARTIAL, rep_bytes + 256, (void**)&replicas))) return rc;
    SRX_HIP(hipMemsetAsync(replicas, 0, rep_bytes + 256, st));
    unsigned long long* queue = (unsigned long long*)((char*)replicas + rep_bytes);
    // (a caller that gives no visiting order gets one that groups the utterances by
    // model: a batch of 16 is walked once per model present in it)
    int64_t E = U;                      // entries of the visiting order
    if (utt_order == nullptr && (rc = build_model_order(h, utt_model, off, U, M, st, &utt_order, &E))) return rc;
    const int64_t n_batches = (E + 15) / 16;
    int64_t* batch_base;
    const int64_t n_blk = (n_batches + 255) / 256;
    if ((rc = srx_ws_ensure(h, WS_MISC, sizeof(int64_t) * (size_t)(n_batches + n_blk), (void**)&batch_base))) return rc;
    int64_t* block_sum = batch_base + n_batches;
    hipLaunchKernelGGL(fused_batch_total_kernel, dim3((unsigned)n_blk), dim3(256), 0, st, off, E, utt_order,
                       n_batches, batch_base, block_sum);
    hipLaunchKernelGGL(fused_block_scan_kernel, dim3(1), dim3(1024), 0, st, n_blk, block_sum);
    hipLaunchKernelGGL(fused_batch_base_kernel, dim3((unsigned)n_blk), dim3(256), 0, st, n_batches,
                       (const int64_t*)block_sum, batch_base);
    hipLaunchKernelGGL(fused_prep_kernel, dim3(24), dim3(256), 0, st, *md, RQ, H, base, cst, A, pi);
    const float* Xf = (const float*)X;
    // (a caller that re-estimates start and transition probabilities only: no sums over the
    // features, and a backward pass that reads the lattice alone)
    const bool with_obs = h->estep_stats != SRX_STATS_TRANSITIONS;
#define SRX_CASE(rq, hh, ntx)                                                                \
    if (RQ == rq && H == hh && NTX == ntx) {                                                  \
        rc = with_obs ? launch<rq, hh, ntx>(h, Xf, off, E, n_frames, utt_model, utt_order, md, \
                                            tb, batch_base, lat, scale, replicas, rec_size,   \
                                            st, nullptr, nullptr, queue)                      \
                      : launch<rq, hh, ntx, false, false>(h, Xf, off, E, n_frames, utt_model, \
                                                          utt_order, md, tb, batch_base, lat, \
                                                          scale, replicas, rec_size, st,      \
                                                          nullptr, nullptr, queue);           \
    } else
    SRX_CASE(0, 1, 1) SRX_CASE(1, 0, 2) SRX_CASE(1, 1, 3) SRX_CASE(2, 0, 4) SRX_CASE(2, 1, 5)
    SRX_CASE(3, 0, 6)
    { return SRX_EUNSUPPORTED; }
#undef SRX_CASE
    if (rc) return rc;
    // (one thread per output sum: each adds its 64 replica values in order)
    const unsigned g_fin = (unsigned)std::min<int64_t>(((int64_t)M * rec_size + 255) / 256, 4096);
    hipLaunchKernelGGL(fused_finish_kernel, dim3(g_fin), dim3(256), 0, st, (const double*)replicas, *md,
                       rec_size, out_stats);
    SRX_HIP(hipGetLastError());
    return SRX_OK;
}

// Posteriors of every frame under its utterance's model with the same kernel
// (POST variant): out_post[frame][state] float32, out_lp[utterance] (0 for an empty one).
int srx_fused_posteriors(srx_context* h, const void* X, const int64_t* off, int64_t U,
                         int64_t n_frames, const int32_t* utt_model, const srx_models_t* md,
                         float* out_post, double* out_lp, hipStream_t st) {
    const int M = md->n_models, S = md->n_states, D = md->n_features;
    if (S > 16 || D > 48 || n_frames <= 0 || U <= 0) return SRX_EUNSUPPORTED;
    const int H = (D % 16 >= 1 && D % 16 <= 8) ? 1 : 0;
    const int RQ = H ? D / 16 : (D + 15) / 16;
    const int KW = 8 * RQ + 4 * H;
    const size_t n_w = (size_t)M * 16 * KW * 4;
    float* base;
    int rc = srx_ws_ensure(h, WS_PARAMS,
                           sizeof(float) * (n_w + (size_t)M * 16 * 2 + (size_t)M * 256),
                           (void**)&base);
    if (rc) return rc;
    FusedTables tb;
    float* cst = base + n_w;
    float* A = cst + (size_t)M * 16;
    float* pi = A + (size_t)M * 256;
    tb.W = base; tb.cst = cst; tb.A = A; tb.pi = pi;
    float *lat, *scale;
    if ((rc = srx_ws_ensure(h, WS_LATTICE, sizeof(float) * (size_t)n_frames * 16, (void**)&lat))) return rc;
    scale = nullptr;
    if (!kRatio && (rc = srx_ws_ensure(h, WS_SCALE, sizeof(float) * (size_t)n_frames, (void**)&scale))) return rc;
    SRX_HIP(hipMemsetAsync(out_lp, 0, sizeof(double) * (size_t)U, st));
    // (the counter that hands out the batches)
    unsigned long long* queue;
    if ((rc = srx_ws_ensure(h, WS_PARTIAL, 256, (void**)&queue))) return rc;
    SRX_HIP(hipMemsetAsync(queue, 0, 256, st));
    // a batch of 16 utterances is walked once per model present in it: group them
    const int64_t* order = nullptr;
    int64_t E = U;
    if ((rc = build_model_order(h, utt_model, off, U, M, st, &order, &E))) return rc;
    const int64_t n_batches = (E + 15) / 16;
    int64_t* batch_base;
    const int64_t n_blk = (n_batches + 255) / 256;
    if ((rc = srx_ws_ensure(h, WS_MISC, sizeof(int64_t) * (size_t)(n_batches + n_blk), (void**)&batch_base))) return rc;
    int64_t* block_sum = batch_base + n_batches;
    hipLaunchKernelGGL(fused_batch_total_kernel, dim3((unsigned)n_blk), dim3(256), 0, st, off, E, order,
                       n_batches, batch_base, block_sum);
    hipLaunchKernelGGL(fused_block_scan_kernel, dim3(1), dim3(1024), 0, st, n_blk, block_sum);
    hipLaunchKernelGGL(fused_batch_base_kernel, dim3((unsigned)n_blk), dim3(256), 0, st, n_batches,
                       (const int64_t*)block_sum, batch_base);
    hipLaunchKernelGGL(fused_prep_kernel, dim3(24), dim3(256), 0, st, *md, RQ, H, base, cst, A, pi);
    const float* Xf = (const float*)X;
#define SRX_CASE(rq, hh)                                                                         \
    if (RQ == rq && H == hh)                                                                     \
        return launch<rq, hh, 1, true>(h, Xf, off, E, n_frames, utt_model, order, md, tb,        \
                                       batch_base, lat, scale, nullptr, 0, st, out_post, out_lp, \
                                       queue);
    SRX_CASE(0, 1) SRX_CASE(1, 0) SRX_CASE(1, 1) SRX_CASE(2, 0) SRX_CASE(2, 1) SRX_CASE(3, 0)
#undef SRX_CASE
    return SRX_EUNSUPPORTED;
}
