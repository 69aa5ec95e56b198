// Fast cepstra kernel (float32, nfft 512 or 1024): 16 lanes per frame, the FFT
// in registers.
//
// A frame's N real samples are packed as H = N/2 complex points z[n] and the
// H-point transform is split H = P x 16 (P = H/16 points per lane):
//   1. lane j of the frame's 16-lane group holds z[16 n1 + j], n1 < P -- the
//      samples 32 n1 + 2j, +1, so each of the P loads of a group reads 128
//      contiguous bytes -- and does a P-point FFT over n1 in registers;
//   2. multiplies by the twiddles W_H^(j k1);
//   3. the group transposes through LDS ([k1][j], rows padded to 17) and every
//      lane does 16-point FFTs over j for its k1 (one for P=16, two for P=32);
//   4. the real-input spectrum is unpacked from Z[k], Z[H-k] -- a lane's own lower half and
//      the upper half of one partner lane, which is all that goes through LDS -- squared and
//      scaled into the power spectrum pw[H+1];
//   5. mel energies and DCT rows on the matrix cores, as 4x4x1 instructions: each
//      is 16 independent 4x4 outer products, one per group of 4 lanes (lane 4b+i
//      gives row i of A and column i of B; D[i][j] lands in register i of lane
//      4b+j).  Rows are 4 filters (then 4 cepstra), columns the wave's 4 frames; a
//      block walks its own range of bins, a quad per step (tables built with the
//      plan: weights per (quad, lane), first bin per block).  Blocks that share
//      filters add up through LDS; log10; the DCT's 16 blocks are 4 groups of
//      cepstra x 4 slices of the filters, added across lanes at the end.
// A wave carries 4 frames at once, a workgroup 16.  Cepstra
// rows live on the packed sample axis exactly as in the generic kernel (mfcc.hip): frame t
// of an utterance that starts at sample s0 is row s0 / hop + t.  A small kernel first writes,
// for every row, the sample its frame starts at (-1: no frame there); the transform kernel
// then takes 16 consecutive rows per step and never looks at the utterance offsets.
#include "srx_internal.h"
#include "fft_regs.h"

#include <math.h>

#include <type_traits>

namespace {

constexpr int kWave = 64;
// Frames a workgroup transforms at once (4 per wave).
template <int P> constexpr int frames_per_block() { return 16; }

struct FastDev {
    int frame_len, hop, n_used, n_mel, n_ceps, nnz;
    float log_mult;
    const float* window;      // [frame_len]
    const float2* tw;         // [N] exp(-2 pi i k / N)
    unsigned hop_m, hop_s;    // n / hop = umulhi(n, hop_m) >> hop_s for n < 2^31
    int nq, kk, groups;       // quads of bins per block, filters per DCT slice, groups of 4 filters
    const float* mm_w;        // [nq][64][4]
    const float* mm_dct;      // [64][kk]
    const int32_t* mm_meta;   // [16] first bin, [16] first block, [16] blocks of a group
};


__device__ __forceinline__ int64_t raw_frames(int64_t n, int L, int hop) {
    return n < L ? 0 : (n - L) / hop + 1;
}

// row_start[s0 / hop + t] = s0 + t * hop for every frame t of every utterance (one wave per
// utterance at a time); the table is filled with -1 before.
__global__ __launch_bounds__(256) void row_start_kernel(const int64_t* __restrict__ so, int64_t U,
                                                        int frame_len, unsigned hop, unsigned hop_m,
                                                        unsigned hop_s, int32_t* __restrict__ row_start) {
    const int lane = threadIdx.x & 63;
    const int64_t n_waves = (int64_t)gridDim.x * (blockDim.x >> 6);
    for (int64_t u = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); u < U; u += n_waves) {
        const int64_t us = so[u];
        const unsigned n = (unsigned)(so[u + 1] - us);
        if (n < (unsigned)frame_len) continue;
        const unsigned T = (__umulhi(n - (unsigned)frame_len, hop_m) >> hop_s) + 1u;
        const unsigned row0 = __umulhi((unsigned)us, hop_m) >> hop_s;
        for (unsigned t = lane; t < T; t += 64) row_start[row0 + t] = (int32_t)((unsigned)us + t * hop);
    }
}


// Per transform size, what measured fastest (8192 x 3 s of audio, whole front-end pass):
//   512 points   plain-register arithmetic (1.16 ms against 1.19 packed), mel / DCT tables in LDS
//                (from global memory, which would make room for a fourth workgroup per CU: 1.22),
//                16 frames per workgroup, three workgroups per CU
//   1024 points  packed arithmetic (2.43 ms against 2.63 plain: 210 against 186 registers, two
//                waves per SIMD either way), mel / DCT tables read from global memory (11 KB that
//                every workgroup reads at the same addresses: they live in the vector cache),
//                which lets a workgroup take 16 frames instead of 12 with two workgroups per CU
//                (2.77 ms with 12 frames and the tables in LDS)
template <int P> struct FastCfg {
    typedef cf cpx;
    static constexpr bool lds_weights = true;
};
template <> struct FastCfg<32> {
    typedef v2f cpx;
    static constexpr bool lds_weights = false;
};

// Workgroups a CU holds at once (registers and LDS are sized for it)
template <int P> constexpr int kWgPerCu = P == 16 ? 3 : 2;

// NR: rows of 32 samples that a frame reaches into (ceil(n_used / 32)); P when not known at
// compile time.  Rows from NR on meet the window's zero tail: they are neither loaded nor
// multiplied, and the first transform stage knows its zeros.
template <int P, int NR>
__global__ __launch_bounds__(16 * frames_per_block<P>(), (kWgPerCu<P> * 16 * frames_per_block<P>() + 255) / 256) void cepstra_fast_kernel(
    const float* __restrict__ audio, const int32_t* __restrict__ row_start, int64_t n_rows, int64_t total,
    FastDev c, float* __restrict__ cep) {
    constexpr int H = 16 * P, N = 2 * H, KB = P / 16;   // KB 16-point FFTs per lane in step 3
    constexpr int YS = P * 17;                          // float2 per frame in the transpose tile
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    // ---- shared tables
    v2f* s_tw = (v2f*)smem_raw;                       // [H/2]: exp(-2 pi i k / N)
    float* s_win = (float*)(s_tw + H / 2);                  // [N], zero past the samples the FFT uses
    typedef typename FastCfg<P>::cpx cpx;
    constexpr bool kLdsW = FastCfg<P>::lds_weights;
    float* s_w4_lds = s_win + N;                            // [nq][64][4]
    float* s_dctA_lds = s_w4_lds + (size_t)c.nq * 256;      // [64][kk]
    const float* s_w4 = kLdsW ? s_w4_lds : c.mm_w;
    const float* s_dctA = kLdsW ? s_dctA_lds : c.mm_dct;
    int* s_meta = kLdsW ? (int*)(s_dctA_lds + 64 * c.kk) : (int*)(s_win + N);   // [48]
    if constexpr (kLdsW) {
        for (int i = threadIdx.x; i < c.nq * 256; i += blockDim.x) s_w4_lds[i] = c.mm_w[i];
        for (int i = threadIdx.x; i < 64 * c.kk; i += blockDim.x) s_dctA_lds[i] = c.mm_dct[i];
    }
    float* s_zero = (float*)(s_meta + 48);            // four zeros: the partial sum a filter group does not have
    v2f* s_tw2 = (v2f*)(s_meta + 52);                 // [P][16]: W_H^(j k1)
    float* s_frames = (float*)(s_tw2 + P * 16);
    // A frame's tile holds the transpose buffer Y / the spectrum Z first; once Z has
    // been read, the same floats hold pw [H+4], the log-mel row [4 kk] and a quarter
    // of the wave's partial mel sums [64].
    constexpr int per_frame = 2 * YS;                       // floats
    constexpr int kPw = 0, kLm = H + 4;
    const int part_off = kLm + 4 * c.kk;
    for (int i = threadIdx.x; i < H / 2; i += blockDim.x) s_tw[i] = v2f{c.tw[i].x, c.tw[i].y};
    for (int i = threadIdx.x; i < N; i += blockDim.x) s_win[i] = i < c.n_used ? c.window[i] : 0.f;
    for (int i = threadIdx.x; i < 52; i += blockDim.x) s_meta[i] = i < 48 ? c.mm_meta[i] : 0;
    for (int i = threadIdx.x; i < P * 16; i += blockDim.x)
    {
        const float2 w = c.tw[2 * (((i & 15) * (i >> 4)) & (H - 1))];
        s_tw2[i] = v2f{w.x, w.y};
    }
    __syncthreads();

    const int wave = threadIdx.x / kWave, lane = threadIdx.x % kWave;
    const int fs = lane >> 4, j = lane & 15;          // frame slot in the wave, lane in the group
    const int item = 4 * wave + fs;                   // frame slot in the workgroup
    float* fr = s_frames + (size_t)item * per_frame;
    v2f* Y = (v2f*)fr;                          // transpose tile, then Z[H]
    float* pw = fr + kPw;                             // [H+1] (+3 of padding)
    // stage 5 sees the wave's 4 frames as columns: lane (block b, column r)
    float* wfr = s_frames + (size_t)(4 * wave) * per_frame;
    const int blk = lane >> 2, col = lane & 3;
    const float* pw_col = wfr + (size_t)col * per_frame + kPw;
    float* lm_col = wfr + (size_t)col * per_frame + kLm;                  // [4 kk]
    // partial sums of block b, column r: 4 floats in the tile of frame b / 4
    auto part_at = [&](int b, int r) {
        return wfr + (size_t)(b >> 2) * per_frame + part_off + (b & 3) * 16 + r * 4;
    };
    const int k0 = s_meta[blk];
    const int grp_first = s_meta[16 + blk], grp_cnt = s_meta[32 + blk];   // (lane as (group, column))
    const int ks = lane >> 4, cg = (lane >> 2) & 3;   // DCT: slice of filters, group of cepstra

    // A workgroup walks a CONTIGUOUS range of steps of FB rows each.  The start sample of a
    // row is read two steps ahead and the samples of step b+1 are requested before step b is
    // transformed, so neither load is waited for where it is issued.
    constexpr int FB = frames_per_block<P>();
    const int64_t n_chunks = (n_rows + FB - 1) / FB;
    const int64_t per_block = (n_chunks + gridDim.x - 1) / gridDim.x;
    const int64_t b_begin = (int64_t)blockIdx.x * per_block;
    const int64_t b_end = b_begin + per_block < n_chunks ? b_begin + per_block : n_chunks;
    if (b_begin >= b_end) return;
    const unsigned hop = (unsigned)c.hop;
    // start sample of this lane group's row in step bq (-1: none)
    auto row_of = [&](int64_t bq) { return bq * FB + item; };
    auto load_start = [&](int64_t bq) -> int {
        const int64_t r = row_of(bq);
        const int v = row_start[r < n_rows ? r : n_rows - 1];
        return r < n_rows ? v : -1;
    };
    // Sample loads are raw buffer loads on a per-chunk descriptor that ends with the
    // packed buffer (a frame's 8-byte pairs are only 4-byte aligned; past the end the
    // range check returns zeros).  All P rows of 32 samples are requested: what lies
    // behind the frame's own samples meets the zero tail of the window table in a
    // multiplication whose zero wins over anything (fmul_legacy), so a frame never
    // sees its neighbour's samples.  A group without a frame reads the chunk's first
    // samples and stores nothing.
    auto fetch = [&](int start, int64_t bq, float (&xr)[2 * P]) {
        // rows of step bq start at samples >= bq * FB * hop
        const int64_t p0 = bq * FB * (int64_t)hop < total ? bq * FB * (int64_t)hop : 0;
        const uintptr_t a = (uintptr_t)(audio + p0);
        const int64_t rest = (total - p0) * 4;
        const unsigned lo32 = __builtin_amdgcn_readfirstlane((unsigned)a);
        const unsigned hi32 = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
        const unsigned nrec = __builtin_amdgcn_readfirstlane((unsigned)(rest < 0xFFFFFF00ll ? rest : 0xFFFFFF00ll));
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(((uintptr_t)hi32 << 32) | lo32), 0, nrec, 0x00020000);
        const unsigned rel = start >= 0 ? (unsigned)start - (unsigned)p0 : 0u;
        const unsigned voff = (rel + 2u * (unsigned)j) * 4u;
#pragma unroll
        for (int n1 = 0; n1 < NR; ++n1) {
            const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rsrc, voff + 128u * n1, 0, 0);
            xr[2 * n1] = __uint_as_float(v[0]);
            xr[2 * n1 + 1] = __uint_as_float(v[1]);
        }
    };
    // one step: the frames whose samples are in xc (start sample `start`, row `row`)
    // (the samples are dead once the window has been applied: the next step's are requested
    // into the same registers right there, a whole step before they are used)
    auto transform = [&](int start, int64_t row, float (&xc)[2 * P], int next_start, int64_t next_b) {
        const bool have = start >= 0;
        // ---- 1. window, bit-reversed register order, P-point FFT over n1
        const bool any = __ballot(have) != 0;     // no frame starts in this chunk for this wave?
        cpx z1[P];
#pragma unroll
        for (int n1 = 0; n1 < P; ++n1) {
            if (n1 >= NR) {
                z1[brev(n1, ilog2(P))] = cpx{0.f, 0.f};
                continue;
            }
            const int i0 = 32 * n1 + 2 * j;
            const float2 wv = *(const float2*)(s_win + i0);
            z1[brev(n1, ilog2(P))] = cpx{mul_zero_wins(xc[2 * n1], wv.x), mul_zero_wins(xc[2 * n1 + 1], wv.y)};
        }
        fetch(next_start, next_b, xc);
        if (any) {
        fft_regs<P, NR>(z1);
        // (The wave's issue priority rises along the step -- 0 in the first register stage, 2 from
        // here, 3 from the spectrum on -- so that of the waves sharing a SIMD the one furthest
        // into its step goes first: its LDS round trips then run under the other waves' transform
        // arithmetic instead of waiting behind it.  Measured on one box: 1.100 -> 1.081 ms per
        // pass at 512 points, 2.02 -> 1.84 ms at 1024.)
        __builtin_amdgcn_s_setprio(2);
        // ---- 2. twiddle W_H^(j k1) and 3. transpose through LDS: Y[k1][j]
#pragma unroll
        for (int k1 = 0; k1 < P; ++k1) Y[k1 * 17 + j] = twiddled(s_tw2[k1 * 16 + j], z1[k1], k1 == 0);
        wave_sync();
        cpx z2[KB][16];
#pragma unroll
        for (int i = 0; i < KB; ++i) {
            const int k1 = j + 16 * i;
#pragma unroll
            for (int n2 = 0; n2 < 16; ++n2) {
                const v2f y = Y[k1 * 17 + n2];
                z2[i][brev(n2, 4)] = cpx{y.x, y.y};
            }
        }
        wave_sync();
#pragma unroll
        for (int i = 0; i < KB; ++i) fft_regs<16>(z2[i]);
        __builtin_amdgcn_s_setprio(3);
        // ---- 4. real-input spectrum, power / N.  Bins k and H-k share their inputs.  Lane j now
        // holds Z[j + 16 m], m < P (m = KB k2 + i).  Its bins k = j + 16 mm, mm < P/2, are its
        // own lower half; their partners H - k = (16 - j) + 16 (P - 1 - mm) are the UPPER half of
        // lane (16 - j) mod 16 -- for j = 0 (H - k = 16 (P - mm)) one place further, Z[0] for
        // mm = 0.  So only half of the spectrum travels: every lane leaves its upper half in LDS
        // (slot u = m - P/2; lanes P + 4 floats apart: 16-byte stores, 8 lanes in 8 different
        // groups of 4 banks) and reads its partner's.
        constexpr int LT = P + 4;
        float* slots = fr;
#pragma unroll
        for (int u = 0; u < P / 2; u += 2) {
            const int m0 = u + P / 2, m1 = m0 + 1;
            const cpx a = z2[m0 % KB][m0 / KB], b = z2[m1 % KB][m1 / KB];
            *(f32x4*)(slots + j * LT + 2 * u) = f32x4{a.x, a.y, b.x, b.y};
        }
        wave_sync();
        cpx zks[P / 2], zcs[P / 2];
        {
            const int jp = (16 - j) & 15;
            const float* sp = slots + jp * LT + (j == 0 ? 2 : 0);
#pragma unroll
            for (int mm = 0; mm < P / 2; ++mm) {
                zks[mm] = z2[mm % KB][mm / KB];
                const v2f t = *(const v2f*)(sp + 2 * (P / 2 - 1 - mm));
                zcs[mm] = cpx{t.x, t.y};
            }
            if (j == 0) zcs[0] = z2[0][0];
        }
        const cpx zh = z2[0][8];                  // Z[H/2], lane 0's (the only lane that uses it)
        wave_sync();                              // the slots are in registers: pw may take their place
#pragma unroll
        for (int mm = 0; mm < P / 2; ++mm) {
            const int k = j + 16 * mm;
            // with E = (Z[k] + conj Z[H-k]) / 2 and O = -i (Z[k] - conj Z[H-k]) / 2:
            // X[k] = E + W^k O ;  X[H-k] = conj(E - W^k O).  The halves are left out: pw holds
            // |2X|^2, and the factor 1 / (4 N), a power of two, sits in the mel weights.
            power_pair(zks[mm], zcs[mm], s_tw[k], 1.f, pw[k], pw[H - k]);
        }
        if (j == 0) {   // k = H/2 is its own partner: E = (zr, 0), O = (zi, 0), W^(H/2) = -i
            pw[H / 2] = (zh.x * zh.x + zh.y * zh.y) * 4.f;
        }
        if (j < 3) pw[H + 1 + j] = 0.f;           // padding that meets zero weights: keep it finite
        wave_sync();
        __builtin_amdgcn_s_setprio(3);
        // ---- 5. mel energies: block `blk` adds its range of bins for 4 filters x 4 frames
        {
            // four accumulators, one per bin of a quad: a 4x4x1 product waits for the one before
            // it only if both add to the same registers, and 4 nq of them in one chain is what
            // this stage used to spend most of its time waiting for
            f32x4 ac[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) ac[e] = f32x4{0.f, 0.f, 0.f, 0.f};
            const float* wq = s_w4 + lane * 4;
            const float* pq = pw_col + k0;
            // (Three or four quads' operands are in flight at any time: an LDS read takes a few
            // hundred cycles to come back while twelve waves queue theirs, a quad's four products
            // a few dozen.  nq is a multiple of 3 or of 4; the refills past the last quad repeat it
            // and are not used.)
            const int nq = c.nq;
            auto walk = [&](auto depth) {
                constexpr int DEPTH = decltype(depth)::value;
                f32x4 wa[DEPTH], pa[DEPTH];
#pragma unroll
                for (int u = 0; u < DEPTH; ++u) {
                    wa[u] = *(const f32x4*)(wq + u * 256);
                    pa[u] = *(const f32x4*)(pq + u * 4);
                }
                for (int q = 0; q < nq; q += DEPTH) {
#pragma unroll
                    for (int u = 0; u < DEPTH; ++u) {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            ac[e] = __builtin_amdgcn_mfma_f32_4x4x1f32(wa[u][e], pa[u][e], ac[e], 0, 0, 0);
                        const int qn = q + DEPTH + u < nq ? q + DEPTH + u : nq - 1;
                        wa[u] = *(const f32x4*)(wq + qn * 256);
                        pa[u] = *(const f32x4*)(pq + qn * 4);
                    }
                }
            };
            // The plans of the two shapes this kernel is built for walk 6 quads (16 kHz, 512
            // points, 40 filters) and 12 (22.05 kHz, 1024 points, 26 filters): with the count
            // known, every LDS address is the lane's base plus an immediate, the first product of
            // each accumulator starts from a zero literal, and no refill is clamped.
            auto walk_fixed = [&](auto count, auto depth) {
                constexpr int NQ = decltype(count)::value, DEPTH = decltype(depth)::value;
                f32x4 wa[DEPTH], pa[DEPTH];
#pragma unroll
                for (int u = 0; u < DEPTH; ++u) {
                    wa[u] = *(const f32x4*)(wq + u * 256);
                    pa[u] = *(const f32x4*)(pq + u * 4);
                }
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    const int u = q % DEPTH;
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        ac[e] = __builtin_amdgcn_mfma_f32_4x4x1f32(
                            wa[u][e], pa[u][e], q == 0 ? f32x4{0.f, 0.f, 0.f, 0.f} : ac[e], 0, 0, 0);
                    if (q + DEPTH < NQ) {
                        wa[u] = *(const f32x4*)(wq + (q + DEPTH) * 256);
                        pa[u] = *(const f32x4*)(pq + (q + DEPTH) * 4);
                    }
                }
            };
            if (nq == 6) walk_fixed(std::integral_constant<int, 6>{}, std::integral_constant<int, 3>{});
            else if (nq == 12) walk_fixed(std::integral_constant<int, 12>{}, std::integral_constant<int, 4>{});
            else if (nq % 4 == 0) walk(std::integral_constant<int, 4>{});
            else walk(std::integral_constant<int, 3>{});
            const f32x4 acc = (ac[0] + ac[1]) + (ac[2] + ac[3]);
            *(f32x4*)part_at(blk, col) = acc;
        }
        wave_sync();
        // blocks of one group of filters add up; exact zeros -> eps; log_mult * log10
        // log_mult * log10(2) as a float and the part of it the float misses
        const double scale_d = (double)c.log_mult * 0.30102999566398120;
        const float log2_scale = (float)scale_d, log2_scale_lo = (float)(scale_d - (double)log2_scale);
        if (blk < c.groups) {
            // (a group's blocks are read at once, not one after the other's answer)
            // (a piece the group does not have is read from four zeros: one select of an
            // address instead of four of values)
            f32x4 e = *(const f32x4*)part_at(grp_first, col);
#pragma unroll
            for (int i = 1; i < 4; ++i)
                e += *(const f32x4*)(i < grp_cnt ? part_at(grp_first + i, col) : s_zero);
            for (int nb = grp_first + 4; nb < grp_first + grp_cnt; ++nb) e += *(const f32x4*)part_at(nb, col);
            f32x4 l;
            // log_mult * log10(v) on the base-2 logarithm instruction.  Exact zeros become eps, and
            // values below 2^-96 (denormals among them, which the instruction does not take) are
            // scaled up first -- on a path of its own that a wave takes only when one of its
            // energies is that small (silence, digital zero), not in every step.
            const bool small = e[0] < 0x1p-96f || e[1] < 0x1p-96f || e[2] < 0x1p-96f || e[3] < 0x1p-96f;
            if (__ballot(small) == 0) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float l2 = __builtin_amdgcn_logf(e[i]);
                    l[i] = fmaf(l2, log2_scale, l2 * log2_scale_lo);
                }
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float v = e[i] == 0.f ? 2.220446049250313e-16f : e[i];
                    const bool tiny = v < 0x1p-96f;
                    const float l2 = __builtin_amdgcn_logf(tiny ? v * 0x1p32f : v) - (tiny ? 32.f : 0.f);
                    l[i] = fmaf(l2, log2_scale, l2 * log2_scale_lo);
                }
            }
            *(f32x4*)(lm_col + 4 * blk) = l;
        } else if (blk < c.kk) {
            *(f32x4*)(lm_col + 4 * blk) = f32x4{0.f, 0.f, 0.f, 0.f};      // padding of the log-mel row
        }
        wave_sync();
        // ---- DCT-II rows: 4 cepstra x 4 frames per block, over one slice of the filters
        {
            f32x4 ac[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) ac[e] = f32x4{0.f, 0.f, 0.f, 0.f};
            const float* aq = s_dctA + lane * c.kk;
            const float* bq = lm_col + ks * c.kk;
            // (kk is a multiple of 4 and at most 32; up to four quads' operands in flight)
            const int kk = c.kk;
            // (the two shapes' plans: 12 filters per slice of 40, 8 per slice of 26 -- all operands
            // read at once from immediate offsets, accumulators started from a zero literal)
            auto dct_fixed = [&](auto count) {
                constexpr int KQ = decltype(count)::value;      // quads of filters
                f32x4 da[KQ], db[KQ];
#pragma unroll
                for (int u = 0; u < KQ; ++u) {
                    da[u] = *(const f32x4*)(aq + 4 * u);
                    db[u] = *(const f32x4*)(bq + 4 * u);
                }
#pragma unroll
                for (int u = 0; u < KQ; ++u)
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        ac[e] = __builtin_amdgcn_mfma_f32_4x4x1f32(
                            da[u][e], db[u][e], u == 0 ? f32x4{0.f, 0.f, 0.f, 0.f} : ac[e], 0, 0, 0);
            };
            if (kk == 12) dct_fixed(std::integral_constant<int, 3>{});
            else if (kk == 8) dct_fixed(std::integral_constant<int, 2>{});
            else {
            f32x4 da[4], db[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int qn = 4 * u < kk ? 4 * u : kk - 4;
                da[u] = *(const f32x4*)(aq + qn);
                db[u] = *(const f32x4*)(bq + qn);
            }
#pragma unroll 1
            for (int q = 0; q < kk; q += 16) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (q + 4 * u < kk) {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            ac[e] = __builtin_amdgcn_mfma_f32_4x4x1f32(da[u][e], db[u][e], ac[e], 0, 0, 0);
                    }
                    const int qn = q + 16 + 4 * u < kk ? q + 16 + 4 * u : kk - 4;
                    if (q + 16 < kk) {
                        da[u] = *(const f32x4*)(aq + qn);
                        db[u] = *(const f32x4*)(bq + qn);
                    }
                }
            }
            }
            f32x4 acc = (ac[0] + ac[1]) + (ac[2] + ac[3]);
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = sum_rows(acc[i]);
            // the frame of column `col` belongs to lane group `col` of the FFT stages
            const int rowi = have ? (int)row : -1;
            const int r0 = __builtin_amdgcn_readlane(rowi, 0), r1 = __builtin_amdgcn_readlane(rowi, 16);
            const int r2 = __builtin_amdgcn_readlane(rowi, 32), r3 = __builtin_amdgcn_readlane(rowi, 48);
            const int rc = col == 0 ? r0 : col == 1 ? r1 : col == 2 ? r2 : r3;
            if (ks == 0 && rc >= 0) {
                float* dst = cep + (int64_t)rc * c.n_ceps + 4 * cg;
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (4 * cg + i < c.n_ceps) dst[i] = acc[i];
            }
        }
        wave_sync();
        __builtin_amdgcn_s_setprio(0);
        }   // any
    };
    float x[2 * P];
    int s0 = load_start(b_begin), s1 = load_start(b_begin + 1);
    fetch(s0, b_begin, x);
    for (int64_t b = b_begin; b < b_end; ++b) {
        const int s2 = load_start(b + 2);
        transform(s0, row_of(b), x, s1, b + 1 < b_end ? b + 1 : b);   // (the last step's repeat is harmless)
        s0 = s1;
        s1 = s2;
    }
}

template <int P, int NR>
int launch_fast(srx_context* h, const float* audio, const int64_t* so, int64_t U, int64_t total,
                const FastDev& d, float* cep, hipStream_t st) {
    constexpr int H = 16 * P, N = 2 * H, YS = P * 17;
    const size_t tables = sizeof(float2) * (H / 2) + sizeof(float2) * P * 16 + 4 * ((size_t)N + 52) +
                          (FastCfg<P>::lds_weights ? 4 * ((size_t)d.nq * 256 + 64 * (size_t)d.kk) : 0);
    const size_t per_frame = 4 * (size_t)2 * YS;
    if ((size_t)(H + 4) + 4 * (size_t)d.kk + 64 > (size_t)2 * YS) return SRX_EUNSUPPORTED;
    const size_t lds = tables + per_frame * frames_per_block<P>();

    if (lds > 160 * 1024) return SRX_EUNSUPPORTED;
    auto kern = cepstra_fast_kernel<P, NR>;
    SRX_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    // start sample of every cepstra row (rows as in srx_mfcc_batch: total / hop + 1)
    const int64_t n_rows = total / d.hop + 1;
    int32_t* row_start;
    int rc = srx_ws_ensure(h, WS_MISC, sizeof(int32_t) * (size_t)n_rows, (void**)&row_start);
    if (rc) return rc;
    SRX_HIP(hipMemsetAsync(row_start, 0xFF, sizeof(int32_t) * (size_t)n_rows, st));
    {
        int64_t g = (U + 3) / 4;
        const int64_t cap = (int64_t)h->n_cu * 8;
        if (g > cap) g = cap;
        hipLaunchKernelGGL(row_start_kernel, dim3((unsigned)g), dim3(256), 0, st, so, U, d.frame_len,
                           (unsigned)d.hop, d.hop_m, d.hop_s, row_start);
    }
    const int64_t n_chunks = (n_rows + frames_per_block<P>() - 1) / frames_per_block<P>();
    // four rounds of the workgroups that fit a CU at once (LDS decides)
    int per_cu = (int)((160 * 1024) / lds);
    per_cu = per_cu < 1 ? 1 : (per_cu > kWgPerCu<P> ? kWgPerCu<P> : per_cu);
    const int64_t want = (int64_t)h->n_cu * per_cu * 4;
    int64_t grid = n_chunks < want ? n_chunks : want;
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(16 * frames_per_block<P>()), lds, st, audio,
                       (const int32_t*)row_start, n_rows, total, d, cep);
    SRX_HIP(hipGetLastError());
    return SRX_OK;
}

}  // namespace


int srx_mfcc_fast_cepstra(srx_context* h, srx_mfcc_plan* p, const float* audio, const int64_t* so,
                          int64_t U, int64_t total, float* cep, hipStream_t st) {
    const srx_mfcc_cfg_t& c = p->cfg;
    if (!p->mm_ok || (c.nfft != 512 && c.nfft != 1024) || total >= 0x7FFFFF00ll || c.hop < 2)
        return SRX_EUNSUPPORTED;
    FastDev d;
    d.frame_len = c.frame_len; d.hop = c.hop;
    d.n_used = c.frame_len < c.nfft ? c.frame_len : c.nfft;
    d.n_mel = c.n_mel; d.n_ceps = c.n_ceps; d.nnz = p->nnz;
    {
        // l = ceil(log2 hop), m = ceil(2^(31+l) / hop) < 2^32: floor(n / hop) = (n m) >> (31 + l)
        int l = 0;
        while ((1ll << l) < c.hop) ++l;
        const unsigned __int128 one = 1;
        d.hop_m = (unsigned)(((one << (31 + l)) + c.hop - 1) / (unsigned)c.hop);
        d.hop_s = (unsigned)(l - 1);
    }
    d.nq = p->mm_nq; d.kk = p->mm_kk; d.groups = p->mm_groups;
    d.mm_w = (const float*)p->mm_w; d.mm_dct = (const float*)p->mm_dct; d.mm_meta = p->mm_meta;
    d.log_mult = (float)c.log_mult;
    d.window = (const float*)p->window[0];
    d.tw = (const float2*)p->tw_full;
    // (the two shapes the plans are made for: 400 samples in 512 points, 551 in 1024)
    const int rows = (d.n_used + 31) / 32;
    if (c.nfft == 512)
        return rows == 13 ? launch_fast<16, 13>(h, audio, so, U, total, d, cep, st)
                          : launch_fast<16, 16>(h, audio, so, U, total, d, cep, st);
    return rows == 18 ? launch_fast<32, 18>(h, audio, so, U, total, d, cep, st)
                      : launch_fast<32, 32>(h, audio, so, U, total, d, cep, st);
}
