// Fused Baum-Welch E-step on the matrix cores (float32 data, S <= 16, D <= 40).
//
// One wave processes 16 utterances of ONE model at a time, step by step:
//   forward   emission contraction (as in score_mfma.hip) -> b~_t, then
//             alpha_t = (A^T alpha_{t-1}) * b~_t / c_t with the 16x16 mat-vec
//             as 4 MFMAs; alpha_t and 1/c_t are spilled ([frames,16] floats).
//   backward  emission again (cheaper than spilling it), beta_t = c_t A (b~ beta)_{t+1}
//             as 4 MFMAs, gamma_t, and three accumulating products whose
//             contraction index is the utterance:
//                 G   += alpha_t   (b~ beta)_{t+1}^T      (16 x 16)     4 MFMAs
//                 obs += gamma_t   [x_t | x_t^2]          (16 x 2 DP)  4R MFMAs
//             (xi sums are A o G, applied once at the end).
// State placement.  MFMA row r of the 16-row tile carries state 4*(r&3)+(r>>2), so
// the accumulator register q of lane group g (rows 4g..4g+3) is state 4q+g --
// exactly the k index that lane group supplies as the B operand of k-step q.
// The lattice therefore feeds the next mat-vec straight from the accumulator
// registers, no lane movement.  Products that contract over utterances need the
// utterance on the k axis instead of the lane: alpha, gamma, b~beta and the
// frame go through small per-wave LDS tiles ([utterance][row]) for that.
// Accumulation: f32 inside the MFMAs over one batch of 16 utterances, then
// added to a per-wave float64 record in LDS (each element owned by one lane),
// which is flushed with f64 atomics into one of kReplicas global records when
// the wave's model changes or its work ends; a last kernel sums the replicas in
// fixed order and applies A to G.
#include "srx_internal.h"

#include <math.h>

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kReplicas = 64;
constexpr int kWavesPerBlock = 4;
constexpr float kNegBig = -1.0e30f;
constexpr double kLn2 = 0.6931471805599453;

__host__ __device__ inline int sigma(int row) { return 4 * (row & 3) + (row >> 2); }

struct FusedTables {
    const float* W;     // [M][16 rows][4R][4]
    const float* cst;   // [M][16 rows]
    const float* A;     // [M][16][16] zero padded, state order
    const float* pi;    // [M][16]
};

__global__ void fused_prep_kernel(srx_models_t md, int R, float* W, float* cst, float* A,
                                  float* pi) {
    const int M = md.n_models, S = md.n_states, D = md.n_features;
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    const int nth = gridDim.x * blockDim.x;
    const int KQ = 4 * R;
    for (int i = tid; i < M * 16 * KQ * 4; i += nth) {
        const int g = i & 3, q = (i >> 2) % KQ, row = ((i >> 2) / KQ) % 16, m = (i >> 2) / KQ / 16;
        const int s = sigma(row);
        const int r = q >> 2, e = q & 3;
        const int d = 8 * r + 2 * g + (e & 1);
        float v = 0.f;
        if (s < S && d < D) {
            const size_t k = ((size_t)m * S + s) * D + d;
            const double var = fmax(md.covars[k], 2.2250738585072014e-308);
            v = (float)((e >> 1) ? (-0.5 / var) * 1.4426950408889634
                                 : (md.means[k] / var) * 1.4426950408889634);
        }
        W[i] = v;
    }
    for (int i = tid; i < M * 16; i += nth) {
        const int row = i % 16, m = i / 16, s = sigma(row);
        float v = kNegBig;
        if (s < S) {
            double acc = D * 1.8378770664093453;
            for (int d = 0; d < D; ++d) {
                const size_t k = ((size_t)m * S + s) * D + d;
                const double var = fmax(md.covars[k], 2.2250738585072014e-308);
                acc += log(var) + md.means[k] * md.means[k] / var;
            }
            v = (float)(-0.5 * acc * 1.4426950408889634);
        }
        cst[i] = v;
        pi[i] = (row < S) ? (float)md.startprob[m * S + row] : 0.f;   // state order
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

__device__ __forceinline__ float sum16(f32x4 v) {   // over the 16 states of an utterance
    float s = (v[0] + v[1]) + (v[2] + v[3]);
    s += __shfl_xor(s, 16, 64);
    s += __shfl_xor(s, 32, 64);
    return s;
}
__device__ __forceinline__ float max16(f32x4 v) {
    float s = fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3]));
    s = fmaxf(s, __shfl_xor(s, 16, 64));
    s = fmaxf(s, __shfl_xor(s, 32, 64));
    return s;
}
__device__ __forceinline__ float sum_u(float v) {   // over the 16 utterance lanes of a group
    v += __shfl_xor(v, 1, 64);
    v += __shfl_xor(v, 2, 64);
    v += __shfl_xor(v, 4, 64);
    v += __shfl_xor(v, 8, 64);
    return v;
}

template <int R>
__global__ __launch_bounds__(256, 2) void estep_fused_kernel(
    const float* __restrict__ X, const int64_t* __restrict__ off, int64_t U, int64_t n_frames,
    const int32_t* __restrict__ utt_model, const int64_t* __restrict__ utt_order, int M, int S,
    int D, FusedTables tb, float* __restrict__ lat, float* __restrict__ scale,
    double* __restrict__ replicas, int rec_size) {
    constexpr int KQ = 4 * R;      // emission k-steps
    constexpr int DP = 8 * R;      // padded feature count
    constexpr int XW = DP;         // row of the frame tile (squares are formed on read)
    constexpr int NTX = R;         // 16-column tiles of the frame tile
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63, g = lane >> 4, ul = lane & 15;
    const int rec_pad = (rec_size + 1) & ~1;
    const size_t wave_bytes = sizeof(double) * rec_pad + sizeof(float) * (3 * 256 + 16 * XW);
    double* acc = (double*)(smem_raw + wave_bytes * wv);
    float* t_gam = (float*)(acc + rec_pad);
    float* t_alp = t_gam + 256;
    float* t_bb = t_alp + 256;
    float* t_x = t_bb + 256;
    for (int i = lane; i < rec_size; i += 64) acc[i] = 0.0;

    const srx_stats_layout L(S, D);
    const int64_t n_groups = (U + 15) / 16;
    const int64_t n_waves = (int64_t)gridDim.x * kWavesPerBlock;
    const int64_t wid = (int64_t)blockIdx.x * kWavesPerBlock + wv;
    const int64_t per = (n_groups + n_waves - 1) / n_waves;
    const int64_t k0 = wid * per, k1 = (k0 + per < n_groups) ? k0 + per : n_groups;
    double* my_rep = replicas + (size_t)(wid % kReplicas) * M * rec_size;

    int cur_model = -1;
    float w[KQ];
    f32x4 c0, pi4;
    float atop[4], aop[4];
#pragma unroll
    for (int q = 0; q < KQ; ++q) w[q] = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) { atop[q] = aop[q] = 0.f; c0[q] = 0.f; pi4[q] = 0.f; }

    auto flush_global = [&]() {
        if (cur_model < 0) return;
        wave_sync();
        double* dst = my_rep + (size_t)cur_model * rec_size;
        for (int i = lane; i < rec_size; i += 64) {
            const double v = acc[i];
            if (v != 0.0) unsafeAtomicAdd(dst + i, v);
            acc[i] = 0.0;
        }
        wave_sync();
    };

    for (int64_t k = k0; k < k1; ++k) {
        const int64_t e = k * 16 + ul;
        const bool valid = e < U;
        const int64_t u = valid ? (utt_order ? utt_order[e] : e) : 0;
        const int m_l = valid ? (utt_model ? utt_model[u] : 0) : -1;
        const int64_t f_start = valid ? off[u] : 0;
        const int T_l = valid ? (int)(off[u + 1] - f_start) : 0;
        bool pending = valid && T_l > 0;
        while (true) {
            const unsigned long long mask = __ballot(pending);
            if (mask == 0) break;
            const int first = __ffsll((long long)mask) - 1;
            const int m_cur = __shfl(m_l, first, 64);
            const bool active = pending && m_l == m_cur;
            pending = pending && !active;
            if (m_cur != cur_model) {
                flush_global();
                cur_model = m_cur;
                const float* Wm = tb.W + (size_t)m_cur * 16 * KQ * 4;
#pragma unroll
                for (int q = 0; q < KQ; ++q) w[q] = Wm[((size_t)ul * KQ + q) * 4 + g];
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
            int64_t f0 = active ? f_start : 0;
            if (f0 > n_frames - 1) f0 = n_frames - 1;
            int Tmax = T;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) Tmax = max(Tmax, __shfl_xor(Tmax, o, 64));

            const float* xbase = X + f0 * D;
            float* latp = lat + f0 * 16 + 4 * g;
            float* scp = scale + f0;
            float xc[R][2], xn[R][2];
            auto load = [&](int t, float (&dst)[R][2]) {
                int tc = t < T ? t : T - 1;
                if (tc < 0) tc = 0;
                const float* p = xbase + (int64_t)tc * D;
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    // features past D meet zero weights: read a valid neighbour
                    // instead of predicating the load
                    dst[r][0] = p[min(8 * r + 2 * g, D - 1)];
                    dst[r][1] = p[min(8 * r + 2 * g + 1, D - 1)];
                }
            };
            auto emission = [&](const float (&x)[R][2]) {
                f32x4 a4 = c0;
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const float x0 = x[r][0], x1 = x[r][1];
                    a4 = __builtin_amdgcn_mfma_f32_16x16x4f32(w[4 * r + 0], x0, a4, 0, 0, 0);
                    a4 = __builtin_amdgcn_mfma_f32_16x16x4f32(w[4 * r + 1], x1, a4, 0, 0, 0);
                    a4 = __builtin_amdgcn_mfma_f32_16x16x4f32(w[4 * r + 2], x0 * x0, a4, 0, 0, 0);
                    a4 = __builtin_amdgcn_mfma_f32_16x16x4f32(w[4 * r + 3], x1 * x1, a4, 0, 0, 0);
                }
                return a4;
            };

            // ------------------------------------------------------ forward
            f32x4 alpha = {0.f, 0.f, 0.f, 0.f};
            double lp = 0.0;
            load(0, xc);
            for (int t = 0; t < Tmax; ++t) {
                if (t + 1 < Tmax) load(t + 1, xn);
                const f32x4 ll = emission(xc);
                const float mx = max16(ll);
                f32x4 b, a4;
#pragma unroll
                for (int r = 0; r < 4; ++r) b[r] = __builtin_amdgcn_exp2f(ll[r] - mx);
                if (t == 0) {
                    a4 = pi4 * b;
                } else {
                    f32x4 p = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        p = __builtin_amdgcn_mfma_f32_16x16x4f32(atop[q], alpha[q], p, 0, 0, 0);
                    a4 = p * b;
                }
                const float c = sum16(a4);
                const float ci = 1.0f / c;
                if (t < T) {
                    lp += (double)(mx + __builtin_amdgcn_logf(c));
                    alpha = a4 * ci;
                    *(f32x4*)(latp + (int64_t)t * 16) = alpha;
                    if (g == 0) scp[t] = ci;
                }
#pragma unroll
                for (int r = 0; r < R; ++r) { xc[r][0] = xn[r][0]; xc[r][1] = xn[r][1]; }
            }

            // ----------------------------------------------------- backward
            f32x4 bb = {0.f, 0.f, 0.f, 0.f}, post = bb, start = bb, G = bb;
            f32x4 O[NTX];
#pragma unroll
            for (int nt = 0; nt < NTX; ++nt) O[nt] = bb;
            if (Tmax > 0) load(Tmax - 1, xc);
            for (int t = Tmax - 1; t >= 0; --t) {
                if (t > 0) load(t - 1, xn);
                const bool live = t < T;
                const f32x4 ll = emission(xc);
                const float mx = max16(ll);
                f32x4 b, al = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int r = 0; r < 4; ++r) b[r] = __builtin_amdgcn_exp2f(ll[r] - mx);
                float ci = 0.f;
                if (live) {
                    al = *(const f32x4*)(latp + (int64_t)t * 16);
                    ci = scp[t];
                }
                f32x4 p = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    p = __builtin_amdgcn_mfma_f32_16x16x4f32(aop[q], bb[q], p, 0, 0, 0);
                f32x4 beta = p * ci;
                if (t == T - 1) beta = f32x4{ci, ci, ci, ci};
                f32x4 gam = al * beta;
                float gs = sum16(gam);
                if (gs == 0.f) gs = 1.f;          // utils.normalize: zero rows divide by 1
                gam = gam / gs;
                // tiles with the utterance as row: [ul][4g .. 4g+3]
                *(f32x4*)(t_gam + ul * 16 + 4 * g) = gam;
                *(f32x4*)(t_alp + ul * 16 + 4 * g) = al;
                *(f32x4*)(t_bb + ul * 16 + 4 * g) = bb;
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const float x0 = live ? xc[r][0] : 0.f, x1 = live ? xc[r][1] : 0.f;
                    *(float2*)(t_x + ul * XW + 8 * r + 2 * g) = make_float2(x0, x1);
                }
                wave_sync();
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int row = (4 * q + g) * 16 + ul;
                    G = __builtin_amdgcn_mfma_f32_16x16x4f32(t_alp[row], t_bb[row], G, 0, 0, 0);
                    const float gq = t_gam[row];
#pragma unroll
                    for (int nt = 0; nt < NTX; ++nt) {
                        // column c of [x | x^2]: c < DP is x_c, else x_(c-DP) squared
                        const int c = 16 * nt + ul;
                        float xv = t_x[(4 * q + g) * XW + (c < DP ? c : c - DP)];
                        if (c >= DP) xv *= xv;
                        O[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(gq, xv, O[nt], 0, 0, 0);
                    }
                }
                post += gam;
                if (t == 0) start = gam;
                bb = live ? b * beta : f32x4{0.f, 0.f, 0.f, 0.f};
                wave_sync();
#pragma unroll
                for (int r = 0; r < R; ++r) { xc[r][0] = xn[r][0]; xc[r][1] = xn[r][1]; }
            }

            // ------------------------ batch sums -> per-wave float64 record
            // G and O are in accumulator layout: lane (g, ul) holds rows 4g+reg, column ul.
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int si = sigma(4 * g + reg), sj = sigma(ul);
                if (si < S && sj < S) acc[L.trans + si * S + sj] += (double)G[reg];
#pragma unroll
                for (int nt = 0; nt < NTX; ++nt) {
                    const int c = 16 * nt + ul;
                    const int d = c < DP ? c : c - DP;
                    if (si < S && d < D)
                        acc[(c < DP ? L.obs : L.obs2) + si * D + d] += (double)O[nt][reg];
                }
            }
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const float ps = sum_u(post[reg]), ss = sum_u(start[reg]);
                const int s = 4 * reg + g;
                if (ul == 0 && s < S) {
                    acc[L.post + s] += (double)ps;
                    acc[L.start + s] += (double)ss;
                }
            }
            {
                double lps = lp;
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) lps += __shfl_xor(lps, o, 64);
                const unsigned long long am = __ballot(active);
                if (lane == 0) {
                    acc[L.logprob] += lps * kLn2;
                    acc[L.nobs] += (double)__popcll(am & 0xFFFFull);
                }
            }
            wave_sync();
        }
    }
    flush_global();
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

template <int R>
int launch(srx_context* h, const float* X, const int64_t* off, int64_t U, int64_t n_frames,
           const int32_t* utt_model, const int64_t* utt_order, const srx_models_t* md,
           FusedTables tb, float* lat, float* scale, double* replicas, int rec_size,
           hipStream_t st) {
    const int rec_pad = (rec_size + 1) & ~1;
    const size_t wave_bytes = sizeof(double) * rec_pad + sizeof(float) * (3 * 256 + 16 * 8 * R);
    const size_t lds = wave_bytes * kWavesPerBlock;
    auto kern = estep_fused_kernel<R>;
    SRX_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds));
    const int64_t n_groups = (U + 15) / 16;
    int64_t grid = (n_groups + kWavesPerBlock - 1) / kWavesPerBlock;
    const int64_t cap = (int64_t)h->n_cu * 2;
    if (grid > cap) grid = cap;
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(64 * kWavesPerBlock), lds, st, X, off, U,
                       n_frames, utt_model, utt_order, md->n_models, md->n_states, md->n_features,
                       tb, lat, scale, replicas, rec_size);
    SRX_HIP(hipGetLastError());
    return SRX_OK;
}

}  // namespace

int srx_fused_estep(srx_context* h, const void* X, const int64_t* off, int64_t U, int64_t n_frames,
                    const int32_t* utt_model, const int64_t* utt_order, const srx_models_t* md,
                    double* out_stats, hipStream_t st) {
    const int M = md->n_models, S = md->n_states, D = md->n_features;
    if (S > 16 || D > 40 || n_frames <= 0 || U <= 0) return SRX_EUNSUPPORTED;
    int R = (D + 7) / 8;
    if (R == 3) R = 4;
    if (R == 1) R = 2;
    const int KQ = 4 * R;
    const int rec_size = (int)srx_stats_layout(S, D).size;
    const size_t n_w = (size_t)M * 16 * KQ * 4;
    float* base;
    int rc = srx_ws_ensure(h, WS_PARAMS, sizeof(float) * (n_w + (size_t)M * 16 * 2 + (size_t)M * 256),
                           (void**)&base);
    if (rc) return rc;
    FusedTables tb;
    float* W = base;
    float* cst = W + n_w;
    float* A = cst + (size_t)M * 16;
    float* pi = A + (size_t)M * 256;
    tb.W = W; tb.cst = cst; tb.A = A; tb.pi = pi;
    float *lat, *scale;
    double* replicas;
    if ((rc = srx_ws_ensure(h, WS_LATTICE, sizeof(float) * (size_t)n_frames * 16, (void**)&lat))) return rc;
    if ((rc = srx_ws_ensure(h, WS_SCALE, sizeof(float) * (size_t)n_frames, (void**)&scale))) return rc;
    const size_t rep_bytes = sizeof(double) * (size_t)kReplicas * M * rec_size;
    if ((rc = srx_ws_ensure(h, WS_PARTIAL, rep_bytes, (void**)&replicas))) return rc;
    SRX_HIP(hipMemsetAsync(replicas, 0, rep_bytes, st));
    hipLaunchKernelGGL(fused_prep_kernel, dim3(8), dim3(256), 0, st, *md, R, W, cst, A, pi);
    const float* Xf = (const float*)X;
    if (R == 2) rc = launch<2>(h, Xf, off, U, n_frames, utt_model, utt_order, md, tb, lat, scale, replicas, rec_size, st);
    else if (R == 4) rc = launch<4>(h, Xf, off, U, n_frames, utt_model, utt_order, md, tb, lat, scale, replicas, rec_size, st);
    else rc = launch<5>(h, Xf, off, U, n_frames, utt_model, utt_order, md, tb, lat, scale, replicas, rec_size, st);
    if (rc) return rc;
    hipLaunchKernelGGL(fused_finish_kernel, dim3(8), dim3(256), 0, st, (const double*)replicas, *md,
                       rec_size, out_stats);
    SRX_HIP(hipGetLastError());
    return SRX_OK;
}
