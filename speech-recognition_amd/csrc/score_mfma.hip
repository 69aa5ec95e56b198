// Fused matrix-core scoring: emission log-densities and the scaled forward
// recursion in one pass over X, for all word models at once.
//
// Emission as a contraction (per frame x, per state column c = model*SP + s):
//     ll2[c] = cst[c] + sum_d  x_d * (mu/var * log2e) + x_d^2 * (-1/(2 var) * log2e)
// i.e. [W | W2] (16*NT x 2D) times [x ; x^2] (2D x 16 utterances), computed with
// v_mfma_f32_16x16x4_f32 (exact f32 FMA chains), the constant as the initial
// accumulator.  Everything is kept in log2 units so v_exp_f32 / v_log_f32 need
// no rescaling; the result is multiplied by ln 2 once at the end.
//
// Wave geometry.  One wave scores 16 utterances against all models, time step
// by time step.  MFMA rows = state columns, MFMA columns = utterances, so the
// accumulator of lane l (g = l>>4, u = l&15) holds state columns 4g..4g+3 of
// utterance u: the forward recursion needs no transpose.  SP = 4: the four
// lane groups are four models with all states in one lane.  SP = 8: a model's
// eight states sit in lanes l and l^16.
// The B operand of k-step q comes from lane group g: feature (q, g) is
//     q = 4r+e:  d = 8r + 2g + (e&1),  x_d (e<2) or x_d^2 (e>=2)
// so every lane reads two adjacent floats per r -- the four groups of an
// utterance together read its whole frame, each byte once.
// The weights (4R k-steps x NT tiles) and the transition coefficients are
// loop-invariant and live in registers for the whole utterance batch.
#include "srx_internal.h"

#include <math.h>

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr float kLog2e = 1.4426950408889634f;
constexpr double kLn2 = 0.6931471805599453;
constexpr float kNegBig = -1.0e30f;

struct MfmaTables {
    const float* W;     // [NT*16][4R][4]
    const float* cst;   // [NT*16]
    const float* A;     // [M][S][S]
    const float* pi;    // [M][S]
};

__global__ void mfma_prep_kernel(srx_models_t md, int SP, int NT, int R, float* W, float* cst,
                                 float* A, float* pi) {
    const int M = md.n_models, S = md.n_states, D = md.n_features;
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    const int nth = gridDim.x * blockDim.x;
    const int KQ = 4 * R;
    for (int i = tid; i < NT * 16 * KQ * 4; i += nth) {
        const int g = i & 3, q = (i >> 2) % KQ, row = (i >> 2) / KQ;
        const int m = row / SP, s = row % SP;
        const int r = q >> 2, e = q & 3;
        const int d = 8 * r + 2 * g + (e & 1);
        float v = 0.f;
        if (m < M && s < S && d < D) {
            const size_t k = ((size_t)m * S + s) * D + d;
            const double var = fmax(md.covars[k], 2.2250738585072014e-308);
            v = (float)((e >> 1) ? (-0.5 / var) * 1.4426950408889634
                                 : (md.means[k] / var) * 1.4426950408889634);
        }
        W[i] = v;
    }
    for (int row = tid; row < NT * 16; row += nth) {
        const int m = row / SP, s = row % SP;
        float v = kNegBig;
        if (m < M && s < S) {
            double acc = D * 1.8378770664093453;
            for (int d = 0; d < D; ++d) {
                const size_t k = ((size_t)m * S + s) * D + d;
                const double var = fmax(md.covars[k], 2.2250738585072014e-308);
                acc += log(var) + md.means[k] * md.means[k] / var;
            }
            v = (float)(-0.5 * acc * 1.4426950408889634);
        }
        cst[row] = v;
    }
    for (int i = tid; i < M * S * S; i += nth) A[i] = (float)md.transmat[i];
    for (int i = tid; i < M * S; i += nth) pi[i] = (float)md.startprob[i];
}

__device__ __forceinline__ float xchg16(float v) { return __shfl_xor(v, 16, 64); }

template <int SP, int NT, int R>
__global__ __launch_bounds__(256, 2) void score_mfma_kernel(
    const float* __restrict__ X, const int64_t* __restrict__ off, int64_t U, int64_t n_frames,
    int M, int S, int D, MfmaTables tb, double* __restrict__ out) {
    constexpr int KQ = 4 * R;
    const int lane = threadIdx.x & 63, g = lane >> 4, ul = lane & 15;
    const int64_t group = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (group * 16 >= U) return;
    const int64_t u = group * 16 + ul;
    const bool valid = u < U;
    const int64_t uc = valid ? u : U - 1;
    int64_t f0 = off[uc];
    const int T = valid ? (int)(off[uc + 1] - f0) : 0;
    if (f0 > n_frames - 1) f0 = n_frames - 1;   // empty trailing utterance: stay in bounds
    int Tmax = T;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) Tmax = max(Tmax, __shfl_xor(Tmax, o, 64));

    // ---- loop-invariant registers
    float w[NT][KQ];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int q = 0; q < KQ; ++q) w[nt][q] = tb.W[(((size_t)(nt * 16 + ul)) * KQ + q) * 4 + g];
    f32x4 c0[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) c0[nt][r] = tb.cst[nt * 16 + 4 * g + r];

    constexpr int NI = SP == 4 ? 4 : 8;        // source states per model
    float a[NT][NI][4];                        // a[nt][i][r]: from source i to own state r
    float pi_[NT][4];
    int model[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int m = SP == 4 ? 4 * nt + g : 2 * nt + (g >> 1);
        const int h = SP == 4 ? 0 : (g & 1);
        model[nt] = m;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int j = 4 * h + r;
            pi_[nt][r] = (m < M && j < S) ? tb.pi[m * S + j] : 0.f;
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                // i < 4: states held by this lane, i >= 4: states held by the partner lane
                const int src = SP == 4 ? i : (i < 4 ? 4 * h + i : 4 * (1 - h) + (i - 4));
                a[nt][i][r] = (m < M && j < S && src < S) ? tb.A[((size_t)m * S + src) * S + j] : 0.f;
            }
        }
    }

    float alpha[NT][4];
    double lp[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        lp[nt] = 0.0;
#pragma unroll
        for (int r = 0; r < 4; ++r) alpha[nt][r] = 0.f;
    }

    // ---- frame loads: two adjacent floats per r
    const float* xbase = X + f0 * D + 2 * g;
    float xc[R][2], xn[R][2];
    auto load = [&](int t, float (&dst)[R][2]) {
        int tc = t < T ? t : T - 1;
        if (tc < 0) tc = 0;
        const float* p = xbase + (int64_t)tc * D;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int d0 = 8 * r + 2 * g;
            dst[r][0] = d0 < D ? p[8 * r] : 0.f;
            dst[r][1] = d0 + 1 < D ? p[8 * r + 1] : 0.f;
        }
    };
    if (Tmax > 0) load(0, xc);

    for (int t = 0; t < Tmax; ++t) {
        if (t + 1 < Tmax) load(t + 1, xn);
        f32x4 acc[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[nt] = c0[nt];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const float x0 = xc[r][0], x1 = xc[r][1];
            const float s0 = x0 * x0, s1 = x1 * x1;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[nt][4 * r + 0], x0, acc[nt], 0, 0, 0);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[nt][4 * r + 1], x1, acc[nt], 0, 0, 0);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[nt][4 * r + 2], s0, acc[nt], 0, 0, 0);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[nt][4 * r + 3], s1, acc[nt], 0, 0, 0);
        }
        const bool live = t < T;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            // shift by the frame's largest log-density over the model's states
            float mx = fmaxf(fmaxf(acc[nt][0], acc[nt][1]), fmaxf(acc[nt][2], acc[nt][3]));
            if (SP == 8) mx = fmaxf(mx, xchg16(mx));
            float b[4], na[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) b[r] = __builtin_amdgcn_exp2f(acc[nt][r] - mx);
            if (t == 0) {
#pragma unroll
                for (int r = 0; r < 4; ++r) na[r] = pi_[nt][r] * b[r];
            } else {
                float pa[4];
                if (SP == 8) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) pa[i] = xchg16(alpha[nt][i]);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float s = 0.f;
#pragma unroll
                    for (int i = 0; i < 4; ++i) s = fmaf(alpha[nt][i], a[nt][i][r], s);
                    if (SP == 8) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) s = fmaf(pa[i], a[nt][4 + i][r], s);
                    }
                    na[r] = s * b[r];
                }
            }
            float c = (na[0] + na[1]) + (na[2] + na[3]);
            if (SP == 8) c += xchg16(c);
            const float ci = __builtin_amdgcn_rcpf(c);
            const double dl = (double)(mx + __builtin_amdgcn_logf(c));
            if (live) {
                lp[nt] += dl;
#pragma unroll
                for (int r = 0; r < 4; ++r) alpha[nt][r] = na[r] * ci;
            }
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            xc[r][0] = xn[r][0];
            xc[r][1] = xn[r][1];
        }
    }

    if (valid) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const bool writer = SP == 4 ? true : (g & 1) == 0;
            if (writer && model[nt] < M) out[u * M + model[nt]] = lp[nt] * kLn2;
        }
    }
}

template <int SP, int NT, int R>
void launch(const float* X, const int64_t* off, int64_t U, int64_t n_frames, int M, int S, int D,
            MfmaTables tb, double* out, hipStream_t st) {
    const int64_t groups = (U + 15) / 16;
    const unsigned grid = (unsigned)((groups + 3) / 4);
    hipLaunchKernelGGL((score_mfma_kernel<SP, NT, R>), dim3(grid), dim3(256), 0, st, X, off, U,
                       n_frames, M, S, D, tb, out);
}

}  // namespace

int srx_mfma_score(srx_context* h, const void* X, const int64_t* off, int64_t U, int64_t n_frames,
                   const srx_models_t* md, double* out, hipStream_t st) {
    const int M = md->n_models, S = md->n_states, D = md->n_features;
    if (S > 8 || D > 40 || n_frames <= 0) return SRX_EUNSUPPORTED;
    const int SP = S <= 4 ? 4 : 8;
    const int NT = (M * SP + 15) / 16;
    if (NT > 3) return SRX_EUNSUPPORTED;
    int R = (D + 7) / 8;
    if (R == 3) R = 4;
    if (R == 1) R = 2;
    const int KQ = 4 * R;
    const size_t n_w = (size_t)NT * 16 * KQ * 4, n_c = (size_t)NT * 16;
    float* base;
    int rc = srx_ws_ensure(h, WS_PARAMS, sizeof(float) * (n_w + n_c + (size_t)M * S * S + (size_t)M * S),
                           (void**)&base);
    if (rc) return rc;
    MfmaTables tb;
    float* W = base;
    float* cst = W + n_w;
    float* A = cst + n_c;
    float* pi = A + (size_t)M * S * S;
    tb.W = W; tb.cst = cst; tb.A = A; tb.pi = pi;
    hipLaunchKernelGGL(mfma_prep_kernel, dim3(8), dim3(256), 0, st, *md, SP, NT, R, W, cst, A, pi);
    const float* Xf = (const float*)X;
#define SRX_CASE(sp, nt, r)                                                        \
    if (SP == sp && NT == nt && R == r) {                                          \
        launch<sp, nt, r>(Xf, off, U, n_frames, M, S, D, tb, out, st);             \
        SRX_HIP(hipGetLastError());                                                \
        return SRX_OK;                                                             \
    }
    SRX_CASE(8, 3, 5) SRX_CASE(8, 2, 5) SRX_CASE(8, 1, 5)
    SRX_CASE(8, 3, 4) SRX_CASE(8, 2, 4) SRX_CASE(8, 1, 4)
    SRX_CASE(8, 3, 2) SRX_CASE(8, 2, 2) SRX_CASE(8, 1, 2)
    SRX_CASE(4, 3, 5) SRX_CASE(4, 2, 5) SRX_CASE(4, 1, 5)
    SRX_CASE(4, 3, 4) SRX_CASE(4, 2, 4) SRX_CASE(4, 1, 4)
    SRX_CASE(4, 3, 2) SRX_CASE(4, 2, 2) SRX_CASE(4, 1, 2)
#undef SRX_CASE
    return SRX_EUNSUPPORTED;
}
