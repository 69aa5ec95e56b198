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


// ---------------------------------------------------------------------------
// Split-bf16 variant.  The emission contraction runs on
// v_mfma_f32_16x16x32_bf16 (16x the f32 matrix rate) with every f32 operand
// written as hi + lo bf16 halves and three of the four partial products kept:
//     x w  ~=  xh wh + xl wh + xh wl          (xl wl < 2^-16 |x w| is dropped)
// hi(x) is the truncated upper 16 bits (exact remainder), lo is rounded.  The
// relative error per product is <= ~3e-5 and averages out over the 2D terms;
// scores agree with the exact-f32 kernel to ~1e-6 relative.
// k-steps (8 bf16 per lane, the same order in the weight operand):
//   HL(r)    [x0h x1h s0h s1h | x0l x1l s0l s1l] . [wh.. | wh..]   r = 0..R-1
//   WL(r,r') [x0h x1h s0h s1h (r) | same (r')]   . [wl(r) | wl(r')]
// with x0,x1 the lane's two features of block r and s = x^2 (formed in f32).
// The 8x8 / 4x4 transition products also go to the matrix core (exact f32
// 16x16x4, block-diagonal per tile): accumulator register q of lane group g is
// tile row 4g+q, which is the k index that group supplies in k-step q, so
// alpha feeds the next step without lane movement; only the per-model max and
// sum cross lanes (one exchange each for 8 states, none for 4).
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct SplitTables {
    const unsigned* Wp;   // [NT*16 rows][KS k-steps][4 lane groups][4 dwords] packed bf16 pairs
    const float* cst;     // [NT*16]
    const float* A;       // [M][S][S]
    const float* pi;      // [M][S]
};

__device__ __host__ inline unsigned bf16_rne(float f) {
    unsigned u = __builtin_bit_cast(unsigned, f);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return u >> 16;
}

// One thread per (row, k-step, lane group): the 8 weights that face the 8
// feature slots of that k-step.
__global__ void split_prep_kernel(srx_models_t md, int SP, int NT, int R, unsigned* Wp, float* cst,
                                  float* A, float* pi) {
    const int M = md.n_models, S = md.n_states, D = md.n_features;
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    const int nth = gridDim.x * blockDim.x;
    const int KS = R + (R + 1) / 2;
    for (int i = tid; i < NT * 16 * KS * 4; i += nth) {
        const int g = i & 3, ks = (i >> 2) % KS, row = (i >> 2) / KS;
        const int m = row / SP, s = row % SP;
        unsigned short v[8];
        for (int j = 0; j < 8; ++j) {
            int r, lo_part;
            if (ks < R) { r = ks; lo_part = 0; }                        // HL: hi weights twice
            else { r = 2 * (ks - R) + (j >> 2); lo_part = 1; }          // WL: lo weights of r, r'
            const int e = j & 3;                                        // x0 x1 s0 s1
            const int d = 8 * r + 2 * g + (e & 1);
            float wv = 0.f;
            if (m < M && s < S && r < R && d < D) {
                const size_t k = ((size_t)m * S + s) * D + d;
                const double var = fmax(md.covars[k], 2.2250738585072014e-308);
                wv = (float)((e >> 1) ? (-0.5 / var) * 1.4426950408889634
                                      : (md.means[k] / var) * 1.4426950408889634);
            }
            const unsigned hi = bf16_rne(wv);
            const float rem = wv - __builtin_bit_cast(float, hi << 16);
            v[j] = (unsigned short)(lo_part ? bf16_rne(rem) : hi);
        }
        for (int q = 0; q < 4; ++q) Wp[(size_t)i * 4 + q] = (unsigned)v[2 * q] | ((unsigned)v[2 * q + 1] << 16);
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

template <int SP, int NT, int R>
__global__ __launch_bounds__(256, 2) void score_split_kernel(
    const float* __restrict__ X, const int64_t* __restrict__ off, int64_t U, int64_t n_frames,
    int M, int S, int D, SplitTables tb, double* __restrict__ out) {
    constexpr int RP = (R + 1) / 2;     // WL k-steps
    constexpr int KS = R + RP;
    const int lane = threadIdx.x & 63, g = lane >> 4, ul = lane & 15;
    const int64_t group = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (group * 16 >= U) return;
    const int64_t u = group * 16 + ul;
    const bool valid = u < U;
    const int64_t uc = valid ? u : U - 1;
    int64_t f0 = off[uc];
    const int T = valid ? (int)(off[uc + 1] - f0) : 0;
    if (f0 > n_frames - 1) f0 = n_frames - 1;
    int Tmax = T;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) Tmax = max(Tmax, __shfl_xor(Tmax, o, 64));

    // ---- loop-invariant registers: packed weights, constants, transition operands
    u32x4 w[NT][KS];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
            w[nt][ks] = *(const u32x4*)(tb.Wp + ((((size_t)(nt * 16 + ul)) * KS + ks) * 4 + g) * 4);
    f32x4 c0[NT];
    float atop[NT][4];
    int model[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        // accumulator rows 4g..4g+3 of the tile; tile row rho = SP*h + s
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            c0[nt][q] = tb.cst[nt * 16 + 4 * g + q];
            // transition operand: out row = ul, source row = 4g+q (k-step q, lane group g)
            const int src = 4 * g + q, dst = ul;
            const int ms = (nt * 16 + src) / SP, md_ = (nt * 16 + dst) / SP;
            const int ss = src % SP, sd = dst % SP;
            atop[nt][q] = (ms == md_ && ms < M && ss < S && sd < S)
                              ? tb.A[((size_t)ms * S + ss) * S + sd] : 0.f;
        }
        model[nt] = (nt * 16 + 4 * g) / SP;
    }

    f32x4 alpha[NT];
    double lp[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        lp[nt] = 0.0;
        alpha[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    // ---- frame loads: two adjacent floats per r, kPrefetch steps in flight
    // (one step is 2.5 KB per wave; HBM latency needs several of them queued)
    constexpr int PF = 4;
    const float* xbase = X + f0 * D + 2 * g;
    float xq[PF][R][2];
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
#pragma unroll
    for (int k = 0; k < PF; ++k)
        if (k < Tmax) load(k, xq[k]);

    auto step = [&](int t, const float (&xc)[R][2]) {
        // ---- split the frame: packed hi / lo halves of x and x^2
        unsigned pxh[R], psh[R], pxl[R], psl[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const float x0 = xc[r][0], x1 = xc[r][1];
            const float s0 = x0 * x0, s1 = x1 * x1;
            const unsigned bx0 = __float_as_uint(x0), bx1 = __float_as_uint(x1);
            const unsigned bs0 = __float_as_uint(s0), bs1 = __float_as_uint(s1);
            pxh[r] = __builtin_amdgcn_perm(bx1, bx0, 0x07060302u);
            psh[r] = __builtin_amdgcn_perm(bs1, bs0, 0x07060302u);
            const bf16x2 xl = {(__bf16)(x0 - __uint_as_float(bx0 & 0xffff0000u)),
                               (__bf16)(x1 - __uint_as_float(bx1 & 0xffff0000u))};
            const bf16x2 sl = {(__bf16)(s0 - __uint_as_float(bs0 & 0xffff0000u)),
                               (__bf16)(s1 - __uint_as_float(bs1 & 0xffff0000u))};
            pxl[r] = __builtin_bit_cast(unsigned, xl);
            psl[r] = __builtin_bit_cast(unsigned, sl);
        }
        f32x4 acc[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[nt] = c0[nt];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            u32x4 f;
            if (ks < R) {
                f = u32x4{pxh[ks], psh[ks], pxl[ks], psl[ks]};
            } else {
                const int r0 = 2 * (ks - R), r1 = r0 + 1 < R ? r0 + 1 : r0;
                f = u32x4{pxh[r0], psh[r0], pxh[r1], psh[r1]};   // a padded r' meets zero weights
            }
            const bf16x8 fb = __builtin_bit_cast(bf16x8, f);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                    __builtin_bit_cast(bf16x8, w[nt][ks]), fb, acc[nt], 0, 0, 0);
        }
        const bool live = t < T;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            float mx = fmaxf(fmaxf(acc[nt][0], acc[nt][1]), fmaxf(acc[nt][2], acc[nt][3]));
            if (SP == 8) mx = fmaxf(mx, xchg16(mx));
            f32x4 b, na;
#pragma unroll
            for (int r = 0; r < 4; ++r) b[r] = __builtin_amdgcn_exp2f(acc[nt][r] - mx);
            if (t == 0) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = nt * 16 + 4 * g + r, m = row / SP, s = row % SP;
                    na[r] = ((m < M && s < S) ? tb.pi[m * S + s] : 0.f) * b[r];
                }
            } else {
                f32x4 p = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    p = __builtin_amdgcn_mfma_f32_16x16x4f32(atop[nt][q], alpha[nt][q], p, 0, 0, 0);
                na = p * b;
            }
            float c = (na[0] + na[1]) + (na[2] + na[3]);
            if (SP == 8) c += xchg16(c);
            const float ci = __builtin_amdgcn_rcpf(c);
            const double dl = (double)(mx + __builtin_amdgcn_logf(c));
            if (live) {
                lp[nt] += dl;
                // alpha is the next step's matrix operand for EVERY row of this
                // utterance's column: a dead chain (c == 0: padding rows of the
                // tile, or a sequence the model cannot emit) must feed zeros,
                // not NaN, or it would poison the other model of the tile.  Its
                // own log-probability stays -inf / NaN as in the reference.
                alpha[nt] = c > 0.f ? na * ci : f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
    };

    for (int t0 = 0; t0 < Tmax; t0 += PF) {
#pragma unroll
        for (int k = 0; k < PF; ++k) {
            if (t0 + k < Tmax) {
                step(t0 + k, xq[k]);
                if (t0 + k + PF < Tmax) load(t0 + k + PF, xq[k]);
            }
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
void launch_split(const float* X, const int64_t* off, int64_t U, int64_t n_frames, int M, int S,
                  int D, SplitTables tb, double* out, hipStream_t st) {
    const int64_t groups = (U + 15) / 16;
    const unsigned grid = (unsigned)((groups + 3) / 4);
    hipLaunchKernelGGL((score_split_kernel<SP, NT, R>), dim3(grid), dim3(256), 0, st, X, off, U,
                       n_frames, M, S, D, tb, out);
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
    if (!h->exact_f32) {
        const int KS = R + (R + 1) / 2;
        const size_t n_wp = (size_t)NT * 16 * KS * 4 * 4, n_c = (size_t)NT * 16;
        unsigned* base;
        int rc = srx_ws_ensure(h, WS_PARAMS,
                               4 * (n_wp + n_c + (size_t)M * S * S + (size_t)M * S), (void**)&base);
        if (rc) return rc;
        SplitTables tb;
        float* cst = (float*)(base + n_wp);
        float* A = cst + n_c;
        float* pi = A + (size_t)M * S * S;
        tb.Wp = base; tb.cst = cst; tb.A = A; tb.pi = pi;
        hipLaunchKernelGGL(split_prep_kernel, dim3(8), dim3(256), 0, st, *md, SP, NT, R, base, cst,
                           A, pi);
        const float* Xf = (const float*)X;
#define SRX_CASE(sp, nt, r)                                                        \
    if (SP == sp && NT == nt && R == r) {                                          \
        launch_split<sp, nt, r>(Xf, off, U, n_frames, M, S, D, tb, out, st);       \
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
