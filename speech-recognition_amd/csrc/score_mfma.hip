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
#include "score_split.h"

#include <stdlib.h>


namespace {


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
    const float* xbase = X + f0 * D;
    float xc[R][2], xn[R][2];
    auto load = [&](int t, float (&dst)[R][2]) {
        int tc = t < T ? t : T - 1;
        if (tc < 0) tc = 0;
        const float* p = xbase + (int64_t)tc * D;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            // features past D meet zero weights: read a valid neighbour instead of
            // predicating the load (a predicated load costs a branch and a full wait)
            dst[r][0] = p[min(8 * r + 2 * g, D - 1)];
            dst[r][1] = p[min(8 * r + 2 * g + 1, D - 1)];
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
// Stand-alone emission log-densities (all states of all models, natural log),
// float32: frames are the MFMA columns here -- a wave takes 16 CONSECUTIVE frames
// (one contiguous 16 D-float piece of X), NTC tiles of 16 state columns, and
// writes out[frame][column].  Same split-bf16 contraction as the scoring kernel.
template <int NTC, int RQ, int H>
__global__ __launch_bounds__(256) void emission_split_kernel(
    const float* __restrict__ X, int64_t n_frames, int C, int D, int NT, SplitTables tb,
    float* __restrict__ out) {
    const int lane = threadIdx.x & 63, g = lane >> 4, ul = lane & 15;
    // (banks of more than NTC tiles: the slices of a group of 16 frames run in neighbouring waves,
    // the slice index fastest, so that the frames come from memory once)
    const int chunks = tb.chunks;
    const int64_t wg = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int nt0 = (chunks == 1 ? 0 : (int)(wg % chunks)) * NTC;
    // (the feature axis is cut as in the scoring kernel: RQ blocks of 16 and, if H, a last
    // block of 8 whose operand carries hi and lo halves -- a 39-feature row is read in 160
    // bytes and takes 8 matrix instructions per tile instead of 9)
    constexpr int QT = RQ + H;
    struct item_t {
        u32x4 v[RQ ? RQ : 1];
        u32x2 h;
    };
    u32x4 wh[NTC][QT], wl[NTC][QT];
    f32x4 c0[NTC];
#pragma unroll
    for (int nt = 0; nt < NTC; ++nt) {
        const bool on = nt0 + nt < NT;
#pragma unroll
        for (int q = 0; q < QT; ++q) {
            const unsigned* p = tb.Wp + ((((size_t)((on ? nt0 + nt : 0) * 16 + ul)) * QT + q) * 2 * 4 + g) * 4;
            wh[nt][q] = on ? *(const u32x4*)p : u32x4{0u, 0u, 0u, 0u};
            wl[nt][q] = on ? *(const u32x4*)(p + 16) : u32x4{0u, 0u, 0u, 0u};
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) c0[nt][r] = on ? tb.cst[(nt0 + nt) * 16 + 4 * g + r] : 0.f;
    }
    const int64_t n_groups = (n_frames + 15) / 16;
    const int64_t n_waves = (int64_t)gridDim.x * (blockDim.x >> 6) / chunks;
    const int64_t wid = chunks == 1 ? wg : wg / chunks;
    const unsigned lane_off = (unsigned)ul * (unsigned)D * 4u + 16u * g;
    const unsigned half_off = (unsigned)ul * (unsigned)D * 4u + 64u * RQ + 8u * g;
    auto load = [&](int64_t grp, item_t& dst) {
        // descriptor over [first frame of the group, end of X): frames past the
        // end read zeros
        const int64_t f0 = grp < n_groups ? grp * 16 : 0;
        const int64_t rest = grp < n_groups ? (n_frames - f0) * (int64_t)D * 4 : 0;
        const uintptr_t a = (uintptr_t)(X + f0 * D);
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a);
        const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
        const unsigned n = (unsigned)(rest < 0xFFFFFF00ll ? rest : 0xFFFFFF00ll);
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(((uintptr_t)hi << 32) | lo), 0, __builtin_amdgcn_readfirstlane(n), 0x00020000);
#pragma unroll
        for (int q = 0; q < RQ; ++q)
            dst.v[q] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane_off + 64u * q, 0, 0);
        if constexpr (H != 0) dst.h = __builtin_amdgcn_raw_buffer_load_b64(rsrc, half_off, 0, 0);
    };
    item_t xa, xb;
    load(wid, xa);
    for (int64_t grp = wid; grp < n_groups; grp += n_waves) {
        load(grp + n_waves, xb);
        f32x4 acc[NTC];
#pragma unroll
        for (int nt = 0; nt < NTC; ++nt) acc[nt] = c0[nt];
#pragma unroll
        for (int q = 0; q < RQ; ++q) {
            bf16x8 bh, bl;
            split_operand(xa.v[q], bh, bl);
#pragma unroll
            for (int nt = 0; nt < NTC; ++nt)
                acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                    __builtin_bit_cast(bf16x8, wh[nt][q]), bh, acc[nt], 0, 0, 0);
#pragma unroll
            for (int nt = 0; nt < NTC; ++nt)
                acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                    __builtin_bit_cast(bf16x8, wh[nt][q]), bl, acc[nt], 0, 0, 0);
#pragma unroll
            for (int nt = 0; nt < NTC; ++nt)
                acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                    __builtin_bit_cast(bf16x8, wl[nt][q]), bh, acc[nt], 0, 0, 0);
        }
        if constexpr (H != 0) {
            const bf16x8 bb = split_half_operand(xa.h);
#pragma unroll
            for (int nt = 0; nt < NTC; ++nt)
                acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                    __builtin_bit_cast(bf16x8, wh[nt][RQ]), bb, acc[nt], 0, 0, 0);
#pragma unroll
            for (int nt = 0; nt < NTC; ++nt)
                acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                    __builtin_bit_cast(bf16x8, wl[nt][RQ]), bb, acc[nt], 0, 0, 0);
        }
        const int64_t f = grp * 16 + ul;
        if (f < n_frames) {
            float* o = out + f * C;
#pragma unroll
            for (int nt = 0; nt < NTC; ++nt) {
                const int col = (nt0 + nt) * 16 + 4 * g;
                const f32x4 v = acc[nt] * (float)kLn2;
                if ((C & 3) == 0 && col + 3 < C) {
                    *(f32x4*)(o + col) = v;
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (col + r < C) o[col + r] = v[r];
                }
            }
        }
        xa = xb;
    }
}

// Emission log-densities of each utterance under ITS model only (Viterbi's input):
// out[frame][state], natural log.  The table holds one 16-row tile per model; a
// wave walks utterances, 16 consecutive frames per step.  The feature axis is cut as in
// the scoring kernel (RQ blocks of 16 features and, if H, a last block of 8 whose operand
// carries hi and lo halves), and the three kinds of partial product have an accumulator
// each -- term for term the arithmetic of the fused Viterbi kernel (viterbi_fused.hip), so
// that the two paths decode to the same states and log-probabilities bit for bit.
template <int RQ, int H>
__global__ __launch_bounds__(256) void emission_utt_split_kernel(
    const float* __restrict__ X, const int64_t* __restrict__ off, int64_t U, int64_t n_frames,
    const int32_t* __restrict__ utt_model, int S, int D, SplitTables tb, float* __restrict__ out) {
    constexpr int QT = RQ + H;
    struct item_t {
        u32x4 v[RQ ? RQ : 1];
        u32x2 h;
    };
    const int lane = threadIdx.x & 63, g = lane >> 4, ul = lane & 15;
    const int64_t n_waves = (int64_t)gridDim.x * (blockDim.x >> 6);
    const int64_t wid = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const unsigned lane_off = (unsigned)ul * (unsigned)D * 4u + 16u * g;
    const unsigned half_off = (unsigned)ul * (unsigned)D * 4u + 64u * RQ + 8u * g;
    int cur = -1;
    u32x4 wh[QT], wl[QT];
    f32x4 c0 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < QT; ++q) wh[q] = wl[q] = u32x4{0u, 0u, 0u, 0u};
    for (int64_t u = wid; u < U; u += n_waves) {
        const int m = utt_model ? utt_model[u] : 0;
        const int64_t f0 = off[u];
        const int64_t T = off[u + 1] - f0;
        if (T <= 0) continue;
        if (m != cur) {
            cur = m;
#pragma unroll
            for (int q = 0; q < QT; ++q) {
                const unsigned* p = tb.Wp + ((((size_t)(m * 16 + ul)) * QT + q) * 2 * 4 + g) * 4;
                wh[q] = *(const u32x4*)p;
                wl[q] = *(const u32x4*)(p + 16);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) c0[r] = tb.cst[m * 16 + 4 * g + r];
        }
        const int64_t n_groups = (T + 15) / 16;
        auto load = [&](int64_t k, item_t& dst) {
            // descriptor over [first frame of the group, end of the utterance)
            const int64_t fk = k < n_groups ? f0 + k * 16 : f0;
            const int64_t rest = k < n_groups ? (f0 + T - fk) * (int64_t)D * 4 : 0;
            const uintptr_t a = (uintptr_t)(X + fk * D);
            const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a);
            const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
            const unsigned n = (unsigned)(rest < 0xFFFFFF00ll ? rest : 0xFFFFFF00ll);
            const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
                (void*)(((uintptr_t)hi << 32) | lo), 0, __builtin_amdgcn_readfirstlane(n), 0x00020000);
#pragma unroll
            for (int q = 0; q < RQ; ++q)
                dst.v[q] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane_off + 64u * q, 0, 0);
            if constexpr (H != 0) dst.h = __builtin_amdgcn_raw_buffer_load_b64(rsrc, half_off, 0, 0);
        };
        item_t xa, xb;
        load(0, xa);
        for (int64_t k = 0; k < n_groups; ++k) {
            load(k + 1, xb);
            f32x4 acc = c0, acc1 = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int q = 0; q < RQ; ++q) {
                bf16x8 bh, bl;
                split_operand(xa.v[q], bh, bl);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wh[q]), bh, acc, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wh[q]), bl, acc1, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wl[q]), bh, acc2, 0, 0, 0);
            }
            if constexpr (H != 0) {
                const bf16x8 bb = split_half_operand(xa.h);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wh[RQ]), bb, acc1, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wl[RQ]), bb, acc2, 0, 0, 0);
            }
            acc += acc1 + acc2;
            const int64_t t = k * 16 + ul;
            if (t < T) {
                float* o = out + (f0 + t) * S;
                const int col = 4 * g;
                const f32x4 v = acc * (float)kLn2;
                if ((S & 3) == 0 && col + 3 < S) {
                    *(f32x4*)(o + col) = v;
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (col + r < S) o[col + r] = v[r];
                }
            }
            xa = xb;
        }
    }
}

// The same output with the contraction in plain float32 on v_mfma_f32_16x16x4_f32
// (posteriors need the log-densities to float32 rounding, not to the 1e-4 of the
// split-bf16 product).  Weight table per model: one 16-row tile,
// W[row][q][x | x^2][g][i] = coefficient of feature 16 q + 4 g + i, natural-log units;
// lane (g, ul) keeps row ul's coefficients for its own k slots in registers.
// With H the last 8 features form a half block of their own, [x | x^2][g][i < 2] = feature
// 16 RQ + 2 g + i, behind the RQ full blocks (the scoring kernel's cut of the feature axis: a
// 39-feature row is read in 160 bytes and takes 20 matrix instructions instead of 24).
__global__ void emission_f32_prep_kernel(srx_models_t md, int RQ, int H, float* W, float* cst) {
    const int M = md.n_models, S = md.n_states, D = md.n_features;
    const int tid = blockIdx.x * blockDim.x + threadIdx.x, nth = gridDim.x * blockDim.x;
    const int per_row = RQ * 32 + H * 16;
    for (int i = tid; i < M * 16 * per_row; i += nth) {
        const int row = i / per_row, w = i % per_row;
        const int hw = w - RQ * 32;                  // >= 0: inside the half block
        const int q = w >> 5;
        const int e = hw >= 0 ? hw >> 3 : (w >> 4) & 1;
        const int d = hw >= 0 ? 16 * RQ + (hw & 7) : 16 * q + (w & 15);
        const int m = row >> 4, s_ = row & 15;
        float v = 0.f;
        if (s_ < S && d < D) {
            const size_t k = ((size_t)m * S + s_) * D + d;
            const double var = fmax(md.covars[k], 2.2250738585072014e-308);
            v = (float)(e ? -0.5 / var : md.means[k] / var);
        }
        W[i] = v;
    }
    // constants: one wave per row, lanes over features (a serial sum of D float64
    // logarithms per row would be most of this kernel's time)
    const int wave = tid >> 6, lane = tid & 63, n_wave = nth >> 6;
    for (int row = wave; row < M * 16; row += n_wave) {
        const int m = row >> 4, s_ = row & 15;
        double acc = 0.0;
        if (s_ < S)
            for (int d = lane; d < D; d += 64) {
                const size_t k = ((size_t)m * S + s_) * D + d;
                const double var = fmax(md.covars[k], 2.2250738585072014e-308);
                acc += log(var) + md.means[k] * md.means[k] / var;
            }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
        if (lane == 0) cst[row] = s_ < S ? (float)(-0.5 * (acc + D * 1.8378770664093453)) : 0.f;
    }
}

template <int RQ, int H>
__global__ __launch_bounds__(256) void emission_utt_f32_kernel(
    const float* __restrict__ X, const int64_t* __restrict__ off, int64_t U, int64_t n_frames,
    const int32_t* __restrict__ utt_model, int S, int D, const float* __restrict__ W,
    const float* __restrict__ cst, float* __restrict__ out) {
    const int lane = threadIdx.x & 63, g = lane >> 4, ul = lane & 15;
    const int64_t n_waves = (int64_t)gridDim.x * (blockDim.x >> 6);
    const int64_t wid = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const unsigned lane_off = (unsigned)ul * (unsigned)D * 4u + 16u * g;
    const unsigned half_off = (unsigned)ul * (unsigned)D * 4u + 64u * RQ + 8u * g;
    constexpr int PER_ROW = RQ * 32 + H * 16;
    int cur = -1;
    f32x4 w1[RQ ? RQ : 1], w2[RQ ? RQ : 1];
    float wh1[2] = {0.f, 0.f}, wh2[2] = {0.f, 0.f};
    f32x4 c0 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < RQ; ++q) w1[q] = w2[q] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int64_t u = wid; u < U; u += n_waves) {
        const int m = utt_model ? utt_model[u] : 0;
        const int64_t f0 = off[u];
        const int64_t T = off[u + 1] - f0;
        if (T <= 0) continue;
        if (m != cur) {
            cur = m;
#pragma unroll
            for (int q = 0; q < RQ; ++q) {
                const float* p = W + ((size_t)(m * 16 + ul)) * PER_ROW + q * 32 + 4 * g;
                w1[q] = *(const f32x4*)p;
                w2[q] = *(const f32x4*)(p + 16);
            }
            if constexpr (H != 0) {
                const float* p = W + ((size_t)(m * 16 + ul)) * PER_ROW + RQ * 32 + 2 * g;
                wh1[0] = p[0]; wh1[1] = p[1];
                wh2[0] = p[8]; wh2[1] = p[9];
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) c0[r] = cst[m * 16 + 4 * g + r];
        }
        const int64_t n_groups = (T + 15) / 16;
        auto load = [&](int64_t k, u32x4 (&dst)[RQ ? RQ : 1], u32x2& dh) {
            // descriptor over [first frame of the group, end of the utterance)
            const int64_t fk = k < n_groups ? f0 + k * 16 : f0;
            const int64_t rest = k < n_groups ? (f0 + T - fk) * (int64_t)D * 4 : 0;
            const uintptr_t a = (uintptr_t)(X + fk * D);
            const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a);
            const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
            const unsigned n = (unsigned)(rest < 0xFFFFFF00ll ? rest : 0xFFFFFF00ll);
            const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
                (void*)(((uintptr_t)hi << 32) | lo), 0, __builtin_amdgcn_readfirstlane(n), 0x00020000);
#pragma unroll
            for (int q = 0; q < RQ; ++q)
                dst[q] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane_off + 64u * q, 0, 0);
            if constexpr (H != 0) dh = __builtin_amdgcn_raw_buffer_load_b64(rsrc, half_off, 0, 0);
        };
        u32x4 xa[RQ ? RQ : 1], xb[RQ ? RQ : 1];
        u32x2 ha = {0u, 0u}, hb = {0u, 0u};
        load(0, xa, ha);
        for (int64_t k = 0; k < n_groups; ++k) {
            load(k + 1, xb, hb);
            f32x4 a1 = c0, a2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int q = 0; q < RQ; ++q) {
                const f32x4 x = __builtin_bit_cast(f32x4, xa[q]);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w1[q][i], x[i], a1, 0, 0, 0);
                    a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(w2[q][i], x[i] * x[i], a2, 0, 0, 0);
                }
            }
            if constexpr (H != 0) {
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const float x = __uint_as_float(ha[i]);
                    a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wh1[i], x, a1, 0, 0, 0);
                    a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(wh2[i], x * x, a2, 0, 0, 0);
                }
            }
            const int64_t t = k * 16 + ul;
            if (t < T) {
                float* o = out + (f0 + t) * S;
                const int col = 4 * g;
                const f32x4 v = a1 + a2;
                if ((S & 3) == 0 && col + 3 < S) {
                    *(f32x4*)(o + col) = v;
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (col + r < S) o[col + r] = v[r];
                }
            }
#pragma unroll
            for (int q = 0; q < RQ; ++q) xa[q] = xb[q];
            ha = hb;
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

int srx_mfma_emission(srx_context* h, const void* X, int64_t n_frames, const srx_models_t* md,
                      float* out, hipStream_t st) {
    const int M = md->n_models, S = md->n_states, D = md->n_features;
    const int64_t C64 = (int64_t)M * S;
    if (h->exact_f32 || D > 48 || n_frames <= 0 || C64 > 4096) return SRX_EUNSUPPORTED;
    const int H = (D % 16 >= 1 && D % 16 <= 8) ? 1 : 0;
    const int C = (int)C64, NT = (C + 15) / 16, RQ = H ? D / 16 : (D + 15) / 16, QT = RQ + H;
    const size_t n_wp = (size_t)NT * 16 * QT * 2 * 4 * 4, n_c = (size_t)NT * 16;
    unsigned* base;
    int rc = srx_ws_ensure(h, WS_PARAMS, 4 * (n_wp + n_c + (size_t)M * S * S + (size_t)M * S),
                           (void**)&base);
    if (rc) return rc;
    SplitTables tb;
    float* cst = (float*)(base + n_wp);
    float* A = cst + n_c;
    float* pi = A + (size_t)M * S * S;
    tb.Wp = base; tb.cst = cst; tb.A = A; tb.pi = pi; tb.order = nullptr;
    // rows of the table = columns of the output: "padded states per model" = S
    hipLaunchKernelGGL(split_prep_kernel, dim3(16), dim3(256), 0, st, *md, S, NT, RQ, H, base, cst, A, pi);
    const int NTC = NT >= 3 ? 3 : NT;
    const int chunks = (NT + NTC - 1) / NTC;
    const int64_t n_groups = (n_frames + 15) / 16;
    int64_t gx = (n_groups + 3) / 4;
    const int64_t cap = (int64_t)h->n_cu * 8;
    if (gx > cap) gx = cap;
    const dim3 grid((unsigned)(gx * chunks));
    tb.chunks = chunks;
    const float* Xf = (const float*)X;
#define SRX_CASE(ntc, rq, hh)                                                                \
    if (NTC == ntc && RQ == rq && H == hh) {                                                 \
        hipLaunchKernelGGL((emission_split_kernel<ntc, rq, hh>), grid, dim3(256), 0, st, Xf, \
                           n_frames, C, D, NT, tb, out);                                     \
        SRX_HIP(hipGetLastError());                                                          \
        return SRX_OK;                                                                       \
    }
#define SRX_CASES(ntc)                                                                       \
    SRX_CASE(ntc, 0, 1) SRX_CASE(ntc, 1, 0) SRX_CASE(ntc, 1, 1) SRX_CASE(ntc, 2, 0)          \
    SRX_CASE(ntc, 2, 1) SRX_CASE(ntc, 3, 0)
    SRX_CASES(3) SRX_CASES(2) SRX_CASES(1)
#undef SRX_CASES
#undef SRX_CASE
    return SRX_EUNSUPPORTED;
}

int srx_mfma_emission_utt(srx_context* h, const void* X, const int64_t* off, int64_t U,
                          int64_t n_frames, const int32_t* utt_model, const srx_models_t* md,
                          float* out, hipStream_t st) {
    const int M = md->n_models, S = md->n_states, D = md->n_features;
    if (h->exact_f32 || S > 16 || D > 48 || M > 4096 || n_frames <= 0 || U <= 0) return SRX_EUNSUPPORTED;
    const int H = (D % 16 >= 1 && D % 16 <= 8) ? 1 : 0;
    const int NT = M, RQ = H ? D / 16 : (D + 15) / 16, QT = RQ + H;
    const size_t n_wp = (size_t)NT * 16 * QT * 2 * 4 * 4, n_c = (size_t)NT * 16;
    unsigned* base;
    int rc = srx_ws_ensure(h, WS_MISC, 4 * (n_wp + n_c + (size_t)M * S * S + (size_t)M * S),
                           (void**)&base);
    if (rc) return rc;
    SplitTables tb;
    float* cst = (float*)(base + n_wp);
    float* A = cst + n_c;
    float* pi = A + (size_t)M * S * S;
    tb.Wp = base; tb.cst = cst; tb.A = A; tb.pi = pi; tb.order = nullptr;
    // one 16-row tile per model
    hipLaunchKernelGGL(split_prep_kernel, dim3(16), dim3(256), 0, st, *md, 16, NT, RQ, H, base, cst, A, pi);
    int64_t gx = (U + 3) / 4;
    const int64_t cap = (int64_t)h->n_cu * 8;
    if (gx > cap) gx = cap;
    const float* Xf = (const float*)X;
#define SRX_CASE(rq, hh)                                                                           \
    if (RQ == rq && H == hh) {                                                                     \
        hipLaunchKernelGGL((emission_utt_split_kernel<rq, hh>), dim3((unsigned)gx), dim3(256), 0,  \
                           st, Xf, off, U, n_frames, utt_model, S, D, tb, out);                    \
        SRX_HIP(hipGetLastError());                                                                \
        return SRX_OK;                                                                             \
    }
    SRX_CASE(0, 1) SRX_CASE(1, 0) SRX_CASE(1, 1) SRX_CASE(2, 0) SRX_CASE(2, 1) SRX_CASE(3, 0)
#undef SRX_CASE
    return SRX_EUNSUPPORTED;
}

int srx_mfma_emission_utt_f32(srx_context* h, const void* X, const int64_t* off, int64_t U,
                              int64_t n_frames, const int32_t* utt_model, const srx_models_t* md,
                              float* out, hipStream_t st) {
    const int M = md->n_models, S = md->n_states, D = md->n_features;
    if (S > 16 || D > 48 || M > 4096 || n_frames <= 0 || U <= 0) return SRX_EUNSUPPORTED;
    const int H = (D % 16 >= 1 && D % 16 <= 8) ? 1 : 0;
    const int RQ = H ? D / 16 : (D + 15) / 16;
    const size_t n_w = (size_t)M * 16 * (RQ * 32 + H * 16), n_c = (size_t)M * 16;
    float* base;
    int rc = srx_ws_ensure(h, WS_MISC, 4 * (n_w + n_c), (void**)&base);
    if (rc) return rc;
    float* cst = base + n_w;
    hipLaunchKernelGGL(emission_f32_prep_kernel, dim3(16), dim3(256), 0, st, *md, RQ, H, base, cst);
    int64_t gx = (U + 3) / 4;
    const int64_t cap = (int64_t)h->n_cu * 8;
    if (gx > cap) gx = cap;
    const float* Xf = (const float*)X;
#define SRX_CASE(rq, hh)                                                                        \
    if (RQ == rq && H == hh) {                                                                  \
        hipLaunchKernelGGL((emission_utt_f32_kernel<rq, hh>), dim3((unsigned)gx), dim3(256), 0, \
                           st, Xf, off, U, n_frames, utt_model, S, D, base, cst, out);          \
        SRX_HIP(hipGetLastError());                                                             \
        return SRX_OK;                                                                          \
    }
    SRX_CASE(0, 1) SRX_CASE(1, 0) SRX_CASE(1, 1) SRX_CASE(2, 0) SRX_CASE(2, 1) SRX_CASE(3, 0)
#undef SRX_CASE
    return SRX_EUNSUPPORTED;
}

// Batches of 16 consecutive utterances whose frames span 4 GiB or more are out of the reach
// of the split kernel's one descriptor per batch (32-bit offsets); it leaves them alone and
// this kernel scores exactly those: one wave per (utterance, model) pair, 64-bit addresses,
// float64 arithmetic, lanes = 4 feature groups x 16 states.  Correct for any length; a
// wave walks its utterance frame by frame, so it is no faster than the recursion is long.
namespace {
__global__ __launch_bounds__(64) void score_far_kernel(const float* __restrict__ X,
                                                       const int64_t* __restrict__ off, int64_t U,
                                                       srx_models_t md, double* __restrict__ out) {
    const int M = md.n_models, S = md.n_states, D = md.n_features;
    const int64_t u_first = (int64_t)blockIdx.x * 16;
    const int64_t u_last = u_first + 15 < U - 1 ? u_first + 15 : U - 1;
    if ((off[u_last + 1] - off[u_first]) * D * 4 < 0xFFFFFF00ll) return;
    const int64_t u = u_first + blockIdx.y;
    if (u >= U) return;
    const int lane = threadIdx.x, g = lane >> 4, s = lane & 15;
    const int64_t f0 = off[u], T = off[u + 1] - f0;
    for (int m = blockIdx.z; m < M; m += gridDim.z) {
        double mu[12], hv[12], Acol[16], cpart = 0.0;
#pragma unroll
        for (int j = 0; j < 12; ++j) {
            const int d = g + 4 * j;
            mu[j] = hv[j] = 0.0;
            if (s < S && d < D) {
                const size_t k = ((size_t)m * S + s) * D + d;
                const double var = fmax(md.covars[k], 2.2250738585072014e-308);
                mu[j] = md.means[k];
                hv[j] = 0.5 / var;
                cpart += log(var);
            }
        }
        cpart += __shfl_xor(cpart, 16, 64);
        cpart += __shfl_xor(cpart, 32, 64);
        const double cst = -0.5 * (cpart + D * 1.8378770664093453);
#pragma unroll
        for (int j = 0; j < 16; ++j)
            Acol[j] = (j < S && s < S) ? md.transmat[((size_t)m * S + j) * S + s] : 0.0;
        const double pi = s < S ? md.startprob[(size_t)m * S + s] : 0.0;
        double alpha = 0.0, lp = 0.0;
        for (int64_t t = 0; t < T; ++t) {
            const float* x = X + (f0 + t) * D;
            double acc = 0.0;
#pragma unroll
            for (int j = 0; j < 12; ++j) {
                const int d = g + 4 * j;
                const double dx = (d < D ? (double)x[d] : 0.0) - mu[j];
                acc += dx * dx * hv[j];
            }
            acc += __shfl_xor(acc, 16, 64);
            acc += __shfl_xor(acc, 32, 64);
            const double ll = s < S ? cst - acc : -INFINITY;
            double mx = ll;
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) mx = fmax(mx, __shfl_xor(mx, o, 64));
            const double b = exp(ll - mx);
            double a;
            if (t == 0) {
                a = pi * b;
            } else {
                a = 0.0;
#pragma unroll
                for (int j = 0; j < 16; ++j) a += __shfl(alpha, (lane & 48) | j, 64) * Acol[j];
                a *= b;
            }
            double c = a;
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) c += __shfl_xor(c, o, 64);
            lp += log(c) + mx;
            alpha = a / c;
        }
        if (lane == 0) out[u * M + m] = lp;
    }
}
}  // namespace

int srx_mfma_score(srx_context* h, const void* X, const int64_t* off, int64_t U, int64_t n_frames,
                   const srx_models_t* md, double* out, hipStream_t st) {
    return srx_mfma_score_tables(h, X, off, U, n_frames, md, out, nullptr, 0, st, nullptr);
}

int srx_mfma_score_tables(srx_context* h, const void* X, const int64_t* off, int64_t U,
                          int64_t n_frames, const srx_models_t* md, double* out, void* tables,
                          size_t tables_bytes, hipStream_t st, const int64_t* order) {
    const int M = md->n_models, S = md->n_states, D = md->n_features;
    // a visiting order puts utterances from anywhere in X into one batch: only the kernel with
    // one descriptor over all of X can follow it, and only while X is within its reach
    if (order && n_frames * (int64_t)D * 4 >= 0xFFFFFF00ll) order = nullptr;
    if (S > 16 || D > 48 || n_frames <= 0) return SRX_EUNSUPPORTED;
    if (tables && h->exact_f32) return SRX_EUNSUPPORTED;
    if (h->exact_f32 && (D > 40 || S > 8)) return SRX_EUNSUPPORTED;
    const int SP = S <= 4 ? 4 : (S <= 8 ? 8 : 16);
    const int NT = (M * SP + 15) / 16;
    int R = (D + 7) / 8;
    if (R == 3) R = 4;
    if (R == 1) R = 2;

    if (!h->exact_f32 && D <= 48 && NT <= 3 * 64) {
        const int H = (D % 16 >= 1 && D % 16 <= 8) ? 1 : 0;
        const int RQ = H ? D / 16 : (D + 15) / 16;
        // banks of more than 3 tiles: slices of the grid take NTC tiles each (X is read
        // once per slice); the tables are padded to whole slices
        const int NTC = NT <= 3 ? NT : (NT == 4 ? 2 : 3);
        const int chunks = (NT + NTC - 1) / NTC;
        const int NTP = chunks * NTC;
        const size_t n_wp = (size_t)NTP * 16 * (RQ + H) * 2 * 4 * 4, n_c = (size_t)NTP * 16;
        unsigned* base;
        const size_t need = 4 * (n_wp + n_c + (size_t)M * S * S + (size_t)M * S);
        if (tables) {
            if (need > tables_bytes) {
                srx_set_error("model bank too large for the caller's table memory: its tables need %zu "
                              "bytes, %zu are there (n_models=%d n_states=%d n_features=%d)",
                              need, tables_bytes, M, S, D);
                return SRX_EINVAL;
            }
            base = (unsigned*)tables;
        } else {
            int rc = srx_ws_ensure(h, WS_PARAMS, need, (void**)&base);
            if (rc) return rc;
        }
        float* cst = (float*)(base + n_wp);
        float* A = cst + n_c;
        float* pi = A + (size_t)M * S * S;
        hipLaunchKernelGGL(split_prep_kernel, dim3(16), dim3(256), 0, st, *md, SP, NTP, RQ, H, base,
                           cst, A, pi);
        const float* Xf = (const float*)X;
        int rc;
        if (SP == 16)
            rc = srx_split_score_sp16(Xf, off, U, n_frames, M, S, D, NTC, chunks, RQ, H, base, cst, A, pi, out, st, order);
        else
            rc = SP == 8 ? srx_split_score_sp8(Xf, off, U, n_frames, M, S, D, NTC, chunks, RQ, H, base, cst, A, pi, out, st, order)
                         : srx_split_score_sp4(Xf, off, U, n_frames, M, S, D, NTC, chunks, RQ, H, base, cst, A, pi, out, st, order);
        if (rc) return rc;
        // only a buffer of 4 GiB or more can hold a batch the split kernel has skipped
        if (n_frames * (int64_t)D * 4 >= 0xFFFFFF00ll) {
            hipLaunchKernelGGL(score_far_kernel, dim3((unsigned)((U + 15) / 16), 16, (unsigned)(M < 4 ? M : 4)),
                               dim3(64), 0, st, Xf, off, U, *md, out);
            SRX_HIP(hipGetLastError());
        }
        return SRX_OK;
    }
    if (NT > 3 || tables) return SRX_EUNSUPPORTED;
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
