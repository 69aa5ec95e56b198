// Float64 scoring on the matrix cores: emission log-densities and the scaled forward
// recursion in one pass over X, every multiply-add of the emission contraction a
// v_mfma_f64_16x16x4_f64.  This is the validation dtype (results agree with the
// reference's float64 arithmetic to rounding) at a rate the shape-generic kernels cannot
// reach: on gfx950 the f64 matrix instruction costs less than twice the f32 one.
//
// Emission as a contraction, per frame x and state column c (log2 units throughout):
//     ll2[c] = cst[c] + sum_d  x_d * (mu/var * log2e) + x_d^2 * (-1/(2 var) * log2e)
// Wave geometry: one wave scores 16 utterances against ONE tile of 16 state columns
// (4 models of <= 4 states, 2 models of <= 8 or 1 model of <= 16), time step by time step; the
// tiles of a bank are the grid's y axis and each re-reads X.  Matrix columns are
// utterances.  The f64 instruction spreads its 16 result rows differently from the f32
// one: register r of lane group g (= lane / 16) is row 4r + g.  State columns are
// therefore placed so that row 4r + g is "logical column" 4g + r = (model 4g/SP, state
// (4g + r) mod SP): a lane then holds four consecutive states of one model; a model of 8
// states sits in lanes l and l ^ 16, one of 16 in all four lane groups of an utterance.  The A operand of
// lane (g, ul) is the weight of row ul, i.e. of logical column 4 (ul & 3) + (ul >> 2).
// The B operand of k-step q comes from lane group g: q = 4r + e is feature
// d = 8r + 2g + (e & 1), x_d for e < 2 and x_d^2 for e >= 2.
#include "srx_internal.h"

#include <math.h>

namespace {

typedef double f64x4 __attribute__((ext_vector_type(4)));
constexpr double kLn2 = 0.6931471805599453;
constexpr double kLog2e = 1.4426950408889634;
constexpr double kNegBig = -1.0e300;

struct F64Tables {
    const double* W;     // [n_tiles*16 logical columns][4R][4 lane groups]
    const double* cst;   // [n_tiles*16]
};

__global__ void f64_prep_kernel(srx_models_t md, int SP, int n_tiles, int R, double* W, double* cst) {
    const int M = md.n_models, S = md.n_states, D = md.n_features;
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    const int nth = gridDim.x * blockDim.x;
    const int KQ = 4 * R;
    for (int i = tid; i < n_tiles * 16 * KQ * 4; i += nth) {
        const int g = i & 3, q = (i >> 2) % KQ, col = (i >> 2) / KQ;
        const int m = col / SP, s = col % SP;
        const int r = q >> 2, e = q & 3;
        const int d = 8 * r + 2 * g + (e & 1);
        double v = 0.0;
        if (m < M && s < S && d < D) {
            const size_t k = ((size_t)m * S + s) * D + d;
            const double var = fmax(md.covars[k], 2.2250738585072014e-308);
            v = (e >> 1) ? (-0.5 / var) * kLog2e : (md.means[k] / var) * kLog2e;
        }
        W[i] = v;
    }
    const int wave = tid >> 6, lane = tid & 63, n_wave = nth >> 6;
    for (int col = wave; col < n_tiles * 16; col += n_wave) {
        const int m = col / SP, s = col % SP;
        const bool on = m < M && s < S;
        double acc = 0.0;
        if (on)
            for (int d = lane; d < D; d += 64) {
                const size_t k = ((size_t)m * S + s) * D + d;
                const double var = fmax(md.covars[k], 2.2250738585072014e-308);
                acc += log(var) + md.means[k] * md.means[k] / var;
            }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
        if (lane == 0) cst[col] = on ? -0.5 * (acc + D * 1.8378770664093453) * kLog2e : kNegBig;
    }
}

template <int SP, int R>
__global__ __launch_bounds__(256) void score_f64_kernel(
    const double* __restrict__ X, const int64_t* __restrict__ off, int64_t U, int64_t n_frames,
    srx_models_t md, F64Tables tb, int n_tiles, double* __restrict__ out) {
    constexpr int KQ = 4 * R;
    const int M = md.n_models, S = md.n_states, D = md.n_features;
    const int lane = threadIdx.x & 63, g = lane >> 4, ul = lane & 15;
    // Work item = (batch of 16 utterances, tile).  Models of at most 8 states: the tile runs
    // fastest, so the waves that take the tiles of one batch sit next to each other (mostly in
    // one workgroup) and walk the same frames at about the same time -- the batch's features
    // come from memory once and from the cache for the other tiles (6 x 8 states: 3.87 -> 3.37
    // ms per 65,536 x 200 frames, 24 x 4: 6.6 -> 5.6).  16-state models (a tile per model, 128
    // transition registers per lane) measured half as fast again that way and keep the tile
    // slowest.
    const int64_t item = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int64_t gp = ((U + 15) / 16 + 3) / 4 * 4;
    const int tile = SP < 16 ? (int)(item % n_tiles) : (int)(item / gp);
    const int64_t group = SP < 16 ? item / n_tiles : item % gp;
    if (group * 16 >= U || tile >= n_tiles) return;
    const int64_t u = group * 16 + ul;
    const bool valid = u < U;
    const int64_t uc = valid ? u : U - 1;
    int64_t f0 = off[uc];
    const int T = valid ? (int)(off[uc + 1] - f0) : 0;
    if (f0 > n_frames - 1) f0 = n_frames - 1;   // empty trailing utterance: stay in bounds
    int Tmax = T;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) Tmax = max(Tmax, __shfl_xor(Tmax, o, 64));

    // ---- loop-invariant registers: this lane's A operands (row ul of the tile) ...
    const int acol = tile * 16 + 4 * (ul & 3) + (ul >> 2);
    double w[KQ];
#pragma unroll
    for (int q = 0; q < KQ; ++q) w[q] = tb.W[((size_t)acol * KQ + q) * 4 + g];
    // ... and, for its own four result rows (logical columns 4g .. 4g+3), the constants,
    // start probabilities and transition coefficients
    constexpr int NB = SP / 4;                  // lane groups (blocks of 4 states) per model
    const int m = (16 / SP) * tile + g / NB;
    const int h = g % NB;                       // this lane's block inside its model
    constexpr int NI = SP;                      // source states per model
    f64x4 c0;
    double pi_[4], a[NI][4];                    // a[i][r]: from source i to own state r
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        c0[r] = tb.cst[tile * 16 + 4 * g + r];
        const int j = 4 * h + r;
        pi_[r] = (m < M && j < S) ? md.startprob[m * S + j] : 0.0;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            // sources 4k .. 4k+3 of the list are the states held by lane l ^ 16k (k = 0: own)
            const int src = 4 * (h ^ (i >> 2)) + (i & 3);
            a[i][r] = (m < M && j < S && src < S) ? md.transmat[((size_t)m * S + src) * S + j] : 0.0;
        }
    }
    // sum / maximum over the lane groups of a model
    auto model_max = [&](double v) {
        if (SP >= 8) v = fmax(v, __shfl_xor(v, 16, 64));
        if (SP == 16) v = fmax(v, __shfl_xor(v, 32, 64));
        return v;
    };
    auto model_sum = [&](double v) {
        if (SP >= 8) v += __shfl_xor(v, 16, 64);
        if (SP == 16) v += __shfl_xor(v, 32, 64);
        return v;
    };

    double alpha[4] = {0.0, 0.0, 0.0, 0.0};
    double lp = 0.0;

    // ---- frame loads: two adjacent values per r
    const double* xbase = X + f0 * D;
    double xc[R][2], xn[R][2];
    auto load = [&](int t, double (&dst)[R][2]) {
        int tc = t < T ? t : T - 1;
        if (tc < 0) tc = 0;
        const double* p = xbase + (int64_t)tc * D;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            // features past D meet zero weights: read a valid neighbour instead
            dst[r][0] = p[min(8 * r + 2 * g, D - 1)];
            dst[r][1] = p[min(8 * r + 2 * g + 1, D - 1)];
        }
    };
    if (Tmax > 0) load(0, xc);

    for (int t = 0; t < Tmax; ++t) {
        if (t + 1 < Tmax) load(t + 1, xn);
        f64x4 acc = c0;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const double x0 = xc[r][0], x1 = xc[r][1];
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(w[4 * r + 0], x0, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(w[4 * r + 1], x1, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(w[4 * r + 2], x0 * x0, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(w[4 * r + 3], x1 * x1, acc, 0, 0, 0);
        }
        // shift by the frame's largest log-density over the model's states
        const double mx = model_max(fmax(fmax(acc[0], acc[1]), fmax(acc[2], acc[3])));
        double b[4], na[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) b[r] = exp2(acc[r] - mx);
        if (t == 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) na[r] = pi_[r] * b[r];
        } else {
            double src[NI];
#pragma unroll
            for (int i = 0; i < 4; ++i) src[i] = alpha[i];
#pragma unroll
            for (int k = 1; k < NB; ++k)
#pragma unroll
                for (int i = 0; i < 4; ++i) src[4 * k + i] = __shfl_xor(alpha[i], 16 * k, 64);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                double s = 0.0;
#pragma unroll
                for (int i = 0; i < NI; ++i) s = fma(src[i], a[i][r], s);
                na[r] = s * b[r];
            }
        }
        const double c = model_sum((na[0] + na[1]) + (na[2] + na[3]));
        // a chain no state can continue (c == 0) stays at zero; its log-probability is -inf
        const double ci = c > 0.0 ? 1.0 / c : 0.0;
        const double dl = mx + log2(c);
        if (t < T) {
            lp += dl;
#pragma unroll
            for (int r = 0; r < 4; ++r) alpha[r] = na[r] * ci;
        }
        if (t + 1 < Tmax) {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                xc[r][0] = xn[r][0];
                xc[r][1] = xn[r][1];
            }
        }
    }

    const bool writer = h == 0;
    if (valid && writer && m < M) out[u * M + m] = lp * kLn2;
}

// ---------------------------------------------------------------------------
// Stand-alone float64 emission log-densities (natural log) of each utterance under its
// own model, [n_frames, S], for the float64 Viterbi / posteriors / E-step chains.  A wave
// takes an utterance, 16 CONSECUTIVE frames as the matrix columns, the model's states as
// one 16-row tile.  Lane group g supplies features 8q + 2g, 8q + 2g + 1 of a frame (one
// 16-byte load per q); weight table per model
//   W[row][q][x | x^2][g][i] = coefficient of feature 8q + 2g + i.
// Result register r of lane (g, frame) is state 4r + g.
__global__ void emission_f64_prep_kernel(srx_models_t md, int RQ, double* W, double* cst) {
    const int M = md.n_models, S = md.n_states, D = md.n_features;
    const int tid = blockIdx.x * blockDim.x + threadIdx.x, nth = gridDim.x * blockDim.x;
    const int per_row = RQ * 16;
    for (int i = tid; i < M * 16 * per_row; i += nth) {
        const int row = i / per_row, w = i % per_row;
        const int q = w >> 4, e = (w >> 3) & 1, d = 8 * q + (w & 7);
        const int m = row >> 4, s_ = row & 15;
        double v = 0.0;
        if (s_ < S && d < D) {
            const size_t k = ((size_t)m * S + s_) * D + d;
            const double var = fmax(md.covars[k], 2.2250738585072014e-308);
            v = e ? -0.5 / var : md.means[k] / var;
        }
        W[i] = v;
    }
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
        if (lane == 0) cst[row] = s_ < S ? -0.5 * (acc + D * 1.8378770664093453) : 0.0;
    }
}

typedef double f64x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int RQ>
__global__ __launch_bounds__(256) void emission_utt_f64_kernel(
    const double* __restrict__ X, const int64_t* __restrict__ off, int64_t U, int64_t n_frames,
    const int32_t* __restrict__ utt_model, int S, int D, const double* __restrict__ W,
    const double* __restrict__ cst, double* __restrict__ out) {
    const int lane = threadIdx.x & 63, g = lane >> 4, ul = lane & 15;
    const int64_t n_waves = (int64_t)gridDim.x * (blockDim.x >> 6);
    const int64_t wid = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const unsigned lane_off = (unsigned)ul * (unsigned)D * 8u + 16u * g;
    int cur = -1;
    f64x2 w1[RQ], w2[RQ];
    f64x4 c0 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int q = 0; q < RQ; ++q) w1[q] = w2[q] = f64x2{0.0, 0.0};
    for (int64_t u = wid; u < U; u += n_waves) {
        const int m = utt_model ? utt_model[u] : 0;
        const int64_t f0 = off[u];
        const int64_t T = off[u + 1] - f0;
        if (T <= 0) continue;
        if (m != cur) {
            cur = m;
#pragma unroll
            for (int q = 0; q < RQ; ++q) {
                const double* p = W + (((size_t)(m * 16 + ul)) * RQ + q) * 16 + 2 * g;
                w1[q] = *(const f64x2*)p;
                w2[q] = *(const f64x2*)(p + 8);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) c0[r] = cst[m * 16 + 4 * r + g];
        }
        const int64_t n_groups = (T + 15) / 16;
        auto load = [&](int64_t k, u32x4 (&dst)[RQ]) {
            // descriptor over [first frame of the group, end of the utterance): rows past
            // the end read as zeros
            const int64_t fk = k < n_groups ? f0 + k * 16 : f0;
            const int64_t rest = k < n_groups ? (f0 + T - fk) * (int64_t)D * 8 : 0;
            const uintptr_t a = (uintptr_t)(X + fk * D);
            const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a);
            const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
            const unsigned n = (unsigned)(rest < 0xFFFFFF00ll ? rest : 0xFFFFFF00ll);
            const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
                (void*)(((uintptr_t)hi << 32) | lo), 0, __builtin_amdgcn_readfirstlane(n), 0x00020000);
#pragma unroll
            for (int q = 0; q < RQ; ++q)
                dst[q] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane_off + 64u * q, 0, 0);
        };
        u32x4 xa[RQ], xb[RQ];
        load(0, xa);
        for (int64_t k = 0; k < n_groups; ++k) {
            load(k + 1, xb);
            f64x4 a1 = c0, a2 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int q = 0; q < RQ; ++q) {
                const f64x2 x = __builtin_bit_cast(f64x2, xa[q]);
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(w1[q][i], x[i], a1, 0, 0, 0);
                    a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(w2[q][i], x[i] * x[i], a2, 0, 0, 0);
                }
            }
            const int64_t t = k * 16 + ul;
            if (t < T) {
                double* o = out + (f0 + t) * S;
                const f64x4 v = a1 + a2;
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (4 * r + g < S) o[4 * r + g] = v[r];
            }
#pragma unroll
            for (int q = 0; q < RQ; ++q) xa[q] = xb[q];
        }
    }
}

// The same log-densities for EVERY state of EVERY model, out[frame][m * S + s]
// (srx_emission_loglik in float64): a wave takes groups of 16 consecutive frames
// (grid-stride), loads them once and walks the models' tiles, whose weights come from the
// table of emission_f64_prep_kernel through the cache.
template <int RQ>
__global__ __launch_bounds__(256) void emission_all_f64_kernel(
    const double* __restrict__ X, int64_t n_frames, int M, int S, int D,
    const double* __restrict__ W, const double* __restrict__ cst, double* __restrict__ out) {
    const int lane = threadIdx.x & 63, g = lane >> 4, ul = lane & 15;
    const int64_t n_waves = (int64_t)gridDim.x * (blockDim.x >> 6);
    const int64_t wid = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int64_t n_groups = (n_frames + 15) / 16;
    const int64_t C = (int64_t)M * S;
    for (int64_t k = wid; k < n_groups; k += n_waves) {
        const int64_t t = k * 16 + ul;
        const bool ok = t < n_frames;
        const double* row = X + (ok ? t : n_frames - 1) * D;
        f64x2 x[RQ];
#pragma unroll
        for (int q = 0; q < RQ; ++q) {
            // features past D meet zero weights: read a valid neighbour instead
            const int d0 = 8 * q + 2 * g;
            x[q][0] = row[d0 < D ? d0 : D - 1];
            x[q][1] = row[d0 + 1 < D ? d0 + 1 : D - 1];
        }
        for (int m = 0; m < M; ++m) {
            f64x4 a1, a2 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int r = 0; r < 4; ++r) a1[r] = cst[m * 16 + 4 * r + g];
#pragma unroll
            for (int q = 0; q < RQ; ++q) {
                const double* p = W + (((size_t)(m * 16 + ul)) * RQ + q) * 16 + 2 * g;
                const f64x2 w1 = *(const f64x2*)p, w2 = *(const f64x2*)(p + 8);
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(w1[i], x[q][i], a1, 0, 0, 0);
                    a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(w2[i], x[q][i] * x[q][i], a2, 0, 0, 0);
                }
            }
            if (ok) {
                double* o = out + t * C + (int64_t)m * S;
                const f64x4 v = a1 + a2;
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (4 * r + g < S) o[4 * r + g] = v[r];
            }
        }
    }
}

// ---------------------------------------------------------------------------
// Observation sums of the float64 E-step (accumulate_sufficient_statistics,
// my_hmmlearn/hmm.py:237-253): post += sum_t gamma_t, obs += gamma^T X, obs**2 += gamma^T X^2
// as matrix products.  One wave per workgroup walks a contiguous slice of the processing
// order; four frames are one k-step: A[row = state][k] = gamma[t0 + k][state],
// B[k][col] = x[t0 + k][16c + col] (and its square) for NC blocks of 16 features, so result
// register r of lane (g, col) is the sum for state 4r + g and feature 16c + col.  The sums
// stay in registers while the model does not change, then go to the wave's own slice of
// `partial` [gridDim][M][S + 2 S D] (layout and fixed-order reduction of obs_stats_kernel).
template <int NC>
__global__ __launch_bounds__(64) void obs_stats_f64_kernel(
    const double* __restrict__ X, const double* __restrict__ gamma, const int64_t* __restrict__ off,
    int64_t U, const int32_t* __restrict__ utt_model, const int64_t* __restrict__ utt_order, int M,
    int S, int D, double* __restrict__ partial) {
    const int lane = threadIdx.x, g = lane >> 4, ul = lane & 15;
    const int SD = S * D, Q = S + 2 * SD;
    double* mine = partial + (size_t)blockIdx.x * M * Q;
    for (int i = lane; i < M * Q; i += 64) mine[i] = 0.0;
    const int64_t per = (U + gridDim.x - 1) / gridDim.x;
    const int64_t k0 = blockIdx.x * per;
    const int64_t k1 = k0 + per < U ? k0 + per : U;
    f64x4 o1[NC], o2[NC];
    double ap = 0.0;
#pragma unroll
    for (int c = 0; c < NC; ++c) o1[c] = o2[c] = f64x4{0.0, 0.0, 0.0, 0.0};
    int cur = -1;
    auto flush = [&](int mm) {
        if (mm < 0) return;
        double* dst = mine + (size_t)mm * Q;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int st_ = 4 * r + g, d = 16 * c + ul;
                if (st_ < S && d < D) {
                    dst[S + st_ * D + d] += o1[c][r];
                    dst[S + SD + st_ * D + d] += o2[c][r];
                }
            }
            o1[c] = o2[c] = f64x4{0.0, 0.0, 0.0, 0.0};
        }
        double p = ap;
        p += __shfl_xor(p, 16, 64);
        p += __shfl_xor(p, 32, 64);
        if (g == 0 && ul < S) dst[ul] += p;
        ap = 0.0;
    };
    // (the slice is zeroed and later updated by different lanes of this one-wave workgroup)
    __syncthreads();
    for (int64_t k = k0; k < k1; ++k) {
        const int64_t u = utt_order ? utt_order[k] : k;
        const int m = utt_model ? utt_model[u] : 0;
        if (m != cur) {
            flush(cur);
            cur = m;
        }
        const int64_t f0 = off[u], T = off[u + 1] - f0;
        const bool st_ok = ul < S;
        auto load = [&](int64_t t0, double& gv, double (&xv)[NC]) {
            const int64_t t = t0 + g;
            const bool ok = t < T;
            const int64_t f = f0 + (ok ? t : 0);
            gv = ok && st_ok ? gamma[f * S + ul] : 0.0;
#pragma unroll
            for (int c = 0; c < NC; ++c) xv[c] = ok && 16 * c + ul < D ? X[f * D + 16 * c + ul] : 0.0;
        };
        double ga, xa[NC], gb, xb[NC];
        if (T > 0) load(0, ga, xa);
        for (int64_t t0 = 0; t0 < T; t0 += 4) {
            if (t0 + 4 < T) load(t0 + 4, gb, xb);
            ap += ga;
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                o1[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(ga, xa[c], o1[c], 0, 0, 0);
                o2[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(ga, xa[c] * xa[c], o2[c], 0, 0, 0);
            }
            ga = gb;
#pragma unroll
            for (int c = 0; c < NC; ++c) xa[c] = xb[c];
        }
    }
    flush(cur);
}

template <int SP, int R>
void launch(const double* X, const int64_t* off, int64_t U, int64_t n_frames, const srx_models_t& md,
            int n_tiles, F64Tables tb, double* out, hipStream_t st) {
    const int64_t groups = (U + 15) / 16;
    const int per_block = 4;
    const int64_t items = SP < 16 ? groups * n_tiles : (groups + 3) / 4 * 4 * n_tiles;
    const unsigned grid = (unsigned)((items + per_block - 1) / per_block);
    hipLaunchKernelGGL((score_f64_kernel<SP, R>), dim3(grid), dim3(64 * per_block), 0, st, X, off,
                       U, n_frames, md, tb, n_tiles, out);
}

}  // namespace

// Models of <= 16 states and <= 48 features; anything else: SRX_EUNSUPPORTED (the caller
// takes the shape-generic kernels).
int srx_mfma_score_f64(srx_context* h, const void* X, const int64_t* off, int64_t U,
                       int64_t n_frames, const srx_models_t* md, double* out, hipStream_t st) {
    const int M = md->n_models, S = md->n_states, D = md->n_features;
    if (S > 16 || D > 48 || D < 1 || n_frames <= 0 || U <= 0) return SRX_EUNSUPPORTED;
    const int SP = S <= 4 ? 4 : (S <= 8 ? 8 : 16);
    const int n_tiles = (M * SP + 15) / 16;
    if ((int64_t)n_tiles * ((U + 15) / 16) > 0x7FFFFFFFll * 4) return SRX_EUNSUPPORTED;
    const int R = (D + 7) / 8;
    const int KQ = 4 * R;
    const size_t n_w = (size_t)n_tiles * 16 * KQ * 4, n_c = (size_t)n_tiles * 16;
    double* base;
    int rc = srx_ws_ensure(h, WS_PARAMS, sizeof(double) * (n_w + n_c), (void**)&base);
    if (rc) return rc;
    F64Tables tb;
    tb.W = base;
    tb.cst = base + n_w;
    hipLaunchKernelGGL(f64_prep_kernel, dim3(16), dim3(256), 0, st, *md, SP, n_tiles, R, base, base + n_w);
    const double* Xd = (const double*)X;
#define SRX_CASE(sp, r)                                                   \
    if (SP == sp && R == r) {                                             \
        launch<sp, r>(Xd, off, U, n_frames, *md, n_tiles, tb, out, st);   \
        SRX_HIP(hipGetLastError());                                       \
        return SRX_OK;                                                    \
    }
    SRX_CASE(8, 1) SRX_CASE(8, 2) SRX_CASE(8, 3) SRX_CASE(8, 4) SRX_CASE(8, 5) SRX_CASE(8, 6)
    SRX_CASE(4, 1) SRX_CASE(4, 2) SRX_CASE(4, 3) SRX_CASE(4, 4) SRX_CASE(4, 5) SRX_CASE(4, 6)
    SRX_CASE(16, 1) SRX_CASE(16, 2) SRX_CASE(16, 3) SRX_CASE(16, 4) SRX_CASE(16, 5) SRX_CASE(16, 6)
#undef SRX_CASE
    return SRX_EUNSUPPORTED;
}

// Float64 log-densities of every utterance under its own model, out[frame][state]
// (<= 16 states, <= 48 features, one utterance < 4 GiB); its tables live in the WS_MISC slot.
int srx_mfma_emission_utt_f64(srx_context* h, const void* X, const int64_t* off, int64_t U,
                              int64_t n_frames, const int32_t* utt_model, const srx_models_t* md,
                              double* out, hipStream_t st) {
    const int M = md->n_models, S = md->n_states, D = md->n_features;
    if (S > 16 || D > 48 || M > 4096 || n_frames <= 0 || U <= 0) return SRX_EUNSUPPORTED;
    const int RQ = (D + 7) / 8;
    const size_t n_w = (size_t)M * 16 * RQ * 16, n_c = (size_t)M * 16;
    double* base;
    int rc = srx_ws_ensure(h, WS_MISC, sizeof(double) * (n_w + n_c), (void**)&base);
    if (rc) return rc;
    double* cst = base + n_w;
    hipLaunchKernelGGL(emission_f64_prep_kernel, dim3(16), dim3(256), 0, st, *md, RQ, base, cst);
    int64_t gx = (U + 3) / 4;
    const int64_t cap = (int64_t)h->n_cu * 8;
    if (gx > cap) gx = cap;
    const double* Xd = (const double*)X;
#define SRX_CASE(rq)                                                                        \
    if (RQ == rq) {                                                                         \
        hipLaunchKernelGGL((emission_utt_f64_kernel<rq>), dim3((unsigned)gx), dim3(256), 0, \
                           st, Xd, off, U, n_frames, utt_model, S, D, base, cst, out);      \
        SRX_HIP(hipGetLastError());                                                         \
        return SRX_OK;                                                                      \
    }
    SRX_CASE(6) SRX_CASE(5) SRX_CASE(4) SRX_CASE(3) SRX_CASE(2) SRX_CASE(1)
#undef SRX_CASE
    return SRX_EUNSUPPORTED;
}

// Observation sums of the float64 E-step into `partial` [n_slices][M][S + 2 S D]
// (<= 16 states, <= 48 features); n_slices workgroups of one wave.
int srx_mfma_obs_stats_f64(srx_context* h, const void* X, const void* gamma, const int64_t* off,
                           int64_t U, const int32_t* utt_model, const int64_t* utt_order,
                           const srx_models_t* md, double* partial, int n_slices, hipStream_t st) {
    const int M = md->n_models, S = md->n_states, D = md->n_features;
    if (S > 16 || D > 48 || U <= 0 || n_slices < 1) return SRX_EUNSUPPORTED;
    const int NC = (D + 15) / 16;
#define SRX_CASE(nc)                                                                              \
    if (NC == nc) {                                                                               \
        hipLaunchKernelGGL((obs_stats_f64_kernel<nc>), dim3((unsigned)n_slices), dim3(64), 0, st, \
                           (const double*)X, (const double*)gamma, off, U, utt_model, utt_order,  \
                           M, S, D, partial);                                                     \
        SRX_HIP(hipGetLastError());                                                               \
        return SRX_OK;                                                                            \
    }
    SRX_CASE(3) SRX_CASE(2) SRX_CASE(1)
#undef SRX_CASE
    return SRX_EUNSUPPORTED;
}

// Float64 log-densities of all state columns of all models, out[frame][m * S + s]
// (<= 16 states, <= 48 features); tables in the WS_MISC slot.
int srx_mfma_emission_f64(srx_context* h, const void* X, int64_t n_frames, const srx_models_t* md,
                          double* out, hipStream_t st) {
    const int M = md->n_models, S = md->n_states, D = md->n_features;
    if (S > 16 || D > 48 || M > 4096 || n_frames <= 0) return SRX_EUNSUPPORTED;
    const int RQ = (D + 7) / 8;
    const size_t n_w = (size_t)M * 16 * RQ * 16, n_c = (size_t)M * 16;
    double* base;
    int rc = srx_ws_ensure(h, WS_MISC, sizeof(double) * (n_w + n_c), (void**)&base);
    if (rc) return rc;
    double* cst = base + n_w;
    hipLaunchKernelGGL(emission_f64_prep_kernel, dim3(16), dim3(256), 0, st, *md, RQ, base, cst);
    int64_t gx = ((n_frames + 15) / 16 + 3) / 4;
    const int64_t cap = (int64_t)h->n_cu * 16;
    if (gx > cap) gx = cap;
#define SRX_CASE(rq)                                                                        \
    if (RQ == rq) {                                                                         \
        hipLaunchKernelGGL((emission_all_f64_kernel<rq>), dim3((unsigned)gx), dim3(256), 0, \
                           st, (const double*)X, n_frames, M, S, D, base, cst, out);        \
        SRX_HIP(hipGetLastError());                                                         \
        return SRX_OK;                                                                      \
    }
    SRX_CASE(6) SRX_CASE(5) SRX_CASE(4) SRX_CASE(3) SRX_CASE(2) SRX_CASE(1)
#undef SRX_CASE
    return SRX_EUNSUPPORTED;
}
