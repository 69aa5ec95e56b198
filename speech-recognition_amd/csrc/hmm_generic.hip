// Shape-generic HMM kernels: any number of models, S <= 64 states, any feature
// count, float or double arithmetic.  This is the validation ("parity") path
// in double and the fallback for shapes the matrix-core kernels do not cover.
//
// Decomposition (all ragged over utterances):
//   prep      model tables in the compute dtype (means, -1/(2 var), constants, log A)
//   emission  log N(x_t; mu_s, var_s) for every (frame, state)        [HBM/VALU]
//   chain     one sub-wave of SP lanes (SP = next pow2 >= S) per (utterance, model):
//             lane j owns state j, cross-lane traffic by width-SP shuffles
//             - score:   scaled forward, log-prob accumulated in double
//             - fb:      forward, backward, posteriors, xi row sums
//             - viterbi: max-plus forward with back-pointers + back-trace
//   stats     gamma^T [X, X^2] per model, per-workgroup partials then a fixed-order sum
//
// Scaling.  The reference exponentiates the log-densities and rescales the
// forward variables by 1/sum (my_hmmlearn/utils.py:139-147).  Here every frame
// is first shifted by its largest log-density m_t (b~ = exp(ll - m_t)); the
// scaled lattices are unchanged, the frame's scale becomes c~_t = c_t exp(m_t)
// and log P = sum_t (m_t - log c~_t), so nothing underflows in float.
#include "srx_internal.h"

#include <type_traits>
#include <utility>

#include <math.h>

__global__ void srx_fill_slices(int64_t* offs, int64_t n_slices, int64_t slice, int64_t n_frames);

namespace {

constexpr int kBlock = 256;
constexpr double kTiny = 2.2250738585072014e-308;  // np.finfo(float).tiny, hmm.py:159

template <typename T>
struct Tables {
    const T* mu;     // [M,S,D]
    const T* hinv;   // [M,S,D]  -0.5 / var
    const T* cst;    // [M,S]    -0.5 (D log 2pi + sum log var)
    const T* A;      // [M,S,S]
    const T* pi;     // [M,S]
    const T* logA;   // [M,S,S]
    const T* logpi;  // [M,S]
};

template <typename T>
size_t tables_bytes(int M, int S, int D) {
    return sizeof(T) * ((size_t)M * S * D * 2 + (size_t)M * S * 3 + (size_t)M * S * S * 2);
}

template <typename T>
Tables<T> carve_tables(void* base, int M, int S, int D) {
    T* p = (T*)base;
    Tables<T> t;
    t.mu = p;      p += (size_t)M * S * D;
    t.hinv = p;    p += (size_t)M * S * D;
    t.cst = p;     p += (size_t)M * S;
    t.A = p;       p += (size_t)M * S * S;
    t.pi = p;      p += (size_t)M * S;
    t.logA = p;    p += (size_t)M * S * S;
    t.logpi = p;
    return t;
}

// beta of the scaled backward recursion is of order 1 / alpha^ for a state: where alpha^ has
// underflowed to zero, beta may pass the dtype's largest value, and 0 x infinity would poison
// the frame's posteriors and every sum they enter.  Held at the largest finite value the
// product is the zero that alpha^ already stands for.
__device__ __forceinline__ float finite_max(float x) { return fminf(x, 3.4028234664e38f); }
__device__ __forceinline__ double finite_max(double x) { return fmin(x, 1.7976931348623157e308); }

template <typename T>
__global__ void prep_kernel(srx_models_t md, T* mu, T* hinv, T* cst, T* A, T* pi, T* logA,
                            T* logpi) {
    const int M = md.n_models, S = md.n_states, D = md.n_features;
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    const int nth = gridDim.x * blockDim.x;
    for (int i = tid; i < M * S * D; i += nth) {
        double v = fmax(md.covars[i], kTiny);
        mu[i] = (T)md.means[i];
        hinv[i] = (T)(-0.5 / v);
    }
    // constants: one wave per state, lanes over features (a serial sum of D float64
    // logarithms per state would be most of this kernel's time)
    const int wave = tid >> 6, lane = tid & 63, n_wave = nth >> 6;
    for (int i = wave; i < M * S; i += n_wave) {
        double acc = 0.0;
        for (int d = lane; d < D; d += 64) acc += log(fmax(md.covars[(size_t)i * D + d], kTiny));
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
        if (lane == 0) {
            cst[i] = (T)(-0.5 * (D * 1.8378770664093453 + acc));
            pi[i] = (T)md.startprob[i];
            logpi[i] = (T)log(md.startprob[i]);
        }
    }
    for (int i = tid; i < M * S * S; i += nth) {
        A[i] = (T)md.transmat[i];
        logA[i] = (T)log(md.transmat[i]);
    }
}

// One workgroup per utterance (grid-stride).  sel < 0: all models, columns
// m*S+s; otherwise only the utterance's own model (utt_model or model 0).
template <typename T>
__global__ __launch_bounds__(kBlock) void emission_kernel(const T* __restrict__ X, const int64_t* __restrict__ off,
                                int64_t U, const int32_t* __restrict__ utt_model, int all_models,
                                int M, int S, int D, Tables<T> tb, T* __restrict__ out) {
    const int NC = all_models ? M * S : S;
    for (int64_t u = blockIdx.x; u < U; u += gridDim.x) {
        const int64_t f0 = off[u];
        const int64_t n = (off[u + 1] - f0) * NC;
        const int mbase = all_models ? 0 : (utt_model ? utt_model[u] : 0);
        for (int64_t idx = threadIdx.x; idx < n; idx += blockDim.x) {
            const int64_t f = f0 + idx / NC;
            const int c = (int)(idx % NC) + mbase * S;
            const T* x = X + f * D;
            const T* mu = tb.mu + (size_t)c * D;
            const T* hi = tb.hinv + (size_t)c * D;
            T acc = 0;
            for (int d = 0; d < D; ++d) {
                T df = x[d] - mu[d];
                acc += df * df * hi[d];
            }
            out[f * NC + (c - mbase * S)] = acc + tb.cst[c];
        }
    }
}

// Lane exchange inside a sub-wave of SP <= 16 lanes, which is (part of) one DPP row: a
// v_mov_b32_dpp per 32-bit half has the latency of a vector instruction, where __shfl /
// __shfl_xor go through the LDS crossbar (ds_bpermute) and cost ten times that on the
// serial path of a chain.
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
    const unsigned u = __float_as_uint(v);
    return __uint_as_float((unsigned)__builtin_amdgcn_update_dpp(u, u, CTRL, 0xf, 0xf, false));
}
template <int CTRL>
__device__ __forceinline__ double dpp_mov(double v) {
    const unsigned long long u = (unsigned long long)__double_as_longlong(v);
    unsigned lo = (unsigned)u, hi = (unsigned)(u >> 32);
    lo = (unsigned)__builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xf, 0xf, false);
    hi = (unsigned)__builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xf, 0xf, false);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
// Partner of each reduction stage: lane 15 - i (row_mirror), 7 - i inside a half row
// (row_half_mirror), i ^ 2 and i ^ 1 (quad_perm).  After every stage the values are
// symmetric under the next stage's pairing, so all lanes end with the same bits.
template <typename T, int SP, typename F>
__device__ __forceinline__ T sub_reduce(T v, F&& op) {
    if constexpr (SP <= 16) {
        if constexpr (SP >= 16) v = op(v, dpp_mov<0x140>(v));
        if constexpr (SP >= 8) v = op(v, dpp_mov<0x141>(v));
        if constexpr (SP >= 4) v = op(v, dpp_mov<0x4E>(v));
        if constexpr (SP >= 2) v = op(v, dpp_mov<0xB1>(v));
    } else {
#pragma unroll
        for (int o = SP / 2; o > 0; o >>= 1) v = op(v, __shfl_xor(v, o, SP));
    }
    return v;
}
template <typename T, int SP>
__device__ __forceinline__ T sub_sum(T v) {
    return sub_reduce<T, SP>(v, [](T a, T b) { return a + b; });
}
template <typename T, int SP>
__device__ __forceinline__ T sub_max(T v) {
    return sub_reduce<T, SP>(v, [](T a, T b) { return (T)fmax(a, b); });
}
// log P(x) = sum_t (mx_t + log c_t) without a logarithm per step (a software f64 log is
// ~50 instructions on the chain's serial path): the product of the c_t is carried as a
// mantissa in [0.5, 1) and a binary exponent, renormalised every step with the two frexp
// instructions; one logarithm at the end.  c == 0 gives -inf, as log(0) did.
struct LogProduct {
    double shift = 0.0, mant = 1.0;
    long long expo = 0;
    __device__ __forceinline__ void step(double mx, double c) {
        shift += mx;
        int e;
        mant = frexp(mant * c, &e);
        expo += e;
    }
    __device__ __forceinline__ double value() const {
        return shift + ((double)expo * 0.6931471805599453 + log(mant));
    }
};

// Value of lane I of the sub-wave in every lane: row_newbcast for a whole row.
template <typename T, int SP, int I>
__device__ __forceinline__ T sub_bcast(T v) {
    if constexpr (SP == 16) return dpp_mov<0x150 + I>(v);
    else return __shfl(v, I, SP);
}
// f(integral_constant<int, i>) for i = 0 .. SP-1, in order.
template <typename F, int... I>
__device__ __forceinline__ void for_lanes_impl(F&& f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int SP, typename F>
__device__ __forceinline__ void for_lanes(F&& f) {
    for_lanes_impl(f, std::make_integer_sequence<int, SP>{});
}

template <typename T> __device__ __forceinline__ T neg_inf();
template <> __device__ __forceinline__ float neg_inf<float>() { return -INFINITY; }
template <> __device__ __forceinline__ double neg_inf<double>() { return -(double)INFINITY; }

// Scaled forward over one chain per sub-wave.  loglik row stride is NC.
template <typename T, int SP>
__global__ __launch_bounds__(kBlock) void score_chain_kernel(const T* __restrict__ loglik, const int64_t* __restrict__ off,
                                   int64_t U, int M, int S, Tables<T> tb,
                                   double* __restrict__ out) {
    const int j = threadIdx.x % SP;
    const int64_t sub = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / SP;
    const int64_t nsub = (int64_t)gridDim.x * blockDim.x / SP;
    const int NC = M * S;
    for (int64_t chain = sub; chain < U * M; chain += nsub) {
        const int64_t u = chain / M;
        const int m = (int)(chain % M);
        const int64_t f0 = off[u];
        const int T_len = (int)(off[u + 1] - f0);
        T acol[SP];
#pragma unroll
        for (int i = 0; i < SP; ++i)
            acol[i] = (i < S && j < S) ? tb.A[((size_t)m * S + i) * S + j] : (T)0;
        const T pij = j < S ? tb.pi[m * S + j] : (T)0;
        const T* ll = loglik + f0 * NC + m * S + j;
        T alpha = 0;
        LogProduct lpp;
        for (int t = 0; t < T_len; ++t) {
            const T l = j < S ? ll[(int64_t)t * NC] : neg_inf<T>();
            const T mx = sub_max<T, SP>(l);
            const T b = j < S ? exp(l - mx) : (T)0;
            T a;
            if (t == 0) {
                a = pij * b;
            } else {
                a = 0;
                for_lanes<SP>([&](auto ic) {
                    constexpr int i = decltype(ic)::value;
                    a += sub_bcast<T, SP, i>(alpha) * acol[i];
                });
                a *= b;
            }
            const T c = sub_sum<T, SP>(a);
            lpp.step((double)mx, (double)c);
            alpha = a / c;
        }
        if (j == 0) out[chain] = lpp.value();
    }
}

// Forward-backward for the utterance's own model.  loglik row stride is S.
// Writes posteriors into `lat` (which first holds the forward lattice), and
// accumulates start / xi / log-prob / nobs into LDS, then one partial record
// per workgroup: partial[blockIdx][M][S*S + S + 2] = {xi, start, logprob, nobs}.
// With partial == NULL only the posteriors and utt_lp[u] (per-utterance
// log-probability, may itself be NULL otherwise) are produced.
template <typename T, int SP>
__global__ __launch_bounds__(kBlock) void fb_chain_kernel(T* __restrict__ loglik, const int64_t* __restrict__ off,
                                int64_t U, const int32_t* __restrict__ utt_model,
                                const int64_t* __restrict__ utt_order, int M, int S,
                                Tables<T> tb, T* __restrict__ lat, T* __restrict__ scale,
                                double* __restrict__ partial, double* __restrict__ utt_lp,
                                int acc_in_lds) {
    extern __shared__ __attribute__((aligned(16))) double acc_lds[];
    const int P = S * S + S + 2;
    // (banks whose M x P sums do not fit the LDS accumulate straight into the workgroup's
    // own slice of `partial` in global memory)
    double* acc = acc_in_lds || !partial ? acc_lds : partial + (size_t)blockIdx.x * M * P;
    if (partial) {
        for (int i = threadIdx.x; i < M * P; i += blockDim.x) acc[i] = 0.0;
        __threadfence_block();
        __syncthreads();
    }

    const int j = threadIdx.x % SP;
    const int64_t sub = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / SP;
    const int64_t nsub = (int64_t)gridDim.x * blockDim.x / SP;
    for (int64_t k = sub; k < U; k += nsub) {
        const int64_t u = utt_order ? utt_order[k] : k;
        const int m = utt_model ? utt_model[u] : 0;
        const int64_t f0 = off[u];
        const int T_len = (int)(off[u + 1] - f0);
        if (T_len <= 0) {
            if (utt_lp && j == 0) utt_lp[u] = 0.0;
            continue;
        }
        T acol[SP], arow[SP];
#pragma unroll
        for (int i = 0; i < SP; ++i) {
            acol[i] = (i < S && j < S) ? tb.A[((size_t)m * S + i) * S + j] : (T)0;
            arow[i] = (i < S && j < S) ? tb.A[((size_t)m * S + j) * S + i] : (T)0;
        }
        const T pij = j < S ? tb.pi[m * S + j] : (T)0;
        T* ll = loglik + f0 * S + j;
        T* la = lat + f0 * S + j;
        T* sc = scale + f0;
        // ---- forward (the shifted densities b~ replace the log-densities in the
        // workspace: the backward pass reads them instead of exponentiating again)
        T alpha = 0;
        LogProduct lpp;
        for (int t = 0; t < T_len; ++t) {
            const T l = j < S ? ll[(int64_t)t * S] : neg_inf<T>();
            const T mx = sub_max<T, SP>(l);
            const T b = j < S ? exp(l - mx) : (T)0;
            if (j < S) ll[(int64_t)t * S] = b;
            T a;
            if (t == 0) {
                a = pij * b;
            } else {
                a = 0;
                for_lanes<SP>([&](auto ic) {
                    constexpr int i = decltype(ic)::value;
                    a += sub_bcast<T, SP, i>(alpha) * acol[i];
                });
                a *= b;
            }
            const T c = sub_sum<T, SP>(a);
            lpp.step((double)mx, (double)c);
            const T ci = (T)1 / c;
            alpha = a * ci;
            if (j < S) la[(int64_t)t * S] = alpha;
            if (j == 0) sc[t] = ci;
        }
        const double lp = lpp.value();
        // ---- backward, posteriors, xi
        T xi[SP];
#pragma unroll
        for (int i = 0; i < SP; ++i) xi[i] = 0;
        T bb = 0;      // b~_{t+1}[j] * beta'_{t+1}[j]
        T gamma0 = 0;
        for (int t = T_len - 1; t >= 0; --t) {
            const T al = j < S ? la[(int64_t)t * S] : (T)0;
            const T ci = sc[t];
            T beta;
            if (t == T_len - 1) {
                beta = ci;
            } else {
                T s = 0;
                for_lanes<SP>([&](auto ic) {
                    constexpr int i = decltype(ic)::value;
                    const T w = arow[i] * sub_bcast<T, SP, i>(bb);
                    s += w;
                    xi[i] += al * w;
                });
                beta = finite_max(s * ci);
            }
            T g = al * beta;
            T gs = sub_sum<T, SP>(g);
            if (gs == (T)0) gs = (T)1;   // utils.normalize: zero rows divide by 1
            g = g / gs;
            if (j < S) la[(int64_t)t * S] = g;
            const T b = j < S ? ll[(int64_t)t * S] : (T)0;
            bb = b * beta;
            gamma0 = g;
        }
        if (utt_lp && j == 0) utt_lp[u] = lp;
        if (partial && j < S) {
            double* a = acc + (size_t)m * P;
            if (T_len > 1)
                for (int i = 0; i < S; ++i) atomicAdd(&a[j * S + i], (double)xi[i]);
            atomicAdd(&a[S * S + j], (double)gamma0);
            if (j == 0) {
                atomicAdd(&a[S * S + S], lp);
                atomicAdd(&a[S * S + S + 1], 1.0);
            }
        }
    }
    if (!partial || !acc_in_lds) return;
    __syncthreads();
    double* dst = partial + (size_t)blockIdx.x * M * P;
    for (int i = threadIdx.x; i < M * P; i += blockDim.x) dst[i] = acc[i];
}

// post / obs / obs2: each workgroup walks a contiguous slice of the processing
// order; thread owns (s,d) pairs and keeps their sums in registers while the
// model does not change, then adds them to its own slice of `partial`
// [gridDim][M][S + 2 S D].
template <typename T, int SLOTS>
__global__ __launch_bounds__(kBlock) void obs_stats_kernel(const T* __restrict__ X, const T* __restrict__ gamma,
                                 const int64_t* __restrict__ off, int64_t U,
                                 const int32_t* __restrict__ utt_model,
                                 const int64_t* __restrict__ utt_order, int M, int S, int D,
                                 double* __restrict__ partial) {
    const int SD = S * D;
    const int Q = S + 2 * SD;
    double* mine = partial + (size_t)blockIdx.x * M * Q;
    for (int i = threadIdx.x; i < M * Q; i += blockDim.x) mine[i] = 0.0;
    const int64_t per = (U + gridDim.x - 1) / gridDim.x;
    const int64_t k0 = blockIdx.x * per;
    const int64_t k1 = k0 + per < U ? k0 + per : U;
    double a1[SLOTS], a2[SLOTS], ap = 0.0;
#pragma unroll
    for (int q = 0; q < SLOTS; ++q) a1[q] = a2[q] = 0.0;
    int cur = -1;
#define SRX_FLUSH(MM)                                        \
    if ((MM) >= 0) {                                         \
        double* dst = mine + (size_t)(MM) * Q;               \
        _Pragma("unroll") for (int q = 0; q < SLOTS; ++q) {  \
            const int p = threadIdx.x + q * blockDim.x;      \
            if (p < SD) {                                    \
                dst[S + p] += a1[q];                         \
                dst[S + SD + p] += a2[q];                    \
            }                                                \
            a1[q] = a2[q] = 0.0;                             \
        }                                                    \
        if (threadIdx.x < S) dst[threadIdx.x] += ap;         \
        ap = 0.0;                                            \
    }
    __syncthreads();
    for (int64_t k = k0; k < k1; ++k) {
        const int64_t u = utt_order ? utt_order[k] : k;
        const int m = utt_model ? utt_model[u] : 0;
        if (m != cur) {
            SRX_FLUSH(cur);
            cur = m;
        }
        const int64_t f0 = off[u], f1 = off[u + 1];
        for (int64_t f = f0; f < f1; ++f) {
            const T* g = gamma + f * S;
            const T* x = X + f * D;
#pragma unroll
            for (int q = 0; q < SLOTS; ++q) {
                const int p = threadIdx.x + q * blockDim.x;
                if (p < SD) {
                    const double gv = (double)g[p / D];
                    const double xv = (double)x[p % D];
                    a1[q] += gv * xv;
                    a2[q] += gv * xv * xv;
                }
            }
            if (threadIdx.x < S) ap += (double)g[threadIdx.x];
        }
    }
    SRX_FLUSH(cur);
#undef SRX_FLUSH
}

// The same sums for float32 data on the matrix cores (at most 16 states, NC <= 3 blocks of
// 16 features): four frames are one k-step of v_mfma_f32_16x16x4_f32, A[row = state][k] =
// gamma[t0 + k][state], B[k][col] = x[t0 + k][16c + col] or its square, so result register r
// of lane (g, col) is the sum for state 4g + r and feature 16c + col.  The float32
// accumulators cover one utterance, then fold into float64 registers that stay with the
// model until it changes; each one-wave workgroup owns a slice of the processing order and
// of `partial` (layout and fixed-order reduction as above).
template <int NC>
__global__ __launch_bounds__(64) void obs_stats_f32_mfma_kernel(
    const float* __restrict__ X, const float* __restrict__ gamma, const int64_t* __restrict__ off,
    int64_t U, const int32_t* __restrict__ utt_model, const int64_t* __restrict__ utt_order, int M,
    int S, int D, double* __restrict__ partial) {
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    const int lane = threadIdx.x, g = lane >> 4, ul = lane & 15;
    const int SD = S * D, Q = S + 2 * SD;
    double* mine = partial + (size_t)blockIdx.x * M * Q;
    for (int i = lane; i < M * Q; i += 64) mine[i] = 0.0;
    const int64_t per = (U + gridDim.x - 1) / gridDim.x;
    const int64_t k0 = blockIdx.x * per;
    const int64_t k1 = k0 + per < U ? k0 + per : U;
    double d1[NC][4], d2[NC][4], dp = 0.0;
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r) d1[c][r] = d2[c][r] = 0.0;
    int cur = -1;
    auto flush = [&](int mm) {
        if (mm < 0) return;
        double* dst = mine + (size_t)mm * Q;
#pragma unroll
        for (int c = 0; c < NC; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int st_ = 4 * g + r, d = 16 * c + ul;
                if (st_ < S && d < D) {
                    dst[S + st_ * D + d] += d1[c][r];
                    dst[S + SD + st_ * D + d] += d2[c][r];
                }
                d1[c][r] = d2[c][r] = 0.0;
            }
        double p = dp;
        p += __shfl_xor(p, 16, 64);
        p += __shfl_xor(p, 32, 64);
        if (g == 0 && ul < S) dst[ul] += p;
        dp = 0.0;
    };
    // (the slice is zeroed and later updated by different lanes of this one-wave workgroup)
    __syncthreads();
    for (int64_t k = k0; k < k1; ++k) {
        const int64_t u = utt_order ? utt_order[k] : k;
        if (u < 0) continue;
        const int m = utt_model ? utt_model[u] : 0;
        if (m != cur) {
            flush(cur);
            cur = m;
        }
        const int64_t f0 = off[u], T = off[u + 1] - f0;
        const bool st_ok = ul < S;
        auto load = [&](int64_t t0, float& gv, float (&xv)[NC]) {
            const int64_t t = t0 + g;
            const bool ok = t < T;
            const int64_t f = f0 + (ok ? t : 0);
            gv = ok && st_ok ? gamma[f * S + ul] : 0.f;
#pragma unroll
            for (int c = 0; c < NC; ++c) xv[c] = ok && 16 * c + ul < D ? X[f * D + 16 * c + ul] : 0.f;
        };
        f32x4 o1[NC], o2[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) o1[c] = o2[c] = f32x4{0.f, 0.f, 0.f, 0.f};
        float ap = 0.f;
        float ga, xa[NC], gb = 0.f, xb[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) xb[c] = 0.f;
        if (T > 0) load(0, ga, xa);
        int64_t folded = 0;
        for (int64_t t0 = 0; t0 < T; t0 += 4) {
            if (t0 + 4 < T) load(t0 + 4, gb, xb);
            ap += ga;
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                o1[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(ga, xa[c], o1[c], 0, 0, 0);
                o2[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(ga, xa[c] * xa[c], o2[c], 0, 0, 0);
            }
            ga = gb;
#pragma unroll
            for (int c = 0; c < NC; ++c) xa[c] = xb[c];
            // (float32 sums cover at most 4096 frames before they are folded)
            if (t0 + 4 >= T || t0 + 4 - folded >= 4096) {
#pragma unroll
                for (int c = 0; c < NC; ++c)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        d1[c][r] += (double)o1[c][r];
                        d2[c][r] += (double)o2[c][r];
                    }
#pragma unroll
                for (int c = 0; c < NC; ++c) o1[c] = o2[c] = f32x4{0.f, 0.f, 0.f, 0.f};
                dp += (double)ap;
                ap = 0.f;
                folded = t0 + 4;
            }
        }
    }
    flush(cur);
}

// Fixed-order sum of the per-workgroup partials into the packed records.
__global__ void reduce_stats_kernel(const double* __restrict__ part_fb, int n_fb,
                                    const double* __restrict__ part_obs, int n_obs, int M, int S,
                                    int D, double* __restrict__ out) {
    const srx_stats_layout L(S, D);
    const int P = S * S + S + 2;
    const int Q = S + 2 * S * D;
    // one wave per output sum: lane l adds partials l, l + 64, ... in order, then the 64
    // lane sums are added in a fixed tree
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int n_wave = (gridDim.x * blockDim.x) >> 6;
    for (int i = wave; i < M * (P + Q); i += n_wave) {
        const bool fb = i < M * P;
        const int k = fb ? i : i - M * P;
        const double* src = fb ? part_fb : part_obs;
        const int n = fb ? n_fb : n_obs;
        const size_t stride = (size_t)M * (fb ? P : Q);
        double s = 0.0;
        for (int b = lane; b < n; b += 64) s += src[(size_t)b * stride + k];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        if (lane != 0) continue;
        int64_t dst;
        int m;
        if (fb) {
            m = k / P;
            const int r = k % P;
            if (r < S * S) dst = L.trans + r;
            else if (r < S * S + S) dst = L.start + (r - S * S);
            else if (r == S * S + S) dst = L.logprob;
            else dst = L.nobs;
        } else {
            m = k / Q;
            const int r = k % Q;
            dst = r < S ? L.post + r : L.obs + (r - S);  // obs2 follows obs
        }
        out[(size_t)m * L.size + dst] = s;
    }
}

template <typename T, int SP>
__global__ __launch_bounds__(kBlock) void viterbi_chain_kernel(const T* __restrict__ loglik,
                                     const int64_t* __restrict__ off, int64_t U,
                                     const int32_t* __restrict__ utt_model, int S, Tables<T> tb,
                                     uint8_t* __restrict__ bp, int32_t* __restrict__ path,
                                     double* __restrict__ out_lp) {
    const int j = threadIdx.x % SP;
    const int64_t sub = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / SP;
    const int64_t nsub = (int64_t)gridDim.x * blockDim.x / SP;
    for (int64_t u = sub; u < U; u += nsub) {
        const int m = utt_model ? utt_model[u] : 0;
        const int64_t f0 = off[u];
        const int T_len = (int)(off[u + 1] - f0);
        if (T_len <= 0) {
            if (j == 0) out_lp[u] = 0.0;
            continue;
        }
        T lacol[SP];
#pragma unroll
        for (int i = 0; i < SP; ++i)
            lacol[i] = (i < S && j < S) ? tb.logA[((size_t)m * S + i) * S + j] : neg_inf<T>();
        const T* ll = loglik + f0 * S + j;
        uint8_t* b = bp + f0 * S + j;
        T v = j < S ? tb.logpi[m * S + j] + ll[0] : neg_inf<T>();
        double shift = 0.0;
        for (int t = 1; t < T_len; ++t) {
            T best = neg_inf<T>();
            int arg = 0;
            for_lanes<SP>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                const T cand = sub_bcast<T, SP, i>(v) + lacol[i];
                if (cand > best) {   // strict: the lowest index wins ties (np.argmax)
                    best = cand;
                    arg = i;
                }
            });
            if (j < S) {
                v = best + ll[(int64_t)t * S];
                b[(int64_t)t * S] = (uint8_t)arg;
            }
            if constexpr (std::is_same<T, float>::value) {
                // (float32: delta kept relative to its largest entry, see viterbi_lane_kernel)
                if ((t & 7) == 0) {
                    const T top = sub_max<T, SP>(v);
                    if (top > neg_inf<T>()) {
                        shift += (double)top;
                        v -= top;
                    }
                }
            }
        }
        T best = neg_inf<T>();
        int arg = 0;
#pragma unroll
        for (int i = 0; i < SP; ++i) {
            const T cand = __shfl(v, i, SP);
            if (i < S && cand > best) {
                best = cand;
                arg = i;
            }
        }
        if (j == 0) {
            out_lp[u] = shift + (double)best;
            int s = arg;
            path[f0 + T_len - 1] = s;
            const uint8_t* bq = bp + f0 * S;
            for (int t = T_len - 2; t >= 0; --t) {
                s = bq[(int64_t)(t + 1) * S + s];
                path[f0 + t] = s;
            }
        }
    }
}

// Viterbi for models of at most 8 states, float32: ONE LANE per utterance.  The lane
// keeps log A (S x S) and delta (S) in registers, so a step is S^2 add / compare /
// select triples without any cross-lane traffic; log-densities are read four frames
// at a time (one 4 S-float piece), the S back-pointers of a frame are packed into one
// word (3 bits each), written four frames at a time, and the backtrace resolves four
// frames per load instead of one dependent byte load per frame.  Ties take the lowest
// state, as np.argmax does (utils.py:243-266).
template <int S>
__global__ __launch_bounds__(64) void viterbi_lane_kernel(
    const float* __restrict__ loglik, const int64_t* __restrict__ off, int64_t U,
    const int32_t* __restrict__ utt_model, Tables<float> tb, uint32_t* __restrict__ bpw,
    int32_t* __restrict__ path, double* __restrict__ out_lp) {
    const int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= U) return;
    const int m = utt_model ? utt_model[u] : 0;
    const int64_t f0 = off[u];
    const int T_len = (int)(off[u + 1] - f0);
    if (T_len <= 0) {
        out_lp[u] = 0.0;
        return;
    }
    struct __attribute__((packed, aligned(4))) Rows4 { float v[4 * S]; };   // rows are 4-byte aligned
    struct __attribute__((packed, aligned(4))) Words4 { uint32_t v[4]; };
    float la[S][S], d[S];
#pragma unroll
    for (int i = 0; i < S; ++i)
#pragma unroll
        for (int j = 0; j < S; ++j) la[i][j] = tb.logA[((size_t)m * S + i) * S + j];
    const float* ll = loglik + f0 * S;
    uint32_t* bw = bpw + f0;
#pragma unroll
    for (int j = 0; j < S; ++j) d[j] = tb.logpi[m * S + j] + ll[j];
    auto step = [&](const float* row) {
        float nd[S];
        uint32_t word = 0;
#pragma unroll
        for (int j = 0; j < S; ++j) {
            float best = neg_inf<float>();
            uint32_t arg = 0;
#pragma unroll
            for (int i = 0; i < S; ++i) {
                const float cand = d[i] + la[i][j];
                arg = cand > best ? (uint32_t)i : arg;       // strict: the lowest index wins ties
                best = cand > best ? cand : best;
            }
            nd[j] = best + row[j];
            word |= arg << (3 * j);
        }
#pragma unroll
        for (int j = 0; j < S; ++j) d[j] = nd[j];
        return word;
    };
    // frames 1 .. T-1 in pieces of four, the next piece requested before this one is used
    int t = 1;
    double shift = 0.0;                  // what has been taken off delta so far
    Rows4 cur, nxt;
    if (t + 4 <= T_len) cur = *(const Rows4*)(ll + (size_t)t * S);
    for (; t + 4 <= T_len; t += 4) {
        if (t + 8 <= T_len) nxt = *(const Rows4*)(ll + (size_t)(t + 4) * S);
        Words4 w;
#pragma unroll
        for (int k = 0; k < 4; ++k) w.v[k] = step(cur.v + k * S);
        *(Words4*)(bw + t) = w;
        cur = nxt;
        // delta is kept relative to its largest entry (the path does not change): a float32
        // lattice that grows by ~50 per frame would compare candidates with an ulp of 1e-3
        // after 200 frames and of 0.03 after 5000
        float top = d[0];
#pragma unroll
        for (int j = 1; j < S; ++j) top = fmaxf(top, d[j]);
        if (top > neg_inf<float>()) {
            shift += (double)top;
#pragma unroll
            for (int j = 0; j < S; ++j) d[j] -= top;
        }
    }
    for (; t < T_len; ++t) {
        float row[S];
#pragma unroll
        for (int j = 0; j < S; ++j) row[j] = ll[(size_t)t * S + j];
        bw[t] = step(row);
    }
    float best = neg_inf<float>();
    int s = 0;
#pragma unroll
    for (int i = 0; i < S; ++i)
        if (d[i] > best) {
            best = d[i];
            s = i;
        }
    out_lp[u] = shift + (double)best;
    // backtrace: path[t] = back-pointer of frame t+1 at the state of frame t+1
    int32_t* p = path + f0;
    p[T_len - 1] = s;
    t = T_len - 1;                       // frame whose word is looked up next
    for (; t >= 4; t -= 4) {
        const Words4 w = *(const Words4*)(bw + t - 3);       // words of frames t-3 .. t
        Words4 o;
#pragma unroll
        for (int k = 3; k >= 0; --k) {
            s = (int)((w.v[k] >> (3 * s)) & 7u);
            o.v[k] = (uint32_t)s;                             // state of frame t-4+k
        }
        *(Words4*)(p + t - 4) = o;
    }
    for (; t >= 1; --t) {
        s = (int)((bw[t] >> (3 * s)) & 7u);
        p[t - 1] = s;
    }
}

// Posteriors for models of at most 8 states, float32: ONE LANE per utterance, like
// viterbi_lane_kernel.  The transition matrix (both orientations of its use come from
// the same S x S registers), the scaled forward vector and b~ beta stay in registers;
// log-densities, the lattice (kept in the output array until the backward pass
// replaces it by the posteriors) and the scale factors move four frames at a time.
// The arithmetic is fb_chain_kernel's: b~ = exp(l - max l), alpha scaled to sum 1,
// beta with the same factors, gamma = alpha beta normalised (zero rows divide by 1).
// STATS: the E-step of small jobs.  The lane also sums xi of its utterance in S x S
// registers (float32 over one utterance, like the fused kernel over a batch) and adds
// them, gamma_0, the log-probability and a count to the workgroup's float64 record in
// LDS -> partial[blockIdx][M][S*S + S + 2], fb_chain_kernel's layout.
template <int S, bool STATS>
__global__ __launch_bounds__(64) void fb_lane_kernel(
    const float* __restrict__ loglik, const int64_t* __restrict__ off, int64_t U,
    const int32_t* __restrict__ utt_model, Tables<float> tb, float* __restrict__ lat,
    float* __restrict__ scale, double* __restrict__ utt_lp, double* __restrict__ partial, int M,
    const double* __restrict__ A64, const double* __restrict__ pi64) {
    // (A64 / pi64: read the transition and start probabilities from the float64 model
    // arrays instead of prepared float32 tables)
    extern __shared__ __attribute__((aligned(16))) double acc_lds[];
    constexpr int P = S * S + S + 2;
    if constexpr (STATS) {
        for (int i = threadIdx.x; i < M * P; i += blockDim.x) acc_lds[i] = 0.0;
        __syncthreads();
    }
    const int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = u < U;
    const int m = valid && utt_model ? utt_model[u] : 0;
    const int64_t f0 = valid ? off[u] : 0;
    const int T_len = valid ? (int)(off[u + 1] - f0) : 0;
    if (T_len <= 0 && valid && utt_lp) utt_lp[u] = 0.0;
    if (T_len > 0) {
    struct __attribute__((packed, aligned(4))) Rows4 { float v[4 * S]; };   // rows are 4-byte aligned
    struct __attribute__((packed, aligned(4))) Four { float v[4]; };
    float A[S][S];
#pragma unroll
    for (int i = 0; i < S; ++i)
#pragma unroll
        for (int j = 0; j < S; ++j)
            A[i][j] = A64 ? (float)A64[((size_t)m * S + i) * S + j] : tb.A[((size_t)m * S + i) * S + j];
    float pi[S];
#pragma unroll
    for (int j = 0; j < S; ++j) pi[j] = pi64 ? (float)pi64[m * S + j] : tb.pi[m * S + j];
    const float* ll = loglik + f0 * S;
    float* la = lat + f0 * S;
    float* sc = scale + f0;
    // b~ of one frame (and the row maximum it was shifted by)
    auto emis = [&](const float* row, float (&b)[S]) {
        float mx = row[0];
#pragma unroll
        for (int j = 1; j < S; ++j) mx = fmaxf(mx, row[j]);
#pragma unroll
        for (int j = 0; j < S; ++j) b[j] = __expf(row[j] - mx);      // (hardware exp2: ~1 ulp)
        return mx;
    };
    // ---- forward
    float alpha[S];
    double lp = 0.0;
    float lpf = 0.f;              // float partial sum of the log-probability, folded every 4 frames
    auto fwd = [&](bool first, const float* row, float* out_row) {
        float b[S], a[S];
        const float mx = emis(row, b);
        if (first) {
#pragma unroll
            for (int j = 0; j < S; ++j) a[j] = pi[j] * b[j];
        } else {
#pragma unroll
            for (int j = 0; j < S; ++j) {
                float acc = 0.f;
#pragma unroll
                for (int i = 0; i < S; ++i) acc += alpha[i] * A[i][j];
                a[j] = acc * b[j];
            }
        }
        float c = 0.f;
#pragma unroll
        for (int j = 0; j < S; ++j) c += a[j];
        lpf += mx + __logf(c);
        const float ci = __builtin_amdgcn_rcpf(c);
#pragma unroll
        for (int j = 0; j < S; ++j) {
            alpha[j] = a[j] * ci;
            out_row[j] = alpha[j];
        }
        return ci;
    };
    {
        float row[S], o[S];
#pragma unroll
        for (int j = 0; j < S; ++j) row[j] = ll[j];
        sc[0] = fwd(true, row, o);
#pragma unroll
        for (int j = 0; j < S; ++j) la[j] = o[j];
    }
    int t = 1;
    Rows4 cur, nxt;
    if (t + 4 <= T_len) cur = *(const Rows4*)(ll + (size_t)t * S);
    for (; t + 4 <= T_len; t += 4) {
        if (t + 8 <= T_len) nxt = *(const Rows4*)(ll + (size_t)(t + 4) * S);
        Rows4 o;
        Four c4;
#pragma unroll
        for (int k = 0; k < 4; ++k) c4.v[k] = fwd(false, cur.v + k * S, o.v + k * S);
        *(Rows4*)(la + (size_t)t * S) = o;
        *(Four*)(sc + t) = c4;
        cur = nxt;
        lp += (double)lpf;
        lpf = 0.f;
    }
    for (; t < T_len; ++t) {
        float row[S], o[S];
#pragma unroll
        for (int j = 0; j < S; ++j) row[j] = ll[(size_t)t * S + j];
        sc[t] = fwd(false, row, o);
#pragma unroll
        for (int j = 0; j < S; ++j) la[(size_t)t * S + j] = o[j];
    }
    lp += (double)lpf;
    if (utt_lp) utt_lp[u] = lp;
    // ---- backward: frames T-1 .. 0; bb = b~_{t+1} beta_{t+1}
    float bb[S], gam0[S];
    float xi[STATS ? S : 1][STATS ? S : 1];
#pragma unroll
    for (int j = 0; j < S; ++j) bb[j] = gam0[j] = 0.f;
    if constexpr (STATS) {
#pragma unroll
        for (int i = 0; i < S; ++i)
#pragma unroll
            for (int j = 0; j < S; ++j) xi[i][j] = 0.f;
    }
    auto bwd = [&](bool last, const float* row, const float* arow, float ci, float* out_row) {
        float beta[S];
        if (last) {
#pragma unroll
            for (int i = 0; i < S; ++i) beta[i] = ci;
        } else {
#pragma unroll
            for (int i = 0; i < S; ++i) {
                float acc = 0.f;
#pragma unroll
                for (int j = 0; j < S; ++j) {
                    const float w = A[i][j] * bb[j];
                    acc += w;
                    if constexpr (STATS) xi[i][j] += arow[i] * w;
                }
                beta[i] = finite_max(acc * ci);
            }
        }
        float g[S], gs = 0.f;
#pragma unroll
        for (int i = 0; i < S; ++i) {
            g[i] = arow[i] * beta[i];
            gs += g[i];
        }
        if (gs == 0.f) gs = 1.f;                 // utils.normalize: zero rows divide by 1
        const float gi = __builtin_amdgcn_rcpf(gs);
        float b[S];
        emis(row, b);
#pragma unroll
        for (int i = 0; i < S; ++i) {
            out_row[i] = g[i] * gi;
            if constexpr (STATS) gam0[i] = out_row[i];
            bb[i] = b[i] * beta[i];
        }
    };
    t = T_len - 1;
    bool last = true;
    // (the piece below the current one is requested before the current one is used)
    Rows4 lr, ar, lr_n, ar_n;
    Four c4, c4_n;
    if (t >= 3) {
        lr = *(const Rows4*)(ll + (size_t)(t - 3) * S);
        ar = *(const Rows4*)(la + (size_t)(t - 3) * S);
        c4 = *(const Four*)(sc + t - 3);
    }
    for (; t >= 3; t -= 4) {
        if (t >= 7) {
            lr_n = *(const Rows4*)(ll + (size_t)(t - 7) * S);
            ar_n = *(const Rows4*)(la + (size_t)(t - 7) * S);
            c4_n = *(const Four*)(sc + t - 7);
        }
        Rows4 o;
#pragma unroll
        for (int k = 3; k >= 0; --k) {
            bwd(last, lr.v + k * S, ar.v + k * S, c4.v[k], o.v + k * S);
            last = false;
        }
        *(Rows4*)(la + (size_t)(t - 3) * S) = o;
        lr = lr_n;
        ar = ar_n;
        c4 = c4_n;
        if constexpr (STATS) {
            // (the float32 xi sums never hold more than 2048 frames)
            if ((t & 2047) < 4) {
                double* a = acc_lds + (size_t)m * P;
#pragma unroll
                for (int i = 0; i < S; ++i)
#pragma unroll
                    for (int j = 0; j < S; ++j) {
                        atomicAdd(&a[i * S + j], (double)xi[i][j]);
                        xi[i][j] = 0.f;
                    }
            }
        }
    }
    for (; t >= 0; --t) {
        float row[S], ar[S], o[S];
#pragma unroll
        for (int j = 0; j < S; ++j) {
            row[j] = ll[(size_t)t * S + j];
            ar[j] = la[(size_t)t * S + j];
        }
        bwd(last, row, ar, sc[t], o);
        last = false;
#pragma unroll
        for (int j = 0; j < S; ++j) la[(size_t)t * S + j] = o[j];
    }
    if constexpr (STATS) {
        double* a = acc_lds + (size_t)m * P;
#pragma unroll
        for (int i = 0; i < S; ++i) {
#pragma unroll
            for (int j = 0; j < S; ++j)
                if (T_len > 1) atomicAdd(&a[i * S + j], (double)xi[i][j]);
            atomicAdd(&a[S * S + i], (double)gam0[i]);
        }
        atomicAdd(&a[S * S + S], lp);
        atomicAdd(&a[S * S + S + 1], 1.0);
    }
    }
    if constexpr (STATS) {
        __syncthreads();
        double* dst = partial + (size_t)blockIdx.x * M * P;
        for (int i = threadIdx.x; i < M * P; i += blockDim.x) dst[i] = acc_lds[i];
    }
}

// 1, 2 or 4 floats as one register group and one memory access
template <int V> struct LaneVec;
template <> struct LaneVec<1> {
    typedef float type;
    static __device__ float get(const type& x, int) { return x; }
    static __device__ void set(type& x, int, float v) { x = v; }
    static __device__ type load(__amdgpu_buffer_rsrc_t r, unsigned off) {
        return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 0));
    }
    static __device__ void store(type x, __amdgpu_buffer_rsrc_t r, unsigned off) {
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(x), r, off, 0, 0);
    }
};
template <> struct LaneVec<2> {
    typedef float type __attribute__((ext_vector_type(2)));
    typedef unsigned bits __attribute__((ext_vector_type(2)));
    static __device__ float get(const type& x, int w) { return x[w]; }
    static __device__ void set(type& x, int w, float v) { x[w] = v; }
    static __device__ type load(__amdgpu_buffer_rsrc_t r, unsigned off) {
        return (type)__builtin_amdgcn_raw_buffer_load_b64(r, off, 0, 0);
    }
    static __device__ void store(type x, __amdgpu_buffer_rsrc_t r, unsigned off) {
        __builtin_amdgcn_raw_buffer_store_b64((bits)x, r, off, 0, 0);
    }
};
template <> struct LaneVec<4> {
    typedef float type __attribute__((ext_vector_type(4)));
    typedef unsigned bits __attribute__((ext_vector_type(4)));
    static __device__ float get(const type& x, int w) { return x[w]; }
    static __device__ void set(type& x, int w, float v) { x[w] = v; }
    static __device__ type load(__amdgpu_buffer_rsrc_t r, unsigned off) {
        return (type)__builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0);
    }
    static __device__ void store(type x, __amdgpu_buffer_rsrc_t r, unsigned off) {
        __builtin_amdgcn_raw_buffer_store_b128((bits)x, r, off, 0, 0);
    }
};

// Orders the LDS accesses of a one-wave workgroup (its lanes run in lockstep and its LDS
// operations complete in order): no hardware barrier, and unlike __syncthreads no wait for
// the wave's outstanding global loads and stores.
__device__ __forceinline__ void tile_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Posteriors for models of at most 8 states, float32, one lane per utterance with
// fb_lane_kernel's arithmetic, arranged around what its memory traffic costs.  (a) Every
// access goes through an LDS tile of C = 8 (16 for S <= 4) frames x 64 utterances that the
// wave fills and drains together, each utterance's piece as one contiguous run of up to
// 256 bytes: a lane reading its own rows straight from memory touches 64 cache lines per
// instruction, and that look-up rate, not the bytes, is what bounds fb_lane_kernel.
// (b) The scaled forward vectors are not kept: the forward pass leaves only the vector
// after the last frame of every tile (in that frame's row of the output array, which the
// backward pass overwrites later) and the log-probability; the backward pass recomputes
// the forward vectors and scale factors of a tile from that checkpoint -- 104 bytes per
// frame through memory at 8 states instead of 160, for one more forward recursion in
// registers.  gamma and loglik must be aligned to the access width (16 bytes when S is a
// multiple of 4, 8 when even) and shorter than 4 GiB (n_bytes = their size).
template <int S>
__global__ __launch_bounds__(64) void posteriors_lane_kernel(
    const float* __restrict__ loglik, const int64_t* __restrict__ off, int64_t U,
    const int32_t* __restrict__ utt_model, Tables<float> tb, float* __restrict__ gamma,
    double* __restrict__ utt_lp, unsigned n_bytes) {
    // frames per tile: 16 for the smallest models, whose recursion over 8 frames is shorter
    // than the memory round trip of the tile fetched meanwhile
    constexpr int C = S <= 4 ? 16 : 8;
    constexpr int V = S % 4 == 0 ? 4 : S % 2 == 0 ? 2 : 1;      // dwords per access
    constexpr int UNITS = C * S / V;                            // accesses per utterance and tile
    constexpr int UPI = 64 / UNITS;                             // utterances per instruction
    constexpr int NI = (64 + UPI - 1) / UPI;                    // instructions per tile
    constexpr int STRIDE = C * S + V;                           // floats per utterance in LDS
    typedef typename LaneVec<V>::type Vec;
    __shared__ __attribute__((aligned(16))) float tile[64 * STRIDE];
    __shared__ uint2 s_span[64];               // an utterance's first byte in the arrays, and its end

    const int lane = threadIdx.x;
    const int64_t u = (int64_t)blockIdx.x * 64 + lane;
    const bool valid = u < U;
    const int m = valid && utt_model ? utt_model[u] : 0;
    const int64_t f0 = valid ? off[u] : 0;
    const int T_len = valid ? (int)(off[u + 1] - f0) : 0;
    s_span[lane] = make_uint2((unsigned)(f0 * S * 4), (unsigned)((f0 + T_len) * S * 4));
    int n_tiles = (T_len + C - 1) / C;
#pragma unroll
    for (int o = 32; o; o >>= 1) n_tiles = max(n_tiles, __shfl_xor(n_tiles, o));
    n_tiles = __builtin_amdgcn_readfirstlane(n_tiles);          // (loop bounds in scalar registers)
    tile_sync();

    float A[S][S], pi[S];
#pragma unroll
    for (int i = 0; i < S; ++i)
#pragma unroll
        for (int j = 0; j < S; ++j) A[i][j] = tb.A[((size_t)m * S + i) * S + j];
#pragma unroll
    for (int j = 0; j < S; ++j) pi[j] = tb.pi[m * S + j];

    // The wave's share of moving a tile: instruction i carries UPI utterances, lane (sub, unit)
    // the unit-th access of utterance i * UPI + sub (UPI is a power of two for every S: the
    // instructions cover the 64 utterances exactly).  Both arrays are addressed as buffers of
    // n_bytes with 32-bit offsets, so an access past the end of the arrays returns zeros or
    // is dropped: a tile may be fetched beyond an utterance's end (into the next utterance's
    // rows, which the lane never uses), only the stores check the utterance's end.
    static_assert(NI * UPI == 64, "tile instructions must cover the wave's utterances");
    const __amdgpu_buffer_rsrc_t r_ll =
        __builtin_amdgcn_make_buffer_rsrc((void*)loglik, 0, n_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_g = __builtin_amdgcn_make_buffer_rsrc((void*)gamma, 0, n_bytes, 0x00020000);
    const int sub = lane / UNITS, unit = lane - sub * UNITS;
    const bool mover = sub < UPI;
    const unsigned lane_off = (unsigned)unit * V * 4;
    auto fetch = [&](int c, Vec (&r)[NI]) {
        const unsigned tile_off = (unsigned)c * (C * S * 4) + lane_off;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            unsigned o = s_span[(i * UPI + sub) & 63].x + tile_off;
            if constexpr (UPI * UNITS < 64) o = mover ? o : 0xffffffffu;
            r[i] = LaneVec<V>::load(r_ll, o);
        }
    };
    auto fill = [&](const Vec (&r)[NI]) {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int e = i * UPI + sub;
            if (mover) *(Vec*)(tile + e * STRIDE + unit * V) = r[i];
        }
    };
    auto drain = [&](int c) {
        const unsigned tile_off = (unsigned)c * (C * S * 4) + lane_off;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int e = (i * UPI + sub) & 63;
            const uint2 sp = s_span[e];
            const unsigned o = sp.x + tile_off;
            const Vec x = *(const Vec*)(tile + e * STRIDE + unit * V);
            LaneVec<V>::store(x, r_g, mover && o < sp.y ? o : 0xffffffffu);
        }
    };
    float* my = tile + lane * STRIDE;
    auto get_row = [&](int k, float (&row)[S]) {
#pragma unroll
        for (int q = 0; q < S / V; ++q) {
            const Vec x = *(const Vec*)(my + k * S + q * V);
#pragma unroll
            for (int w = 0; w < V; ++w) row[q * V + w] = LaneVec<V>::get(x, w);
        }
    };
    // b~ = exp(l - max l) of one frame and the maximum it was shifted by
    auto emis = [&](const float (&row)[S], float (&b)[S]) {
        float mx = row[0];
#pragma unroll
        for (int j = 1; j < S; ++j) mx = fmaxf(mx, row[j]);
#pragma unroll
        for (int j = 0; j < S; ++j) b[j] = __expf(row[j] - mx);
        return mx;
    };
    // one forward step on alpha; returns 1 / c_t, leaves c_t in c_out
    float alpha[S];
    auto fwd = [&](bool first, const float (&b)[S], float& c_out) {
        float a[S];
        if (first) {
#pragma unroll
            for (int j = 0; j < S; ++j) a[j] = pi[j] * b[j];
        } else {
#pragma unroll
            for (int j = 0; j < S; ++j) {
                float acc = 0.f;
#pragma unroll
                for (int i = 0; i < S; ++i) acc += alpha[i] * A[i][j];
                a[j] = acc * b[j];
            }
        }
        float c = 0.f;
#pragma unroll
        for (int j = 0; j < S; ++j) c += a[j];
        c_out = c;
        const float ci = __builtin_amdgcn_rcpf(c);
#pragma unroll
        for (int j = 0; j < S; ++j) alpha[j] = a[j] * ci;
        return ci;
    };

    // ---- forward: log-probability and a checkpoint per tile
    Vec R[NI];
    double lp = 0.0;
    float lpf = 0.f;                                  // folded into lp after frames 4, 8, ...
    if (n_tiles > 0) fetch(0, R);
    for (int c = 0; c < n_tiles; ++c) {
        fill(R);
        tile_sync();
        if (c + 1 < n_tiles) fetch(c + 1, R);
#pragma unroll
        for (int k = 0; k < C; ++k) {
            const int t = c * C + k;
            if (t < T_len) {
                float row[S], b[S], cs;
                get_row(k, row);
                const float mx = emis(row, b);
                fwd(c == 0 && k == 0, b, cs);
                lpf += mx + __logf(cs);
                if ((k & 3) == 0 && t > 0) {
                    lp += (double)lpf;
                    lpf = 0.f;
                }
            }
        }
        if ((c + 1) * C < T_len) {
            float* o = gamma + (f0 + (int64_t)c * C + C - 1) * S;
#pragma unroll
            for (int q = 0; q < S / V; ++q) {
                Vec x;
#pragma unroll
                for (int w = 0; w < V; ++w) LaneVec<V>::set(x, w, alpha[q * V + w]);
                *(Vec*)(o + q * V) = x;
            }
        }
        tile_sync();                              // the tile is refilled next
    }
    lp += (double)lpf;
    if (valid && utt_lp) utt_lp[u] = lp;

    // ---- backward: tiles from the last to the first; bb = b~_{t+1} beta_{t+1}
    float bb[S];
#pragma unroll
    for (int j = 0; j < S; ++j) bb[j] = 0.f;
    Vec ck[S / V], ck_n[S / V];
    auto checkpoint = [&](int c, Vec (&r)[S / V]) {
        const float* p = c > 0 && c * C < T_len ? gamma + (f0 + (int64_t)c * C - 1) * S : gamma;
#pragma unroll
        for (int q = 0; q < S / V; ++q) r[q] = *(const Vec*)(p + q * V);
    };
    if (n_tiles > 0) {
        fetch(n_tiles - 1, R);
        checkpoint(n_tiles - 1, ck);
    }
    for (int c = n_tiles - 1; c >= 0; --c) {
        fill(R);
        tile_sync();
        if (c > 0) {
            fetch(c - 1, R);
            checkpoint(c - 1, ck_n);
        }
        // the forward vectors, scale factors and b~ of the tile's frames again
        float al[C][S], bq[C][S], cinv[C];
#pragma unroll
        for (int j = 0; j < S; ++j) alpha[j] = LaneVec<V>::get(ck[j / V], j % V);
#pragma unroll
        for (int k = 0; k < C; ++k) {
            const int t = c * C + k;
            if (t < T_len) {
                float row[S], cs;
                get_row(k, row);
                emis(row, bq[k]);
                cinv[k] = fwd(c == 0 && k == 0, bq[k], cs);
#pragma unroll
                for (int j = 0; j < S; ++j) al[k][j] = alpha[j];
            }
        }
#pragma unroll
        for (int k = C - 1; k >= 0; --k) {
            const int t = c * C + k;
            if (t < T_len) {
                const bool last = t == T_len - 1;
                float g[S], gs = 0.f, beta[S];
#pragma unroll
                for (int i = 0; i < S; ++i) {
                    float acc = 0.f;
#pragma unroll
                    for (int j = 0; j < S; ++j) acc += A[i][j] * bb[j];
                    beta[i] = last ? cinv[k] : finite_max(acc * cinv[k]);
                    g[i] = al[k][i] * beta[i];
                    gs += g[i];
                }
                if (gs == 0.f) gs = 1.f;              // utils.normalize: zero rows divide by 1
                const float gi = __builtin_amdgcn_rcpf(gs);
#pragma unroll
                for (int q = 0; q < S / V; ++q) {
                    Vec x;
#pragma unroll
                    for (int w = 0; w < V; ++w) LaneVec<V>::set(x, w, g[q * V + w] * gi);
                    *(Vec*)(my + k * S + q * V) = x;
                }
#pragma unroll
                for (int i = 0; i < S; ++i) bb[i] = bq[k][i] * beta[i];
            }
        }
        tile_sync();
        drain(c);
        tile_sync();
#pragma unroll
        for (int q = 0; q < S / V; ++q) ck[q] = ck_n[q];
    }
}

// Float64 E-step of small jobs.  A pass over all frames first replaces every row of
// log-densities by b~ = exp(l - max l) and keeps the maxima (the software float64
// exponentials run in parallel there instead of inside the recursion); then one lane per
// utterance runs the two recursions on b~, like fb_lane_kernel<S, true> -- product of the
// scale factors as mantissa and exponent (LogProduct), one logarithm per utterance.
__global__ void shift_exp_f64_kernel(double* __restrict__ ll, int64_t n_frames, int S,
                                     double* __restrict__ mx_out) {
    const int64_t f = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= n_frames) return;
    double* row = ll + f * S;
    double mx = row[0];
    for (int j = 1; j < S; ++j) mx = fmax(mx, row[j]);
    for (int j = 0; j < S; ++j) row[j] = exp(row[j] - mx);
    mx_out[f] = mx;
}

template <int S>
__global__ __launch_bounds__(64) void fb_lane_f64_kernel(
    const double* __restrict__ btil, const double* __restrict__ mxv, const int64_t* __restrict__ off,
    int64_t U, const int32_t* __restrict__ utt_model, const double* __restrict__ A64,
    const double* __restrict__ pi64, double* __restrict__ lat, double* __restrict__ scale,
    double* __restrict__ partial, int M) {
    extern __shared__ __attribute__((aligned(16))) double acc_lds[];
    constexpr int P = S * S + S + 2;
    for (int i = threadIdx.x; i < M * P; i += blockDim.x) acc_lds[i] = 0.0;
    __syncthreads();
    const int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = u < U;
    const int m = valid && utt_model ? utt_model[u] : 0;
    const int64_t f0 = valid ? off[u] : 0;
    const int T_len = valid ? (int)(off[u + 1] - f0) : 0;
    if (T_len > 0) {
        struct Rows4 { double v[4 * S]; };
        struct Four { double v[4]; };
        double A[S][S];
#pragma unroll
        for (int i = 0; i < S; ++i)
#pragma unroll
            for (int j = 0; j < S; ++j) A[i][j] = A64[((size_t)m * S + i) * S + j];
        const double* bt = btil + f0 * S;
        const double* mx = mxv + f0;
        double* la = lat + f0 * S;
        double* sc = scale + f0;
        // ---- forward (rows move four frames at a time, the next four requested before
        // the current four are used)
        double alpha[S];
        LogProduct lpp;
        auto fwd = [&](bool first, const double* row, double mxt, double* out_row) {
            double a[S], c = 0.0;
#pragma unroll
            for (int j = 0; j < S; ++j) {
                double acc;
                if (first) {
                    acc = pi64[m * S + j];
                } else {
                    acc = 0.0;
#pragma unroll
                    for (int i = 0; i < S; ++i) acc += alpha[i] * A[i][j];
                }
                a[j] = acc * row[j];
                c += a[j];
            }
            lpp.step(mxt, c);
            const double ci = 1.0 / c;
#pragma unroll
            for (int j = 0; j < S; ++j) {
                alpha[j] = a[j] * ci;
                out_row[j] = alpha[j];
            }
            return ci;
        };
        {
            double row[S], o[S];
#pragma unroll
            for (int j = 0; j < S; ++j) row[j] = bt[j];
            sc[0] = fwd(true, row, mx[0], o);
#pragma unroll
            for (int j = 0; j < S; ++j) la[j] = o[j];
        }
        int t = 1;
        Rows4 cur, nxt;
        Four mcur, mnxt;
        if (t + 4 <= T_len) {
            cur = *(const Rows4*)(bt + (size_t)t * S);
            mcur = *(const Four*)(mx + t);
        }
        for (; t + 4 <= T_len; t += 4) {
            if (t + 8 <= T_len) {
                nxt = *(const Rows4*)(bt + (size_t)(t + 4) * S);
                mnxt = *(const Four*)(mx + t + 4);
            }
            Rows4 o;
            Four c4;
#pragma unroll
            for (int k = 0; k < 4; ++k) c4.v[k] = fwd(false, cur.v + k * S, mcur.v[k], o.v + k * S);
            *(Rows4*)(la + (size_t)t * S) = o;
            *(Four*)(sc + t) = c4;
            cur = nxt;
            mcur = mnxt;
        }
        for (; t < T_len; ++t) {
            double row[S], o[S];
#pragma unroll
            for (int j = 0; j < S; ++j) row[j] = bt[(size_t)t * S + j];
            sc[t] = fwd(false, row, mx[t], o);
#pragma unroll
            for (int j = 0; j < S; ++j) la[(size_t)t * S + j] = o[j];
        }
        const double lp = lpp.value();
        // ---- backward: bb = b~_{t+1} beta_{t+1}
        double bb[S], gam0[S], xi[S][S];
#pragma unroll
        for (int i = 0; i < S; ++i) {
            bb[i] = gam0[i] = 0.0;
#pragma unroll
            for (int j = 0; j < S; ++j) xi[i][j] = 0.0;
        }
        auto bwd = [&](bool last, const double* row, const double* arow, double ci, double* out_row) {
            double beta[S], g[S], gs = 0.0;
#pragma unroll
            for (int i = 0; i < S; ++i) {
                if (last) {
                    beta[i] = ci;
                } else {
                    double acc = 0.0;
#pragma unroll
                    for (int j = 0; j < S; ++j) {
                        const double w = A[i][j] * bb[j];
                        acc += w;
                        xi[i][j] += arow[i] * w;
                    }
                    beta[i] = finite_max(acc * ci);
                }
                g[i] = arow[i] * beta[i];
                gs += g[i];
            }
            if (gs == 0.0) gs = 1.0;                  // utils.normalize: zero rows divide by 1
            const double gi = 1.0 / gs;
#pragma unroll
            for (int i = 0; i < S; ++i) {
                out_row[i] = g[i] * gi;
                gam0[i] = out_row[i];
                bb[i] = row[i] * beta[i];
            }
        };
        t = T_len - 1;
        bool last = true;
        Rows4 lr, ar, lr_n, ar_n;
        Four c4, c4_n;
        if (t >= 3) {
            lr = *(const Rows4*)(bt + (size_t)(t - 3) * S);
            ar = *(const Rows4*)(la + (size_t)(t - 3) * S);
            c4 = *(const Four*)(sc + t - 3);
        }
        for (; t >= 3; t -= 4) {
            if (t >= 7) {
                lr_n = *(const Rows4*)(bt + (size_t)(t - 7) * S);
                ar_n = *(const Rows4*)(la + (size_t)(t - 7) * S);
                c4_n = *(const Four*)(sc + t - 7);
            }
            Rows4 o;
#pragma unroll
            for (int k = 3; k >= 0; --k) {
                bwd(last, lr.v + k * S, ar.v + k * S, c4.v[k], o.v + k * S);
                last = false;
            }
            *(Rows4*)(la + (size_t)(t - 3) * S) = o;
            lr = lr_n;
            ar = ar_n;
            c4 = c4_n;
        }
        for (; t >= 0; --t) {
            double row[S], ar1[S], o[S];
#pragma unroll
            for (int j = 0; j < S; ++j) {
                row[j] = bt[(size_t)t * S + j];
                ar1[j] = la[(size_t)t * S + j];
            }
            bwd(last, row, ar1, sc[t], o);
            last = false;
#pragma unroll
            for (int j = 0; j < S; ++j) la[(size_t)t * S + j] = o[j];
        }
        double* a = acc_lds + (size_t)m * P;
#pragma unroll
        for (int i = 0; i < S; ++i) {
#pragma unroll
            for (int j = 0; j < S; ++j)
                if (T_len > 1) atomicAdd(&a[i * S + j], xi[i][j]);
            atomicAdd(&a[S * S + i], gam0[i]);
        }
        atomicAdd(&a[S * S + S], lp);
        atomicAdd(&a[S * S + S + 1], 1.0);
    }
    __syncthreads();
    double* dst = partial + (size_t)blockIdx.x * M * P;
    for (int i = threadIdx.x; i < M * P; i += blockDim.x) dst[i] = acc_lds[i];
}

// One workgroup per model.
__global__ void mstep_kernel(const double* __restrict__ stats, srx_priors_t pr, int mode,
                             srx_models_t md) {
    const int S = md.n_states, D = md.n_features, m = blockIdx.x;
    const srx_stats_layout L(S, D);
    const double* st = stats + (size_t)m * L.size;
    double* pi = md.startprob + (size_t)m * S;
    double* A = md.transmat + (size_t)m * S * S;
    __shared__ double rowsum[65];
    // start probabilities (hmm.py:445-448): structural zeros stay zero; the 1-D
    // normalise divides by the plain sum.
    if (threadIdx.x == 0) {
        double s = 0.0;
        for (int i = 0; i < S; ++i) {
            double v = fmax(pr.startprob_prior - 1.0 + st[L.start + i], 0.0);
            if (pi[i] == 0.0) v = 0.0;
            pi[i] = v;
            s += v;
        }
        rowsum[64] = s;
    }
    // transition rows (hmm.py:449-451): zero rows divide by 1.
    if (threadIdx.x < S) {
        const int i = threadIdx.x;
        double s = 0.0;
        for (int j = 0; j < S; ++j) {
            double v = fmax(pr.transmat_prior - 1.0 + st[L.trans + i * S + j], 0.0);
            if (A[i * S + j] == 0.0) v = 0.0;
            A[i * S + j] = v;
            s += v;
        }
        rowsum[i] = s == 0.0 ? 1.0 : s;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < S; i += blockDim.x) pi[i] /= rowsum[64];
    for (int i = threadIdx.x; i < S * S; i += blockDim.x) A[i] /= rowsum[i / S];
    if (mode != SRX_MSTEP_FULL) return;
    // means and diagonal covariances (hmm.py:168-184).
    double* mu = md.means + (size_t)m * S * D;
    double* cv = md.covars + (size_t)m * S * D;
    for (int i = threadIdx.x; i < S * D; i += blockDim.x) {
        const double denom = st[L.post + i / D];
        const double obs = st[L.obs + i], obs2 = st[L.obs2 + i];
        const double mean = (pr.means_weight * pr.means_prior + obs) / (pr.means_weight + denom);
        const double diff = mean - pr.means_prior;
        const double cn = pr.means_weight * diff * diff + obs2 - 2.0 * mean * obs +
                          mean * mean * denom;
        const double cd = fmax(pr.covars_weight - 1.0, 0.0) + denom;
        mu[i] = mean;
        cv[i] = (pr.covars_prior + cn) / fmax(cd, 1e-5);
    }
}

int sub_width(int S) {
    int sp = 2;
    while (sp < S) sp <<= 1;
    return sp;
}

template <typename T>
int prepare(srx_context* h, const srx_models_t* md, Tables<T>* tb, hipStream_t st) {
    void* base;
    int rc = srx_ws_ensure(h, WS_PARAMS, tables_bytes<T>(md->n_models, md->n_states, md->n_features),
                           &base);
    if (rc) return rc;
    *tb = carve_tables<T>(base, md->n_models, md->n_states, md->n_features);
    hipLaunchKernelGGL(prep_kernel<T>, dim3(8), dim3(kBlock), 0, st, *md, (T*)tb->mu, (T*)tb->hinv,
                       (T*)tb->cst, (T*)tb->A, (T*)tb->pi, (T*)tb->logA, (T*)tb->logpi);
    SRX_HIP(hipGetLastError());
    return SRX_OK;
}

int grid_for(srx_context* h, int64_t items, int per_block) {
    int64_t g = (items + per_block - 1) / per_block;
    const int64_t cap = (int64_t)h->n_cu * 8;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}

#define SP_DISPATCH(SPV, CALL)                     \
    switch (SPV) {                                 \
        case 2: { constexpr int SP = 2; CALL; } break;   \
        case 4: { constexpr int SP = 4; CALL; } break;   \
        case 8: { constexpr int SP = 8; CALL; } break;   \
        case 16: { constexpr int SP = 16; CALL; } break; \
        case 32: { constexpr int SP = 32; CALL; } break; \
        default: { constexpr int SP = 64; CALL; } break; \
    }

template <typename T>
int emission_t(srx_context* h, const void* X, int64_t n_frames, const srx_models_t* md, void* out,
               hipStream_t st) {
    Tables<T> tb;
    int rc = prepare<T>(h, md, &tb, st);
    if (rc) return rc;
    // A single pseudo-utterance spanning all frames would serialise on one
    // workgroup; cut the frame range into fixed slices instead.
    const int64_t slice = 256;
    const int64_t n_slices = (n_frames + slice - 1) / slice;
    int64_t* offs;
    rc = srx_ws_ensure(h, WS_MISC, sizeof(int64_t) * (n_slices + 1), (void**)&offs);
    if (rc) return rc;
    hipLaunchKernelGGL(srx_fill_slices, dim3(grid_for(h, n_slices + 1, kBlock)), dim3(kBlock), 0, st,
                       offs, n_slices, slice, n_frames);
    hipLaunchKernelGGL(emission_kernel<T>, dim3(grid_for(h, n_slices, 1)), dim3(kBlock), 0, st,
                       (const T*)X, (const int64_t*)offs, n_slices, (const int32_t*)nullptr, 1,
                       md->n_models, md->n_states, md->n_features, tb, (T*)out);
    SRX_HIP(hipGetLastError());
    return SRX_OK;
}

template <typename T>
int score_t(srx_context* h, const void* X, const int64_t* off, int64_t U, int64_t n_frames,
            const srx_models_t* md, double* out, hipStream_t st) {
    const int M = md->n_models, S = md->n_states, D = md->n_features;
    Tables<T> tb;
    int rc = prepare<T>(h, md, &tb, st);
    if (rc) return rc;
    T* ll;
    rc = srx_ws_ensure(h, WS_LOGLIK, sizeof(T) * (size_t)n_frames * M * S, (void**)&ll);
    if (rc) return rc;
    hipLaunchKernelGGL(emission_kernel<T>, dim3(grid_for(h, U, 1)), dim3(kBlock), 0, st,
                       (const T*)X, off, U, (const int32_t*)nullptr, 1, M, S, D, tb, ll);
    const int sp = sub_width(S);
    SP_DISPATCH(sp, hipLaunchKernelGGL((score_chain_kernel<T, SP>),
                                       dim3(grid_for(h, U * M, kBlock / SP)), dim3(kBlock), 0, st,
                                       (const T*)ll, off, U, M, S, tb, out));
    SRX_HIP(hipGetLastError());
    return SRX_OK;
}

template <typename T>
int viterbi_t(srx_context* h, const void* X, const int64_t* off, int64_t U, int64_t n_frames,
              const int32_t* utt_model, const srx_models_t* md, int32_t* path, double* lp,
              hipStream_t st) {
    const int M = md->n_models, S = md->n_states, D = md->n_features;
    Tables<T> tb;
    int rc = prepare<T>(h, md, &tb, st);
    if (rc) return rc;
    if constexpr (std::is_same<T, float>::value) {
        // emission and recursion in one kernel where the shape allows it
        rc = srx_fused_viterbi(h, X, off, U, n_frames, utt_model, md, tb.logA, tb.logpi, path, lp, st);
        if (rc != SRX_EUNSUPPORTED) return rc;
    }
    T* ll;
    uint8_t* bp;
    rc = srx_ws_ensure(h, WS_LOGLIK, sizeof(T) * (size_t)n_frames * S, (void**)&ll);
    if (rc) return rc;
    rc = srx_ws_ensure(h, WS_BACKPTR, (size_t)n_frames * S, (void**)&bp);
    if (rc) return rc;
    bool done = false;
    if constexpr (std::is_same<T, float>::value) {
        // matrix-core emission where the shape allows it
        rc = srx_mfma_emission_utt(h, X, off, U, n_frames, utt_model, md, (float*)ll, st);
        if (rc == SRX_OK) done = true;
        else if (rc != SRX_EUNSUPPORTED) return rc;
    }
    if constexpr (std::is_same<T, double>::value) {
        rc = srx_mfma_emission_utt_f64(h, X, off, U, n_frames, utt_model, md, (double*)ll, st);
        if (rc == SRX_OK) done = true;
        else if (rc != SRX_EUNSUPPORTED) return rc;
    }
    if (!done)
        hipLaunchKernelGGL(emission_kernel<T>, dim3(grid_for(h, U, 1)), dim3(kBlock), 0, st,
                           (const T*)X, off, U, utt_model, 0, M, S, D, tb, ll);
    if constexpr (std::is_same<T, float>::value) {
        if (S <= 8) {
            // (the back-pointer buffer holds one packed word per frame here)
            uint32_t* bpw;
            rc = srx_ws_ensure(h, WS_BACKPTR, sizeof(uint32_t) * (size_t)n_frames, (void**)&bpw);
            if (rc) return rc;
            const dim3 grid((unsigned)((U + 63) / 64));
#define SRX_CASE(s_)                                                                          \
    if (S == s_)                                                                              \
        hipLaunchKernelGGL((viterbi_lane_kernel<s_>), grid, dim3(64), 0, st, (const float*)ll, \
                           off, U, utt_model, tb, bpw, path, lp);
            SRX_CASE(1) SRX_CASE(2) SRX_CASE(3) SRX_CASE(4) SRX_CASE(5) SRX_CASE(6) SRX_CASE(7)
            SRX_CASE(8)
#undef SRX_CASE
            SRX_HIP(hipGetLastError());
            return SRX_OK;
        }
    }
    const int sp = sub_width(S);
    SP_DISPATCH(sp, hipLaunchKernelGGL((viterbi_chain_kernel<T, SP>),
                                       dim3(grid_for(h, U, kBlock / SP)), dim3(kBlock), 0, st,
                                       (const T*)ll, off, U, utt_model, S, tb, bp, path, lp));
    SRX_HIP(hipGetLastError());
    return SRX_OK;
}

#ifdef SRX_POSTERIORS_DIRECT
constexpr bool kPosteriorsTiled = false;     // (measurement builds: fb_lane_kernel instead)
#else
constexpr bool kPosteriorsTiled = true;
#endif

// Posteriors of every frame under its utterance's model: the forward-backward
// chain with the caller's array as the lattice, no statistics.
template <typename T>
int posteriors_t(srx_context* h, const void* X, const int64_t* off, int64_t U, int64_t n_frames,
                 const int32_t* utt_model, const srx_models_t* md, void* gamma, double* lp,
                 hipStream_t st) {
    const int M = md->n_models, S = md->n_states, D = md->n_features;
    Tables<T> tb;
    int rc = prepare<T>(h, md, &tb, st);
    if (rc) return rc;
    T *ll, *sc;
    rc = srx_ws_ensure(h, WS_LOGLIK, sizeof(T) * (size_t)n_frames * S, (void**)&ll);
    if (rc) return rc;
    rc = srx_ws_ensure(h, WS_SCALE, sizeof(T) * (size_t)n_frames, (void**)&sc);
    if (rc) return rc;
    // (plain float32 emission: the split-bfloat16 matrix-core emission is good to
    // 1e-4 in the log-density, which shows in small posteriors)
    bool done = false;
    if constexpr (std::is_same<T, float>::value) {
        rc = srx_mfma_emission_utt_f32(h, X, off, U, n_frames, utt_model, md, (float*)ll, st);
        if (rc == SRX_OK) done = true;
        else if (rc != SRX_EUNSUPPORTED) return rc;
    }
    if constexpr (std::is_same<T, double>::value) {
        rc = srx_mfma_emission_utt_f64(h, X, off, U, n_frames, utt_model, md, (double*)ll, st);
        if (rc == SRX_OK) done = true;
        else if (rc != SRX_EUNSUPPORTED) return rc;
    }
    if (!done)
        hipLaunchKernelGGL(emission_kernel<T>, dim3(grid_for(h, U, 1)), dim3(kBlock), 0, st,
                           (const T*)X, off, U, utt_model, 0, M, S, D, tb, ll);
    if constexpr (std::is_same<T, float>::value) {
        // (the tiled kernel addresses both arrays with 32-bit byte offsets)
        if (S <= 8 && ((uintptr_t)gamma & 15) == 0 && (uint64_t)n_frames * S * 4 < (1ull << 32) &&
            kPosteriorsTiled) {
            const dim3 grid((unsigned)((U + 63) / 64));
#define SRX_CASE(s_)                                                                       \
    if (S == s_)                                                                           \
        hipLaunchKernelGGL((posteriors_lane_kernel<s_>), grid, dim3(64), 0, st,            \
                           (const float*)ll, off, U, utt_model, tb, (float*)gamma, lp,        \
                           (unsigned)((uint64_t)n_frames * S * 4));
            SRX_CASE(1) SRX_CASE(2) SRX_CASE(3) SRX_CASE(4) SRX_CASE(5) SRX_CASE(6) SRX_CASE(7)
            SRX_CASE(8)
#undef SRX_CASE
            SRX_HIP(hipGetLastError());
            return SRX_OK;
        }
        if (S <= 8) {
            const int bs = 64;
            const dim3 grid((unsigned)((U + bs - 1) / bs));
#define SRX_CASE(s_)                                                                       \
    if (S == s_)                                                                           \
        hipLaunchKernelGGL((fb_lane_kernel<s_, false>), grid, dim3(bs), 0, st, (const float*)ll, \
                           off, U, utt_model, tb, (float*)gamma, (float*)sc, lp, (double*)nullptr, M, \
                           (const double*)nullptr, (const double*)nullptr);
            SRX_CASE(1) SRX_CASE(2) SRX_CASE(3) SRX_CASE(4) SRX_CASE(5) SRX_CASE(6) SRX_CASE(7)
            SRX_CASE(8)
#undef SRX_CASE
            SRX_HIP(hipGetLastError());
            return SRX_OK;
        }
    }
    const int sp = sub_width(S);
    SP_DISPATCH(sp, hipLaunchKernelGGL((fb_chain_kernel<T, SP>),
                                       dim3(grid_for(h, U, kBlock / SP)), dim3(kBlock), 0, st,
                                       (T*)ll, off, U, utt_model, (const int64_t*)nullptr, M,
                                       S, tb, (T*)gamma, sc, (double*)nullptr, lp, 1));
    SRX_HIP(hipGetLastError());
    return SRX_OK;
}

template <typename T>
int estep_t(srx_context* h, const void* X, const int64_t* off, int64_t U, int64_t n_frames,
            const int32_t* utt_model, const int64_t* utt_order, const srx_models_t* md,
            double* out_stats, hipStream_t st) {
    const int M = md->n_models, S = md->n_states, D = md->n_features;
    Tables<T> tb;
    int rc = prepare<T>(h, md, &tb, st);
    if (rc) return rc;
    T *ll, *lat, *sc;
    rc = srx_ws_ensure(h, WS_LOGLIK, sizeof(T) * (size_t)n_frames * S, (void**)&ll);
    if (rc) return rc;
    rc = srx_ws_ensure(h, WS_LATTICE, sizeof(T) * (size_t)n_frames * S, (void**)&lat);
    if (rc) return rc;
    rc = srx_ws_ensure(h, WS_SCALE, sizeof(T) * (size_t)n_frames, (void**)&sc);
    if (rc) return rc;
    const int sp = sub_width(S);
    const int P = S * S + S + 2, Q = S + 2 * S * D;
    const int g_fb = grid_for(h, U, kBlock / sp);
    int g_obs = grid_for(h, U, 4);
    if (g_obs > 1024) g_obs = 1024;
    // float64 on the matrix cores: one-wave workgroups, so twice as many slices
    const bool obs_mfma = std::is_same<T, double>::value && S <= 16 && D <= 48 && U > 0;
    if (obs_mfma) g_obs = (int)std::min<int64_t>(U, 2048);
    double* part;
    rc = srx_ws_ensure(h, WS_PARTIAL, sizeof(double) * ((size_t)g_fb * M * P + (size_t)g_obs * M * Q),
                       (void**)&part);
    if (rc) return rc;
    double* part_obs = part + (size_t)g_fb * M * P;
    size_t lds = sizeof(double) * (size_t)M * P;
    const int acc_in_lds = lds <= 64 * 1024;
    if (!acc_in_lds) lds = 0;
    bool emitted = false;
    if constexpr (std::is_same<T, double>::value) {
        // matrix-core emission where the shape allows it
        rc = srx_mfma_emission_utt_f64(h, X, off, U, n_frames, utt_model, md, (double*)ll, st);
        if (rc == SRX_OK) emitted = true;
        else if (rc != SRX_EUNSUPPORTED) return rc;
    }
    if (!emitted)
        hipLaunchKernelGGL(emission_kernel<T>, dim3(grid_for(h, U, 1)), dim3(kBlock), 0, st,
                           (const T*)X, off, U, utt_model, 0, M, S, D, tb, ll);
    SP_DISPATCH(sp, hipLaunchKernelGGL((fb_chain_kernel<T, SP>), dim3(g_fb), dim3(kBlock), lds, st,
                                       (T*)ll, off, U, utt_model, utt_order, M, S, tb, lat,
                                       sc, part, (double*)nullptr, acc_in_lds));
    const int slots = (S * D + kBlock - 1) / kBlock;
    SRX_REQUIRE(slots <= 16, "estep: n_states*n_features = %d exceeds %d", S * D, 16 * kBlock);
#define OBS(SL)                                                                                   \
    hipLaunchKernelGGL((obs_stats_kernel<T, SL>), dim3(g_obs), dim3(kBlock), 0, st, (const T*)X,  \
                       (const T*)lat, off, U, utt_model, utt_order, M, S, D, part_obs)
    // (a caller that re-estimates start and transition probabilities only gets zeros for
    // the sums over the features: no pass over X)
    const bool with_obs = h->estep_stats != SRX_STATS_TRANSITIONS;
    if (!with_obs) {
    } else if (obs_mfma) {
        rc = srx_mfma_obs_stats_f64(h, X, lat, off, U, utt_model, utt_order, md, part_obs, g_obs, st);
        if (rc) return rc;
    } else if (slots <= 1) OBS(1);
    else if (slots <= 2) OBS(2);
    else if (slots <= 4) OBS(4);
    else if (slots <= 8) OBS(8);
    else OBS(16);
#undef OBS
    // one thread per output sum
    const int64_t n_out = (int64_t)M * (P + Q);                  // one wave per output sum
    const int g_red = (int)std::min<int64_t>((n_out * 64 + kBlock - 1) / kBlock, 4096);
    hipLaunchKernelGGL(reduce_stats_kernel, dim3(g_red), dim3(kBlock), 0, st, (const double*)part, g_fb,
                       (const double*)part_obs, with_obs ? g_obs : 0, M, S, D, out_stats);
    SRX_HIP(hipGetLastError());
    return SRX_OK;
}

// E-step of small float32 jobs with at most 8 states: matrix-core emission of every frame
// at once, then ONE LANE per utterance for the two recursions (a few dozen instructions
// per frame where the fused kernel spends 72 matrix instructions per step of a wave that a
// small job cannot fill), then the observation sums over all frames.  A job this path does
// not want returns SRX_EUNSUPPORTED.
int estep_lane(srx_context* h, const void* X, const int64_t* off, int64_t U, int64_t n_frames,
               const int32_t* utt_model, const int64_t* utt_order, const srx_models_t* md,
               double* out_stats, hipStream_t st) {
    const int M = md->n_models, S = md->n_states, D = md->n_features;
    const bool asked = h->estep_path == SRX_ESTEP_LANE;      // any number of utterances then
    // measured crossovers against the fused kernel (utterances of 60 to 240 frames): 8192
    // utterances for 3 states, about 6000 for 5, 4096 for 8
    const int64_t max_utts = S <= 4 ? 8192 : (S <= 6 ? 4096 : 3072);
    const int P = S * S + S + 2, Q = S + 2 * S * D;
    const size_t lds = sizeof(double) * (size_t)M * P;
    if (S > 8 || D > 48 || U <= 0 || (U > max_utts && !asked) || n_frames <= 0 || lds > 32 * 1024)
        return SRX_EUNSUPPORTED;
    Tables<float> tb = {};           // (the lane kernel takes A and pi from the float64 model arrays)
    int rc;
    float *ll, *lat, *sc;
    if ((rc = srx_ws_ensure(h, WS_LOGLIK, sizeof(float) * (size_t)n_frames * S, (void**)&ll))) return rc;
    if ((rc = srx_ws_ensure(h, WS_LATTICE, sizeof(float) * (size_t)n_frames * S, (void**)&lat))) return rc;
    if ((rc = srx_ws_ensure(h, WS_SCALE, sizeof(float) * (size_t)n_frames, (void**)&sc))) return rc;
    const int g_fb = (int)((U + 63) / 64);
    const int g_obs = (int)std::min<int64_t>(U, 1024);      // one-wave workgroups, a slice of utterances each
    double* part;
    if ((rc = srx_ws_ensure(h, WS_PARTIAL, sizeof(double) * ((size_t)g_fb * M * P + (size_t)g_obs * M * Q),
                            (void**)&part)))
        return rc;
    double* part_obs = part + (size_t)g_fb * M * P;
    rc = srx_mfma_emission_utt_f32(h, X, off, U, n_frames, utt_model, md, ll, st);
    if (rc) return rc;               // (SRX_EUNSUPPORTED included: the caller goes on to the other paths)
#define SRX_CASE(s_)                                                                           \
    if (S == s_)                                                                               \
        hipLaunchKernelGGL((fb_lane_kernel<s_, true>), dim3(g_fb), dim3(64), lds, st, (const float*)ll, \
                           off, U, utt_model, tb, lat, sc, (double*)nullptr, part, M, md->transmat,     \
                           md->startprob);
    SRX_CASE(1) SRX_CASE(2) SRX_CASE(3) SRX_CASE(4) SRX_CASE(5) SRX_CASE(6) SRX_CASE(7) SRX_CASE(8)
#undef SRX_CASE
#define OBS(NC)                                                                                   \
    hipLaunchKernelGGL((obs_stats_f32_mfma_kernel<NC>), dim3(g_obs), dim3(64), 0, st, (const float*)X, \
                       (const float*)lat, off, U, utt_model, utt_order, M, S, D, part_obs)
    // (a caller that re-estimates start and transition probabilities only gets zeros for
    // the sums over the features: no pass over X)
    const bool with_obs = h->estep_stats != SRX_STATS_TRANSITIONS;
    if (!with_obs) {
    } else if (D <= 16) OBS(1);
    else if (D <= 32) OBS(2);
    else OBS(3);
#undef OBS
    const int64_t n_out = (int64_t)M * (P + Q);                  // one wave per output sum
    const int g_red = (int)std::min<int64_t>((n_out * 64 + kBlock - 1) / kBlock, 4096);
    hipLaunchKernelGGL(reduce_stats_kernel, dim3(g_red), dim3(kBlock), 0, st, (const double*)part, g_fb,
                       (const double*)part_obs, with_obs ? g_obs : 0, M, S, D, out_stats);
    SRX_HIP(hipGetLastError());
    return SRX_OK;
}

int estep_lane_f64(srx_context* h, const void* X, const int64_t* off, int64_t U, int64_t n_frames,
                   const int32_t* utt_model, const int64_t* utt_order, const srx_models_t* md,
                   double* out_stats, hipStream_t st) {
    const int M = md->n_models, S = md->n_states, D = md->n_features;
    const bool asked = h->estep_path == SRX_ESTEP_LANE;
    const int P = S * S + S + 2, Q = S + 2 * S * D;
    const size_t lds = sizeof(double) * (size_t)M * P;
    // (5 to 8 states: the sub-wave chain kernel is faster than one float64 lane)
    if (S > (asked ? 8 : 4) || D > 48 || U <= 0 || (U > 4096 && !asked) || n_frames <= 0 || lds > 32 * 1024)
        return SRX_EUNSUPPORTED;
    int rc;
    double *ll, *lat, *sc;
    if ((rc = srx_ws_ensure(h, WS_LOGLIK, sizeof(double) * (size_t)n_frames * S, (void**)&ll))) return rc;
    if ((rc = srx_ws_ensure(h, WS_LATTICE, sizeof(double) * (size_t)n_frames * S, (void**)&lat))) return rc;
    if ((rc = srx_ws_ensure(h, WS_SCALE, sizeof(double) * (size_t)n_frames * 2, (void**)&sc))) return rc;
    double* mxv = sc + n_frames;
    const int g_fb = (int)((U + 63) / 64);
    const int g_obs = (int)std::min<int64_t>(U, 1024);
    double* part;
    if ((rc = srx_ws_ensure(h, WS_PARTIAL, sizeof(double) * ((size_t)g_fb * M * P + (size_t)g_obs * M * Q),
                            (void**)&part)))
        return rc;
    double* part_obs = part + (size_t)g_fb * M * P;
    rc = srx_mfma_emission_utt_f64(h, X, off, U, n_frames, utt_model, md, ll, st);
    if (rc) return rc;
    hipLaunchKernelGGL(shift_exp_f64_kernel, dim3((unsigned)((n_frames + 255) / 256)), dim3(256), 0, st, ll,
                       n_frames, S, mxv);
#define SRX_CASE(s_)                                                                              \
    if (S == s_)                                                                                  \
        hipLaunchKernelGGL((fb_lane_f64_kernel<s_>), dim3(g_fb), dim3(64), lds, st, (const double*)ll, \
                           (const double*)mxv, off, U, utt_model, md->transmat, md->startprob, lat, sc, \
                           part, M);
    SRX_CASE(1) SRX_CASE(2) SRX_CASE(3) SRX_CASE(4) SRX_CASE(5) SRX_CASE(6) SRX_CASE(7) SRX_CASE(8)
#undef SRX_CASE
    // (a caller that re-estimates start and transition probabilities only gets zeros for
    // the sums over the features: no pass over X)
    const bool with_obs = h->estep_stats != SRX_STATS_TRANSITIONS;
    if (with_obs) {
        rc = srx_mfma_obs_stats_f64(h, X, lat, off, U, utt_model, utt_order, md, part_obs, g_obs, st);
        if (rc) return rc;
    }
    const int64_t n_out = (int64_t)M * (P + Q);
    const int g_red = (int)std::min<int64_t>((n_out * 64 + kBlock - 1) / kBlock, 4096);
    hipLaunchKernelGGL(reduce_stats_kernel, dim3(g_red), dim3(kBlock), 0, st, (const double*)part, g_fb,
                       (const double*)part_obs, with_obs ? g_obs : 0, M, S, D, out_stats);
    SRX_HIP(hipGetLastError());
    return SRX_OK;
}

}  // namespace

int srx_lane_estep_f64(srx_context* h, const void* X, const int64_t* off, int64_t U, int64_t n_frames,
                       const int32_t* utt_model, const int64_t* utt_order, const srx_models_t* md,
                       double* out_stats, hipStream_t st) {
    return estep_lane_f64(h, X, off, U, n_frames, utt_model, utt_order, md, out_stats, st);
}

int srx_lane_estep(srx_context* h, const void* X, const int64_t* off, int64_t U, int64_t n_frames,
                   const int32_t* utt_model, const int64_t* utt_order, const srx_models_t* md,
                   double* out_stats, hipStream_t st) {
    return estep_lane(h, X, off, U, n_frames, utt_model, utt_order, md, out_stats, st);
}

__global__ void srx_fill_slices(int64_t* offs, int64_t n_slices, int64_t slice, int64_t n_frames) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t step = (int64_t)gridDim.x * blockDim.x;
    for (int64_t k = i; k <= n_slices; k += step) {
        const int64_t v = k * slice;
        offs[k] = v < n_frames ? v : n_frames;
    }
}

int srx_generic_emission(srx_context* h, const void* X, int dtype, int64_t n_frames,
                         const srx_models_t* md, void* out, hipStream_t st) {
    return dtype == SRX_F64 ? emission_t<double>(h, X, n_frames, md, out, st)
                            : emission_t<float>(h, X, n_frames, md, out, st);
}

int srx_generic_score(srx_context* h, const void* X, int dtype, const int64_t* off, int64_t U,
                      int64_t n_frames, const srx_models_t* md, double* out, hipStream_t st) {
    return dtype == SRX_F64 ? score_t<double>(h, X, off, U, n_frames, md, out, st)
                            : score_t<float>(h, X, off, U, n_frames, md, out, st);
}

int srx_generic_viterbi(srx_context* h, const void* X, int dtype, const int64_t* off, int64_t U,
                        int64_t n_frames, const int32_t* utt_model, const srx_models_t* md,
                        int32_t* path, double* lp, hipStream_t st) {
    return dtype == SRX_F64 ? viterbi_t<double>(h, X, off, U, n_frames, utt_model, md, path, lp, st)
                            : viterbi_t<float>(h, X, off, U, n_frames, utt_model, md, path, lp, st);
}

int srx_generic_posteriors(srx_context* h, const void* X, int dtype, const int64_t* off, int64_t U,
                           int64_t n_frames, const int32_t* utt_model, const srx_models_t* md,
                           void* gamma, double* lp, hipStream_t st) {
    return dtype == SRX_F64
               ? posteriors_t<double>(h, X, off, U, n_frames, utt_model, md, gamma, lp, st)
               : posteriors_t<float>(h, X, off, U, n_frames, utt_model, md, gamma, lp, st);
}

int srx_generic_estep(srx_context* h, const void* X, int dtype, const int64_t* off, int64_t U,
                      int64_t n_frames, const int32_t* utt_model, const int64_t* utt_order,
                      const srx_models_t* md, double* out_stats, hipStream_t st) {
    return dtype == SRX_F64
               ? estep_t<double>(h, X, off, U, n_frames, utt_model, utt_order, md, out_stats, st)
               : estep_t<float>(h, X, off, U, n_frames, utt_model, utt_order, md, out_stats, st);
}

int srx_generic_mstep(srx_context* h, const double* stats, const srx_priors_t* pr, int mode,
                      const srx_models_t* md, hipStream_t st) {
    SRX_REQUIRE(md->n_states <= 64, "mstep: n_states %d > 64", md->n_states);
    hipLaunchKernelGGL(mstep_kernel, dim3(md->n_models), dim3(kBlock), 0, st, stats, *pr, mode, *md);
    SRX_HIP(hipGetLastError());
    return SRX_OK;
}
