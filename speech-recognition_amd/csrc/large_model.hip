// Large emission model: S states (up to 1024), each a mixture of up to 16 diagonal
// Gaussians, and Viterbi decoding over a dense S x S transition matrix.
//
// gmm_emission_kernel   out[f][s] = ln sum_m w_sm N(x_f; mu_sm, var_sm).  Four
//     states share four 16-row MFMA tiles: row 4g+r of tile k is component 4k+r of
//     state 4Q+g (absent components are zero-weight rows), 16 frames are the
//     columns, so the 16 components of a state land in the 16 accumulator registers
//     of ONE lane and the log-sum-exp needs no cross-lane step.  The contraction
//     [W | W2] . [x ; x^2] runs as split-bf16 on v_mfma_f32_16x16x32_bf16 exactly as
//     in score_mfma.hip.  A wave keeps FB blocks of 16 frames as B operands in
//     registers and walks ALL states; the four waves of a workgroup share the
//     weight tiles through LDS (double-buffered), so the weight stream leaves L2
//     once per 256 frames.  Results are staged in LDS and leave as 64-byte runs
//     per frame.
// viterbi_dense_kernel  one 1024-thread workgroup decodes UB utterances step by
//     step: thread j owns state j, delta_{t-1} of the UB utterances sits in LDS
//     (double-buffered) and is read as broadcasts, row i of log A is one coalesced
//     load; two utterances per packed add / max.  delta_t of every frame goes to a
//     scratch lattice.  viterbi_backtrace_kernel then recomputes the arg max along
//     the best path only (lowest index on ties), as utils.viterbi does
//     (my_hmmlearn/utils.py:243-266), from the transposed log A.
#include "srx_internal.h"

#include <math.h>

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr float kNegBig = -1.0e30f;
constexpr double kLn2 = 0.6931471805599453;
constexpr double kLog2e = 1.4426950408889634;

__device__ __host__ inline unsigned bf16_rne(float f) {
    unsigned u = __builtin_bit_cast(unsigned, f);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return u >> 16;
}

// ------------------------------------------------------------------ GMM tables
// Wp[tile][q RQ + H][part 2][g 4][row 16][4 dwords]: the 8 bf16 weights facing the
// slots [x0 x1 x2 x3 s0 s1 s2 s3] of features 16q+4g .. +3, hi and lo parts, in
// the order the lanes of one operand read consume them (lane = 16 g + row).  If H, the
// feature axis ends in a block of 8 (lane group g: features 16 RQ + 2g, +1) whose operand
// carries hi AND lo halves, [xh0 xh1 xl0 xl1 | sh0 sh1 sl0 sl1]: the hi weights are repeated
// in slots 2, 3 and 6, 7 and the lo weights face zeros there -- two matrix instructions for
// that block instead of three (39 features: 8 per tile and frame block instead of 9).
// Row m of tile t is component 4 (t & 3) + (m & 3) of state 4 (t >> 2) + (m >> 2).
// cst[tile][row] = log2e (ln w - 0.5 (D ln 2pi + sum ln var + sum mu^2 / var)).
__global__ void gmm_prep_kernel(srx_gmm_t gm, int RQ, int H, unsigned* __restrict__ Wp,
                                float* __restrict__ cst) {
    const int S = gm.n_states, MX = gm.n_mix, D = gm.n_features;
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t nth = (int64_t)gridDim.x * blockDim.x;
    const int64_t n_tiles = 4 * (((int64_t)S + 3) / 4);
    const int QT = RQ + H;
    for (int64_t i = tid; i < n_tiles * 16 * QT * 2 * 4; i += nth) {
        const int row = i & 15, g = (i >> 4) & 3, part = (i >> 6) & 1, q = (int)((i >> 7) % QT);
        const int64_t tile = (i >> 7) / QT;
        const int64_t s = 4 * (tile >> 2) + (row >> 2);
        const int m = 4 * (int)(tile & 3) + (row & 3);
        unsigned short v[8];
        for (int j = 0; j < 8; ++j) {
            int d = 16 * q + 4 * g + (j & 3);
            if (q == RQ) d = ((j & 3) < 2 || part == 0) ? 16 * RQ + 2 * g + (j & 1) : D;   // half block
            float wv = 0.f;
            if (s < S && m < MX && d < D) {
                const size_t k = ((size_t)s * MX + m) * D + d;
                const double var = fmax(gm.covars[k], 2.2250738585072014e-308);
                wv = (float)((j >> 2) ? (-0.5 / var) * kLog2e : (gm.means[k] / var) * kLog2e);
            }
            const unsigned hi = bf16_rne(wv);
            const float rem = wv - __builtin_bit_cast(float, hi << 16);
            v[j] = (unsigned short)(part ? bf16_rne(rem) : hi);
        }
        for (int k = 0; k < 4; ++k) Wp[(size_t)i * 4 + k] = (unsigned)v[2 * k] | ((unsigned)v[2 * k + 1] << 16);
    }
    const int64_t wave = tid >> 6, n_wave = nth >> 6;
    const int lane = (int)(tid & 63);
    for (int64_t row = wave; row < n_tiles * 16; row += n_wave) {
        const int64_t tile = row >> 4;
        const int64_t s = 4 * (tile >> 2) + ((row & 15) >> 2);
        const int m = 4 * (int)(tile & 3) + (int)(row & 3);
        const bool on = s < S && m < MX;
        double acc = 0.0;
        if (on)
            for (int d = lane; d < D; d += 64) {
                const size_t k = ((size_t)s * MX + m) * D + d;
                const double var = fmax(gm.covars[k], 2.2250738585072014e-308);
                acc += log(var) + gm.means[k] * gm.means[k] / var;
            }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
        if (lane == 0) {
            float c = kNegBig;
            if (on) {
                const double w = gm.weights[(size_t)s * MX + m];
                if (w > 0.0) c = (float)((log(w) - 0.5 * (acc + D * 1.8378770664093453)) * kLog2e);
            }
            cst[row] = c;
        }
    }
}

__device__ __forceinline__ void split_operand(const u32x4& xc, bf16x8& bh, bf16x8& bl) {
    float x[4], sq[4];
    unsigned xl_[2], sl_[2];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        x[j] = __uint_as_float(xc[j]);
        sq[j] = x[j] * x[j];
    }
    u32x4 fh;
    fh[0] = __builtin_amdgcn_perm(xc[1], xc[0], 0x07060302u);
    fh[1] = __builtin_amdgcn_perm(xc[3], xc[2], 0x07060302u);
    fh[2] = __builtin_amdgcn_perm(__float_as_uint(sq[1]), __float_as_uint(sq[0]), 0x07060302u);
    fh[3] = __builtin_amdgcn_perm(__float_as_uint(sq[3]), __float_as_uint(sq[2]), 0x07060302u);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const bf16x2 a = {(__bf16)(x[2 * j] - __uint_as_float(xc[2 * j] & 0xffff0000u)),
                          (__bf16)(x[2 * j + 1] - __uint_as_float(xc[2 * j + 1] & 0xffff0000u))};
        const bf16x2 b = {
            (__bf16)(sq[2 * j] - __uint_as_float(__float_as_uint(sq[2 * j]) & 0xffff0000u)),
            (__bf16)(sq[2 * j + 1] - __uint_as_float(__float_as_uint(sq[2 * j + 1]) & 0xffff0000u))};
        xl_[j] = __builtin_bit_cast(unsigned, a);
        sl_[j] = __builtin_bit_cast(unsigned, b);
    }
    const u32x4 fl = {xl_[0], xl_[1], sl_[0], sl_[1]};
    bh = __builtin_bit_cast(bf16x8, fh);
    bl = __builtin_bit_cast(bf16x8, fl);
}

constexpr int kFB = 4;                 // 16-frame blocks per wave
constexpr int kWaves = 4;              // waves per workgroup
constexpr int kFramesPerBlock = 16 * kFB * kWaves;

// The last block of 8 features as one operand (see gmm_prep_kernel).
__device__ __forceinline__ bf16x8 split_half_operand(unsigned h0, unsigned h1) {
    const float x0 = __uint_as_float(h0), x1 = __uint_as_float(h1);
    const float s0 = x0 * x0, s1 = x1 * x1;
    const bf16x2 a = {(__bf16)(x0 - __uint_as_float(h0 & 0xffff0000u)), (__bf16)(x1 - __uint_as_float(h1 & 0xffff0000u))};
    const bf16x2 b = {(__bf16)(s0 - __uint_as_float(__float_as_uint(s0) & 0xffff0000u)),
                      (__bf16)(s1 - __uint_as_float(__float_as_uint(s1) & 0xffff0000u))};
    const u32x4 fb = {__builtin_amdgcn_perm(h1, h0, 0x07060302u), __builtin_bit_cast(unsigned, a),
                      __builtin_amdgcn_perm(__float_as_uint(s1), __float_as_uint(s0), 0x07060302u),
                      __builtin_bit_cast(unsigned, b)};
    return __builtin_bit_cast(bf16x8, fb);
}

template <int RQ, int H>
__global__ __launch_bounds__(256, 2) void gmm_emission_kernel(
    const float* __restrict__ X, int64_t n_frames, int S, int D, const unsigned* __restrict__ Wp,
    const float* __restrict__ cst, float* __restrict__ out) {
    constexpr int QT = RQ + H;
    constexpr int TILE = 16 * QT * 2 * 4 * 4;          // dwords of one tile's weights
    __shared__ __attribute__((aligned(16))) unsigned s_w[2][4 * TILE];
    __shared__ float s_c[2][64];
    __shared__ float s_res[kWaves][16][16 * kFB];       // [wave][state in group][frame]
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63, g = lane >> 4, ul = lane & 15;
    const int64_t f_wave = (int64_t)blockIdx.x * kFramesPerBlock + (int64_t)wv * 16 * kFB;

    // ---- this wave's frames as split operands, loaded once
    bf16x8 bh[kFB][RQ ? RQ : 1], bl[kFB][RQ ? RQ : 1], bb[kFB];
    {
        const int64_t f0 = f_wave < n_frames ? f_wave : 0;
        const int64_t rest = f_wave < n_frames ? (n_frames - f0) * (int64_t)D * 4 : 0;
        const uintptr_t a = (uintptr_t)(X + f0 * D);
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a);
        const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
        const unsigned n = (unsigned)(rest < 0xFFFFFF00ll ? rest : 0xFFFFFF00ll);
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(((uintptr_t)hi << 32) | lo), 0, __builtin_amdgcn_readfirstlane(n), 0x00020000);
#pragma unroll
        for (int fb = 0; fb < kFB; ++fb) {
#pragma unroll
            for (int q = 0; q < RQ; ++q) {
                const unsigned o = (unsigned)(fb * 16 + ul) * (unsigned)D * 4u + 16u * g + 64u * q;
                const u32x4 x = __builtin_amdgcn_raw_buffer_load_b128(rsrc, o, 0, 0);   // past the end: 0
                split_operand(x, bh[fb][q], bl[fb][q]);
            }
            if constexpr (H != 0) {
                const unsigned o = (unsigned)(fb * 16 + ul) * (unsigned)D * 4u + 64u * RQ + 8u * g;
                const auto x = __builtin_amdgcn_raw_buffer_load_b64(rsrc, o, 0, 0);
                bb[fb] = split_half_operand(x[0], x[1]);
            }
        }
    }

    // ---- weight stream: the workgroup copies the next four tiles (four states)
    // while it computes the current ones; one barrier per four states
    const int n_quads = (S + 3) / 4;
    // (the copy is split: global loads into registers before the MFMAs, LDS writes
    // after them, so the loads' latency hides behind the matrix work)
    constexpr int NST = (4 * TILE + 1023) / 1024;      // 16-byte pieces per thread
    u32x4 stg[NST];
    float stc = 0.f;
    auto fetch = [&](int Q) {
        const unsigned* src = Wp + (size_t)Q * 4 * TILE;
#pragma unroll
        for (int i = 0; i < NST; ++i) {
            const int o = (i * 256 + threadIdx.x) * 4;
            stg[i] = o < 4 * TILE ? *(const u32x4*)(src + o) : u32x4{0u, 0u, 0u, 0u};
        }
        if (threadIdx.x < 64) stc = cst[(size_t)Q * 64 + threadIdx.x];
    };
    auto commit = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NST; ++i) {
            const int o = (i * 256 + threadIdx.x) * 4;
            if (o < 4 * TILE) *(u32x4*)(&s_w[buf][o]) = stg[i];
        }
        if (threadIdx.x < 64) s_c[buf][threadIdx.x] = stc;
    };
    fetch(0);
    commit(0);
    __syncthreads();
    for (int Q = 0; Q < n_quads; ++Q) {
        const int buf = Q & 1;
        if (Q + 1 < n_quads) fetch(Q + 1);
        f32x4 acc[4][kFB];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            u32x4 wh[QT], wl[QT];
#pragma unroll
            for (int q = 0; q < QT; ++q) {
                const unsigned* p = &s_w[buf][k * TILE + ((q * 2) * 64 + lane) * 4];
                wh[q] = *(const u32x4*)p;
                wl[q] = *(const u32x4*)(p + 256);
            }
            f32x4 c0;
#pragma unroll
            for (int r = 0; r < 4; ++r) c0[r] = s_c[buf][k * 16 + 4 * g + r];
#pragma unroll
            for (int fb = 0; fb < kFB; ++fb) acc[k][fb] = c0;
#pragma unroll
            for (int q = 0; q < RQ; ++q) {
                const bf16x8 ah = __builtin_bit_cast(bf16x8, wh[q]), al = __builtin_bit_cast(bf16x8, wl[q]);
#pragma unroll
                for (int fb = 0; fb < kFB; ++fb)
                    acc[k][fb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh[fb][q], acc[k][fb], 0, 0, 0);
#pragma unroll
                for (int fb = 0; fb < kFB; ++fb)
                    acc[k][fb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl[fb][q], acc[k][fb], 0, 0, 0);
#pragma unroll
                for (int fb = 0; fb < kFB; ++fb)
                    acc[k][fb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh[fb][q], acc[k][fb], 0, 0, 0);
            }
            if constexpr (H != 0) {
                const bf16x8 ah = __builtin_bit_cast(bf16x8, wh[RQ]), al = __builtin_bit_cast(bf16x8, wl[RQ]);
#pragma unroll
                for (int fb = 0; fb < kFB; ++fb)
                    acc[k][fb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bb[fb], acc[k][fb], 0, 0, 0);
#pragma unroll
                for (int fb = 0; fb < kFB; ++fb)
                    acc[k][fb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bb[fb], acc[k][fb], 0, 0, 0);
            }
        }
        // ---- log-sum-exp over this lane's 16 components of state 4Q+g; log2 units
#pragma unroll
        for (int fb = 0; fb < kFB; ++fb) {
            float mx = fmaxf(fmaxf(acc[0][fb][0], acc[0][fb][1]), fmaxf(acc[0][fb][2], acc[0][fb][3]));
#pragma unroll
            for (int k = 1; k < 4; ++k)
                mx = fmaxf(mx, fmaxf(fmaxf(acc[k][fb][0], acc[k][fb][1]), fmaxf(acc[k][fb][2], acc[k][fb][3])));
            float sum = 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int r = 0; r < 4; ++r) sum += __builtin_amdgcn_exp2f(acc[k][fb][r] - mx);
            s_res[wv][(4 * Q + g) & 15][fb * 16 + ul] = (mx + __builtin_amdgcn_logf(sum)) * (float)kLn2;
        }
        // ---- every 16 states: lane l writes 64 bytes of frame l
        const int s_last = 4 * Q + 3 < S - 1 ? 4 * Q + 3 : S - 1;
        if ((s_last & 15) == 15 || s_last == S - 1) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const int sg = s_last & ~15, ns = s_last - sg + 1;
            const int64_t f = f_wave + lane;
            if (f < n_frames) {
                float* o = out + f * S + sg;
                if (ns == 16 && (S & 3) == 0) {
#pragma unroll
                    for (int k = 0; k < 16; k += 4)
                        *(f32x4*)(o + k) = f32x4{s_res[wv][k][lane], s_res[wv][k + 1][lane],
                                                 s_res[wv][k + 2][lane], s_res[wv][k + 3][lane]};
                } else {
                    for (int k = 0; k < ns; ++k) o[k] = s_res[wv][k][lane];
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        if (Q + 1 < n_quads) commit(buf ^ 1);
        __syncthreads();
    }
}

// shape-generic mixture densities (any sizes, float or double): one thread per
// (frame, state), serial over components and features
template <typename T>
__global__ void gmm_emission_generic_kernel(const T* __restrict__ X, int64_t n_frames, srx_gmm_t gm,
                                            T* __restrict__ out) {
    const int S = gm.n_states, MX = gm.n_mix, D = gm.n_features;
    const int64_t n = n_frames * S;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t f = i / S;
        const int s = (int)(i % S);
        const T* x = X + f * D;
        double best = -INFINITY, sum = 0.0;
        for (int m = 0; m < MX; ++m) {
            const double w = gm.weights[(size_t)s * MX + m];
            if (!(w > 0.0)) continue;
            double acc = D * 1.8378770664093453;
            for (int d = 0; d < D; ++d) {
                const size_t k = ((size_t)s * MX + m) * D + d;
                const double var = fmax(gm.covars[k], 2.2250738585072014e-308);
                const double df = (double)x[d] - gm.means[k];
                acc += log(var) + df * df / var;
            }
            const double v = log(w) - 0.5 * acc;
            if (v > best) {
                sum = sum * exp(best - v) + 1.0;
                best = v;
            } else {
                sum += exp(v - best);
            }
        }
        out[i] = (T)(best + log(sum));
    }
}

// ---------------------------------------------------------------- dense Viterbi
template <typename T>
__global__ void dense_prep_kernel(const double* __restrict__ pi, const double* __restrict__ A, int S,
                                  T* __restrict__ logpi, T* __restrict__ logA,
                                  T* __restrict__ logAT) {
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t nth = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = tid; i < (int64_t)S * S; i += nth) {
        const T v = (T)log(A[i]);                       // log 0 = -inf, as np.log does
        logA[i] = v;
        logAT[(i % S) * S + i / S] = v;
    }
    for (int64_t i = tid; i < S; i += nth) logpi[i] = (T)log(pi[i]);
}

template <typename T>
struct Neg;
template <>
struct Neg<float> {
    static __device__ float inf() { return -INFINITY; }
};
template <>
struct Neg<double> {
    static __device__ double inf() { return -INFINITY; }
};

// UB utterances per workgroup, thread j <-> state j (+ k * blockDim for S > blockDim)
template <typename T, int UB>
__global__ __launch_bounds__(1024) void viterbi_dense_kernel(
    const T* __restrict__ ll, const int64_t* __restrict__ off, int64_t U, int S,
    const T* __restrict__ logpi, const T* __restrict__ logA, T* __restrict__ delta_out,
    double* __restrict__ shift_out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char vraw[];
    T* dl = (T*)vraw;                        // [2][S][UB]: delta of the previous / this step
    // Every kRenorm steps delta of an utterance is made relative to its largest entry (what was
    // taken off goes to shift_out[u] for the log-probability): a float32 lattice that grows by
    // tens per frame would otherwise lose the resolution to tell candidates apart.  The stored
    // rows stay consistent: the backtrace only compares entries of one row.
    constexpr int kRenorm = 64;
    __shared__ T red[UB];
    __shared__ double shift_acc[UB];
    if (threadIdx.x < UB) shift_acc[threadIdx.x] = 0.0;
    const int j = threadIdx.x;
    const int64_t u0 = (int64_t)blockIdx.x * UB;
    int64_t f0[UB];
    int Tl[UB];
    int Tmax = 0;
#pragma unroll
    for (int b = 0; b < UB; ++b) {
        const int64_t u = u0 + b;
        f0[b] = u < U ? off[u] : 0;
        Tl[b] = u < U ? (int)(off[u + 1] - f0[b]) : 0;
        Tmax = max(Tmax, Tl[b]);
    }
    const bool own = j < S;
    // t = 0
    if (own) {
#pragma unroll
        for (int b = 0; b < UB; ++b) {
            T v = Neg<T>::inf();
            if (Tl[b] > 0) {
                v = logpi[j] + ll[f0[b] * S + j];
                delta_out[f0[b] * S + j] = v;
            }
            dl[(size_t)j * UB + b] = v;
        }
    }
    __syncthreads();
    for (int t = 1; t < Tmax; ++t) {
        const T* prev = dl + (size_t)((t - 1) & 1) * S * UB;
        T* cur = dl + (size_t)(t & 1) * S * UB;
        if (own) {
            T best[UB];
#pragma unroll
            for (int b = 0; b < UB; ++b) best[b] = Neg<T>::inf();
            const T* acol = logA + j;
#pragma unroll 4
            for (int i = 0; i < S; ++i) {
                const T a = acol[(size_t)i * S];
                const T* p = prev + (size_t)i * UB;       // same address in every lane: a broadcast
#pragma unroll
                for (int b = 0; b < UB; ++b) best[b] = fmax(best[b], p[b] + a);
            }
#pragma unroll
            for (int b = 0; b < UB; ++b) {
                T v = prev[(size_t)j * UB + b];           // finished utterances keep their last column
                if (t < Tl[b]) {
                    v = best[b] + ll[(f0[b] + t) * S + j];
                    delta_out[(f0[b] + t) * S + j] = v;
                }
                cur[(size_t)j * UB + b] = v;
            }
        }
        __syncthreads();
        if (sizeof(T) == 4 && t % kRenorm == 0) {      // (float64 has the headroom: left as it is)
            // largest entry per utterance, found by the first wave alone
            if (threadIdx.x < 64) {
                for (int b = 0; b < UB; ++b) {
                    T v = Neg<T>::inf();
                    for (int i = threadIdx.x; i < S; i += 64) v = fmax(v, cur[(size_t)i * UB + b]);
#pragma unroll
                    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
                    // (an utterance at or past its last frame keeps what it has: its last row is stored)
                    const bool on = t < Tl[b] - 1 && v > Neg<T>::inf();
                    if (threadIdx.x == 0) {
                        red[b] = on ? v : (T)0;
                        if (on) shift_acc[b] += (double)v;
                    }
                }
            }
            __syncthreads();
            if (own) {
#pragma unroll
                for (int b = 0; b < UB; ++b) cur[(size_t)j * UB + b] -= red[b];
            }
            __syncthreads();
        }
    }
    if (threadIdx.x < UB && u0 + threadIdx.x < U) shift_out[u0 + threadIdx.x] = shift_acc[threadIdx.x];
}

// One wave per utterance: final arg max, then s_{t-1} = argmax_i delta_{t-1}(i) + log A(i, s_t)
template <typename T>
__global__ void viterbi_backtrace_kernel(const T* __restrict__ delta, const int64_t* __restrict__ off,
                                         int64_t U, int S, const T* __restrict__ logAT,
                                         const double* __restrict__ shift,
                                         int32_t* __restrict__ path, double* __restrict__ out_lp) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t n_wave = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t u = wave; u < U; u += n_wave) {
        const int64_t f0 = off[u];
        const int Tl = (int)(off[u + 1] - f0);
        if (Tl <= 0) {
            if (lane == 0) out_lp[u] = 0.0;
            continue;
        }
        auto argmax = [&](const T* d, const T* arow, T* val) {
            T best = Neg<T>::inf();
            int arg = 0x7fffffff;
            for (int i = lane; i < S; i += 64) {
                const T v = arow ? d[i] + arow[i] : d[i];
                if (v > best || arg == 0x7fffffff) {      // strict: the lowest index wins ties
                    best = v;
                    arg = i;
                }
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const T ob = __shfl_xor(best, o, 64);
                const int oa = __shfl_xor(arg, o, 64);
                if (ob > best || (ob == best && oa < arg)) {
                    best = ob;
                    arg = oa;
                }
            }
            *val = best;
            return arg;
        };
        T v;
        int s = argmax(delta + (f0 + Tl - 1) * S, nullptr, &v);
        if (lane == 0) {
            out_lp[u] = shift[u] + (double)v;
            path[f0 + Tl - 1] = s;
        }
        for (int t = Tl - 2; t >= 0; --t) {
            s = argmax(delta + (f0 + t) * S, logAT + (size_t)s * S, &v);
            if (lane == 0) path[f0 + t] = s;
        }
    }
}

template <typename T>
int viterbi_dense_t(srx_context* h, const void* ll, const int64_t* off, int64_t U, int64_t n_frames,
                    int S, const double* pi, const double* A, int32_t* path, double* lp,
                    hipStream_t st) {
    if (S > 1024) return SRX_EUNSUPPORTED;
    T *tabs, *delta;
    int rc = srx_ws_ensure(h, WS_PARAMS, sizeof(T) * ((size_t)2 * S * S + S), (void**)&tabs);
    if (rc) return rc;
    rc = srx_ws_ensure(h, WS_LATTICE, sizeof(T) * (size_t)n_frames * S, (void**)&delta);
    if (rc) return rc;
    double* shift;
    rc = srx_ws_ensure(h, WS_SCALE, sizeof(double) * (size_t)(U > 0 ? U : 1), (void**)&shift);
    if (rc) return rc;
    T* logA = tabs;
    T* logAT = logA + (size_t)S * S;
    T* logpi = logAT + (size_t)S * S;
    hipLaunchKernelGGL(dense_prep_kernel<T>, dim3(256), dim3(256), 0, st, pi, A, S, logpi, logA, logAT);
    const int threads = ((S + 63) / 64) * 64;
    // utterances per workgroup: as many as LDS holds twice over, at most 16
    constexpr int UBmax = sizeof(T) == 4 ? 16 : 8;
    const size_t lds16 = sizeof(T) * 2 * (size_t)S * UBmax;
    if (lds16 <= 144 * 1024) {
        auto kern = viterbi_dense_kernel<T, UBmax>;
        SRX_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)lds16));
        hipLaunchKernelGGL(kern, dim3((unsigned)((U + UBmax - 1) / UBmax)), dim3(threads), lds16, st,
                           (const T*)ll, off, U, S, logpi, logA, delta, shift);
    } else {
        return SRX_EUNSUPPORTED;
    }
    hipLaunchKernelGGL(viterbi_backtrace_kernel<T>, dim3((unsigned)((U + 3) / 4)), dim3(256), 0, st,
                       (const T*)delta, off, U, S, logAT, (const double*)shift, path, lp);
    SRX_HIP(hipGetLastError());
    return SRX_OK;
}

}  // namespace

int srx_large_gmm_emission(srx_context* h, const void* X, int dtype, int64_t n_frames,
                           const srx_gmm_t* gm, void* out, hipStream_t st) {
    const int S = gm->n_states, MX = gm->n_mix, D = gm->n_features;
    if (dtype == SRX_F32 && !h->exact_f32 && MX <= 16 && D <= 48) {
        const int H = (D % 16 >= 1 && D % 16 <= 8) ? 1 : 0;
        const int RQ = H ? D / 16 : (D + 15) / 16, QT = RQ + H;
        const size_t n_tiles = 4 * (((size_t)S + 3) / 4);
        const size_t n_wp = n_tiles * 16 * QT * 2 * 4 * 4, n_c = n_tiles * 16;
        unsigned* base;
        int rc = srx_ws_ensure(h, WS_PARAMS, 4 * (n_wp + n_c), (void**)&base);
        if (rc) return rc;
        float* cst = (float*)(base + n_wp);
        hipLaunchKernelGGL(gmm_prep_kernel, dim3(256), dim3(256), 0, st, *gm, RQ, H, base, cst);
        const unsigned grid = (unsigned)((n_frames + kFramesPerBlock - 1) / kFramesPerBlock);
        const float* Xf = (const float*)X;
        float* of = (float*)out;
#define SRX_CASE(rq, hh)                                                                          \
    if (RQ == rq && H == hh)                                                                      \
        hipLaunchKernelGGL((gmm_emission_kernel<rq, hh>), dim3(grid), dim3(256), 0, st, Xf, n_frames, \
                           S, D, base, cst, of);
        SRX_CASE(0, 1) SRX_CASE(1, 0) SRX_CASE(1, 1) SRX_CASE(2, 0) SRX_CASE(2, 1) SRX_CASE(3, 0)
#undef SRX_CASE
        SRX_HIP(hipGetLastError());
        return SRX_OK;
    }
    const int64_t n = n_frames * S;
    unsigned grid = (unsigned)((n + 255) / 256 < (int64_t)h->n_cu * 32 ? (n + 255) / 256 : (int64_t)h->n_cu * 32);
    if (grid < 1) grid = 1;
    if (dtype == SRX_F64)
        hipLaunchKernelGGL(gmm_emission_generic_kernel<double>, dim3(grid), dim3(256), 0, st,
                           (const double*)X, n_frames, *gm, (double*)out);
    else
        hipLaunchKernelGGL(gmm_emission_generic_kernel<float>, dim3(grid), dim3(256), 0, st,
                           (const float*)X, n_frames, *gm, (float*)out);
    SRX_HIP(hipGetLastError());
    return SRX_OK;
}

int srx_large_viterbi_dense(srx_context* h, const void* ll, int dtype, const int64_t* off, int64_t U,
                            int64_t n_frames, int S, const double* pi, const double* A,
                            int32_t* path, double* lp, hipStream_t st) {
    return dtype == SRX_F64
               ? viterbi_dense_t<double>(h, ll, off, U, n_frames, S, pi, A, path, lp, st)
               : viterbi_dense_t<float>(h, ll, off, U, n_frames, S, pi, A, path, lp, st);
}
