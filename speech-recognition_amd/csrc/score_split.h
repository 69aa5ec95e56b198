// Split-bf16 scoring kernel and its tables, shared by the translation units that
// instantiate it (one per padded state count, so they compile in parallel) and by
// the emission kernels of score_mfma.hip.
#pragma once
#include "srx_internal.h"

#include <math.h>
#include <type_traits>

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr double kLn2 = 0.6931471805599453;
constexpr float kNegBig = -1.0e30f;

__device__ __forceinline__ float xchg16(float v) { return __shfl_xor(v, 16, 64); }

// max / sum of a value with its partner 16 lanes away (lanes l and l^16), in
// both lanes, without touching LDS: v_permlane16_swap exchanges the odd
// 16-lane rows of one operand with the even rows of the other, so after
// swapping two copies of v the pair (a, b) holds {v[l], v[l^16]} in some order.
__device__ __forceinline__ float pair_max(float v) {
    const unsigned u = __float_as_uint(v);
    const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float pair_sum(float v) {
    const unsigned u = __float_as_uint(v);
    const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
// The same over all four 16-lane rows (a model of 9..16 states fills a whole tile).
__device__ __forceinline__ float quad_max(float v) {
    v = pair_max(v);
    const unsigned u = __float_as_uint(v);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float quad_sum(float v) {
    v = pair_sum(v);
    const unsigned u = __float_as_uint(v);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}


// ---------------------------------------------------------------------------
// Split-bf16 variant.  The emission contraction runs on
// v_mfma_f32_16x16x32_bf16 (16x the f32 matrix rate) with every f32 operand
// written as hi + lo bf16 halves and three of the four partial products kept:
//     x w  ~=  xh wh + xl wh + xh wl          (xl wl < 2^-16 |x w| is dropped)
// hi(x) is the truncated upper 16 bits (exact remainder), lo is rounded.  The
// relative error per product is <= ~3e-5 and averages out over the 2D terms;
// scores agree with the exact-f32 kernel to ~1e-6 relative.
//
// Memory.  A lane reads 16 contiguous bytes per load: lane group g of
// utterance u takes features 16q+4g .. 16q+4g+3 (q < RQ), so one wave
// instruction moves 16 rows x 64 B instead of 16 x 32 B in 4-byte pieces (the
// vector L1 handles a request per row segment, not per byte).  Loads are raw
// buffer loads on a per-wave descriptor: rows are only 4-byte aligned and the
// last features of a frame run into the next frame (they meet zero weights);
// past the end of X the hardware range check returns zeros instead of
// faulting.  A batch of 16 consecutive utterances must span < 4 GiB.
//
// k-steps per q (8 bf16 per lane, x = the lane's 4 features, s = x^2):
//   [xh0..3 sh0..3] . wh     [xl0..3 sl0..3] . wh     [xh0..3 sh0..3] . wl
// The transition products also go to the matrix core (exact f32 16x16x4,
// block-diagonal per tile): accumulator register q of lane group g is tile
// row 4g+q, which is the k index that group supplies in k-step q, so alpha
// feeds the next step without lane movement; only the per-model max and sum
// cross lanes (one v_permlane16_swap each for 8 states, none for 4).
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

struct SplitTables {
    const unsigned* Wp;   // [NT*16 rows][RQ][2 (hi, lo)][4 lane groups][4 dwords] packed bf16 pairs
    const float* cst;     // [NT*16]
    const float* A;       // [M][S][S]
    const float* pi;      // [M][S]
    const int64_t* order; // scoring: visiting order of the utterances (a permutation), or NULL
    int chunks = 1;       // slices of NT tiles that the bank is scored in (1 up to NT tiles)
};

__device__ __host__ inline unsigned bf16_rne(float f) {
    unsigned u = __builtin_bit_cast(unsigned, f);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return u >> 16;
}

// One thread per (row, q, part, lane group): the 8 weights facing the slots
// [x0 x1 x2 x3 s0 s1 s2 s3] of features 16q+4g .. 16q+4g+3.
// The feature axis is cut into RQ blocks of 16 (lane group g: features 16q+4g..+3)
// and, if H, one last block of 8 (lane group g: features 16RQ+2g, +1).  Its operand
// carries hi AND lo halves, [xh0 xh1 xl0 xl1 | sh0 sh1 sl0 sl1], so the hi weights are
// repeated in slots 2, 3 and 6, 7 (w_hi x_hi + w_hi x_lo in one instruction) and the lo
// weights face zeros there: two instructions per tile for this block instead of three.
__global__ void split_prep_kernel(srx_models_t md, int SP, int NT, int RQ, int H, unsigned* Wp,
                                  float* cst, float* A, float* pi) {
    const int M = md.n_models, S = md.n_states, D = md.n_features;
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    const int nth = gridDim.x * blockDim.x;
    const int QT = RQ + H;
    for (int i = tid; i < NT * 16 * QT * 2 * 4; i += nth) {
        const int g = i & 3, part = (i >> 2) & 1, q = (i >> 3) % QT, row = (i >> 3) / QT;
        const int m = row / SP, s = row % SP;
        unsigned short v[8];
        for (int j = 0; j < 8; ++j) {
            int d = 16 * q + 4 * g + (j & 3);
            if (q == RQ) d = ((j & 3) < 2 || part == 0) ? 16 * RQ + 2 * g + (j & 1) : D;   // half block
            float wv = 0.f;
            if (m < M && s < S && d < D) {
                const size_t k = ((size_t)m * S + s) * D + d;
                const double var = fmax(md.covars[k], 2.2250738585072014e-308);
                wv = (float)((j >> 2) ? (-0.5 / var) * 1.4426950408889634
                                      : (md.means[k] / var) * 1.4426950408889634);
            }
            const unsigned hi = bf16_rne(wv);
            const float rem = wv - __builtin_bit_cast(float, hi << 16);
            v[j] = (unsigned short)(part ? bf16_rne(rem) : hi);
        }
        for (int k = 0; k < 4; ++k) Wp[(size_t)i * 4 + k] = (unsigned)v[2 * k] | ((unsigned)v[2 * k + 1] << 16);
    }
    // constants: one wave per row, lanes over features (the serial sum of D
    // logarithms per row would be most of this kernel's time)
    const int wave = tid >> 6, lane = tid & 63, n_wave = nth >> 6;
    for (int row = wave; row < NT * 16; row += n_wave) {
        const int m = row / SP, s = row % SP;
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
        if (lane == 0)
            cst[row] = on ? (float)(-0.5 * (acc + D * 1.8378770664093453) * 1.4426950408889634) : kNegBig;
    }
    for (int i = tid; i < M * S * S; i += nth) A[i] = (float)md.transmat[i];
    for (int i = tid; i < M * S; i += nth) pi[i] = (float)md.startprob[i];
}

template <int SP, int NT, int RQ, int H>
__global__ __launch_bounds__(256, 2) void score_split_kernel(
    const float* __restrict__ X, const int64_t* __restrict__ off, int64_t U, int64_t n_frames,
    int M, int S, int D, SplitTables tb, double* __restrict__ out) {
    const int lane = threadIdx.x & 63, g = lane >> 4, ul = lane & 15;
    // Work item = (batch of 16 utterances, slice of NT tiles), the slice running fastest: a bank
    // of more than NT tiles reads its features once per slice, and with the slices of a batch
    // in neighbouring waves the second reading comes from the cache.
    const int64_t item = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int chunks = tb.chunks;
    const int64_t group = chunks == 1 ? item : item / chunks;
    if (group * 16 >= U) return;
    const int tile0 = chunks == 1 ? 0 : (int)(item % chunks) * NT;
    // (with a visiting order -- the caller's way of putting utterances of similar length into one
    // batch -- lane ul takes entry group*16 + ul of it; the host passes one only when all of X
    // is within a descriptor's reach, since the utterances of a batch then lie anywhere)
    const bool valid = group * 16 + ul < U;
    const int64_t u = valid ? (tb.order ? tb.order[group * 16 + ul] : group * 16 + ul) : 0;
    const int64_t uc = valid ? u : U - 1;
    const int64_t f0 = off[uc];
    const int T = valid ? (int)(off[uc + 1] - f0) : 0;
    int Tmax = T;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) Tmax = max(Tmax, __shfl_xor(Tmax, o, 64));

    // ---- per-wave buffer descriptor over [first row of the batch, end of X)
    const int64_t u_last = group * 16 + 15 < U - 1 ? group * 16 + 15 : U - 1;
    const int64_t fbase = tb.order ? 0 : off[group * 16];
    const int64_t span = tb.order ? 0 : (off[u_last + 1] - fbase) * D * 4;           // bytes, uniform
    const int64_t rest = (n_frames - fbase) * (int64_t)D * 4;
    const unsigned nrec = (unsigned)(rest < 0xFFFFFF00ll ? rest : 0xFFFFFF00ll);
    // One descriptor reaches 4 GiB.  A batch whose frames span more is left to
    // score_far_kernel (launched after this one whenever X is that large).
    if (span >= 0xFFFFFF00ll) return;
    const unsigned base_lo = __builtin_amdgcn_readfirstlane((unsigned)((uintptr_t)(X + fbase * D)));
    const unsigned base_hi = __builtin_amdgcn_readfirstlane((unsigned)(((uintptr_t)(X + fbase * D)) >> 32));
    const float* xb = (const float*)(((uintptr_t)base_hi << 32) | base_lo);
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)xb, 0, __builtin_amdgcn_readfirstlane(nrec), 0x00020000);
    const unsigned row_off = (unsigned)((f0 - fbase) * D * 4) + 16u * g;   // this lane, frame 0, q = 0
    const unsigned row_bytes = (unsigned)D * 4u;

    // ---- loop-invariant registers: packed weights, constants, transition operands
    constexpr int QT = RQ + H;
    u32x4 wh[NT][QT], wl[NT][QT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int q = 0; q < QT; ++q) {
            const unsigned* p = tb.Wp + ((((size_t)((tile0 + nt) * 16 + ul)) * QT + q) * 2 * 4 + g) * 4;
            wh[nt][q] = *(const u32x4*)p;
            wl[nt][q] = *(const u32x4*)(p + 16);
        }
    // Transition operands, also split-bf16 on the K = 32 instruction: slot j < 4 of
    // lane group g faces hi(alpha) of source row 4g+j, slot j >= 4 its lo part, so
    //   [A_hi | A_hi] . [alpha_hi ; alpha_lo]  +  [A_lo | 0] . [alpha_hi ; *]
    // is two MFMAs (32 cycles) where exact f32 needs four (128); the A_lo alpha_lo
    // term (2^-18 relative) is dropped.
    f32x4 c0[NT];
    u32x4 a_hh[NT], a_l0[NT];
    int model[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        // accumulator rows 4g..4g+3 of the tile; tile row rho = SP*h + s
        unsigned short hi[4], lo[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            c0[nt][q] = tb.cst[(tile0 + nt) * 16 + 4 * g + q];
            // out row = ul, source row = 4g+q
            const int src = 4 * g + q, dst = ul;
            const int ms = ((tile0 + nt) * 16 + src) / SP, md_ = ((tile0 + nt) * 16 + dst) / SP;
            const int ss = src % SP, sd = dst % SP;
            const bool ok = ms == md_ && ms < M && ss < S && sd < S;
            const float a = tb.A[ok ? ((size_t)ms * S + ss) * S + sd : 0] * (ok ? 1.f : 0.f);
            const unsigned h = bf16_rne(a);
            hi[q] = (unsigned short)h;
            lo[q] = (unsigned short)bf16_rne(a - __uint_as_float(h << 16));
        }
        const unsigned h01 = hi[0] | ((unsigned)hi[1] << 16), h23 = hi[2] | ((unsigned)hi[3] << 16);
        a_hh[nt] = u32x4{h01, h23, h01, h23};
        a_l0[nt] = u32x4{lo[0] | ((unsigned)lo[1] << 16), lo[2] | ((unsigned)lo[3] << 16), 0u, 0u};
        model[nt] = ((tile0 + nt) * 16 + 4 * g) / SP;
    }

    // alpha is kept UNNORMALISED: alpha_t = (A^T alpha_{t-1}) b~_t / c_{t-1}, with the
    // division folded into the exponent of b~_t (shift = row maximum + log2 c_{t-1}),
    // so c_t = sum(alpha_t) is the usual scale factor and costs no multiplies.
    f32x4 alpha[NT];
    float lc[NT];          // log2 c_{t-1}; 0 for a dead chain (its alpha is all zero)
    double lp[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        lp[nt] = 0.0;
        lc[nt] = 0.f;
        alpha[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    // ---- frame queue: PF steps in flight.  The loop body is branch-free so the
    // compiler keeps counted waits on it; steps past an utterance's end re-read
    // its last frame and change nothing.
    constexpr int PF = 4;
    struct frame_t {
        u32x4 v[RQ ? RQ : 1];
        u32x2 h;
    };
    frame_t xq[PF];
    const unsigned half_off = row_off - 16u * g + 64u * RQ + 8u * g;
    auto load = [&](int t, frame_t& dst) {
        int tc = t < T ? t : T - 1;
        if (tc < 0) tc = 0;
        const unsigned o = (unsigned)tc * row_bytes;
#pragma unroll
        for (int q = 0; q < RQ; ++q)
            dst.v[q] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, row_off + o + 64u * q, 0, 0);
        if constexpr (H != 0) dst.h = __builtin_amdgcn_raw_buffer_load_b64(rsrc, half_off + o, 0, 0);
    };
#pragma unroll
    for (int k = 0; k < PF; ++k) load(k, xq[k]);

    // first: std::true_type for t == 0 (start probabilities instead of the transition
    // product); ragged: std::false_type when every lane of the wave is known to be
    // inside its utterance at step t (no selects); lpf: float partial sums of the
    // log-probability, folded into the double by the caller.
    auto step = [&](auto first, auto ragged, int t, const frame_t& xf, float (&lpf)[NT]) {
        f32x4 acc[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[nt] = c0[nt];
#pragma unroll
        for (int q = 0; q < RQ; ++q) {
            // ---- split the lane's 4 features and their squares into bf16 halves
            float x[4], sq[4];
            unsigned xl_[2], sl_[2];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                x[j] = __uint_as_float(xf.v[q][j]);
                sq[j] = x[j] * x[j];
            }
            u32x4 fh, fl;
            fh[0] = __builtin_amdgcn_perm(xf.v[q][1], xf.v[q][0], 0x07060302u);
            fh[1] = __builtin_amdgcn_perm(xf.v[q][3], xf.v[q][2], 0x07060302u);
            fh[2] = __builtin_amdgcn_perm(__float_as_uint(sq[1]), __float_as_uint(sq[0]), 0x07060302u);
            fh[3] = __builtin_amdgcn_perm(__float_as_uint(sq[3]), __float_as_uint(sq[2]), 0x07060302u);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const bf16x2 a = {(__bf16)(x[2 * j] - __uint_as_float(xf.v[q][2 * j] & 0xffff0000u)),
                                  (__bf16)(x[2 * j + 1] - __uint_as_float(xf.v[q][2 * j + 1] & 0xffff0000u))};
                const bf16x2 b = {
                    (__bf16)(sq[2 * j] - __uint_as_float(__float_as_uint(sq[2 * j]) & 0xffff0000u)),
                    (__bf16)(sq[2 * j + 1] - __uint_as_float(__float_as_uint(sq[2 * j + 1]) & 0xffff0000u))};
                xl_[j] = __builtin_bit_cast(unsigned, a);
                sl_[j] = __builtin_bit_cast(unsigned, b);
            }
            fl = u32x4{xl_[0], xl_[1], sl_[0], sl_[1]};
            const bf16x8 bh = __builtin_bit_cast(bf16x8, fh), bl = __builtin_bit_cast(bf16x8, fl);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                    __builtin_bit_cast(bf16x8, wh[nt][q]), bh, acc[nt], 0, 0, 0);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                    __builtin_bit_cast(bf16x8, wh[nt][q]), bl, acc[nt], 0, 0, 0);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                    __builtin_bit_cast(bf16x8, wl[nt][q]), bh, acc[nt], 0, 0, 0);
        }
        if constexpr (H != 0) {
            // ---- last block of 8 features: two per lane; hi halves in slots 0, 1 (x) and
            // 4, 5 (x^2), lo halves in slots 2, 3 and 6, 7
            const float x0 = __uint_as_float(xf.h[0]), x1 = __uint_as_float(xf.h[1]);
            const float s0 = x0 * x0, s1 = x1 * x1;
            const bf16x2 a = {(__bf16)(x0 - __uint_as_float(xf.h[0] & 0xffff0000u)),
                              (__bf16)(x1 - __uint_as_float(xf.h[1] & 0xffff0000u))};
            const bf16x2 b = {(__bf16)(s0 - __uint_as_float(__float_as_uint(s0) & 0xffff0000u)),
                              (__bf16)(s1 - __uint_as_float(__float_as_uint(s1) & 0xffff0000u))};
            const u32x4 fb = {__builtin_amdgcn_perm(xf.h[1], xf.h[0], 0x07060302u),
                              __builtin_bit_cast(unsigned, a),
                              __builtin_amdgcn_perm(__float_as_uint(s1), __float_as_uint(s0), 0x07060302u),
                              __builtin_bit_cast(unsigned, b)};
            const bf16x8 bb = __builtin_bit_cast(bf16x8, fb);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                    __builtin_bit_cast(bf16x8, wh[nt][RQ]), bb, acc[nt], 0, 0, 0);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                    __builtin_bit_cast(bf16x8, wl[nt][RQ]), bb, acc[nt], 0, 0, 0);
        }
        // (The recursion is a chain of dependent instructions, the emission products before it are
        // bulk work: of the two waves of a SIMD the one inside its chain issues first and the
        // other one's products fill the gaps.  0.543 -> 0.526 ms per 13.1 M frames on one box;
        // the reverse assignment and finer grades measured less.)
        __builtin_amdgcn_s_setprio(2);
        // ---- recursion, all tiles stage by stage (independent chains interleave)
        const bool live = t < T;
        float mx[NT], c[NT];
        f32x4 na[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            mx[nt] = fmaxf(fmaxf(acc[nt][0], acc[nt][1]), fmaxf(acc[nt][2], acc[nt][3]));
            if (SP == 8) mx[nt] = pair_max(mx[nt]);
            if (SP == 16) mx[nt] = quad_max(mx[nt]);
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            f32x4 b;
            const float sh = decltype(first)::value ? mx[nt] : mx[nt] + lc[nt];
#pragma unroll
            for (int r = 0; r < 4; ++r) b[r] = __builtin_amdgcn_exp2f(acc[nt][r] - sh);
            if constexpr (decltype(first)::value) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = (tile0 + nt) * 16 + 4 * g + r, m = row / SP, s_ = row % SP;
                    const bool ok = m < M && s_ < S;
                    na[nt][r] = tb.pi[ok ? m * S + s_ : 0] * (ok ? 1.f : 0.f) * b[r];
                }
            } else {
                const f32x4 al = alpha[nt];
                const unsigned u0 = __float_as_uint(al[0]), u1 = __float_as_uint(al[1]);
                const unsigned u2 = __float_as_uint(al[2]), u3 = __float_as_uint(al[3]);
                const bf16x2 l01 = {(__bf16)(al[0] - __uint_as_float(u0 & 0xffff0000u)),
                                    (__bf16)(al[1] - __uint_as_float(u1 & 0xffff0000u))};
                const bf16x2 l23 = {(__bf16)(al[2] - __uint_as_float(u2 & 0xffff0000u)),
                                    (__bf16)(al[3] - __uint_as_float(u3 & 0xffff0000u))};
                const u32x4 bq = {__builtin_amdgcn_perm(u1, u0, 0x07060302u),
                                  __builtin_amdgcn_perm(u3, u2, 0x07060302u),
                                  __builtin_bit_cast(unsigned, l01), __builtin_bit_cast(unsigned, l23)};
                const bf16x8 bv = __builtin_bit_cast(bf16x8, bq);
                f32x4 p = {0.f, 0.f, 0.f, 0.f};
                p = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a_hh[nt]), bv, p, 0, 0, 0);
                p = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a_l0[nt]), bv, p, 0, 0, 0);
                na[nt] = p * b;
            }
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            c[nt] = (na[nt][0] + na[nt][1]) + (na[nt][2] + na[nt][3]);
            if (SP == 8) c[nt] = pair_sum(c[nt]);
            if (SP == 16) c[nt] = quad_sum(c[nt]);
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            // A dead chain (c == 0: padding rows of the tile, or a sequence the
            // model cannot emit) has alpha == 0 from here on; its shift stays finite
            // so that the other model of the tile never meets inf or NaN.  Its own
            // log-probability becomes -inf as in the reference.
            const float l2 = __builtin_amdgcn_logf(c[nt]);
            const float lcn = c[nt] > 0.f ? l2 : 0.f;
            if constexpr (decltype(ragged)::value) {
                lpf[nt] += live ? mx[nt] + l2 : 0.f;
                lc[nt] = live ? lcn : lc[nt];
#pragma unroll
                for (int r = 0; r < 4; ++r) alpha[nt][r] = live ? na[nt][r] : alpha[nt][r];
            } else {
                lpf[nt] += mx[nt] + l2;
                lc[nt] = lcn;
                alpha[nt] = na[nt];
            }
        }
        __builtin_amdgcn_s_setprio(0);
    };
    auto fold = [&](float (&lpf)[NT]) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            lp[nt] += (double)lpf[nt];
            lpf[nt] = 0.f;
        }
    };

    // t = 0, then groups of PF steps with static queue slots (slot = t mod PF):
    // first the groups every lane is live in, then the ragged tail.
    int Tmin = T;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) Tmin = min(Tmin, __shfl_xor(Tmin, o, 64));
    float lpf[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) lpf[nt] = 0.f;
    if (Tmax > 0) {
        step(std::true_type{}, std::true_type{}, 0, xq[0], lpf);
        fold(lpf);
        load(PF, xq[0]);
    }
    int t0 = 1;
    for (; t0 + PF <= Tmin; t0 += PF) {
#pragma unroll
        for (int k = 0; k < PF; ++k) {
            step(std::false_type{}, std::false_type{}, t0 + k, xq[(1 + k) % PF], lpf);
            load(t0 + k + PF, xq[(1 + k) % PF]);
            // memory instructions stay on their side of this point (everything else may
            // cross): the loads of a step are issued before the next step begins
            __builtin_amdgcn_sched_barrier(0x78F);
        }
        fold(lpf);
    }
    for (; t0 < Tmax; t0 += PF) {
#pragma unroll
        for (int k = 0; k < PF; ++k) {
            step(std::false_type{}, std::true_type{}, t0 + k, xq[(1 + k) % PF], lpf);
            load(t0 + k + PF, xq[(1 + k) % PF]);
        }
        fold(lpf);
    }

    if (valid) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const bool writer = SP == 4 ? true : (SP == 8 ? (g & 1) == 0 : g == 0);
            if (writer && model[nt] < M) out[u * M + model[nt]] = lp[nt] * kLn2;
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

// Operand of the last block of 8 features, two per lane: hi halves in slots 0, 1 (x) and
// 4, 5 (x^2), lo halves in slots 2, 3 and 6, 7 (the table repeats the hi weights there).
__device__ __forceinline__ bf16x8 split_half_operand(const u32x2& xh) {
    const float x0 = __uint_as_float(xh[0]), x1 = __uint_as_float(xh[1]);
    const float s0 = x0 * x0, s1 = x1 * x1;
    const bf16x2 a = {(__bf16)(x0 - __uint_as_float(xh[0] & 0xffff0000u)),
                      (__bf16)(x1 - __uint_as_float(xh[1] & 0xffff0000u))};
    const bf16x2 b = {(__bf16)(s0 - __uint_as_float(__float_as_uint(s0) & 0xffff0000u)),
                      (__bf16)(s1 - __uint_as_float(__float_as_uint(s1) & 0xffff0000u))};
    const u32x4 fb = {__builtin_amdgcn_perm(xh[1], xh[0], 0x07060302u), __builtin_bit_cast(unsigned, a),
                      __builtin_amdgcn_perm(__float_as_uint(s1), __float_as_uint(s0), 0x07060302u),
                      __builtin_bit_cast(unsigned, b)};
    return __builtin_bit_cast(bf16x8, fb);
}

template <int SP, int NT, int RQ, int H>
void launch_split(const float* X, const int64_t* off, int64_t U, int64_t n_frames, int M, int S,
                  int D, int chunks, SplitTables tb, double* out, hipStream_t st) {
    const int64_t groups = (U + 15) / 16;
    const unsigned grid = (unsigned)((groups * chunks + 3) / 4);
    tb.chunks = chunks;
    hipLaunchKernelGGL((score_split_kernel<SP, NT, RQ, H>), dim3(grid), dim3(256), 0,
                       st, X, off, U, n_frames, M, S, D, tb, out);
}


}  // namespace

// Scoring dispatch for one padded state count (SP = 16: score_split16.hip, SP = 8: score_split8.hip, SP = 4:
// score_split4.hip).  RQ full 16-feature blocks, H = 1 for an 8-feature tail; the
// bank's tiles are taken NT at a time by `chunks` slices of the grid (the tables
// hold chunks * NT tiles).
int srx_split_score_sp8(const float* X, const int64_t* off, int64_t U, int64_t n_frames, int M,
                        int S, int D, int NT, int chunks, int RQ, int H, const unsigned* Wp, const float* cst,
                        const float* A, const float* pi, double* out, hipStream_t st, const int64_t* order);
int srx_split_score_sp4(const float* X, const int64_t* off, int64_t U, int64_t n_frames, int M,
                        int S, int D, int NT, int chunks, int RQ, int H, const unsigned* Wp, const float* cst,
                        const float* A, const float* pi, double* out, hipStream_t st, const int64_t* order);
int srx_split_score_sp16(const float* X, const int64_t* off, int64_t U, int64_t n_frames, int M,
                         int S, int D, int NT, int chunks, int RQ, int H, const unsigned* Wp, const float* cst,
                         const float* A, const float* pi, double* out, hipStream_t st, const int64_t* order);
