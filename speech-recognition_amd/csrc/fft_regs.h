// Register-resident FFT pieces shared by the front-end kernels that keep a frame's transform in
// registers: complex types, radix-2 transforms of 16 / 32 points with compile-time twiddles,
// the real-input unpack to power bins, and the small cross-lane helpers around them.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

// a * b where a zero factor gives 0 whatever the other one is (inf, NaN)
__device__ __forceinline__ float mul_zero_wins(float a, float b) {
    float r;
    asm("v_mul_legacy_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

__device__ __forceinline__ float sum_rows(float v) {
    // v[l] + v[l^16] + v[l^32] + v[l^48]
    unsigned u = __float_as_uint(v);
    auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    u = __float_as_uint(v);
    r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

constexpr int ilog2(int v) { return v <= 1 ? 0 : 1 + ilog2(v >> 1); }
constexpr int brev(int i, int bits) {
    int r = 0;
    for (int b = 0; b < bits; ++b) r |= ((i >> b) & 1) << (bits - 1 - b);
    return r;
}

// Complex numbers are (re, im) register pairs so that additions and the two halves
// of a complex product are packed-f32 instructions.
typedef float v2f __attribute__((ext_vector_type(2)));

__device__ __forceinline__ v2f cmul(v2f w, v2f z) {
    // (w.x z.x - w.y z.y, w.x z.y + w.y z.x)
    const v2f wyy = {-w.y, w.y};
    return w.xx * z + wyy * z.yx;
}

// The same product for operands in vector registers, as the two packed instructions it is:
// half swaps and sign changes ride on the operand modifiers (the compiler spends an xor and a
// move per product on building (-w.y, w.y)).
__device__ __forceinline__ v2f cmul_v(v2f w, v2f z) {
    v2f t, r;
    // t = (-w.y z.y, w.y z.x)
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,0]"
        : "=v"(t) : "v"(w), "v"(z));
    // r = (w.x z.x, w.x z.y) + t
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,1]"
        : "=v"(r) : "v"(w), "v"(z), "v"(t));
    return r;
}

// W_32^q = exp(-2 pi i q / 32), q < 16, rounded to float from the double values (the
// same numbers the plan's twiddle table holds).  The register transforms take their
// twiddles from here: compile-time constants sit in scalar registers or literals, a
// table read at run time would cost a vector register pair per twiddle.
__device__ constexpr float kCos32[16] = {
    1.0f, 0.9807852506637573f, 0.9238795042037964f, 0.8314695954322815f, 0.7071067690849304f,
    0.5555702447891235f, 0.3826834261417389f, 0.19509032368659973f, 6.123234262925839e-17f,
    -0.19509032368659973f, -0.3826834261417389f, -0.5555702447891235f, -0.7071067690849304f,
    -0.8314695954322815f, -0.9238795042037964f, -0.9807852506637573f};
__device__ constexpr float kSin32[16] = {
    -0.0f, -0.19509032368659973f, -0.3826834261417389f, -0.5555702447891235f, -0.7071067690849304f,
    -0.8314695954322815f, -0.9238795042037964f, -0.9807852506637573f, -1.0f, -0.9807852506637573f,
    -0.9238795042037964f, -0.8314695954322815f, -0.7071067690849304f, -0.5555702447891235f,
    -0.3826834261417389f, -0.19509032368659973f};

// In-place radix-2 DIT on registers (Q = 16 or 32).  Input must already sit in
// bit-reversed order (a[brev(n)] = x[n]); the twiddles W_Q^q are compile-time constants.
// NR < Q: inputs n >= NR are known to be zero (a frame shorter than the transform: the tail of
// the zero padding), so the first-stage butterflies whose second input is one of them are
// copies.
template <int Q, int NR = Q>
__device__ __forceinline__ void fft_regs(v2f (&z)[Q]) {
#pragma unroll
    for (int s = 1; s <= ilog2(Q); ++s) {
        const int m = 1 << s, half = m >> 1, tstep = Q / m;
#pragma unroll
        for (int k = 0; k < Q; k += m) {
#pragma unroll
            for (int jj = 0; jj < half; ++jj) {
                const int i0 = k + jj, i1 = i0 + half;
                v2f t;
                if (s == 1 && brev(i1, ilog2(Q)) >= NR) {
                    z[i1] = z[i0];
                    continue;
                }
                if (jj == 0) {
                    t = z[i1];
                } else if (2 * jj == half) {     // W = -i: t = (z.y, -z.x), folded into the two adds
                    v2f a, b;
                    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,0] neg_hi:[0,1]"
                        : "=v"(a) : "v"(z[i0]), "v"(z[i1]));
                    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,0]"
                        : "=v"(b) : "v"(z[i0]), "v"(z[i1]));
                    z[i0] = a;
                    z[i1] = b;
                    continue;
                } else {
                    const int q32 = jj * tstep * (32 / Q);      // folds to a constant once unrolled
                    t = cmul(v2f{kCos32[q32], kSin32[q32]}, z[i1]);
                }
                z[i1] = z[i0] - t;
                z[i0] = z[i0] + t;
            }
        }
    }
}

// The same transforms on separate real and imaginary registers with plain (one value per lane)
// instructions.  A packed instruction costs two plain ones in issue time, so the arithmetic is
// priced alike, but plain operands may sit in any register: none of the moves that line pairs
// up for the packed forms, sign changes and half swaps are free (operand order, v_sub, negated
// fma operands), and a twiddle of (1 -+ i) / sqrt 2 is two additions and four fused
// multiply-adds instead of a full complex product.
struct cf {
    float x, y;
};

template <int Q, int NR = Q>
__device__ __forceinline__ void fft_regs(cf (&z)[Q]) {
#pragma unroll
    for (int s = 1; s <= ilog2(Q); ++s) {
        const int m = 1 << s, half = m >> 1, tstep = Q / m;
#pragma unroll
        for (int k = 0; k < Q; k += m) {
#pragma unroll
            for (int jj = 0; jj < half; ++jj) {
                const int i0 = k + jj, i1 = i0 + half;
                const cf a = z[i0], b = z[i1];
                if (s == 1 && brev(i1, ilog2(Q)) >= NR) {      // (zero input: see the packed form)
                    z[i1] = a;
                } else if (jj == 0) {
                    z[i0] = cf{a.x + b.x, a.y + b.y};
                    z[i1] = cf{a.x - b.x, a.y - b.y};
                } else if (2 * jj == half) {     // W = -i: t = (b.y, -b.x)
                    z[i0] = cf{a.x + b.y, a.y - b.x};
                    z[i1] = cf{a.x - b.y, a.y + b.x};
                } else if (4 * jj == half) {     // W = (1 - i) / sqrt 2: t = c (b.x + b.y, b.y - b.x)
                    const float c = 0.7071067690849304f;
                    const float u = b.x + b.y, v = b.y - b.x;
                    z[i0] = cf{fmaf(c, u, a.x), fmaf(c, v, a.y)};
                    z[i1] = cf{fmaf(-c, u, a.x), fmaf(-c, v, a.y)};
                } else if (4 * jj == 3 * half) { // W = (-1 - i) / sqrt 2: t = c (b.y - b.x, -(b.x + b.y))
                    const float c = 0.7071067690849304f;
                    const float u = b.y - b.x, v = b.x + b.y;
                    z[i0] = cf{fmaf(c, u, a.x), fmaf(-c, v, a.y)};
                    z[i1] = cf{fmaf(-c, u, a.x), fmaf(c, v, a.y)};
                } else {
                    const int q32 = jj * tstep * (32 / Q);      // folds to a constant once unrolled
                    const float wr = kCos32[q32], wi = kSin32[q32];
                    // t = (wr b.x - wi b.y, wr b.y + wi b.x), a +- t as chains of two fma each
                    z[i0] = cf{fmaf(wr, b.x, fmaf(-wi, b.y, a.x)), fmaf(wr, b.y, fmaf(wi, b.x, a.y))};
                    z[i1] = cf{fmaf(-wr, b.x, fmaf(wi, b.y, a.x)), fmaf(-wr, b.y, fmaf(-wi, b.x, a.y))};
                }
            }
        }
    }
}

// Y = W z for the transpose tile (unit: W is 1)
__device__ __forceinline__ v2f twiddled(v2f w, v2f z, bool) { return cmul_v(w, z); }
__device__ __forceinline__ v2f twiddled(v2f w, cf z, bool unit) {
    return unit ? v2f{z.x, z.y} : v2f{fmaf(w.x, z.x, -w.y * z.y), fmaf(w.x, z.y, w.y * z.x)};
}
// Real-input unpack of the bin pair (k, H - k) from zk = Z[k], zc = Z[H-k] and w = W_N^k:
// with E = (Z[k] + conj Z[H-k]) / 2 and O = -i (Z[k] - conj Z[H-k]) / 2, X[k] = E + W^k O and
// X[H-k] = conj(E - W^k O); the halves are left to `scale` (|2X|^2 / 4).
__device__ __forceinline__ void power_pair(v2f zk, v2f zc, v2f w, float scale, float& pk, float& pc) {
    v2f e2, dd, t1, pp;
    // e2 = Z[k] + conj Z[H-k], dd = Z[k] - conj Z[H-k]
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,0] neg_hi:[0,1]" : "=v"(e2) : "v"(zk), "v"(zc));
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,0]" : "=v"(dd) : "v"(zk), "v"(zc));
    // pp = W^k (dd.y, -dd.x) = (w.x dd.y + w.y dd.x, w.y dd.y - w.x dd.x)
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1]" : "=v"(t1) : "v"(w), "v"(dd));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[0,0,1] neg_lo:[0,0,0] neg_hi:[0,1,0]"
        : "=v"(pp) : "v"(w), "v"(dd), "v"(t1));
    const v2f xa = e2 + pp, xb = e2 - pp;
    const v2f qa = xa * xa, qb = xb * xb;
    pk = (qa.x + qa.y) * scale;
    pc = (qb.x + qb.y) * scale;
}
__device__ __forceinline__ void power_pair(cf zk, cf zc, v2f w, float scale, float& pk, float& pc) {
    const float ex = zk.x + zc.x, ey = zk.y - zc.y, dx = zk.x - zc.x, dy = zk.y + zc.y;
    const float px = fmaf(w.x, dy, w.y * dx), py = fmaf(w.y, dy, -w.x * dx);
    const float ax = ex + px, ay = ey + py, bx = ex - px, by = ey - py;
    pk = fmaf(ax, ax, ay * ay) * scale;
    pc = fmaf(bx, bx, by * by) * scale;
}

}  // namespace
