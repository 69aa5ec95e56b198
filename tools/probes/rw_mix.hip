// Probe: what HBM delivers to the fused E-step's traffic without its arithmetic.  A wave walks
// batches of 16 utterance streams (200 rows x 156 bytes each) step by step, three rows ahead,
// and per step (forward) writes one contiguous 1 KiB lattice block plus 64 bytes of scale
// factors, or (backward) reads them back; 2 waves per SIMD like the kernel.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
struct row_t { u32x4 a, b; u32x2 c; };

// MODE bit 0: read X, bit 1: write lattice, bit 2: read lattice, bit 3: descending time
// BURST > 0: no scale factors, and the lattice blocks of BURST steps (a multiple of 3) are
// written back to back after the last of them (what staging them in LDS would give)
template <int MODE, int NV = 0, int NM = 0, int EARLY = 0, int SAUX = 0, int LAUX = 0, int BURST = 0>
__global__ __launch_bounds__(256, 2) void k(const float* X, int64_t n_batches, int T, int D, float* lat,
                                            float* scale, float* out) {
    const int lane = threadIdx.x & 63, g = lane >> 4, ul = lane & 15;
    const int64_t n_waves = (int64_t)gridDim.x * 4, wid = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t per = (n_batches + n_waves - 1) / n_waves;
    unsigned acc = 0;
    for (int64_t b = wid * per; b < (wid + 1) * per && b < n_batches; ++b) {
        const float* base = X + b * 16 * (int64_t)T * D;
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
            (void*)base, 0, (unsigned)(16ll * T * D * 4), 0x00020000);
        const __amdgpu_buffer_rsrc_t rl = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(lat + b * 16 * (int64_t)T * 16), 0, (unsigned)(16ll * T * 64), 0x00020000);
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(scale + b * 16 * (int64_t)T), 0, (unsigned)(16ll * T * 4), 0x00020000);
        const unsigned row_off = (unsigned)(ul * T * D * 4) + 16u * g, rb = (unsigned)D * 4u;
        auto tt = [&](int t) { t = t < T ? t : T - 1; return (MODE & 8) ? T - 1 - t : t; };
        auto load = [&](int t, row_t& r, u32x4& al, unsigned& sc) {
            const unsigned o = row_off + (unsigned)tt(t) * rb;
            if (MODE & 1) {
                r.a = __builtin_amdgcn_raw_buffer_load_b128(rsrc, o, 0, 0);
                r.b = __builtin_amdgcn_raw_buffer_load_b128(rsrc, o + 64u, 0, 0);
                r.c = __builtin_amdgcn_raw_buffer_load_b64(rsrc, o - 16u * g + 128u + 8u * g, 0, 0);
            }
            if (MODE & 4) {
                al = __builtin_amdgcn_raw_buffer_load_b128(rl, (unsigned)tt(t) * 1024u + ul * 64u + 16u * g, 0, LAUX);
                sc = __builtin_amdgcn_raw_buffer_load_b32(rs, (unsigned)tt(t) * 64u + ul * 4u, 0, LAUX);
            }
        };
        row_t s[3] = {};
        u32x4 al[3] = {};
        unsigned sc[3] = {};
#pragma unroll
        for (int i = 0; i < 3; ++i) load(i, s[i], al[i], sc[i]);
        for (int t0 = 0; t0 < T; t0 += 3) {
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const row_t& r = s[i];
                unsigned v = r.a[0] ^ r.a[1] ^ r.a[2] ^ r.a[3] ^ r.b[0] ^ r.b[1] ^ r.b[2] ^ r.b[3] ^ r.c[0] ^ r.c[1] ^
                             al[i][0] ^ al[i][1] ^ al[i][2] ^ al[i][3] ^ sc[i];
                acc += v;
                if (EARLY) load(t0 + i + 3, s[i], al[i], sc[i]);
                // arithmetic stand-in: NV dependent-free vector FMAs, NM f32 matrix instructions
                if (NV) {
                    float f[6] = {__uint_as_float(v), 1.f, 2.f, 3.f, 4.f, 5.f};
#pragma unroll
                    for (int j = 0; j < NV; ++j) asm volatile("v_fma_f32 %0, %1, %1, %0" : "+v"(f[j % 6]) : "v"(f[(j + 1) % 6]));
                    v ^= __float_as_uint(f[0] + f[1] + f[2] + f[3] + f[4] + f[5]);
                }
                if (NM) {
                    typedef float f32x4 __attribute__((ext_vector_type(4)));
                    f32x4 c[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
#pragma unroll
                    for (int j = 0; j < NM; ++j)
                        c[j & 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(v), 1.0f, c[j & 1], 0, 0, 0);
                    v ^= __float_as_uint(c[0][0] + c[1][1]);
                }
                if ((MODE & 2) && t0 + i < T && BURST == 0) {
                    __builtin_amdgcn_raw_buffer_store_b128(u32x4{v, acc, v, acc}, rl,
                                                           (unsigned)(t0 + i) * 1024u + ul * 64u + 16u * g, 0, SAUX);
                    __builtin_amdgcn_raw_buffer_store_b32(v, rs, g == 0 ? (unsigned)(t0 + i) * 64u + ul * 4u : 0xFFFFFF00u, 0, SAUX);
                }
                if constexpr (BURST > 0) {
                    if ((MODE & 2) && i == 2 && ((t0 + 3) % BURST == 0 || t0 + 3 >= T)) {
                        const int first = (t0 + 3 - 1) / BURST * BURST;
#pragma unroll
                        for (int q = 0; q < BURST; ++q)
                            __builtin_amdgcn_raw_buffer_store_b128(
                                u32x4{v, acc, v, acc}, rl,
                                first + q < T ? (unsigned)(first + q) * 1024u + ul * 64u + 16u * g : 0xFFFFFF00u, 0, SAUX);
                    }
                }
                if (!EARLY) load(t0 + i + 3, s[i], al[i], sc[i]);
            }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = (float)acc;
}

template <int MODE, int NV = 0, int NM = 0, int EARLY = 0, int SAUX = 0, int LAUX = 0, int BURST = 0>
void run(const char* what, double bytes_per_frame, const float* X, int64_t U, int T, int D, float* lat, float* scale,
         float* out) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int i = 0; i < 5; ++i) k<MODE, NV, NM, EARLY, SAUX, LAUX, BURST><<<512, 256>>>(X, U / 16, T, D, lat, scale, out);
    (void)hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) k<MODE, NV, NM, EARLY, SAUX, LAUX, BURST><<<512, 256>>>(X, U / 16, T, D, lat, scale, out);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    ms /= 10;
    printf("%-46s %.4f ms  %6.1f B/frame  %.3f TB/s\n", what, ms, bytes_per_frame, (double)U * T * bytes_per_frame / ms / 1e9);
}

int main() {
    const int64_t U = 131072; const int T = 200, D = 39;
    float *X, *out, *lat, *scale;
    (void)hipMalloc(&X, U * T * D * 4 + 4096);
    (void)hipMemset(X, 1, U * T * D * 4 + 4096);
    (void)hipMalloc(&lat, U * T * 64);
    (void)hipMalloc(&scale, U * T * 4);
    (void)hipMemset(lat, 0, U * T * 64);
    (void)hipMemset(scale, 0, U * T * 4);
    (void)hipMalloc(&out, 4 * 256 * 512);
    run<3, 100, 0>("forward + 100 vector FMAs per step", 224, X, U, T, D, lat, scale, out);
    run<3, 200, 0>("forward + 200 vector FMAs per step", 224, X, U, T, D, lat, scale, out);
    run<3, 0, 24>("forward + 24 matrix instructions per step", 224, X, U, T, D, lat, scale, out);
    run<3, 100, 24>("forward + 100 FMAs + 24 matrix", 224, X, U, T, D, lat, scale, out);
    run<0, 100, 24>("no memory: 100 FMAs + 24 matrix", 224, X, U, T, D, lat, scale, out);
    run<0, 100, 0>("no memory: 100 FMAs", 224, X, U, T, D, lat, scale, out);
    run<0, 0, 24>("no memory: 24 matrix", 224, X, U, T, D, lat, scale, out);
    run<1, 100, 24>("read X + 100 FMAs + 24 matrix", 156, X, U, T, D, lat, scale, out);
    run<5 | 8, 150, 48>("backward + 150 FMAs + 48 matrix", 224, X, U, T, D, lat, scale, out);
    run<0, 150, 48>("no memory: 150 FMAs + 48 matrix", 224, X, U, T, D, lat, scale, out);
    // the lattice blocks of several steps written back to back (no scale factors)
    run<3, 0, 0, 0, 0, 0, 3>("forward, lattice of 3 steps at a time", 220, X, U, T, D, lat, scale, out);
    run<3, 0, 0, 0, 0, 0, 6>("forward, lattice of 6 steps at a time", 220, X, U, T, D, lat, scale, out);
    run<3, 0, 0, 0, 0, 0, 12>("forward, lattice of 12 steps at a time", 220, X, U, T, D, lat, scale, out);
    run<3, 0, 0, 0, 0, 0, 24>("forward, lattice of 24 steps at a time", 220, X, U, T, D, lat, scale, out);
    // cache-policy bits of the lattice stores / loads (bit 0 sc0, bit 1 nt, bit 4 sc1)
    run<3, 0, 0, 0, 2, 0>("forward, lattice stores nt", 224, X, U, T, D, lat, scale, out);
    run<3, 0, 0, 0, 1, 0>("forward, lattice stores sc0", 224, X, U, T, D, lat, scale, out);
    run<3, 0, 0, 0, 17, 0>("forward, lattice stores sc0 sc1", 224, X, U, T, D, lat, scale, out);
    run<3, 0, 0, 0, 19, 0>("forward, lattice stores sc0 sc1 nt", 224, X, U, T, D, lat, scale, out);
    run<5 | 8, 0, 0, 0, 0, 2>("backward, lattice loads nt", 224, X, U, T, D, lat, scale, out);
    for (int rep = 0; rep < 1; ++rep) {
        run<1>("read X", 156, X, U, T, D, lat, scale, out);
        run<2>("write lattice + scale", 68, X, U, T, D, lat, scale, out);
        run<3>("forward: read X, write lattice + scale", 224, X, U, T, D, lat, scale, out);
        run<4>("read lattice + scale", 68, X, U, T, D, lat, scale, out);
        run<5 | 8>("backward: read X, lattice + scale, descending", 224, X, U, T, D, lat, scale, out);
    }
    return 0;
}
