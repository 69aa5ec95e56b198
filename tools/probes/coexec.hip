// Probe: do matrix (MFMA) and vector instructions co-execute on a gfx950 SIMD, within one
// wave (interleaved stream) and across two waves of one SIMD?
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define MFMA(acc) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))
#define FMA(r) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(r) : "v"(x), "v"(y))
#define FMA6 FMA(r0); FMA(r1); FMA(r2); FMA(r3); FMA(r4); FMA(r5);
#define EXP(r) asm volatile("v_exp_f32 %0, %0" : "+v"(r))

// MODE 0: 8 MFMA / iter.  1: 48 FMA / iter.  2: 8 x (MFMA, 6 FMA) interleaved.
// 3: even waves MODE 0, odd waves MODE 1 (needs >= 2 waves per SIMD: 512 threads).
// 4: 8 MFMA then 48 FMA (same wave, not interleaved)
template <int MODE>
__global__ __launch_bounds__(512) void k(float* out, int iters) {
    f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 1e-3f); b[i] = (__bf16)(i * 1e-3f); }
    float r0 = threadIdx.x, r1 = 1, r2 = 2, r3 = 3, r4 = 4, r5 = 5, x = 0.999f, y = 1e-3f;
    const int wave = threadIdx.x >> 6;
    const bool odd = (wave ^ (wave >> 2)) & 1;   // one wave of each kind per SIMD, whether wave w sits on SIMD w % 4 or w / 2
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0 || (MODE == 3 && !odd)) {
            MFMA(c0); MFMA(c1); MFMA(c2); MFMA(c3); MFMA(c0); MFMA(c1); MFMA(c2); MFMA(c3);
        } else if (MODE == 1 || (MODE == 3 && odd)) {
            FMA6 FMA6 FMA6 FMA6 FMA6 FMA6 FMA6 FMA6
        } else if (MODE == 2) {
            MFMA(c0); FMA6 MFMA(c1); FMA6 MFMA(c2); FMA6 MFMA(c3); FMA6
            MFMA(c0); FMA6 MFMA(c1); FMA6 MFMA(c2); FMA6 MFMA(c3); FMA6
        } else if (MODE == 4) {
            MFMA(c0); MFMA(c1); MFMA(c2); MFMA(c3); MFMA(c0); MFMA(c1); MFMA(c2); MFMA(c3);
            FMA6 FMA6 FMA6 FMA6 FMA6 FMA6 FMA6 FMA6
        } else if (MODE == 6) {
            MFMA(c0); MFMA(c1); MFMA(c2); MFMA(c3); FMA6 FMA6 FMA6 FMA6
            MFMA(c0); MFMA(c1); MFMA(c2); MFMA(c3); FMA6 FMA6 FMA6 FMA6
        } else if (MODE == 7) {           // (2 MFMA, 12 FMA) x 4
            MFMA(c0); MFMA(c1); FMA6 FMA6 MFMA(c2); MFMA(c3); FMA6 FMA6
            MFMA(c0); MFMA(c1); FMA6 FMA6 MFMA(c2); MFMA(c3); FMA6 FMA6
        } else if (MODE == 5) {           // 12 transcendentals
            EXP(r0); EXP(r1); EXP(r2); EXP(r3); EXP(r4); EXP(r5); EXP(r0); EXP(r1); EXP(r2); EXP(r3); EXP(r4); EXP(r5);
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3] + r0 + r1 + r2 + r3 + r4 + r5;
}

template <int MODE>
void run(const char* name, int threads, int blocks, float* out, int iters) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<MODE><<<blocks, threads>>>(out, 16);
    (void)hipEventRecord(e0);
    k<MODE><<<blocks, threads>>>(out, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double waves_per_simd = (double)blocks * threads / 64 / 1024;
    printf("%-40s %4.0f waves/SIMD: %8.3f ms = %7.2f ns per wave-iteration\n", name, waves_per_simd, ms,
           ms * 1e6 / iters / waves_per_simd);
}

int main() {
    float* out; (void)hipMalloc(&out, 4 * 512 * 2048 * 4);
    const int it = 20000;
    for (int wps = 1; wps <= 4; wps *= 2) {
        const int blocks = 256 * wps;
        run<0>("8 MFMA 16x16x32 bf16", 256, blocks, out, it);
        run<1>("48 v_fma_f32", 256, blocks, out, it);
        run<5>("12 v_exp_f32", 256, blocks, out, it);
        run<2>("8 x (MFMA, 6 FMA)", 256, blocks, out, it);
        run<7>("4 x (2 MFMA, 12 FMA)", 256, blocks, out, it);
        run<6>("2 x (4 MFMA, 24 FMA)", 256, blocks, out, it);
        run<4>("8 MFMA then 48 FMA", 256, blocks, out, it);
    }
    run<3>("half the waves 8 MFMA, half 48 FMA", 512, 256, out, it);
    run<3>("half the waves 8 MFMA, half 48 FMA", 512, 512, out, it);
    return 0;
}
