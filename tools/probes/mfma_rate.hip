// Probe: cost of the matrix instructions the kernels use, on a saturated gfx950 SIMD
// (4 accumulators in rotation, 2 and 4 waves per SIMD).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

#define M_BF16(acc) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))
#define M_BF16K16(acc) asm volatile("v_mfma_f32_16x16x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a4), "v"(b4))
#define M_F32(acc) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(x), "v"(y))
#define M_F32S(acc) asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(x), "v"(y))
#define M_F64(acc) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(d##acc) : "v"(dx), "v"(dy))
#define REP8(S) S(c0); S(c1); S(c2); S(c3); S(c0); S(c1); S(c2); S(c3);

typedef double f64x4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    f64x4 dc0 = {0, 0, 0, 0}, dc1 = dc0, dc2 = dc0, dc3 = dc0;
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 1e-3f); b[i] = (__bf16)(i * 1e-3f); }
    bf16x4 a4, b4;
    for (int i = 0; i < 4; ++i) { a4[i] = a[i]; b4[i] = b[i]; }
    float x = threadIdx.x * 1e-3f, y = 1e-3f;
    double dx = x, dy = y;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) { REP8(M_BF16) }
        if (MODE == 1) { REP8(M_F32) }
        if (MODE == 2) { REP8(M_F32S) }
        if (MODE == 3) { REP8(M_F64) }
        if (MODE == 4) { REP8(M_BF16K16) }
    }
    out[blockIdx.x * 256 + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3] + (float)(dc0[0] + dc1[1] + dc2[2] + dc3[3]);
}

template <int MODE>
void run(const char* name, float* out) {
    const int iters = 20000;
    double res[2];
    for (int w = 0; w < 2; ++w) {
        const int blocks = 256 * (w ? 4 : 2);
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        k<MODE><<<blocks, 256>>>(out, 16);
        (void)hipEventRecord(e0);
        k<MODE><<<blocks, 256>>>(out, iters);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        res[w] = ms * 1e6 / iters / 8 / (w ? 4 : 2);
    }
    printf("%-28s %6.2f ns per instruction at 2 waves/SIMD, %6.2f at 4\n", name, res[0], res[1]);
}
int main() {
    float* out; (void)hipMalloc(&out, 4 * 256 * 1024);
    run<0>("v_mfma_f32_16x16x32_bf16", out);
    run<1>("v_mfma_f32_16x16x4_f32", out);
    run<2>("v_mfma_f32_4x4x1_16b_f32", out);
    run<3>("v_mfma_f64_16x16x4_f64", out);
    run<4>("v_mfma_f32_16x16x16_bf16", out);
    return 0;
}
