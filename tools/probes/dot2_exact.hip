// Probe: is  x - trunc_bf16(x)  reproduced exactly by v_dot2_f32_bf16 with a -1 selector?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
__global__ void k(const float* x, float* ref, float* got, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (2 * i + 1 >= n) return;
    unsigned u0 = __float_as_uint(x[2 * i]), u1 = __float_as_uint(x[2 * i + 1]);
    ref[2 * i] = x[2 * i] - __uint_as_float(u0 & 0xffff0000u);
    ref[2 * i + 1] = x[2 * i + 1] - __uint_as_float(u1 & 0xffff0000u);
    unsigned hp = __builtin_amdgcn_perm(u1, u0, 0x07060302u);
    bf16x2 h = __builtin_bit_cast(bf16x2, hp);
    unsigned c0 = 0x0000bf80u, c1 = 0xbf800000u;
    asm volatile("" : "+v"(c0), "+v"(c1));                 // keep the selectors out of the inline-constant encoding
    bf16x2 s0 = __builtin_bit_cast(bf16x2, c0);   // {-1, 0}
    bf16x2 s1 = __builtin_bit_cast(bf16x2, c1);   // {0, -1}
    got[2 * i] = __builtin_amdgcn_fdot2_f32_bf16(h, s0, x[2 * i], false);
    got[2 * i + 1] = __builtin_amdgcn_fdot2_f32_bf16(h, s1, x[2 * i + 1], false);
}
int main() {
    const int n = 1 << 22;
    float* hx = (float*)malloc(n * 4);
    srand(1);
    for (int i = 0; i < n; ++i) {
        double r = (rand() / (double)RAND_MAX - 0.5) * 2;
        int e = rand() % 80 - 40;
        hx[i] = (float)ldexp(r, e);
    }
    hx[0] = 0.f; hx[1] = 1e-38f; hx[2] = 3e38f; hx[3] = -1e-40f; hx[4] = 1.0f; hx[5] = -65504.f;
    float *x, *a, *b;
    hipMalloc(&x, n * 4); hipMalloc(&a, n * 4); hipMalloc(&b, n * 4);
    hipMemcpy(x, hx, n * 4, hipMemcpyHostToDevice);
    k<<<n / 2 / 256, 256>>>(x, a, b, n);
    float* ha = (float*)malloc(n * 4); float* hb = (float*)malloc(n * 4);
    hipMemcpy(ha, a, n * 4, hipMemcpyDeviceToHost); hipMemcpy(hb, b, n * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < n; ++i)
        if (memcmp(&ha[i], &hb[i], 4) != 0 && !(ha[i] == 0.f && hb[i] == 0.f)) {
            if (bad < 10) printf("x=%g ref=%g got=%g\n", hx[i], ha[i], hb[i]);
            ++bad;
        }
    printf("mismatches %d of %d\n", bad, n);
    return 0;
}
