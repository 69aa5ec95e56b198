// Probe: register layout of v_mfma_f64_16x16x4_f64.  Assumed inputs: lane l supplies
// A[row = l % 16][k = l / 16] and B[k = l / 16][col = l % 16].  With A = indicator of (R0, K0)
// and B[K0][col] = 1 + col, D[R0][col] = 1 + col and zero elsewhere: print where it lands.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double f64x4 __attribute__((ext_vector_type(4)));
__global__ void k(double* out, int R0, int K0) {
    const int l = threadIdx.x;
    const double a = (l % 16 == R0 && l / 16 == K0) ? 1.0 : 0.0;
    const double b = (l / 16 == K0) ? 1.0 + l % 16 : 0.0;
    f64x4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    for (int i = 0; i < 4; ++i) out[l * 4 + i] = c[i];
}
int main() {
    double* d; (void)hipMalloc(&d, 64 * 4 * 8);
    double h[256];
    const int tests[4][2] = {{0, 0}, {5, 2}, {9, 1}, {15, 3}};
    for (auto& t : tests) {
        k<<<1, 64>>>(d, t[0], t[1]);
        (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("R0=%d K0=%d:", t[0], t[1]);
        int n = 0;
        for (int l = 0; l < 64; ++l)
            for (int i = 0; i < 4; ++i)
                if (h[l * 4 + i] != 0.0) { if (n < 16) printf(" (lane %d reg %d)=%g", l, i, h[l * 4 + i]); ++n; }
        printf("  [%d nonzero]\n", n);
    }
    return 0;
}
