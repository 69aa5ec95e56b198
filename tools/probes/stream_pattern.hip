// Probe: 65,536 utterance streams of 200 rows x 156 bytes, read the way the scoring kernel
// reads them (a wave = 16 streams, lane group g takes bytes 64q+16g.. of a row).  Does HBM
// deliver more when a stream's rows are requested in bursts of 4 or 8 consecutive rows
// (the same DRAM page) instead of one row per step?  Loads only, summed to keep them alive.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
struct row_t { u32x4 a, b; u32x2 c; };

template <int BURST>
__global__ __launch_bounds__(256) void k(const float* X, int64_t U, int T, int D, float* out) {
    const int lane = threadIdx.x & 63, g = lane >> 4, ul = lane & 15;
    const int64_t group = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (group * 16 >= U) return;
    const int64_t u = group * 16 + ul;
    const float* base = X + group * 16 * (int64_t)T * D;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)base, 0, (unsigned)(16ll * T * D * 4), 0x00020000);
    const unsigned row_off = (unsigned)(ul * T * D * 4) + 16u * g, rb = (unsigned)D * 4u;
    (void)u;
    auto load = [&](int t, row_t& r) {
        const unsigned o = row_off + (unsigned)(t < T ? t : T - 1) * rb;
        r.a = __builtin_amdgcn_raw_buffer_load_b128(rsrc, o, 0, 0);
        r.b = __builtin_amdgcn_raw_buffer_load_b128(rsrc, o + 64u, 0, 0);
        r.c = __builtin_amdgcn_raw_buffer_load_b64(rsrc, o - 16u * g + 128u + 8u * g, 0, 0);
    };
    unsigned acc = 0;
    auto use = [&](const row_t& r) {
        acc += r.a[0] ^ r.a[1] ^ r.a[2] ^ r.a[3] ^ r.b[0] ^ r.b[1] ^ r.b[2] ^ r.b[3] ^ r.c[0] ^ r.c[1];
    };
    row_t s[2 * BURST < 4 ? 4 : 2 * BURST];
    if (BURST == 1) {
#pragma unroll
        for (int i = 0; i < 4; ++i) load(i, s[i]);
        for (int t0 = 0; t0 < T; t0 += 4) {
#pragma unroll
            for (int i = 0; i < 4; ++i) { use(s[i]); load(t0 + i + 4, s[i]); }
        }
    } else {
#pragma unroll
        for (int i = 0; i < BURST; ++i) load(i, s[i]);
        for (int t0 = 0; t0 < T; t0 += 2 * BURST) {
#pragma unroll
            for (int i = 0; i < BURST; ++i) load(t0 + BURST + i, s[BURST + i]);
#pragma unroll
            for (int i = 0; i < BURST; ++i) use(s[i]);
#pragma unroll
            for (int i = 0; i < BURST; ++i) load(t0 + 2 * BURST + i, s[i]);
#pragma unroll
            for (int i = 0; i < BURST; ++i) use(s[BURST + i]);
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = (float)acc;
}

// For scale: the same bytes as one contiguous, fully coalesced stream (16 bytes per lane).
__global__ __launch_bounds__(256) void k_seq(const u32x4* X, int64_t n, float* out) {
    unsigned acc = 0;
    const int64_t stride = (int64_t)gridDim.x * 256 * 4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        u32x4 v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = i + j * (stride / 4) < n ? X[i + j * (stride / 4)] : u32x4{0, 0, 0, 0};
#pragma unroll
        for (int j = 0; j < 4; ++j) acc += v[j][0] ^ v[j][1] ^ v[j][2] ^ v[j][3];
    }
    out[blockIdx.x * 256 + threadIdx.x] = (float)acc;
}

template <int BURST>
void run(const float* X, int64_t U, int T, int D, float* out, int lds = 0) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const unsigned grid = (unsigned)((U / 16 + 3) / 4);
    k<BURST><<<grid, 256, lds>>>(X, U, T, D, out);
    (void)hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) k<BURST><<<grid, 256, lds>>>(X, U, T, D, out);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    ms /= 10;
    printf("burst %d rows, %s: %.4f ms  %.3f TB/s\n", BURST,
           lds >= 65536 ? "2 waves/SIMD (the scoring kernel's occupancy)" : lds >= 40000 ? "3 waves/SIMD" : "4 waves/SIMD",
           ms, (double)U * T * D * 4 / ms / 1e9);
}

int main() {
    const int64_t U = 65536; const int T = 200, D = 39;
    float *X, *out;
    (void)hipMalloc(&X, U * T * D * 4 + 4096);
    (void)hipMemset(X, 1, U * T * D * 4 + 4096);
    (void)hipMalloc(&out, 4 * 256 * 8192);      // one float per thread of the largest grid (8192 workgroups)
    for (int blocks = 1024; blocks <= 8192; blocks *= 2) {
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        const int64_t n = U * T * D / 4;
        k_seq<<<blocks, 256>>>((const u32x4*)X, n, out);
        (void)hipEventRecord(e0);
        for (int i = 0; i < 10; ++i) k_seq<<<blocks, 256>>>((const u32x4*)X, n, out);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        ms /= 10;
        printf("contiguous read, %5d workgroups: %.4f ms  %.3f TB/s\n", blocks, ms, (double)n * 16 / ms / 1e9);
    }
    for (int rep = 0; rep < 2; ++rep) {
        run<1>(X, U, T, D, out);
        run<2>(X, U, T, D, out);
        run<4>(X, U, T, D, out);
        run<8>(X, U, T, D, out);
    }
    // the same with the occupancy held down by an unused LDS allocation per workgroup
    for (int lds = 48 * 1024; lds <= 64 * 1024; lds += 16 * 1024) {
        run<1>(X, U, T, D, out, lds);
        run<2>(X, U, T, D, out, lds);
        run<4>(X, U, T, D, out, lds);
    }
    return 0;
}
