// Probe: the fused Viterbi kernel's access pattern without its arithmetic.  65,536 utterances of
// 200 rows x 156 bytes; a wave owns 16 utterances and visits them in turn, 16 consecutive rows
// (2,496 contiguous bytes) per visit, RING visits requested ahead.  Loads only, summed to keep
// them alive.  Variables: waves per workgroup / workgroups per CU (occupancy), RING, rows per
// visit as 2 x 16 B + 8 B per lane group (exact rows) and a descriptor per visit.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
struct item_t { u32x4 a, b; u32x2 c; };

template <int RING, bool SEQ, int ST>
__global__ __launch_bounds__(1024) void k(const float* X, int64_t U, int T, int D, float* out, unsigned* bw) {
    const int lane = threadIdx.x & 63, g = lane >> 4, ul = lane & 15;
    const int wpb = blockDim.x >> 6;
    const int64_t n_groups = U / 16, n_waves = (int64_t)gridDim.x * wpb;
    unsigned acc = 0;
    const unsigned lane_off = (unsigned)ul * (unsigned)D * 4u + 16u * g;
    const unsigned half_off = (unsigned)ul * (unsigned)D * 4u + 128u + 8u * g;
    for (int64_t grp = (int64_t)blockIdx.x * wpb + (threadIdx.x >> 6); grp < n_groups; grp += n_waves) {
        const int n_chunks = (T + 15) / 16, n_items = n_chunks * 16;
        auto issue = [&](int i, item_t& d) {
            // SEQ: all chunks of utterance 0, then of utterance 1, ...; else chunk k of all 16 in turn
            const int k_ = SEQ ? i % n_chunks : i >> 4, e = SEQ ? i / n_chunks : i & 15;
            const bool on = i < n_items;
            const int64_t f = (grp * 16 + e) * T + 16 * k_;
            const int64_t rest = on ? (int64_t)(T - 16 * k_) * D * 4 : 0;
            const uintptr_t a = (uintptr_t)(X + (on ? f * D : 0));
            const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a);
            const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
            const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
                (void*)(((uintptr_t)hi << 32) | lo), 0, __builtin_amdgcn_readfirstlane((unsigned)rest), 0x00020000);
            d.a = __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane_off, 0, 0);
            d.b = __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane_off + 64u, 0, 0);
            d.c = __builtin_amdgcn_raw_buffer_load_b64(rsrc, half_off, 0, 0);
        };
        auto use = [&](const item_t& r) {
            acc += r.a[0] ^ r.a[1] ^ r.a[2] ^ r.a[3] ^ r.b[0] ^ r.b[1] ^ r.b[2] ^ r.b[3] ^ r.c[0] ^ r.c[1];
        };
        item_t s[RING];
#pragma unroll
        for (int j = 0; j < RING; ++j) issue(j, s[j]);
        for (int i = 0; i < n_items; i += RING) {
#pragma unroll
            for (int j = 0; j < RING; ++j) {
                use(s[j]);
                issue(i + j + RING, s[j]);
            }
            // ST = 1: after every 16 visits, lanes 0..15 store 16 words per utterance (the
            // back-pointer words of a chunk), as four 16-byte stores; ST = 2: as non-temporal stores
            if (ST == 4 && !SEQ && ((i + RING) & 63) == 0) {
                // ST = 4: the words of four chunks at once (4 KiB per group, contiguous and aligned)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    unsigned* p = bw + ((grp * 13 + (i >> 4) - 3 + q) * 256 + 4 * lane);
                    *(u32x4*)p = u32x4{acc, acc + 1, acc + 2, acc + (unsigned)q};
                }
            }
            if (ST == 3 && !SEQ && ((i + RING) & 15) == 0) {
                // ST = 3: the same 1 KiB per chunk as ONE store instruction to a chunk-major array
                // (a contiguous, aligned KiB per group and chunk)
                unsigned* p = bw + ((grp * 13 + (i >> 4)) * 256 + 4 * lane);
                *(u32x4*)p = u32x4{acc, acc + 1, acc + 2, acc + 3};
            }
            if (ST && ST < 3 && !SEQ && ((i + RING) & 15) == 0 && lane < 16) {
                unsigned* p = bw + ((grp * 16 + lane) * T + (i >> 4) * 16);
                if ((i >> 4) * 16 + 16 <= T) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const u32x4 v = {acc, acc + 1, acc + 2, acc + (unsigned)q};
                        if (ST == 2) __builtin_nontemporal_store(v, (u32x4*)(p + 4 * q));
                        else *(u32x4*)(p + 4 * q) = v;
                    }
                }
            }
        }
    }
    out[blockIdx.x * 1024 + threadIdx.x] = (float)acc;
}

template <int RING, bool SEQ, int ST = 0>
void run(const float* X, int64_t U, int T, int D, float* out, int wpb, int per_cu, int lds, unsigned* bw = nullptr) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const unsigned grid = 256u * per_cu;
    (void)hipFuncSetAttribute((const void*)k<RING, SEQ, ST>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    for (int i = 0; i < 20; ++i) k<RING, SEQ, ST><<<grid, 64 * wpb, lds>>>(X, U, T, D, out, bw);
    (void)hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) k<RING, SEQ, ST><<<grid, 64 * wpb, lds>>>(X, U, T, D, out, bw);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    ms /= 10;
    printf("ring %d %s%s, %2d waves per workgroup x %d per CU: %.4f ms  %.3f TB/s\n", RING,
           SEQ ? "utterance by utterance" : "16 utterances in turn",
           ST == 1 ? " + 4 B/row stored" : ST == 2 ? " + 4 B/row stored non-temporally" :
           ST == 3 ? " + 4 B/row stored as one aligned KiB per chunk" :
           ST == 4 ? " + 4 B/row stored as four aligned KiB every four chunks" : "", wpb, per_cu, ms,
           (double)U * T * D * 4 / ms / 1e9);
}

int main() {
    const int64_t U = 65536; const int T = 200, D = 39;
    float *X, *out;
    (void)hipMalloc(&X, U * T * D * 4 + 4096);
    (void)hipMemset(X, 1, U * T * D * 4 + 4096);
    (void)hipMalloc(&out, 4 * 1024 * 1024);
    run<4, false>(X, U, T, D, out, 16, 1, 150 * 1024);
    run<4, false>(X, U, T, D, out, 4, 4, 38 * 1024);
    run<4, false>(X, U, T, D, out, 4, 3, 50 * 1024);
    run<4, false>(X, U, T, D, out, 4, 2, 70 * 1024);
    run<2, false>(X, U, T, D, out, 16, 1, 150 * 1024);
    run<8, false>(X, U, T, D, out, 16, 1, 150 * 1024);
    run<8, false>(X, U, T, D, out, 4, 2, 70 * 1024);
    unsigned* bw;
    (void)hipMalloc(&bw, U * 13 * 16 * 4 + 4096);
    run<4, false, 3>(X, U, T, D, out, 16, 1, 150 * 1024, bw);
    run<4, false, 3>(X, U, T, D, out, 4, 3, 50 * 1024, bw);
    run<4, false, 4>(X, U, T, D, out, 16, 1, 150 * 1024, bw);
    run<4, false, 4>(X, U, T, D, out, 4, 3, 50 * 1024, bw);
    run<4, false, 1>(X, U, T, D, out, 16, 1, 150 * 1024, bw);
    run<4, false, 2>(X, U, T, D, out, 16, 1, 150 * 1024, bw);
    run<4, false, 1>(X, U, T, D, out, 4, 3, 50 * 1024, bw);
    run<4, true>(X, U, T, D, out, 16, 1, 150 * 1024);
    run<4, true>(X, U, T, D, out, 4, 2, 70 * 1024);
    return 0;
}
