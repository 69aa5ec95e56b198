// Fast cepstra kernel (float32, nfft 512 or 1024): 16 lanes per frame, the FFT
// in registers.
//
// A frame's N real samples are packed as H = N/2 complex points z[n] and the
// H-point transform is split H = P x 16 (P = H/16 points per lane):
//   1. lane j of the frame's 16-lane group holds z[16 n1 + j], n1 < P -- the
//      samples 32 n1 + 2j, +1, so each of the P loads of a group reads 128
//      contiguous bytes -- and does a P-point FFT over n1 in registers;
//   2. multiplies by the twiddles W_H^(j k1);
//   3. the group transposes through LDS ([k1][j], rows padded to 17) and every
//      lane does 16-point FFTs over j for its k1 (one for P=16, two for P=32);
//   4. Z goes back to LDS in natural order; the real-input spectrum is unpacked
//      from Z[k], Z[H-k], squared and scaled; mel triangles (each a contiguous
//      run of bins), log10, DCT rows -- tables for all of it live in LDS.
// A wave carries 4 frames at once, a workgroup 16; work items are cut from the
// packed sample axis exactly as in the generic kernel (mfcc.hip).
#include "srx_internal.h"

#include <math.h>

namespace {

constexpr int kWave = 64;
constexpr int kFramesPerBlock = 16;

struct FastDev {
    int frame_len, hop, n_used, n_mel, n_ceps, nnz;
    float log_mult;
    const float* window;      // [frame_len]
    const float2* tw;         // [N] exp(-2 pi i k / N)
    const int32_t* mel_ptr;   // [n_mel+1]
    const int32_t* mel_start; // [n_mel] first bin of each filter's run
    const float* mel_w;       // [nnz]
    const float* dct;         // [n_ceps][n_mel]
};

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

// In-place radix-2 DIT on registers.  Input must already sit in bit-reversed
// order (a[brev(n)] = x[n]); tw[q] = W_Q^q for q < Q/2.
template <int Q>
__device__ __forceinline__ void fft_regs(float (&re)[Q], float (&im)[Q], const float2 (&tw)[Q / 2]) {
#pragma unroll
    for (int s = 1; s <= ilog2(Q); ++s) {
        const int m = 1 << s, half = m >> 1, tstep = Q / m;
#pragma unroll
        for (int k = 0; k < Q; k += m) {
#pragma unroll
            for (int jj = 0; jj < half; ++jj) {
                const int i0 = k + jj, i1 = i0 + half;
                float tr, ti;
                if (jj == 0) {
                    tr = re[i1];
                    ti = im[i1];
                } else if (2 * jj == half) {     // W = -i
                    tr = im[i1];
                    ti = -re[i1];
                } else {
                    const float2 w = tw[jj * tstep];
                    tr = w.x * re[i1] - w.y * im[i1];
                    ti = w.x * im[i1] + w.y * re[i1];
                }
                re[i1] = re[i0] - tr;
                im[i1] = im[i0] - ti;
                re[i0] += tr;
                im[i0] += ti;
            }
        }
    }
}

__device__ __forceinline__ int64_t raw_frames(int64_t n, int L, int hop) {
    return n < L ? 0 : (n - L) / hop + 1;
}

template <int P>
__global__ __launch_bounds__(256, 2) void cepstra_fast_kernel(
    const float* __restrict__ audio, const int64_t* __restrict__ so, int64_t U, int64_t total,
    FastDev c, float* __restrict__ cep) {
    constexpr int H = 16 * P, N = 2 * H, KB = P / 16;   // KB 16-point FFTs per lane in step 3
    constexpr int YS = P * 17;                          // float2 per frame in the transpose tile
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    // ---- shared tables
    float2* s_tw = (float2*)smem_raw;                       // [N]
    float* s_win = (float*)(s_tw + N);                      // [N], zero past the samples the FFT uses
    const int lpad = N;
    float* s_melw = s_win + lpad;                           // [nnz rounded to 4]
    const int wpad = (c.nnz + 3) & ~3;
    int* s_melptr = (int*)(s_melw + wpad);                  // [n_mel+1]
    int* s_melstart = s_melptr + ((c.n_mel + 1 + 3) & ~3);  // [n_mel]
    const int mpad = (c.n_mel + 3) & ~3;
    float* s_dct = (float*)(s_melstart + mpad);             // [n_ceps][n_mel+1]
    const int drow = c.n_mel + 1;
    float2* s_tw2 = (float2*)(s_dct + ((c.n_ceps * drow + 3) & ~3));   // [P][16]: W_H^(j k1)
    float* s_frames = (float*)(s_tw2 + P * 16);
    const int per_frame = 2 * YS + (H + 4) + ((c.n_mel + 3) & ~3);   // floats
    for (int i = threadIdx.x; i < N; i += blockDim.x) s_tw[i] = c.tw[i];
    for (int i = threadIdx.x; i < N; i += blockDim.x) s_win[i] = i < c.n_used ? c.window[i] : 0.f;
    for (int i = threadIdx.x; i < c.nnz; i += blockDim.x) s_melw[i] = c.mel_w[i];
    for (int i = threadIdx.x; i <= c.n_mel; i += blockDim.x) s_melptr[i] = c.mel_ptr[i];
    for (int i = threadIdx.x; i < c.n_mel; i += blockDim.x) s_melstart[i] = c.mel_start[i];
    for (int i = threadIdx.x; i < c.n_ceps * c.n_mel; i += blockDim.x)
        s_dct[(i / c.n_mel) * drow + i % c.n_mel] = c.dct[i];
    for (int i = threadIdx.x; i < P * 16; i += blockDim.x)
        s_tw2[i] = c.tw[2 * (((i & 15) * (i >> 4)) & (H - 1))];
    __syncthreads();

    const int wave = threadIdx.x / kWave, lane = threadIdx.x % kWave;
    const int fs = lane >> 4, j = lane & 15;          // frame slot in the wave, lane in the group
    const int item = 4 * wave + fs;                   // frame slot in the workgroup
    float* fr = s_frames + (size_t)item * per_frame;
    float2* Y = (float2*)fr;                          // transpose tile, then Z[H]
    float* pw = fr + 2 * YS;                          // [H+1]
    float* lm = pw + (H + 4);                         // [n_mel]

    // twiddles of the register FFTs: W_P^q (first pass) and W_16^q (second pass)
    float2 twp[P / 2], tw16[8];
#pragma unroll
    for (int q = 0; q < P / 2; ++q) twp[q] = s_tw[q * (N / P)];
#pragma unroll
    for (int q = 0; q < 8; ++q) tw16[q] = s_tw[q * (N / 16)];

    // A workgroup walks a CONTIGUOUS range of chunks: the utterance under its
    // position then advances monotonically (offsets cached in registers, no search
    // per chunk), and the samples of chunk b+1 are requested before chunk b is
    // transformed, so neither the offset lookups nor the sample loads stall it.
    const int64_t chunk = (int64_t)kFramesPerBlock * c.hop;
    const int64_t n_chunks = (total + chunk - 1) / chunk;
    const int64_t per_block = (n_chunks + gridDim.x - 1) / gridDim.x;
    const int64_t b_begin = (int64_t)blockIdx.x * per_block;
    const int64_t b_end = b_begin + per_block < n_chunks ? b_begin + per_block : n_chunks;
    if (b_begin >= b_end) return;
    int64_t lo = 0;
    {
        int64_t hi = U;
        const int64_t p0 = b_begin * chunk;
        while (hi - lo > 1) {
            const int64_t mid = (lo + hi) >> 1;
            if (so[mid] <= p0) lo = mid; else hi = mid;
        }
    }
    int64_t so_lo = so[lo], so_nx = so[lo + 1];
    const unsigned hop = (unsigned)c.hop;
    struct Loc {
        bool have;
        int64_t s0;
        unsigned t, row0;
    };
    // which frame does this lane group own?  (item-th frame starting in [p0, p1))
    // The packed buffer holds < 2^32 samples here, so 32-bit division is enough.
    auto locate = [&](int64_t bq) {
        const int64_t p0 = bq * chunk;
        const int64_t p1 = p0 + chunk < total ? p0 + chunk : total;
        while (lo + 1 < U && so_nx <= p0) {      // wave-uniform
            ++lo;
            so_lo = so_nx;
            so_nx = so[lo + 1];
        }
        Loc r{false, 0, 0u, 0u};
        int cnt = 0;
        int64_t us = so_lo, ue = so_nx;
        for (int64_t u = lo; u < U && us < p1 && !r.have; ++u) {
            const unsigned n = (unsigned)(ue - us);
            const unsigned Tu = n < (unsigned)c.frame_len ? 0u : (n - (unsigned)c.frame_len) / hop + 1u;
            const unsigned a = p0 > us ? (unsigned)(p0 - us) : 0u;
            const unsigned ta = (a + hop - 1u) / hop;
            unsigned tb = ((unsigned)(p1 - us) + hop - 1u) / hop;
            if (tb > Tu) tb = Tu;
            const int nn = tb > ta ? (int)(tb - ta) : 0;
            if (item < cnt + nn) {
                r.have = true;
                r.s0 = us;
                r.t = ta + (unsigned)(item - cnt);
                r.row0 = (unsigned)us / hop;
            }
            cnt += nn;
            us = ue;
            if (u + 2 <= U && us < p1 && !r.have) ue = so[u + 2];
        }
        return r;
    };
    // Loads are unconditional (a predicated load costs a branch and a full wait):
    // indices are clamped into the frame and the zero tail of the window table
    // removes what the FFT must not see.  A group without a frame reads the start
    // of the buffer and stores nothing.
    auto fetch = [&](const Loc& l, float (&xr)[2 * P]) {
        const float* x = audio + l.s0 + (int64_t)(l.t * hop);
        const int lim = l.have ? c.n_used - 1
                               : (int)(total - 1 < (int64_t)c.n_used - 1 ? total - 1 : c.n_used - 1);
#pragma unroll
        for (int n1 = 0; n1 < P; ++n1) {
            const int i0 = 32 * n1 + 2 * j;
            xr[2 * n1] = x[min(i0, lim)];
            xr[2 * n1 + 1] = x[min(i0 + 1, lim)];
        }
    };
    Loc cur = locate(b_begin);
    float xc[2 * P];
    fetch(cur, xc);
    for (int64_t b = b_begin; b < b_end; ++b) {
        Loc nxt = cur;
        float xn[2 * P];
        if (b + 1 < b_end) nxt = locate(b + 1);
        fetch(nxt, xn);                           // (the last chunk's repeat is harmless)
        const bool have = cur.have;
        const unsigned t = cur.t, row0 = cur.row0;
        // ---- 1. window, bit-reversed register order, P-point FFT over n1
        const bool any = __ballot(have) != 0;     // no frame starts in this chunk for this wave?
        float re[P], im[P];
#pragma unroll
        for (int n1 = 0; n1 < P; ++n1) {
            const int i0 = 32 * n1 + 2 * j;
            const float2 wv = *(const float2*)(s_win + i0);
            re[brev(n1, ilog2(P))] = xc[2 * n1] * wv.x;
            im[brev(n1, ilog2(P))] = xc[2 * n1 + 1] * wv.y;
        }
        if (any) {
        fft_regs<P>(re, im, twp);
        // ---- 2. twiddle W_H^(j k1) and 3. transpose through LDS: Y[k1][j]
#pragma unroll
        for (int k1 = 0; k1 < P; ++k1) {
            const float2 w = s_tw2[k1 * 16 + j];
            float2 v;
            v.x = w.x * re[k1] - w.y * im[k1];
            v.y = w.x * im[k1] + w.y * re[k1];
            Y[k1 * 17 + j] = v;
        }
        wave_sync();
        float zr[KB][16], zi[KB][16];
#pragma unroll
        for (int i = 0; i < KB; ++i) {
            const int k1 = j + 16 * i;
#pragma unroll
            for (int n2 = 0; n2 < 16; ++n2) {
                const float2 v = Y[k1 * 17 + n2];
                zr[i][brev(n2, 4)] = v.x;
                zi[i][brev(n2, 4)] = v.y;
            }
        }
        wave_sync();
#pragma unroll
        for (int i = 0; i < KB; ++i) {
            fft_regs<16>(zr[i], zi[i], tw16);
            // Z[k1 + P k2] in natural order over the same LDS region
#pragma unroll
            for (int k2 = 0; k2 < 16; ++k2) Y[j + 16 * i + P * k2] = make_float2(zr[i][k2], zi[i][k2]);
        }
        wave_sync();
        // ---- 4. real-input spectrum, power / N.  Bins k and H-k share their inputs.
        const float inv_n = 1.0f / (float)N;
#pragma unroll
        for (int mm = 0; mm < P / 2; ++mm) {
            const int k = j + 16 * mm;            // 0 .. H/2-1 ; partner H-k
            const float2 zk = Y[k], zc = Y[(H - k) & (H - 1)];
            const float er = 0.5f * (zk.x + zc.x), ei = 0.5f * (zk.y - zc.y);
            const float orr = 0.5f * (zk.y + zc.y), oi = -0.5f * (zk.x - zc.x);
            const float2 w = s_tw[k];
            // X[k] = E + W^k O ;  X[H-k] = conj(E - W^k O)
            const float pr = w.x * orr - w.y * oi, pi_ = w.x * oi + w.y * orr;
            const float xr = er + pr, xi = ei + pi_;
            const float yr = er - pr, yi = ei - pi_;
            pw[k] = (xr * xr + xi * xi) * inv_n;
            pw[H - k] = (yr * yr + yi * yi) * inv_n;
        }
        if (j == 0) {   // k = H/2 is its own partner
            const float2 z = Y[H / 2];
            // E = (zr, 0), O = (zi, 0) for the self-paired bin; W^(H/2) = -i
            pw[H / 2] = (z.x * z.x + z.y * z.y) * inv_n;
        }
        wave_sync();
        // ---- mel triangles (contiguous runs), exact zeros -> eps, log_mult * log10
        for (int m = j; m < c.n_mel; m += 16) {
            const int q0 = s_melptr[m], q1 = s_melptr[m + 1], b0 = s_melstart[m];
            float e = 0.f;
            const float* pb = pw + b0 - q0;
#pragma unroll 8
            for (int q = q0; q < q1; ++q) e += pb[q] * s_melw[q];
            if (e == 0.f) e = 2.220446049250313e-16f;
            lm[m] = c.log_mult * __log10f(e);
        }
        wave_sync();
        // ---- DCT-II rows
        if (have) {
            float* dst = cep + (int64_t)(row0 + t) * c.n_ceps;
            for (int k = j; k < c.n_ceps; k += 16) {
                float acc = 0.f;
                const float* row = s_dct + k * drow;
#pragma unroll 8
                for (int m = 0; m < c.n_mel; ++m) acc += row[m] * lm[m];
                dst[k] = acc;
            }
        }
        wave_sync();
        }   // any
        // hand the prefetched chunk over only now: touching xn earlier would wait
        // for its loads before the transform instead of after it
        cur = nxt;
#pragma unroll
        for (int i = 0; i < 2 * P; ++i) xc[i] = xn[i];
    }
}

template <int P>
int launch_fast(srx_context* h, const float* audio, const int64_t* so, int64_t U, int64_t total,
                const FastDev& d, float* cep, hipStream_t st) {
    constexpr int H = 16 * P, N = 2 * H, YS = P * 17;
    const int lpad = N, wpad = (d.nnz + 3) & ~3;
    const int mpad = (d.n_mel + 3) & ~3;
    const size_t tables = sizeof(float2) * N + sizeof(float2) * P * 16 +
                          4 * ((size_t)lpad + wpad + ((d.n_mel + 1 + 3) & ~3) + mpad +
                               ((d.n_ceps * (d.n_mel + 1) + 3) & ~3));
    const size_t per_frame = 4 * ((size_t)2 * YS + (H + 4) + mpad);
    const size_t lds = tables + per_frame * kFramesPerBlock;
    if (lds > 160 * 1024) return SRX_EUNSUPPORTED;
    auto kern = cepstra_fast_kernel<P>;
    SRX_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int64_t chunk = (int64_t)kFramesPerBlock * d.hop;
    const int64_t n_chunks = (total + chunk - 1) / chunk;
    int64_t grid = n_chunks < (int64_t)h->n_cu * 4 ? n_chunks : (int64_t)h->n_cu * 4;
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), lds, st, audio, so, U, total, d, cep);
    SRX_HIP(hipGetLastError());
    return SRX_OK;
}

}  // namespace

int srx_mfcc_fast_cepstra(srx_context* h, srx_mfcc_plan* p, const float* audio, const int64_t* so,
                          int64_t U, int64_t total, float* cep, hipStream_t st) {
    const srx_mfcc_cfg_t& c = p->cfg;
    if (!p->fast_ok || (c.nfft != 512 && c.nfft != 1024) || total >= 0xFFFFFFFFll) return SRX_EUNSUPPORTED;
    FastDev d;
    d.frame_len = c.frame_len; d.hop = c.hop;
    d.n_used = c.frame_len < c.nfft ? c.frame_len : c.nfft;
    d.n_mel = c.n_mel; d.n_ceps = c.n_ceps; d.nnz = p->nnz;
    d.log_mult = (float)c.log_mult;
    d.window = (const float*)p->window[0];
    d.tw = (const float2*)p->tw_full;
    d.mel_ptr = p->mel_ptr; d.mel_start = p->mel_start;
    d.mel_w = (const float*)p->mel_w[0];
    d.dct = (const float*)p->dct[0];
    return c.nfft == 512 ? launch_fast<16>(h, audio, so, U, total, d, cep, st)
                         : launch_fast<32>(h, audio, so, U, total, d, cep, st);
}
