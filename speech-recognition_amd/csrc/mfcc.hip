// MFCC front-end: frames -> Hamming -> real FFT -> |.|^2/N -> mel triangles ->
// log10 -> DCT-II -> regression deltas -> per-utterance z-score.
//
// Two kernels over ragged, packed utterances:
//   cepstra_kernel  one wave per frame.  The frame is read once from HBM
//                   (coalesced), windowed, packed as N/2 complex points and
//                   transformed in LDS (radix-2, bit-reversed load), unpacked to
//                   the N/2+1 real-input bins, reduced through the sparse mel
//                   triangles (every bin feeds at most two filters) and the
//                   DCT rows kept in L1/L2.  Work items are cut from the packed
//                   SAMPLE axis, so a few long recordings and many short ones
//                   load the chip equally.
//   finish_kernel   one workgroup per utterance: deltas with edge clamping,
//                   trim of delta_n frames at both ends, z-score (two-pass).
#include "srx_internal.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <type_traits>
#include <vector>

namespace {

constexpr int kWave = 64;
constexpr int kChunkFrames = 16;  // frames per work item of cepstra_kernel

template <typename T> struct cplx { T re, im; };

// Lanes of one wave exchange data through LDS: the hardware executes a wave's
// LDS instructions in order, so only the compiler has to be kept from moving
// accesses across the hand-over points.
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <typename T> __device__ __forceinline__ T eps_of();
template <> __device__ __forceinline__ float eps_of<float>() { return 1.1920929e-07f; }
template <> __device__ __forceinline__ double eps_of<double>() { return 2.220446049250313e-16; }

struct MfccDev {
    int frame_len, hop, nfft, log2h, n_mel, n_ceps;   // log2h = log2(nfft/2)
    int n_used;                                        // min(frame_len, nfft)
    int center;                                        // 0 / 1 (zeros) / 2 (reflection)
    const int32_t* mel_ptr;
    const int32_t* mel_bin;
    double log_mult;
    double power_scale;                                // 1/nfft or 1
    double log_floor;                                  // 0: exact zeros -> eps
    void* logmel;                                      // [rows, n_mel] instead of the DCT (top_db)
};

// Frames of an utterance of n samples before the delta trim.  Centred frames
// (one per hop, the first on sample 0) need no whole frame inside the signal.
__device__ __forceinline__ int64_t raw_frames(const MfccDev& c, int64_t n) {
    if (c.center) return n > 0 ? n / c.hop + 1 : 0;
    return n < c.frame_len ? 0 : (n - c.frame_len) / c.hop + 1;
}

// Work items and cepstra rows are cut from the packed sample axis.  With centred
// frames an utterance has a frame ON its end, so it is given one hop of room:
// utterance u starts at virtual position so[u] + u * hop.
__device__ __forceinline__ int64_t vpos(const MfccDev& c, const int64_t* so, int64_t u) {
    return so[u] + (c.center ? u * (int64_t)c.hop : 0);
}

// DCT rows of one frame by a whole wave: coefficient k is shared by `parts` adjacent
// lanes (64 / the next power of two >= n_ceps), each summing every parts-th filter;
// the partial sums meet by xor shuffles.  lm: the frame's log-mel values in LDS.
template <typename T>
__device__ __forceinline__ void dct_rows(const T* __restrict__ dct, const T* lm, int n_mel,
                                         int n_ceps, int lane, T* __restrict__ dst) {
    int kp = 1;
    while (kp < n_ceps) kp <<= 1;               // n_ceps <= 32
    const int parts = kWave / kp;
    const int k = lane / parts, part = lane % parts;
    T acc = 0;
    if (k < n_ceps) {
        const T* row = dct + (size_t)k * n_mel;
#pragma unroll 4
        for (int m = part; m < n_mel; m += parts) acc += row[m] * lm[m];
    }
    for (int o = parts >> 1; o > 0; o >>= 1) acc += __shfl_xor(acc, o, kWave);
    if (k < n_ceps && part == 0) dst[k] = acc;
}

template <typename T>
__global__ void cepstra_kernel(const T* __restrict__ audio, const int64_t* __restrict__ so,
                               int64_t U, int64_t total, MfccDev c, const T* __restrict__ window,
                               const cplx<T>* __restrict__ tw, const T* __restrict__ mel_w,
                               const T* __restrict__ dct, T* __restrict__ cep) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int H = c.nfft >> 1;
    const int wave = threadIdx.x / kWave, lane = threadIdx.x % kWave;
    const int n_waves = blockDim.x / kWave;
    // per-wave LDS: H complex, n_mel log energies.  The power bins take the place of the
    // transform: bin k < H in the real slot of point k, bin H in the imaginary slot of point 0.
    const size_t per_wave = (sizeof(T) * ((size_t)2 * H + c.n_mel + 1) + 15) & ~(size_t)15;
    cplx<T>* z = (cplx<T>*)(smem_raw + per_wave * wave);
    T* lm = (T*)(z + H);

    const int64_t chunk = (int64_t)kChunkFrames * c.hop;
    const int64_t n_chunks = (total + chunk - 1) / chunk;
    // (total = the virtual length, see vpos)
    for (int64_t b = blockIdx.x; b < n_chunks; b += gridDim.x) {
        const int64_t p0 = b * chunk;
        const int64_t p1 = p0 + chunk < total ? p0 + chunk : total;
        // last utterance whose first sample is <= p0
        int64_t lo = 0, hi = U;
        while (hi - lo > 1) {
            const int64_t mid = (lo + hi) >> 1;
            if (vpos(c, so, mid) <= p0) lo = mid; else hi = mid;
        }
        int item = 0;
        for (int64_t u = lo; u < U && vpos(c, so, u) < p1; ++u) {
            const int64_t s0 = so[u], n = so[u + 1] - s0;
            const int64_t v0 = vpos(c, so, u);
            const int64_t Tu = raw_frames(c, n);
            const int64_t a = p0 > v0 ? p0 - v0 : 0;
            int64_t ta = (a + c.hop - 1) / c.hop;
            int64_t tb = (p1 - v0 + c.hop - 1) / c.hop;
            if (tb > Tu) tb = Tu;
            const int64_t row0 = v0 / c.hop;   // first cepstra row of this utterance
            for (int64_t t = ta; t < tb; ++t, ++item) {
                if (item % n_waves != wave) continue;
                // (a centred frame that lies inside its utterance needs no padding: it is
                // read like an uncentred frame that starts half a transform earlier)
                const bool padded = c.center && (t * c.hop < H || t * c.hop - H + c.nfft > n);
                const T* x = audio + s0 + t * c.hop - (c.center ? H : 0);
                // ---- load, window, pack even/odd samples, bit-reversed store
                // (the loops of this kernel are unrolled by 4 so that the global and LDS
                // loads of four iterations are in flight together: one at a time, every
                // iteration pays a full cache round trip)
#pragma unroll 4
                for (int k = lane; k < H; k += kWave) {
                    const int n0 = 2 * k, n1 = 2 * k + 1;
                    cplx<T> v;
                    if (padded) {
                        // sample i of the frame is signal sample t*hop - nfft/2 + i
                        auto at = [&](int i) -> T {
                            int64_t j = t * c.hop - H + i;
                            if (c.center == 2) {
                                if (j < 0) j = -j;
                                if (j >= n) j = 2 * (n - 1) - j;
                                j = j < 0 ? 0 : (j >= n ? n - 1 : j);
                            } else if (j < 0 || j >= n) {
                                return (T)0;
                            }
                            return audio[s0 + j] * window[i];
                        };
                        v.re = at(n0);
                        v.im = at(n1);
                    } else {
                        v.re = n0 < c.n_used ? x[n0] * window[n0] : (T)0;
                        v.im = n1 < c.n_used ? x[n1] * window[n1] : (T)0;
                    }
                    const int r = (int)(__brev((unsigned)k) >> (32 - c.log2h));
                    z[r] = v;
                }
                wave_sync();
                // ---- DIT over H points on the bit-reversed tile; tw[k] = exp(-2 pi i k / nfft).
                // Two radix-2 stages are taken together as one radix-4 pass (half the LDS
                // round trips and wave syncs): in the second of the two, the lower pair's
                // twiddle is the upper pair's times W^(nfft/4) = -i, so a pass loads two
                // twiddles for four points.  An odd stage count starts with one radix-2 stage.
                int s0st = 0;
                if (c.log2h & 1) {
#pragma unroll 4
                    for (int q = lane; q < (H >> 1); q += kWave) {
                        const int i = q << 1, j = i + 1;          // (twiddle 1)
                        const cplx<T> zi = z[i], zj = z[j];
                        z[i].re = zi.re + zj.re; z[i].im = zi.im + zj.im;
                        z[j].re = zi.re - zj.re; z[j].im = zi.im - zj.im;
                    }
                    wave_sync();
                    s0st = 1;
                }
                if (s0st == 0 && c.log2h >= 2) {
                    // the first radix-4 pass has twiddles 1 and -i only: no table, no products
#pragma unroll 4
                    for (int q = lane; q < (H >> 2); q += kWave) {
                        const int i0 = q << 2;
                        const cplx<T> x0 = z[i0], x1 = z[i0 + 1], x2 = z[i0 + 2], x3 = z[i0 + 3];
                        const T a0r = x0.re + x1.re, a0i = x0.im + x1.im, a1r = x0.re - x1.re, a1i = x0.im - x1.im;
                        const T a2r = x2.re + x3.re, a2i = x2.im + x3.im, a3r = x2.re - x3.re, a3i = x2.im - x3.im;
                        z[i0].re = a0r + a2r;     z[i0].im = a0i + a2i;
                        z[i0 + 1].re = a1r + a3i; z[i0 + 1].im = a1i - a3r;
                        z[i0 + 2].re = a0r - a2r; z[i0 + 2].im = a0i - a2i;
                        z[i0 + 3].re = a1r - a3i; z[i0 + 3].im = a1i + a3r;
                    }
                    wave_sync();
                    s0st = 2;
                }
                for (int s = s0st; s < c.log2h; s += 2) {
                    const int half = 1 << s;
                    const int ts1 = c.nfft >> (s + 1), ts2 = c.nfft >> (s + 2);
#pragma unroll 4
                    for (int q = lane; q < (H >> 2); q += kWave) {
                        const int pos = q & (half - 1);
                        const int i0 = ((q >> s) << (s + 2)) + pos;
                        const cplx<T> w1 = tw[pos * ts1], w2 = tw[pos * ts2];
                        const cplx<T> x0 = z[i0], x1 = z[i0 + half], x2 = z[i0 + 2 * half],
                                      x3 = z[i0 + 3 * half];
                        // first stage of the pass: (x0, x1) and (x2, x3) with w1
                        const T t1r = w1.re * x1.re - w1.im * x1.im, t1i = w1.re * x1.im + w1.im * x1.re;
                        const T t3r = w1.re * x3.re - w1.im * x3.im, t3i = w1.re * x3.im + w1.im * x3.re;
                        const T a0r = x0.re + t1r, a0i = x0.im + t1i, a1r = x0.re - t1r, a1i = x0.im - t1i;
                        const T a2r = x2.re + t3r, a2i = x2.im + t3i, a3r = x2.re - t3r, a3i = x2.im - t3i;
                        // second stage: (a0, a2) with w2, (a1, a3) with -i w2
                        const T u2r = w2.re * a2r - w2.im * a2i, u2i = w2.re * a2i + w2.im * a2r;
                        const T vr = w2.re * a3r - w2.im * a3i, vi = w2.re * a3i + w2.im * a3r;
                        const T u3r = vi, u3i = -vr;
                        z[i0].re = a0r + u2r;            z[i0].im = a0i + u2i;
                        z[i0 + half].re = a1r + u3r;     z[i0 + half].im = a1i + u3i;
                        z[i0 + 2 * half].re = a0r - u2r; z[i0 + 2 * half].im = a0i - u2i;
                        z[i0 + 3 * half].re = a1r - u3r; z[i0 + 3 * half].im = a1i - u3i;
                    }
                    wave_sync();
                }
                // ---- unpack to the real-input spectrum, power / nfft.  Bins k and H - k come
                // from the same two points, so a lane takes the pair and leaves the two powers
                // where it read the points: no other lane touches them.  Four pairs at a time,
                // all loads before the first store, so that the loads are in flight together.
                const T inv_n = (T)c.power_scale;
                auto power = [&](const cplx<T>& zk, const cplx<T>& zc, const cplx<T>& w) -> T {
                    const T er = (T)0.5 * (zk.re + zc.re), ei = (T)0.5 * (zk.im - zc.im);
                    const T orr = (T)0.5 * (zk.im + zc.im), oi = (T)-0.5 * (zk.re - zc.re);
                    const T xr = er + w.re * orr - w.im * oi;
                    const T xi = ei + w.re * oi + w.im * orr;
                    return (xr * xr + xi * xi) * inv_n;
                };
                for (int k0 = 0; k0 <= (H >> 1); k0 += 4 * kWave) {
                    cplx<T> zk[4], zc[4], wk[4], wc[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        int k = k0 + j * kWave + lane;
                        k = k <= (H >> 1) ? k : (H >> 1);
                        zk[j] = z[k];
                        zc[j] = z[(H - k) & (H - 1)];
                        wk[j] = tw[k];
                        wc[j] = tw[H - k];
                    }
                    T pa[4], pb[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        pa[j] = power(zk[j], zc[j], wk[j]);
                        pb[j] = power(zc[j], zk[j], wc[j]);
                    }
                    wave_sync();                // (clamped lanes re-read the middle pair)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int k = k0 + j * kWave + lane;
                        if (k > (H >> 1)) continue;
                        z[k].re = pa[j];
                        if (k == 0) z[0].im = pb[j];
                        else z[H - k].re = pb[j];
                    }
                }
                wave_sync();
                // ---- mel triangles (CSR), exact zeros -> eps, log_mult * log10
                // (filters widen with their index: odd rounds hand them out backwards, so
                // a lane that took a narrow filter takes a wide one next)
                for (int m0 = 0, rnd = 0; m0 < c.n_mel; m0 += kWave, ++rnd) {
                    const int cnt = c.n_mel - m0 < kWave ? c.n_mel - m0 : kWave;
                    if (lane >= cnt) continue;
                    const int m = m0 + ((rnd & 1) ? cnt - 1 - lane : lane);
                    T e = 0;
#pragma unroll 4
                    for (int q = c.mel_ptr[m]; q < c.mel_ptr[m + 1]; ++q) {
                        const int bin = c.mel_bin[q];
                        e += ((const T*)z)[bin < H ? 2 * bin : 1] * mel_w[q];
                    }
                    if (c.log_floor > 0.0) e = e < (T)c.log_floor ? (T)c.log_floor : e;
                    else if (e == (T)0) e = (T)2.220446049250313e-16;   // np.finfo(float).eps, mfcc.py:43
                    lm[m] = (T)c.log_mult * log10(e);
                    if (c.logmel) ((T*)c.logmel)[(row0 + t) * c.n_mel + m] = lm[m];
                }
                wave_sync();
                if (c.logmel) continue;         // the DCT follows the utterance-wide clamp
                // ---- DCT-II rows
                dct_rows<T>(dct, lm, c.n_mel, c.n_ceps, lane, cep + (row0 + t) * c.n_ceps);
                wave_sync();
            }
        }
    }
}

__device__ __forceinline__ int64_t clamp_t(int64_t t, int64_t Tu) {
    return t < 0 ? 0 : (t >= Tu ? Tu - 1 : t);
}

// Feature block `ord` (0 = cepstrum, 1 = delta, 2 = delta of the untrimmed
// delta) at raw frame t, coefficient k; edge frames are repeated (np.pad
// mode='edge', mfcc.py:93).
template <typename T>
__device__ __forceinline__ T delta_at(const T* __restrict__ cep, int K, int64_t Tu, int64_t t, int k,
                                      int ord, int N, T inv_den) {
    if (ord == 0) return cep[t * K + k];
    if (ord == 1) {
        T acc = 0;
        for (int j = -N; j <= N; ++j) acc += (T)j * cep[clamp_t(t + j, Tu) * K + k];
        return acc * inv_den;
    }
    T acc2 = 0;
    for (int i = -N; i <= N; ++i) {
        const int64_t ti = clamp_t(t + i, Tu);
        T acc = 0;
        for (int j = -N; j <= N; ++j) acc += (T)j * cep[clamp_t(ti + j, Tu) * K + k];
        acc2 += (T)i * (acc * inv_den);
    }
    return acc2 * inv_den;
}

// One workgroup per utterance.  Utterances whose cepstra and delta blocks fit
// the LDS budget are finished there (one global read of the cepstra, one global
// write of the features); longer ones go through global memory.
template <typename T>
__global__ void finish_kernel(const T* __restrict__ cep, const int64_t* __restrict__ so,
                              const int64_t* __restrict__ fo, int64_t U, MfccDev c, int order,
                              int N, int zscore, int lds_elems, T* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char fin_raw[];
    T* blk = (T*)fin_raw;                       // [1+order][Tu][K]
    __shared__ double red[256];
    __shared__ T col_mean[128], col_scale[128];
    const int K = c.n_ceps, NF = K * (1 + order);
    int den = 0;
    for (int i = 1; i <= N; ++i) den += 2 * i * i;
    const T inv_den = (T)1 / (T)den;
    for (int64_t u = blockIdx.x; u < U; u += gridDim.x) {
        const int64_t s0 = so[u], n = so[u + 1] - s0;
        const int64_t Tu = raw_frames(c, n);
        const int trim = order > 0 ? N : 0;
        const int64_t To = Tu - 2 * trim > 0 ? Tu - 2 * trim : 0;
        const T* cu = cep + (vpos(c, so, u) / c.hop) * K;
        T* ou = out + fo[u] * NF;
        if (To == 0) continue;
        const bool in_lds = Tu * K * (1 + order) <= lds_elems;
        const int TK = (int)(Tu * K);
        if (in_lds) {
            for (int i = threadIdx.x; i < TK; i += blockDim.x) blk[i] = cu[i];
            // thread -> (coefficient k, row group r): no division in the loops
            const int gk = blockDim.x / K, k = threadIdx.x % K, r = threadIdx.x / K;
            const int Ti = (int)Tu;
            for (int o = 1; o <= order; ++o) {
                __syncthreads();
                const T* src = blk + (size_t)(o - 1) * TK + k;
                T* dst = blk + (size_t)o * TK + k;
                if (r < gk)
                    for (int t = r; t < Ti; t += gk) {
                        T acc = 0;
                        for (int j = -N; j <= N; ++j) {
                            int tt = t + j;
                            tt = tt < 0 ? 0 : (tt >= Ti ? Ti - 1 : tt);
                            acc += (T)j * src[tt * K];
                        }
                        dst[t * K] = acc * inv_den;
                    }
            }
            __syncthreads();
        } else {
            for (int64_t idx = threadIdx.x; idx < To * NF; idx += blockDim.x) {
                const int64_t t = idx / NF;
                const int f = (int)(idx % NF);
                ou[idx] = delta_at<T>(cu, K, Tu, t + trim, f % K, f / K, N, inv_den);
            }
            __syncthreads();
        }
        // feature (t, f) of the kept frames
        auto feat = [&](int64_t t, int f) -> T {
            return in_lds ? blk[(size_t)(f / K) * TK + (t + trim) * K + f % K] : ou[t * NF + f];
        };
        if (zscore) {
            // column sums: thread -> (column f, row group r)
            const int groups = blockDim.x / NF;
            const int f = threadIdx.x % NF, r = threadIdx.x / NF;
            const bool active = r < groups;
            for (int pass = 0; pass < 2; ++pass) {
                double acc = 0.0;
                if (active)
                    for (int64_t t = r; t < To; t += groups) {
                        const double v = (double)feat(t, f);
                        if (pass == 0) acc += v;
                        else { const double d = v - (double)col_mean[f]; acc += d * d; }
                    }
                red[threadIdx.x] = acc;
                __syncthreads();
                if (threadIdx.x < NF) {
                    double s = 0.0;
                    for (int g = 0; g < groups; ++g) s += red[g * NF + threadIdx.x];
                    if (pass == 0) {
                        col_mean[threadIdx.x] = (T)(s / (double)To);
                    } else {
                        const T var = (T)(s / (double)To);
                        const T mean = col_mean[threadIdx.x];
                        const T e = eps_of<T>();
                        const T bound = (T)To * e * var + ((T)To * mean * e) * ((T)To * mean * e);
                        T sc = sqrt(var);
                        if (var <= bound || sc == (T)0) sc = (T)1;   // constant column keeps scale 1
                        col_scale[threadIdx.x] = sc;
                    }
                }
                __syncthreads();
            }
        } else if (threadIdx.x < NF) {
            col_mean[threadIdx.x] = (T)0;
            col_scale[threadIdx.x] = (T)1;
        }
        __syncthreads();
        if (in_lds) {
            const int groups = blockDim.x / NF;
            const int f = threadIdx.x % NF, r = threadIdx.x / NF;
            if (r < groups) {
                const T* src = blk + (size_t)(f / K) * TK + (size_t)trim * K + f % K;
                const T mean = col_mean[f], inv = (T)1 / col_scale[f];
                const int Tn = (int)To;
                for (int t = r; t < Tn; t += groups) ou[(int64_t)t * NF + f] = (src[t * K] - mean) * inv;
            }
        } else if (zscore) {
            for (int64_t idx = threadIdx.x; idx < To * NF; idx += blockDim.x) {
                const int ff = (int)(idx % NF);
                ou[idx] = (ou[idx] - col_mean[ff]) / col_scale[ff];
            }
        }
        __syncthreads();
    }
}

// The same stage for delta_n = N known at compile time, without the delta blocks
// in LDS: a workgroup stages a tile of the utterance's cepstra (plus 2N rows either
// side), thread (coefficient k, row segment) walks down its rows with the 2N+1
// first-order deltas around the current row in registers -- one new delta (2N LDS
// reads) per row -- and emits the row's c, delta and delta-delta.  The walk is
// repeated for the column means, the column variances and the write; only the
// tile is re-read from global memory, and only when the utterance has several.
// Summation orders are those of finish_kernel.
// KC: the number of cepstra when it is known at compile time (13, the reference's), else 0:
// row strides and the thread-to-coefficient map are then immediates instead of scalar
// registers and divisions.
// Workgroups of finish_stream_kernel a CU holds (158 registers with the next utterance's
// cepstra in them; at four, 128 registers, the kernel spills)
#define FIN_WG 3
// Workgroup barrier that orders LDS accesses only: unlike __syncthreads it does not wait for
// the global loads in flight (the next utterance's cepstra).
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
template <typename T, int N, int KC>
__global__ __launch_bounds__(256, FIN_WG) void finish_stream_kernel(
    const T* __restrict__ cep, const int64_t* __restrict__ so, const int64_t* __restrict__ fo,
    int64_t U, MfccDev c, int order, int zscore, int tile_rows, T* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char fin_raw[];
    T* tile = (T*)fin_raw;                      // [tile_rows + 4N][K]
    __shared__ double red[3 * 256];
    __shared__ T col_mean[128], col_scale[128];
    const int K = KC ? KC : c.n_ceps, NF = K * (1 + order);
    const int nseg = blockDim.x / K, k = threadIdx.x % K, seg = threadIdx.x / K;
    int den = 0;
    for (int i = 1; i <= N; ++i) den += 2 * i * i;
    const T inv_den = (T)1 / (T)den;
    const int trim = order > 0 ? N : 0;
    constexpr int RMAX = 16;                    // rows a thread keeps in registers (short utterances)
    // The cepstra of the workgroup's NEXT utterance, requested while the current one is worked
    // on (float32 utterances short enough for the one-walk path; 16 KB of tile = 16 values per
    // thread).  Every load is issued whatever the utterance is (an index outside it reads the
    // array's first element): a load under a branch would be waited for where the branch ends.
    // An utterance's shape (frames, first cepstra row, first output row) comes from the offset
    // arrays two utterances ahead, so that neither the request above nor the walk waits for it.
    constexpr int PRE = sizeof(T) == 4 ? 16 : 1;
    T pre[PRE], pre_first = (T)0, pre_last = (T)0;
    struct Shape {
        int64_t Tu, To, row0, out0;
        bool quick;                             // float32, one tile, at most RMAX rows per thread
    };
    auto shape = [&](int64_t u) {
        const int64_t uu = u < U ? u : U - 1;
        Shape sh;
        sh.Tu = u < U ? raw_frames(c, so[uu + 1] - so[uu]) : 0;
        sh.To = sh.Tu - 2 * trim > 0 ? sh.Tu - 2 * trim : 0;
        sh.row0 = vpos(c, so, uu) / c.hop;
        sh.out0 = fo[uu];
        sh.quick = sizeof(T) == 4 && sh.To > 0 && sh.To <= tile_rows && (sh.To + nseg - 1) / nseg <= RMAX;
        return sh;
    };
    auto prefetch = [&](const Shape& sh) {
        if constexpr (sizeof(T) == 4) {
            const bool ok = sh.quick;
            const T* src = cep + (ok ? sh.row0 * K : 0);
            const int cnt = ok ? (int)sh.Tu * K : 0;
#pragma unroll
            for (int q = 0; q < PRE; ++q) {
                const int i = threadIdx.x + q * 256;
                pre[q] = src[i < cnt ? i : 0];
            }
            const int e = (int)threadIdx.x < K ? (int)threadIdx.x : 0;
            pre_first = src[cnt ? e : 0];
            pre_last = src[cnt ? cnt - K + e : 0];
        }
    };
    Shape sh_cur = shape(blockIdx.x), sh_nxt = shape((int64_t)blockIdx.x + gridDim.x);
    prefetch(sh_cur);
    for (int64_t u = blockIdx.x; u < U; u += gridDim.x) {
        const Shape sh = sh_cur, sh_ahead = sh_nxt;
        sh_cur = sh_nxt;
        sh_nxt = shape(u + 2 * (int64_t)gridDim.x);
        const int64_t Tu = sh.Tu, To = sh.To;
        if (To == 0) {
            prefetch(sh_ahead);
            continue;
        }
        const T* cu = cep + sh.row0 * K;
        T* ou = out + sh.out0 * NF;
        const int64_t n_tiles = (To + tile_rows - 1) / tile_rows;
        // column sums of a phase -> col_mean (phase 0) or col_scale (phase 1)
        auto reduce = [&](int phase, const double (&acc)[3]) {
#pragma unroll
            for (int o = 0; o < 3; ++o) red[o * 256 + threadIdx.x] = seg < nseg ? acc[o] : 0.0;
            lds_barrier();
            if (threadIdx.x < NF) {
                const int o = threadIdx.x / K, kk = threadIdx.x % K;
                double s = 0.0;
                for (int g = 0; g < nseg; ++g) s += red[o * 256 + g * K + kk];
                if (phase == 0) {
                    col_mean[threadIdx.x] = (T)(s / (double)To);
                } else {
                    const T var = (T)(s / (double)To);
                    const T mu = col_mean[threadIdx.x];
                    const T e = eps_of<T>();
                    const T bound = (T)To * e * var + ((T)To * mu * e) * ((T)To * mu * e);
                    T sc = sqrt(var);
                    if (var <= bound || sc == (T)0) sc = (T)1;   // constant column keeps scale 1
                    col_scale[threadIdx.x] = sc;
                }
            }
            lds_barrier();
        };
        // rows [lo, hi) of the utterance's cepstra -> tile, from tile row `pad` on; with pad > 0
        // the pad rows before them are copies of row lo and the pad rows behind them copies of
        // row hi - 1 (the edge rule of the deltas, laid out so that no index needs a clamp)
        auto load_tile = [&](int64_t lo, int64_t hi, int pad = 0) {
            __syncthreads();                    // walkers of the previous tile
            const int cnt = (int)(hi - lo) * K;
            const T* src = cu + lo * K;
            T* tile = (T*)fin_raw + pad * K;
            // (the edge rows are requested first and stored last: one round trip with the rest)
            const bool edge = pad > 0 && (int)threadIdx.x < K;
            T e_first = (T)0, e_last = (T)0;
            if (edge) {
                e_first = src[threadIdx.x];
                e_last = src[cnt - K + threadIdx.x];
            }
            // (8 loads in flight per thread: one at a time, the copy would take a
            // memory round trip per 256 elements)
            for (int i0 = threadIdx.x; i0 < cnt; i0 += 8 * blockDim.x) {
                T v[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int i = i0 + q * blockDim.x;
                    v[q] = i < cnt ? src[i] : (T)0;
                }
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int i = i0 + q * blockDim.x;
                    if (i < cnt) tile[i] = v[q];
                }
            }
            if (edge)
                for (int p_ = 1; p_ <= pad; ++p_) {
                    tile[-p_ * K + (int)threadIdx.x] = e_first;
                    tile[cnt + (p_ - 1) * K + (int)threadIdx.x] = e_last;
                }
            __syncthreads();
        };
        // One walk of rows [a, b) (utterance rows; the tile starts at row lo): emit(q, t, v)
        // gets the q-th row's c, delta, delta-delta.  UNROLL > 0 unrolls the walk that many
        // times under a row-count predicate, so that emit may index registers by q.
        auto walk = [&](auto unroll, int64_t lo, int64_t a, int64_t b, auto&& emit) {
            constexpr int UN = decltype(unroll)::value;
            // (32-bit row numbers relative to the tile: a clamp is one instruction)
            const T* col = tile + k;
            const int r_lo = (int)(0 - lo), r_hi = (int)(Tu - 1 - lo);   // rows 0 and Tu-1
            auto d1 = [&](int s) {
                s = min(max(s, r_lo), r_hi);
                T v = 0;
#pragma unroll
                for (int j = -N; j <= N; ++j) {
                    if (j == 0) continue;
                    const int tt = min(max(s + j, r_lo), r_hi);
                    v += (T)j * col[tt * K];
                }
                return v * inv_den;
            };
            const int ra = (int)(a - lo), rb = (int)(b - lo);
            T w[2 * N + 1];
#pragma unroll
            for (int i = 0; i <= 2 * N; ++i) w[i] = (T)0;
            if (order >= 2) {
#pragma unroll
                for (int i = 0; i < 2 * N; ++i) w[i + 1] = d1(ra - N + i);
            } else if (order == 1) {
                w[N + 1] = d1(ra);              // (becomes the centre after the first shift)
            }
            auto row = [&](int q, int t) {
                T v[3];
                v[0] = col[t * K];
#pragma unroll
                for (int i = 0; i < 2 * N; ++i) w[i] = w[i + 1];
                if (order >= 2) {
                    w[2 * N] = d1(t + N);
                    T a2 = 0;
#pragma unroll
                    for (int i = -N; i <= N; ++i) a2 += (T)i * w[N + i];
                    v[2] = a2 * inv_den;
                } else {
                    v[2] = (T)0;
                    if (order == 1) w[N + 1] = d1(t + 1);
                }
                v[1] = w[N];
                emit(q, t, v);
            };
            if constexpr (UN > 0) {
#pragma unroll
                for (int q = 0; q < UN; ++q)
                    if (ra + q < rb) row(q, ra + q);
            } else {
                for (int t = ra; t < rb; ++t) row(0, t);
            }
        };
        // The walk over a tile whose pad rows hold the edge copies (whole utterance in the tile,
        // tile row = utterance row + N): rows [ra, ra + UN) of the tile below rb, no clamps and
        // no index products -- a pointer that moves down a row per step.  Same sums, in the
        // same order, as walk.
        auto walk_padded = [&](auto unroll, int ra, int rb, auto&& emit) {
            constexpr int UN = decltype(unroll)::value;
            auto d1 = [&](const T* p_) {
                T v = 0;
#pragma unroll
                for (int j = -N; j <= N; ++j) {
                    if (j == 0) continue;
                    v += (T)j * p_[j * K];
                }
                return v * inv_den;
            };
            T w[2 * N + 1];
#pragma unroll
            for (int i = 0; i <= 2 * N; ++i) w[i] = (T)0;
            const T* pc = tile + k + ra * K;
            if (order >= 2) {
#pragma unroll
                for (int i = 0; i < 2 * N; ++i) w[i + 1] = d1(pc + (i - N) * K);
            } else if (order == 1) {
                w[N + 1] = d1(pc);
            }
#pragma unroll
            for (int q = 0; q < UN; ++q)
                if (ra + q < rb) {
                    T v[3];
                    v[0] = *pc;
#pragma unroll
                    for (int i = 0; i < 2 * N; ++i) w[i] = w[i + 1];
                    if (order >= 2) {
                        w[2 * N] = d1(pc + N * K);
                        T a2 = 0;
#pragma unroll
                        for (int i = -N; i <= N; ++i) a2 += (T)i * w[N + i];
                        v[2] = a2 * inv_den;
                    } else {
                        v[2] = (T)0;
                        if (order == 1) w[N + 1] = d1(pc + K);
                    }
                    v[1] = w[N];
                    emit(q, v);
                    pc += K;
                }
        };
        const int first = zscore ? 0 : 2;
        // ---- short utterances, float32: one walk, the rows stay in registers
        if constexpr (sizeof(T) == 4) {
            const int64_t per = (To + nseg - 1) / nseg;
            if (n_tiles == 1 && per <= RMAX) {
                {
                    // the tile from the registers filled one utterance ago (rows from tile row N
                    // on, N edge copies either side), then the request for the next utterance
                    lds_barrier();              // walkers of the previous tile
                    const int cnt = (int)Tu * K;
                    T* body = tile + N * K;
#pragma unroll
                    for (int q = 0; q < PRE; ++q) {
                        const int i = threadIdx.x + q * 256;
                        if (i < cnt) body[i] = pre[q];
                    }
                    if ((int)threadIdx.x < K)
                        for (int p_ = 1; p_ <= N; ++p_) {
                            body[-p_ * K + (int)threadIdx.x] = pre_first;
                            body[cnt + (p_ - 1) * K + (int)threadIdx.x] = pre_last;
                        }
                    lds_barrier();
                }
                prefetch(sh_ahead);
                const int64_t a = trim + seg * per;
                const int64_t b = a + per < Tu - trim ? a + per : Tu - trim;
                const bool mine = seg < nseg && a < b;
                T r[RMAX][3];
#pragma unroll
                for (int q = 0; q < RMAX; ++q) r[q][0] = r[q][1] = r[q][2] = (T)0;
                if (mine)
                    walk_padded(std::integral_constant<int, RMAX>{}, (int)a + N, (int)b + N,
                                [&](int q, const T (&v)[3]) {
                                    r[q][0] = v[0];
                                    r[q][1] = v[1];
                                    r[q][2] = v[2];
                                });
                const int nrow = mine ? (int)(b - a) : 0;
                T mean[3] = {(T)0, (T)0, (T)0}, inv[3] = {(T)1, (T)1, (T)1};
                if (zscore) {
                    double acc[3] = {0.0, 0.0, 0.0};
#pragma unroll
                    for (int q = 0; q < RMAX; ++q)
                        if (q < nrow) {
#pragma unroll
                            for (int o = 0; o < 3; ++o) acc[o] += (double)r[q][o];
                        }
                    lds_barrier();              // (red / col_* of the previous utterance)
                    reduce(0, acc);
#pragma unroll
                    for (int o = 0; o < 3; ++o) {
                        mean[o] = o <= order ? col_mean[o * K + k] : (T)0;
                        acc[o] = 0.0;
                    }
#pragma unroll
                    for (int q = 0; q < RMAX; ++q)
                        if (q < nrow) {
#pragma unroll
                            for (int o = 0; o < 3; ++o) {
                                const double d = (double)r[q][o] - (double)mean[o];
                                acc[o] += d * d;
                            }
                        }
                    reduce(1, acc);
#pragma unroll
                    for (int o = 0; o < 3; ++o) inv[o] = o <= order ? (T)1 / col_scale[o * K + k] : (T)1;
                }
                T* orow = ou + (a - trim) * NF + k;
#pragma unroll
                for (int q = 0; q < RMAX; ++q)
                    if (q < nrow) {
#pragma unroll
                        for (int o = 0; o < 3; ++o)
                            if (o <= order) orow[(int64_t)q * NF + o * K] = (r[q][o] - mean[o]) * inv[o];
                    }
                continue;
            }
        }
        // ---- any length: a walk per phase
        prefetch(sh_ahead);
        if (threadIdx.x < NF) {
            col_mean[threadIdx.x] = (T)0;
            col_scale[threadIdx.x] = (T)1;
        }
        for (int phase = first; phase < 3; ++phase) {
            double acc[3] = {0.0, 0.0, 0.0};
            T mean[3], inv[3];
            __syncthreads();                    // col_mean / col_scale of the previous phase
#pragma unroll
            for (int o = 0; o < 3; ++o) {
                mean[o] = o <= order ? col_mean[o * K + k] : (T)0;
                inv[o] = o <= order ? (T)1 / col_scale[o * K + k] : (T)1;
            }
            for (int64_t ti = 0; ti < n_tiles; ++ti) {
                const int64_t t0 = trim + ti * tile_rows;
                const int64_t t1 = t0 + tile_rows < Tu - trim ? t0 + tile_rows : Tu - trim;
                const int64_t lo = t0 - 2 * N > 0 ? t0 - 2 * N : 0;
                const int64_t hi = t1 + 2 * N < Tu ? t1 + 2 * N : Tu;
                if (n_tiles > 1 || phase == first) load_tile(lo, hi);
                if (seg >= nseg) continue;
                const int64_t per = (t1 - t0 + nseg - 1) / nseg;
                const int64_t a = t0 + seg * per;
                const int64_t b = a + per < t1 ? a + per : t1;
                if (a >= b) continue;
                T* orow = ou + (a - trim) * NF + k;
                walk(std::integral_constant<int, 0>{}, lo, a, b, [&](int, int, const T (&v)[3]) {
#pragma unroll
                    for (int o = 0; o < 3; ++o) {
                        if (o > order) continue;
                        if (phase == 0) {
                            acc[o] += (double)v[o];
                        } else if (phase == 1) {
                            const double d = (double)v[o] - (double)mean[o];
                            acc[o] += d * d;
                        } else {
                            orow[o * K] = (v[o] - mean[o]) * inv[o];
                        }
                    }
                    orow += NF;
                });
            }
            if (phase == 2) break;
            reduce(phase, acc);
        }
        __syncthreads();
    }
}

// The clamp of librosa.power_to_db(top_db=...) sits between the logarithm and the
// DCT and needs the largest log-mel value of the whole utterance: one workgroup per
// utterance finds it over the stored log-mel rows, then each wave takes frames, raises
// the row to (maximum - top_db) in its LDS strip and lanes k < n_ceps take DCT row k.
// rowmax, rowmin (optional, together): the largest and smallest value of every row, left by a
// kernel that wrote the rows AND the cepstra of the unclamped rows: only rows whose minimum
// lies below the threshold are done again here (their log-mel row is not even read otherwise).
template <typename T>
__global__ __launch_bounds__(256) void topdb_dct_kernel(const T* __restrict__ logmel,
                                                        const int64_t* __restrict__ so, int64_t U,
                                                        MfccDev c, T top_db,
                                                        const T* __restrict__ dct,
                                                        const T* __restrict__ rowmax,
                                                        const T* __restrict__ rowmin,
                                                        T* __restrict__ cep) {
    __shared__ T red[256];
    __shared__ T red_min[256];
    constexpr int kMinRows = 1024;
    __shared__ T s_min[kMinRows];
    __shared__ T strip[4][128];
    const int wave = threadIdx.x / kWave, lane = threadIdx.x % kWave;
    // The lane's share of its DCT row (the split of dct_rows, summed in the same order)
    // stays in registers over all frames when it fits: n_mel a multiple of `parts`, at
    // most 32 weights per lane and 128 filters.
    int kp = 1;
    while (kp < c.n_ceps) kp <<= 1;
    const int parts = kWave / kp, k = lane / parts, part = lane % parts;
    const int held_n = c.n_mel / parts;
    const bool held = c.n_mel % parts == 0 && held_n <= 32 && c.n_mel <= 2 * kWave;
    T w[32];
#pragma unroll
    for (int i = 0; i < 32; ++i)
        w[i] = held && i < held_n && k < c.n_ceps ? dct[(size_t)k * c.n_mel + part + i * parts] : (T)0;
    for (int64_t u = blockIdx.x; u < U; u += gridDim.x) {
        const int64_t n = so[u + 1] - so[u];
        const int64_t Tu = raw_frames(c, n);
        if (Tu == 0) continue;
        const int64_t row0 = vpos(c, so, u) / c.hop;
        const T* lm = logmel + row0 * c.n_mel;
        const int64_t cnt = Tu * c.n_mel;
        T m = lm[0];
        T mn = lm[0];
        if (rowmax) {       // the rows' extrema are known: Tu values instead of Tu * n_mel
            for (int64_t i = threadIdx.x; i < Tu; i += blockDim.x) {
                const T hi = rowmax[row0 + i], lo = rowmin[row0 + i];
                m = hi > m ? hi : m;
                mn = lo < mn ? lo : mn;
                if (i < kMinRows) s_min[i] = lo;
            }
        } else {
            for (int64_t i = threadIdx.x; i < cnt; i += blockDim.x) m = lm[i] > m ? lm[i] : m;
        }
        red[threadIdx.x] = m;
        red_min[threadIdx.x] = mn;
        __syncthreads();
        for (int s = 128; s > 0; s >>= 1) {
            if ((int)threadIdx.x < s) {
                if (red[threadIdx.x + s] > red[threadIdx.x]) red[threadIdx.x] = red[threadIdx.x + s];
                if (red_min[threadIdx.x + s] < red_min[threadIdx.x]) red_min[threadIdx.x] = red_min[threadIdx.x + s];
            }
            __syncthreads();
        }
        const T thr = red[0] - top_db;
        if (rowmax && !(red_min[0] < thr)) {    // nothing of this utterance lies below the threshold
            __syncthreads();
            continue;
        }
        // does row t need the clamp (always, when the rows' minima are not known)
        auto need = [&](int64_t t) -> bool {
            if (t >= Tu) return false;
            if (!rowmax) return true;
            return (t < kMinRows ? s_min[t] : rowmin[row0 + t]) < thr;
        };
        if (held) {
            // the row of frame t + 4 is requested before the sums of frame t
            T nxt[2];
            auto request = [&](int64_t t) {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int q = lane + j * kWave;
                    const int64_t tc = t < Tu ? t : Tu - 1;
                    nxt[j] = lm[tc * c.n_mel + (q < c.n_mel ? q : c.n_mel - 1)];
                }
            };
            bool cur = need(wave);
            if (cur) request(wave);
            for (int64_t t = wave; t < Tu; t += 4) {
                const bool nxt_needed = need(t + 4);
                if (cur) {
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const int q = lane + j * kWave;
                        if (q < c.n_mel) strip[wave][q] = nxt[j] < thr ? thr : nxt[j];
                    }
                }
                if (nxt_needed) request(t + 4);
                if (cur) {
                    wave_sync();
                    T acc = 0;
#pragma unroll
                    for (int i = 0; i < 32; ++i)
                        if (i < held_n) acc += w[i] * strip[wave][part + i * parts];
                    for (int o = parts >> 1; o > 0; o >>= 1) acc += __shfl_xor(acc, o, kWave);
                    if (k < c.n_ceps && part == 0) cep[(row0 + t) * c.n_ceps + k] = acc;
                    wave_sync();
                }
                cur = nxt_needed;
            }
        } else {
            for (int64_t t = wave; t < Tu; t += 4) {
                if (!need(t)) continue;
                for (int q = lane; q < c.n_mel; q += kWave) {
                    const T v = lm[t * c.n_mel + q];
                    strip[wave][q] = v < thr ? thr : v;
                }
                wave_sync();
                dct_rows<T>(dct, strip[wave], c.n_mel, c.n_ceps, lane, cep + (row0 + t) * c.n_ceps);
                wave_sync();
            }
        }
        __syncthreads();                        // red of the next utterance
    }
}

// Launches finish_stream_kernel when delta_n has an instantiation; false otherwise.
template <typename T>
bool launch_finish_stream(srx_context* h, const void* cep, const int64_t* so, const int64_t* fo,
                          int64_t U, const MfccDev& d, const srx_mfcc_cfg_t& c, void* out,
                          hipStream_t st) {
    const int N = c.delta_order > 0 ? c.delta_n : 1;
    if (N > 3) return false;
    const size_t lds = 16 * 1024;
    const int rows = (int)(lds / (sizeof(T) * (size_t)c.n_ceps)) - 4 * N;
    if (rows < 16) return false;
    // (float32: the workgroups a CU holds, each walking its utterances with the next one's
    // cepstra already requested)
    const int64_t cap = (int64_t)h->n_cu * (sizeof(T) == 4 ? FIN_WG : 64);
    int64_t grid = U < cap ? U : cap;
#define SRX_CASE(n)                                                                              \
    if (N == n && c.n_ceps == 13)                                                                \
        hipLaunchKernelGGL((finish_stream_kernel<T, n, 13>), dim3((unsigned)grid), dim3(256), lds, st, \
                           (const T*)cep, so, fo, U, d, c.delta_order, c.zscore, rows, (T*)out); \
    else if (N == n)                                                                             \
        hipLaunchKernelGGL((finish_stream_kernel<T, n, 0>), dim3((unsigned)grid), dim3(256), lds, st, \
                           (const T*)cep, so, fo, U, d, c.delta_order, c.zscore, rows, (T*)out);
    SRX_CASE(1) SRX_CASE(2) SRX_CASE(3)
#undef SRX_CASE
    return true;
}

template <typename T>
int upload(const std::vector<double>& v, void** dst) {
    std::vector<T> tmp(v.begin(), v.end());
    SRX_HIP(hipMalloc(dst, sizeof(T) * (tmp.size() ? tmp.size() : 1)));
    SRX_HIP(hipMemcpy(*dst, tmp.data(), sizeof(T) * tmp.size(), hipMemcpyHostToDevice));
    return SRX_OK;
}

// Host-side dense filterbank exactly as the reference builds it (mfcc.py:47-61).
// Slaney's mel scale (linear below 1 kHz, logarithmic above), the default of
// librosa.filters.mel (htk=False).
double slaney_hz_to_mel(double f) {
    const double f_sp = 200.0 / 3, min_log_hz = 1000.0, logstep = log(6.4) / 27.0;
    return f >= min_log_hz ? min_log_hz / f_sp + log(f / min_log_hz) / logstep : f / f_sp;
}
double slaney_mel_to_hz(double m) {
    const double f_sp = 200.0 / 3, min_log_hz = 1000.0, logstep = log(6.4) / 27.0;
    const double min_log_mel = min_log_hz / f_sp;
    return m >= min_log_mel ? min_log_hz * exp(logstep * (m - min_log_mel)) : f_sp * m;
}

// librosa.filters.mel(sr, n_fft, n_mels, fmin=0, fmax=sr/2, htk=False, norm='slaney'):
// triangles over the bins' centre frequencies between n_mel + 2 points equally spaced on
// the Slaney scale, each scaled by 2 / (width in Hz).
void build_filterbank_slaney(const srx_mfcc_cfg_t& c, std::vector<double>& fb, int nbins) {
    const double sr = c.mel_sample_rate;
    const double m_lo = slaney_hz_to_mel(0.0), m_hi = slaney_hz_to_mel(sr / 2);
    std::vector<double> f(c.n_mel + 2);
    for (int i = 0; i < c.n_mel + 2; ++i)
        f[i] = slaney_mel_to_hz(m_lo + (m_hi - m_lo) * i / (c.n_mel + 1));
    for (int m = 0; m < c.n_mel; ++m) {
        const double enorm = 2.0 / (f[m + 2] - f[m]);
        for (int k = 0; k < nbins; ++k) {
            const double hz = (sr / 2) * k / (nbins - 1);
            const double lower = (hz - f[m]) / (f[m + 1] - f[m]);
            const double upper = (f[m + 2] - hz) / (f[m + 2] - f[m + 1]);
            const double w = lower < upper ? lower : upper;
            fb[(size_t)m * nbins + k] = w > 0 ? w * enorm : 0.0;
        }
    }
}

void build_filterbank(const srx_mfcc_cfg_t& c, std::vector<double>& fb, int& nbins) {
    nbins = c.nfft / 2 + 1;
    fb.assign((size_t)c.n_mel * nbins, 0.0);
    if (c.mel_style == 1) {
        build_filterbank_slaney(c, fb, nbins);
        return;
    }
    const double top = 2595.0 * log10(1.0 + (double)c.mel_sample_rate / 2 / 700);
    std::vector<double> edge(c.n_mel + 2);
    const double step = top / (c.n_mel + 1);
    for (int i = 0; i < c.n_mel + 2; ++i) {
        const double mel = i == c.n_mel + 1 ? top : i * step;
        const double hz = 700 * (pow(10.0, mel / 2595) - 1);
        edge[i] = floor((c.nfft + 1) * hz / c.mel_sample_rate);
    }
    for (int m = 0; m < c.n_mel; ++m) {
        const double lo = edge[m], mid = edge[m + 1], hi = edge[m + 2];
        for (int k = (int)lo; k < (int)mid && k < nbins; ++k) fb[(size_t)m * nbins + k] = (k - lo) / (mid - lo);
        for (int k = (int)mid; k < (int)hi && k < nbins; ++k) fb[(size_t)m * nbins + k] = (hi - k) / (hi - mid);
    }
}

int check_cfg(const srx_mfcc_cfg_t* c) {
    SRX_REQUIRE(c != nullptr, "mfcc: NULL cfg");
    SRX_REQUIRE(c->nfft >= 64 && c->nfft <= 4096 && (c->nfft & (c->nfft - 1)) == 0,
                "mfcc: nfft %d must be a power of two in [64, 4096]", c->nfft);
    SRX_REQUIRE(c->frame_len >= 1 && c->hop >= 1 && c->hop <= c->frame_len,
                "mfcc: need 1 <= hop (%d) <= frame_len (%d)", c->hop, c->frame_len);
    SRX_REQUIRE(c->n_mel >= 1 && c->n_mel <= 128, "mfcc: n_mel %d outside [1, 128]", c->n_mel);
    SRX_REQUIRE(c->n_ceps >= 1 && c->n_ceps <= c->n_mel && c->n_ceps <= 32,
                "mfcc: n_ceps %d outside [1, min(n_mel, 32)]", c->n_ceps);
    SRX_REQUIRE(c->delta_order >= 0 && c->delta_order <= 2, "mfcc: delta_order %d outside [0, 2]",
                c->delta_order);
    SRX_REQUIRE(c->delta_n >= 1 && c->delta_n <= 8, "mfcc: delta_n %d outside [1, 8]", c->delta_n);
    SRX_REQUIRE(c->mel_sample_rate > 0 && c->sample_rate > 0, "mfcc: sample rates must be positive");
    SRX_REQUIRE(c->window >= 0 && c->window <= 1, "mfcc: window %d outside [0, 1]", c->window);
    SRX_REQUIRE(c->center >= 0 && c->center <= 2, "mfcc: center %d outside [0, 2]", c->center);
    SRX_REQUIRE(c->mel_style >= 0 && c->mel_style <= 1, "mfcc: mel_style %d outside [0, 1]", c->mel_style);
    SRX_REQUIRE(c->power_norm >= 0 && c->power_norm <= 1, "mfcc: power_norm %d outside [0, 1]", c->power_norm);
    SRX_REQUIRE(c->log_floor >= 0.0 && c->top_db >= 0.0, "mfcc: log_floor and top_db must not be negative");
    SRX_REQUIRE(!c->center || c->frame_len <= c->nfft,
                "mfcc: centred frames need frame_len (%d) <= nfft (%d)", c->frame_len, c->nfft);
    return SRX_OK;
}

}  // namespace

extern "C" {

int64_t srx_mfcc_out_frames(const srx_mfcc_cfg_t* c, int64_t n_samples) {
    if (!c) return 0;
    if (c->center ? n_samples <= 0 : n_samples < c->frame_len) return 0;
    const int64_t t = c->center ? n_samples / c->hop + 1 : (n_samples - c->frame_len) / c->hop + 1;
    if (c->delta_order == 0) return t;
    return t > 2 * c->delta_n ? t - 2 * c->delta_n : 0;
}

// Builds the tables of a zero-initialised plan; on failure the caller destroys the plan
// with whatever tables it already holds.
static int plan_fill(srx_mfcc_plan* p, const srx_mfcc_cfg_t* cfg) {
    int rc = SRX_OK;
    int l2 = 0;
    while ((1 << l2) < cfg->nfft) ++l2;
    p->log2n = l2;
    const int N = cfg->nfft, H = N / 2;
    const int used = cfg->frame_len < N ? cfg->frame_len : N;   // rfft(frames, NFFT) crops
    std::vector<double> win(cfg->frame_len), tw(2 * (size_t)(H + 1));
    for (int i = 0; i < cfg->frame_len; ++i)
        win[i] = cfg->window == 1 ? 0.5 - 0.5 * cos(2.0 * M_PI * i / cfg->frame_len)
                 : cfg->frame_len == 1 ? 1.0
                                     : 0.54 - 0.46 * cos(2.0 * M_PI * i / (cfg->frame_len - 1));
    (void)used;
    if (cfg->center) {
        // a centred frame always has nfft points: the window sits in their middle
        std::vector<double> padded(N, 0.0);
        const int lpad = (N - cfg->frame_len) / 2;
        for (int i = 0; i < cfg->frame_len; ++i) padded[lpad + i] = win[i];
        win.swap(padded);
    }
    for (int k = 0; k <= H; ++k) {
        tw[2 * k] = cos(2.0 * M_PI * k / N);
        tw[2 * k + 1] = -sin(2.0 * M_PI * k / N);
    }
    std::vector<double> fb;
    int nbins;
    build_filterbank(*cfg, fb, nbins);
    std::vector<double> w;
    std::vector<int32_t> bins, ptr(cfg->n_mel + 1, 0);
    for (int m = 0; m < cfg->n_mel; ++m) {
        for (int k = 0; k < nbins; ++k)
            if (fb[(size_t)m * nbins + k] != 0.0) {
                w.push_back(fb[(size_t)m * nbins + k]);
                bins.push_back(k);
            }
        ptr[m + 1] = (int32_t)w.size();
    }
    p->nnz = (int)w.size();
    std::vector<int32_t> starts(cfg->n_mel, 0);
    p->fast_ok = 1;
    for (int m = 0; m < cfg->n_mel; ++m) {
        const int q0 = ptr[m], q1 = ptr[m + 1];
        starts[m] = q1 > q0 ? bins[q0] : 0;
        for (int q = q0; q < q1; ++q)
            if (bins[q] != starts[m] + (q - q0)) p->fast_ok = 0;
    }
    // (Both register-transform kernels leave |2 X|^2 in their power bins: the factor 1/4, and 1/N
    // where the spectrum is divided by N, both powers of two, are part of these tables' weights.)
    // ---- block tables of the fast kernel's mel and DCT stages.  A 4x4x1 matrix
    // instruction is 16 independent 4x4 outer products, one per group of 4 lanes:
    // a block = 4 consecutive filters x a range of bins (first bin a multiple of 4,
    // mm_nq quads of bins).  One block per group of 4 filters to begin with, then the
    // longest ranges are halved until all 16 blocks are in use.
    p->mm_ok = 0;
    const bool plain = !cfg->window && !cfg->center && !cfg->power_norm && cfg->log_floor == 0.0 &&
                       cfg->top_db == 0.0;   // (the register-FFT kernel knows mfcc.py's conventions only)
    if (plain && p->fast_ok && cfg->n_mel <= 64 && cfg->n_ceps <= 16 && (N == 512 || N == 1024)) {
        struct Blk { int g, lo, hi; };
        const int G = (cfg->n_mel + 3) / 4, PW = H + 4;
        std::vector<Blk> blk;
        for (int g = 0; g < G; ++g) {
            int lo = 1 << 30, hi = 0;
            for (int m = 4 * g; m < 4 * g + 4 && m < cfg->n_mel; ++m)
                if (ptr[m + 1] > ptr[m]) {
                    lo = std::min(lo, (int)starts[m]);
                    hi = std::max(hi, (int)starts[m] + (ptr[m + 1] - ptr[m]));
                }
            if (hi == 0) lo = 0;
            blk.push_back({g, lo & ~3, hi});
        }
        auto quads = [](const Blk& b) { return (b.hi - b.lo + 3) / 4; };
        // The kernel walks nq quads in every block, so nq is what the stage costs: take the
        // smallest piece length that cuts the groups' ranges into at most 16 pieces, among the
        // lengths the kernel can walk (multiples of 3 or 4: it keeps that many quads' operands
        // in flight), and cut every group into equal pieces of at most that length.
        int nq = 1;
        for (const Blk& b : blk) nq = std::max(nq, quads(b));
        for (int cand = 3; cand <= nq; ++cand) {
            if (cand % 3 != 0 && cand % 4 != 0) continue;
            int pieces = 0;
            for (const Blk& b : blk) pieces += std::max(1, (quads(b) + cand - 1) / cand);
            if (pieces <= 16) {
                nq = cand;
                break;
            }
        }
        while (nq % 3 != 0 && nq % 4 != 0) ++nq;
        {
            std::vector<Blk> cut;
            for (const Blk& b : blk) {
                const int q = quads(b), n = std::max(1, (q + nq - 1) / nq);
                for (int i = 0; i < n; ++i) {
                    const int q0 = q * i / n, q1 = q * (i + 1) / n;      // (pieces of q / n quads, at most nq)
                    cut.push_back({b.g, b.lo + 4 * q0, i + 1 == n ? b.hi : b.lo + 4 * q1});
                }
            }
            if (cut.size() <= 16) blk = cut;
        }
        const int kk = (((cfg->n_mel + 3) / 4) + 3) & ~3;      // filters per DCT split, a multiple of 4
        std::vector<float> w4((size_t)nq * 64 * 4, 0.f), dctA((size_t)64 * kk, 0.f);
        std::vector<int32_t> meta(48, 0);
        for (size_t b = 0; b < blk.size(); ++b) {
            int k0 = blk[b].lo;
            if (k0 + 4 * nq > PW) k0 = PW - 4 * nq;
            meta[b] = k0;
            if (b == 0 || blk[b - 1].g != blk[b].g) meta[16 + blk[b].g] = (int32_t)b;
            meta[32 + blk[b].g] += 1;
            for (int i = 0; i < 4; ++i) {
                const int m = 4 * blk[b].g + i;
                if (m >= cfg->n_mel) continue;
                for (int q = ptr[m]; q < ptr[m + 1]; ++q) {
                    const int bin = bins[q];
                    if (bin < blk[b].lo || bin >= blk[b].hi) continue;
                    const int rel = bin - k0;
                    w4[((size_t)(rel / 4) * 64 + 4 * b + i) * 4 + rel % 4] = (float)w[q] * (0.25f / (float)N);
                }
            }
        }
        for (int l = 0; l < 64; ++l) {
            const int ks = l >> 4, cg = (l >> 2) & 3, i = l & 3, k = 4 * cg + i;
            for (int s = 0; s < kk; ++s) {
                const int m = ks * kk + s;
                if (k < cfg->n_ceps && m < cfg->n_mel) {
                    double v = sqrt(2.0 / cfg->n_mel) * cos(M_PI * k * (2 * m + 1) / (2.0 * cfg->n_mel));
                    if (k == 0) v *= 1.0 / sqrt(2.0);
                    dctA[(size_t)l * kk + s] = (float)v;
                }
            }
        }
        if (4 * nq <= PW) {
            SRX_HIP(hipMalloc(&p->mm_w, sizeof(float) * w4.size()));
            SRX_HIP(hipMemcpy(p->mm_w, w4.data(), sizeof(float) * w4.size(), hipMemcpyHostToDevice));
            SRX_HIP(hipMalloc(&p->mm_dct, sizeof(float) * dctA.size()));
            SRX_HIP(hipMemcpy(p->mm_dct, dctA.data(), sizeof(float) * dctA.size(), hipMemcpyHostToDevice));
            SRX_HIP(hipMalloc((void**)&p->mm_meta, sizeof(int32_t) * meta.size()));
            SRX_HIP(hipMemcpy(p->mm_meta, meta.data(), sizeof(int32_t) * meta.size(), hipMemcpyHostToDevice));
            p->mm_nq = nq; p->mm_kk = kk; p->mm_groups = G; p->mm_ok = 1;
        }
    }
    // ---- mel stage of the 2048-point kernel: every filter's bin range as whole quads of 4
    // bins, filter-major, dealt to 64 lanes in equal runs; a lane's stretch of one filter ends
    // in one partial sum (slot), and a filter's slots are consecutive.
    p->wd_ok = 0;
    if (N == 2048 && cfg->center && p->fast_ok && cfg->n_mel <= 128) {
        struct Quad { int q, f, last; };
        std::vector<Quad> quads;
        for (int m = 0; m < cfg->n_mel; ++m) {
            const int len = ptr[m + 1] - ptr[m];
            const int q0 = starts[m] / 4, q1 = len > 0 ? (starts[m] + len - 1) / 4 : q0;
            for (int q = q0; q <= q1; ++q) quads.push_back({q, m, q == q1});
        }
        // (quads per lane: a multiple of 12, the kernel requests its table entries in two
        // alternating batches of 6)
        const int lq = ((((int)quads.size() + 63) / 64 + 11) / 12) * 12;
        std::vector<float> w4((size_t)lq * 64 * 4, 0.f);
        std::vector<int32_t> meta((size_t)lq * 64, 0), pptr(cfg->n_mel + 1, 0);
        int slot = -1, prev_f = -1, prev_lane = -1;
        for (size_t i = 0; i < quads.size(); ++i) {
            const int lane = (int)(i / lq), s = (int)(i % lq);
            const Quad& qd = quads[i];
            if (qd.f != prev_f || lane != prev_lane) ++slot;
            if (qd.f != prev_f) pptr[qd.f] = slot;
            prev_f = qd.f;
            prev_lane = lane;
            // the stretch ends with the filter or with the lane's run
            const bool ends = qd.last || s == lq - 1;
            meta[(size_t)s * 64 + lane] = qd.q | (slot << 16) | (ends ? (int32_t)0x80000000 : 0);
            for (int q = ptr[qd.f]; q < ptr[qd.f + 1]; ++q)
                if (bins[q] / 4 == qd.q)
                    w4[((size_t)s * 64 + lane) * 4 + bins[q] % 4] = (float)w[q] * (cfg->power_norm ? 0.25f : 0.25f / (float)N);
        }
        const int n_parts = slot + 1;
        pptr[cfg->n_mel] = n_parts;
        // runs shorter than lq (the tail of the last lanes): zero weights into the spare slot
        for (size_t i = quads.size(); i < (size_t)lq * 64; ++i)
            meta[(i % lq) * 64 + i / lq] = n_parts << 16;
        if (n_parts + 1 <= 256 && lq >= 1) {
            SRX_HIP(hipMalloc(&p->wd_w, sizeof(float) * w4.size()));
            SRX_HIP(hipMemcpy(p->wd_w, w4.data(), sizeof(float) * w4.size(), hipMemcpyHostToDevice));
            SRX_HIP(hipMalloc((void**)&p->wd_meta, sizeof(int32_t) * meta.size()));
            SRX_HIP(hipMemcpy(p->wd_meta, meta.data(), sizeof(int32_t) * meta.size(), hipMemcpyHostToDevice));
            SRX_HIP(hipMalloc((void**)&p->wd_ptr, sizeof(int32_t) * pptr.size()));
            SRX_HIP(hipMemcpy(p->wd_ptr, pptr.data(), sizeof(int32_t) * pptr.size(), hipMemcpyHostToDevice));
            std::vector<float> dpad((size_t)cfg->n_ceps * 128, 0.f);
            for (int k = 0; k < cfg->n_ceps; ++k)
                for (int n = 0; n < cfg->n_mel; ++n) {
                    double v = sqrt(2.0 / cfg->n_mel) * cos(M_PI * k * (2 * n + 1) / (2.0 * cfg->n_mel));
                    if (k == 0) v *= 1.0 / sqrt(2.0);
                    dpad[(size_t)k * 128 + n] = (float)v;
                }
            SRX_HIP(hipMalloc(&p->wd_dct, sizeof(float) * dpad.size()));
            SRX_HIP(hipMemcpy(p->wd_dct, dpad.data(), sizeof(float) * dpad.size(), hipMemcpyHostToDevice));
            p->wd_lq = lq;
            p->wd_ok = 1;
        }
    }
    std::vector<float> twf(2 * (size_t)N);
    for (int k = 0; k < N; ++k) {
        twf[2 * k] = (float)cos(2.0 * M_PI * k / N);
        twf[2 * k + 1] = (float)(-sin(2.0 * M_PI * k / N));
    }
    SRX_HIP(hipMalloc(&p->tw_full, sizeof(float) * twf.size()));
    SRX_HIP(hipMemcpy(p->tw_full, twf.data(), sizeof(float) * twf.size(), hipMemcpyHostToDevice));
    SRX_HIP(hipMalloc((void**)&p->mel_start, sizeof(int32_t) * starts.size()));
    SRX_HIP(hipMemcpy(p->mel_start, starts.data(), sizeof(int32_t) * starts.size(), hipMemcpyHostToDevice));
    std::vector<double> dct((size_t)cfg->n_ceps * cfg->n_mel);
    for (int k = 0; k < cfg->n_ceps; ++k)
        for (int n = 0; n < cfg->n_mel; ++n) {
            double v = sqrt(2.0 / cfg->n_mel) * cos(M_PI * k * (2 * n + 1) / (2.0 * cfg->n_mel));
            if (k == 0) v *= 1.0 / sqrt(2.0);
            dct[(size_t)k * cfg->n_mel + n] = v;
        }
    rc = upload<float>(win, &p->window[0]);   if (rc) return rc;
    rc = upload<double>(win, &p->window[1]);  if (rc) return rc;
    rc = upload<float>(tw, &p->twiddle[0]);   if (rc) return rc;
    rc = upload<double>(tw, &p->twiddle[1]);  if (rc) return rc;
    rc = upload<float>(w, &p->mel_w[0]);      if (rc) return rc;
    rc = upload<double>(w, &p->mel_w[1]);     if (rc) return rc;
    rc = upload<float>(dct, &p->dct[0]);      if (rc) return rc;
    rc = upload<double>(dct, &p->dct[1]);     if (rc) return rc;
    SRX_HIP(hipMalloc((void**)&p->mel_bin, sizeof(int32_t) * (bins.size() ? bins.size() : 1)));
    SRX_HIP(hipMemcpy(p->mel_bin, bins.data(), sizeof(int32_t) * bins.size(), hipMemcpyHostToDevice));
    SRX_HIP(hipMalloc((void**)&p->mel_ptr, sizeof(int32_t) * ptr.size()));
    SRX_HIP(hipMemcpy(p->mel_ptr, ptr.data(), sizeof(int32_t) * ptr.size(), hipMemcpyHostToDevice));
    return SRX_OK;
}

int srx_mfcc_plan_destroy(srx_mfcc_plan_t p);

int srx_mfcc_plan_create(srx_handle_t h, const srx_mfcc_cfg_t* cfg, srx_mfcc_plan_t* out) {
    SRX_REQUIRE(h != nullptr && out != nullptr, "srx_mfcc_plan_create: NULL argument");
    *out = nullptr;
    int rc = check_cfg(cfg);
    if (rc) return rc;
    SRX_HIP(hipSetDevice(h->device));
    srx_mfcc_plan* p = new srx_mfcc_plan();
    memset(p, 0, sizeof(*p));
    p->ctx = h;
    p->cfg = *cfg;
    rc = plan_fill(p, cfg);
    if (rc) {
        srx_mfcc_plan_destroy(p);      // frees the tables made so far; the message stays
        return rc;
    }
    *out = p;
    return SRX_OK;
}

int srx_mfcc_plan_destroy(srx_mfcc_plan_t p) {
    if (!p) return SRX_OK;
    (void)hipSetDevice(p->ctx->device);
    for (int i = 0; i < 2; ++i) {
        (void)hipFree(p->window[i]);
        (void)hipFree(p->twiddle[i]);
        (void)hipFree(p->mel_w[i]);
        (void)hipFree(p->dct[i]);
    }
    (void)hipFree(p->mel_bin);
    (void)hipFree(p->mel_ptr);
    (void)hipFree(p->tw_full);
    (void)hipFree(p->mel_start);
    (void)hipFree(p->mm_w);
    (void)hipFree(p->mm_dct);
    (void)hipFree(p->mm_meta);
    (void)hipFree(p->wd_w);
    (void)hipFree(p->wd_meta);
    (void)hipFree(p->wd_ptr);
    (void)hipFree(p->wd_dct);
    delete p;
    return SRX_OK;
}

int srx_mfcc_filterbank_host(const srx_mfcc_cfg_t* cfg, double* out_dense) {
    int rc = check_cfg(cfg);
    if (rc) return rc;
    SRX_REQUIRE(out_dense != nullptr, "srx_mfcc_filterbank_host: NULL out");
    std::vector<double> fb;
    int nbins;
    build_filterbank(*cfg, fb, nbins);
    memcpy(out_dense, fb.data(), sizeof(double) * fb.size());
    return SRX_OK;
}

int srx_mfcc_batch(srx_handle_t h, srx_mfcc_plan_t p, const void* audio, int dtype,
                   const int64_t* sample_offsets, int64_t n_utts, int64_t n_samples,
                   void* out_feats, const int64_t* frame_offsets, void* stream) {
    SRX_REQUIRE(h != nullptr && p != nullptr, "srx_mfcc_batch: NULL context or plan");
    SRX_REQUIRE(p->ctx == h, "srx_mfcc_batch: plan belongs to another context");
    SRX_REQUIRE(dtype == SRX_F32 || dtype == SRX_F64, "srx_mfcc_batch: unknown dtype %d", dtype);
    SRX_REQUIRE(n_utts >= 0 && n_samples >= 0, "srx_mfcc_batch: negative size");
    if (n_utts == 0 || n_samples == 0) return SRX_OK;
    SRX_REQUIRE(audio && sample_offsets && frame_offsets, "srx_mfcc_batch: NULL buffer");
    if (out_feats == nullptr) return SRX_OK;   // no utterance yields a frame
    SRX_HIP(hipSetDevice(h->device));
    hipStream_t st = (hipStream_t)stream;
    const srx_mfcc_cfg_t& c = p->cfg;
    const int H = c.nfft / 2;
    MfccDev d;
    d.frame_len = c.frame_len; d.hop = c.hop; d.nfft = c.nfft; d.log2h = p->log2n - 1;
    d.n_mel = c.n_mel; d.n_ceps = c.n_ceps;
    d.n_used = c.frame_len < c.nfft ? c.frame_len : c.nfft;
    d.mel_ptr = p->mel_ptr; d.mel_bin = p->mel_bin; d.log_mult = c.log_mult;
    d.center = c.center;
    d.power_scale = c.power_norm ? 1.0 : 1.0 / c.nfft;
    d.log_floor = c.log_floor;
    d.logmel = nullptr;
    if (c.center) {
        d.frame_len = c.nfft;               // the window table is padded to nfft points
        d.n_used = c.nfft;
    }
    const size_t esz = dtype == SRX_F64 ? 8 : 4;
    // cepstra rows and work items live on the virtual sample axis (see vpos)
    const int64_t v_samples = n_samples + (c.center ? n_utts * (int64_t)c.hop : 0);
    const int64_t rows = v_samples / c.hop + 1;
    void* cep;
    int rc = srx_ws_ensure(h, WS_CEPSTRA, esz * (size_t)rows * c.n_ceps, &cep);
    if (rc) return rc;
    if (c.top_db > 0.0) {
        rc = srx_ws_ensure(h, WS_LOGLIK, esz * (size_t)rows * c.n_mel, &d.logmel);
        if (rc) return rc;
    }
    const size_t per_wave = (esz * ((size_t)2 * H + c.n_mel + 1) + 15) & ~(size_t)15;
    // waves per workgroup: the count that leaves the most waves on a CU's 160 KB of LDS
    // within the 64 KB a workgroup may take (long transforms are bound by LDS, not registers)
    int waves = 1, best = 0;
    for (int w = 1; w <= 8; ++w) {
        if (per_wave * w > 64 * 1024) break;
        int per_cu = (int)(160 * 1024 / (per_wave * w)) * w;
        per_cu = per_cu > 32 ? 32 : per_cu;
        if (per_cu > best || (per_cu == best && w <= 4)) { best = per_cu; waves = w; }
    }
    SRX_REQUIRE(per_wave * waves <= 64 * 1024, "srx_mfcc_batch: nfft %d needs %zu bytes of LDS",
                c.nfft, per_wave);
    const int64_t chunk = (int64_t)kChunkFrames * c.hop;
    int64_t n_chunks = (v_samples + chunk - 1) / chunk;
    int64_t grid = n_chunks < (int64_t)h->n_cu * 16 ? n_chunks : (int64_t)h->n_cu * 16;
    int64_t grid2 = n_utts < (int64_t)h->n_cu * 8 ? n_utts : (int64_t)h->n_cu * 8;
    const int idx = dtype == SRX_F64 ? 1 : 0;
    const size_t fin_lds = 60 * 1024;
    const int fin_elems = (int)(fin_lds / esz);
    SRX_HIP(hipFuncSetAttribute((const void*)finish_kernel<float>,
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)fin_lds));
    SRX_HIP(hipFuncSetAttribute((const void*)finish_kernel<double>,
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)fin_lds));
    if (dtype == SRX_F64) {
        hipLaunchKernelGGL(cepstra_kernel<double>, dim3((unsigned)grid), dim3(waves * kWave),
                           per_wave * waves, st, (const double*)audio, sample_offsets, n_utts,
                           v_samples, d, (const double*)p->window[idx],
                           (const cplx<double>*)p->twiddle[idx], (const double*)p->mel_w[idx],
                           (const double*)p->dct[idx], (double*)cep);
        if (d.logmel)
            hipLaunchKernelGGL(topdb_dct_kernel<double>, dim3((unsigned)grid2), dim3(256), 0, st,
                               (const double*)d.logmel, sample_offsets, n_utts, d, c.top_db,
                               (const double*)p->dct[idx], (const double*)nullptr, (const double*)nullptr,
                               (double*)cep);
        if (!launch_finish_stream<double>(h, cep, sample_offsets, frame_offsets, n_utts, d, c, out_feats, st))
            hipLaunchKernelGGL(finish_kernel<double>, dim3((unsigned)grid2), dim3(256), fin_lds, st,
                               (const double*)cep, sample_offsets, frame_offsets, n_utts, d,
                               c.delta_order, c.delta_n, c.zscore, fin_elems, (double*)out_feats);
    } else {
        // centred 2048-point frames: cepstra of the unclamped log-mel rows from the register-FFT
        // kernel; with top_db, the rows that reach below the threshold are done again
        float *rowmax = nullptr, *rowmin = nullptr;
        void* wide_logmel = nullptr;
        rc = SRX_EUNSUPPORTED;
        if (!h->exact_f32 && p->wd_ok) {
            if (c.top_db > 0.0) {
                rc = srx_ws_ensure(h, WS_LOGLIK, esz * (size_t)rows * c.n_mel, &wide_logmel);
                if (rc) return rc;
            }
            rc = srx_mfcc_wide_cepstra(h, p, (const float*)audio, sample_offsets, n_utts, n_samples,
                                       c.top_db > 0.0 ? (float*)wide_logmel : nullptr, (float*)cep,
                                       &rowmax, &rowmin, st);
            if (rc == SRX_OK) d.logmel = c.top_db > 0.0 ? wide_logmel : nullptr;
        }
        if (rc == SRX_EUNSUPPORTED && !h->exact_f32)
            rc = srx_mfcc_fast_cepstra(h, p, (const float*)audio, sample_offsets, n_utts, n_samples,
                                       (float*)cep, st);
        if (rc == SRX_EUNSUPPORTED)
            hipLaunchKernelGGL(cepstra_kernel<float>, dim3((unsigned)grid), dim3(waves * kWave),
                               per_wave * waves, st, (const float*)audio, sample_offsets, n_utts,
                               v_samples, d, (const float*)p->window[idx],
                               (const cplx<float>*)p->twiddle[idx], (const float*)p->mel_w[idx],
                               (const float*)p->dct[idx], (float*)cep);
        else if (rc)
            return rc;
        if (d.logmel)
            hipLaunchKernelGGL(topdb_dct_kernel<float>, dim3((unsigned)grid2), dim3(256), 0, st,
                               (const float*)d.logmel, sample_offsets, n_utts, d, (float)c.top_db,
                               (const float*)p->dct[idx], (const float*)rowmax, (const float*)rowmin,
                               (float*)cep);
        if (!launch_finish_stream<float>(h, cep, sample_offsets, frame_offsets, n_utts, d, c, out_feats, st))
            hipLaunchKernelGGL(finish_kernel<float>, dim3((unsigned)grid2), dim3(256), fin_lds, st,
                               (const float*)cep, sample_offsets, frame_offsets, n_utts, d,
                               c.delta_order, c.delta_n, c.zscore, fin_elems, (float*)out_feats);
    }
    SRX_HIP(hipGetLastError());
    return SRX_OK;
}

}  // extern "C"
