// End-pointing in front of the MFCC path: the energy / zero-crossing block features
// and the run search of the reference's AudioMonitor (new_speech_detect.py:30-61,
// :80-89, :197-343), for many recordings at once.
//
//   endpoint_blocks_kernel   16 lanes per 10 ms block: sum of |x| (float64) and the
//                            number of sign changes inside the block.  One pass over
//                            the samples: 4 (or 8) bytes read per sample, 12 bytes
//                            written per block.
//   endpoint_detect_kernel   16 lanes per recording: peak / floor energies, the
//                            three thresholds, the sequential run search over the
//                            recording's blocks, widening by the zero-crossing rule.
#include "srx_internal.h"

#include <math.h>
#include <type_traits>

namespace {

// Value of lane (j + 1) mod 16 of the same row of 16 lanes (DPP row rotation: no LDS).
__device__ __forceinline__ float rot1(float v) {
    return __uint_as_float((unsigned)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(v), 0x12F, 0xF, 0xF, false));
}
__device__ __forceinline__ double rot1(double v) {
    const unsigned long long u = __double_as_longlong(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)u, 0x12F, 0xF, 0xF, false);
    const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(u >> 32), 0x12F, 0xF, 0xF, false);
    return __longlong_as_double(((unsigned long long)hi << 32) | lo);
}

// Value of lane (j - N) mod 16 of the same row of 16 lanes (DPP row rotation).
template <int N>
__device__ __forceinline__ int ror(int v) {
    return __builtin_amdgcn_update_dpp(0, v, 0x120 + N, 0xF, 0xF, false);
}
template <int N>
__device__ __forceinline__ double ror(double v) {
    const unsigned long long u = __double_as_longlong(v);
    const unsigned lo = (unsigned)ror<N>((int)(unsigned)u), hi = (unsigned)ror<N>((int)(unsigned)(u >> 32));
    return __longlong_as_double(((unsigned long long)hi << 32) | lo);
}
// Sum over the 16 lanes of a row, in every lane.  Rotations by 8, 4, 2, 1 add the same pairs
// as an xor butterfly does (a + b is b + a), so the float64 sums are those bit for bit.
template <typename V>
__device__ __forceinline__ V row_sum(V v) {
    v += ror<8>(v);
    v += ror<4>(v);
    v += ror<2>(v);
    v += ror<1>(v);
    return v;
}

template <typename T>
__global__ __launch_bounds__(256) void endpoint_blocks_kernel(
    const T* __restrict__ audio, const int64_t* __restrict__ so, const int64_t* __restrict__ bo,
    int64_t n_rec, int64_t n_blocks, int block, double* __restrict__ energy,
    int32_t* __restrict__ zcr) {
    const int j = threadIdx.x & 15;
    // Results of a wave's run wait in LDS and leave together at the end: even 12 bytes stored per
    // 640 read, scattered between the loads, cost a read stream a fifth of its rate on this
    // chip (tools/probes/chunk_stream.hip).  Runs longer than the buffer store directly.
    constexpr int kKeep = 512;
    __shared__ double s_e[4][kKeep];
    __shared__ int32_t s_z[4][kKeep];
    const int wv = threadIdx.x >> 6;
    // A wave walks a CONTIGUOUS run of blocks, its four groups of 16 lanes taking four
    // consecutive blocks per trip: one trip reads 4 blocks' samples in a row, and the recording
    // under a group's block changes once in hundreds of trips.  (Dealt out with the stride of
    // the whole grid, every trip landed in another recording and paid a binary search of
    // dependent loads before it could ask for its samples: 0.60 ms for 8192 x 3 s against 0.36.)
    const int grp = (threadIdx.x >> 4) & 3;
    const int64_t wave_id = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    const int64_t per_wave = (((n_blocks + n_waves - 1) / n_waves) + 3) & ~(int64_t)3;
    const int64_t b_first = wave_id * per_wave;
    const int64_t b_end = b_first + per_wave < n_blocks ? b_first + per_wave : n_blocks;
    // The recording under a block advances monotonically with the block index: its
    // bounds stay in registers and are looked up again only when a block leaves them.
    int64_t rec = 0, b_lo = 0, b_hi = 0, s_lo = 0, s_hi = 0;
    constexpr int kBatch = 10;                 // loads in flight per lane (160 samples = 1 piece)
    // first sample and length of block b
    auto locate = [&](int64_t b, const T*& x, int& len) {
        if (!(b_lo <= b && b < b_hi)) {
            int64_t lo = b >= b_lo ? rec : 0, hi = n_rec;
            // (the next recording first: that is where a block behind this one's end lies
            // unless recordings without blocks sit in between)
            if (b >= b_hi && b_hi > b_lo && rec + 1 < n_rec && b < bo[rec + 2]) {
                lo = rec + 1;
                hi = rec + 2;
            }
            while (hi - lo > 1) {
                const int64_t mid = (lo + hi) >> 1;
                if (bo[mid] <= b) lo = mid; else hi = mid;
            }
            rec = lo;
            b_lo = bo[rec];
            b_hi = bo[rec + 1];
            s_lo = so[rec];
            s_hi = so[rec + 1];
        }
        const int64_t s_begin = s_lo + (b - b_lo) * block;
        len = (int)(s_hi - s_begin < block ? s_hi - s_begin : block);
        x = audio + s_begin;
    };
    // lane j takes samples j, j + 16, ...: the sample after lane j's is lane j+1's of
    // the same round, or lane 0's of the next round (a round more is loaded for that).
    // Every lane loads in every round, from a clamped index, and what lies past the block is
    // zeroed afterwards: loads under a branch cannot be counted, and the wave would wait for
    // the next piece's loads before it sums this one.
    auto fetch = [&](const T* x, int len, int base, T (&v)[kBatch + 1]) {
        const int last = len > 0 ? len - 1 : 0;
#pragma unroll
        for (int q = 0; q <= kBatch; ++q) {
            const int i = base + 16 * q + j;
            v[q] = x[i < last ? i : last];
        }
    };
    // A block is walked in pieces of 160 samples; the loads of the next piece (of this block
    // or of the group's next block) are issued before this piece is summed.
    const bool keep = per_wave <= kKeep;
    int64_t b = b_first + grp;
    if (b_first >= b_end) return;             // (the whole wave: the groups meet again below)
    if (b >= b_end) b = b_end;
    const T* x = audio;
    int len = 0, base = 0;
    if (b < b_end) locate(b, x, len);
    T cur[kBatch + 1], nxt[kBatch + 1];
    fetch(x, len, 0, cur);
    double e = 0.0;
    int z = 0;
    while (b < b_end) {
        int64_t nb = b;
        int nbase = base + 16 * kBatch, nlen = len;
        const T* nx = x;
        if (nbase >= len) {
            nb = b + 4;
            nbase = 0;
            nlen = 0;
            if (nb < b_end) locate(nb, nx, nlen);
        }
        fetch(nx, nlen, nbase, nxt);
#pragma unroll
        for (int q = 0; q <= kBatch; ++q) {
            const int i = base + 16 * q + j;
            cur[q] = i < len && (q < kBatch || j == 0) ? cur[q] : (T)0;
        }
#pragma unroll
        for (int q = 0; q < kBatch; ++q) {
            const int i = base + 16 * q + j;
            const T nxt_same = rot1(cur[q]);
            const T nxt_round = rot1(cur[q + 1]);
            const T nv = j == 15 ? nxt_round : nxt_same;
            e += (double)fabs(cur[q]);
            z += (i + 1 < len) && ((cur[q] >= (T)0) != (nv >= (T)0));
        }
        if (nb != b) {
            e = row_sum(e);
            z = row_sum(z);
            if (j == 0) {
                if (keep) {
                    s_e[wv][b - b_first] = e;
                    s_z[wv][b - b_first] = z;
                } else {
                    energy[b] = e;
                    zcr[b] = z;
                }
            }
            e = 0.0;
            z = 0;
        }
        b = nb;
        base = nbase;
        len = nlen;
        x = nx;
#pragma unroll
        for (int q = 0; q <= kBatch; ++q) cur[q] = nxt[q];
    }
    if (keep) {
        // (only this wave wrote its rows: a wave-level fence is enough)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const int n = (int)(b_end - b_first);
        for (int i = threadIdx.x & 63; i < n; i += 64) {
            energy[b_first + i] = s_e[wv][i];
            zcr[b_first + i] = s_z[wv][i];
        }
    }
}

__global__ void endpoint_detect_kernel(const double* __restrict__ energy,
                                       const int32_t* __restrict__ zcr,
                                       const int64_t* __restrict__ bo, int64_t n_rec,
                                       srx_endpoint_cfg_t c, int block, int max_seg,
                                       int32_t* __restrict__ segs, int32_t* __restrict__ counts,
                                       double* __restrict__ thr) {
    // 16 lanes per recording: they load 16 blocks at a time and all walk the same
    // sequential search on values passed round by shuffles (no lane waits on memory
    // for every single block); lane 0 of the group writes.
    const int j = threadIdx.x & 15;
    const int64_t r = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    if (r >= n_rec) return;
    const int64_t b0 = bo[r];
    const int n = (int)(bo[r + 1] - b0);
    const double* e = energy + b0;
    const int32_t* z = zcr + b0;
    int32_t* out = segs + r * (int64_t)max_seg * 4;
    double* t = thr + r * 3;
    if (j == 0) {
        counts[r] = 0;
        t[0] = t[1] = t[2] = 0.0;
    }
    if (n <= 0) return;
    double imx = e[0], imn = e[0];
    for (int i = j; i < n; i += 16) {
        imx = fmax(imx, e[i]);
        imn = fmin(imn, e[i]);
    }
#pragma unroll
    for (int o = 8; o >= 1; o >>= 1) {
        imx = fmax(imx, __shfl_xor(imx, o, 16));
        imn = fmin(imn, __shfl_xor(imn, o, 16));
    }
    if (j == 0) t[0] = imx;                      // (reported when nothing is found)
    if (imx < c.noise_peak) return;
    const double itl = fmin(0.03 * (imx - imn) + imn, 4.0 * imn);
    const double itu = 5.0 * itl;
    // zero-crossing threshold from the blocks that begin in the first 100 ms (blocks
    // past the end of a short recording count as zero)
    const int first = (int)(0.1 * (double)c.sample_rate);
    const int n0 = (first + block - 1) / block;
    // (the group's lanes load the counts side by side: one round trip to memory per 16
    // blocks instead of one per block; the counts are small integers, their sum is exact)
    double mean = 0.0;
    for (int i0 = 0; i0 < n0; i0 += 16) mean += i0 + j < n0 && i0 + j < n ? (double)z[i0 + j] : 0.0;
    mean = row_sum(mean);
    mean = n0 > 0 ? mean / n0 : 0.0;
    double var = 0.0;
    for (int i0 = 0; i0 < n0; i0 += 16) {
        const double d = (i0 + j < n ? (double)z[i0 + j] : 0.0) - mean;
        var += i0 + j < n0 ? d * d : 0.0;
    }
    var = row_sum(var);
    var = n0 > 0 ? var / n0 : 0.0;
    const double izct = fmin((double)c.zcr_cap, mean + 2.0 * sqrt(var));
    // run search: `up` counts accepted runs above ITU, `cnt` accepted runs above ITL
    int up = 0, up_last = 0, cnt = 0;
    int u0 = -1, uc = 0, l0 = -1, ld = 0;
    auto accept = [&](int n1, int d) {
        const int n2 = n1 + d;
        // blocks [lo, hi) whose zero-crossing count is over the threshold: how many, the first
        // and the last -- 16 blocks per round trip, the group's share of the wave's ballot
        const int shift = (int)(threadIdx.x & 48);
        auto scan = [&](int lo, int hi, int& hits, int& first_hit, int& last_hit) {
            hits = 0;
            for (int i0 = lo; i0 < hi; i0 += 16) {
                const bool over = i0 + j < hi && (double)z[i0 + j] > izct;
                const unsigned m = (unsigned)(__ballot(over) >> shift) & 0xFFFFu;
                if (m) {
                    if (hits == 0) first_hit = i0 + __builtin_ctz(m);
                    last_hit = i0 + 31 - __builtin_clz(m);
                    hits += __builtin_popcount(m);
                }
            }
        };
        int hits, first_hit = n1, last_hit = n2, unused;
        scan(n1 - 25 > 0 ? n1 - 25 : 0, n1 + 1, hits, first_hit, unused);
        const int n1p = hits >= 3 ? first_hit : n1;
        scan(n2 + 1, n2 + 25 < n ? n2 + 25 : n, hits, unused, last_hit);
        const int n2p = hits >= 3 ? last_hit : n2;
        if (cnt < max_seg && j == 0) {
            out[4 * cnt + 0] = n1p - 10 > 0 ? n1p - 10 : 0;
            out[4 * cnt + 1] = n1;
            out[4 * cnt + 2] = n2;
            out[4 * cnt + 3] = n2p;
        }
        ++cnt;
    };
    for (int base = 0; base < n; base += 16) {
      const double mine = base + j < n ? e[base + j] : 0.0;
      const int lim = n - base < 16 ? n - base : 16;
      for (int k = 0; k < lim; ++k) {
        const int i = base + k;
        const double v = __shfl(mine, k, 16);
        if (v > itu) {
            if (u0 == -1) u0 = i;
            ++uc;
            if (i == n - 1) {
                ++up;
                up_last = u0;
                u0 = -1;
                uc = 0;
            }
        } else {
            if (uc > 5 || u0 == 0) {
                ++up;
                up_last = u0;
            }
            u0 = -1;
            uc = 0;
        }
        if (v > itl) {
            if (l0 == -1) l0 = i;
            ++ld;
            if (i == n - 1 && up > cnt && l0 <= up_last) {
                accept(l0, ld);
                l0 = -1;
                ld = 0;
            }
        } else {
            if (ld > 5 && up > cnt && l0 <= up_last) accept(l0, ld);
            l0 = -1;
            ld = 0;
        }
      }
    }
    if (j != 0) return;
    counts[r] = cnt;
    if (cnt > 0) {
        t[0] = itu;
        t[1] = itl;
        t[2] = izct;
    }
}

}  // namespace

extern "C" {

int64_t srx_endpoint_blocks(const srx_endpoint_cfg_t* c, int64_t n_samples) {
    if (!c || n_samples <= 0) return 0;
    const int64_t block = (int64_t)((double)c->sample_rate * c->block_ms / 1000.0);
    if (block < 1) return 0;
    return (n_samples + block - 1) / block;
}

int srx_endpoint_detect(srx_handle_t h, const void* audio, int dtype, const int64_t* sample_offsets,
                        int64_t n_recordings, int64_t n_samples, const srx_endpoint_cfg_t* cfg,
                        const int64_t* block_offsets, int64_t n_blocks, double* out_energy,
                        int32_t* out_zcr, int32_t* out_segments, int32_t* out_counts,
                        double* out_thresholds, int32_t max_segments, void* stream) {
    SRX_REQUIRE(h != nullptr && cfg != nullptr, "srx_endpoint_detect: NULL context or cfg");
    SRX_REQUIRE(dtype == SRX_F32 || dtype == SRX_F64, "srx_endpoint_detect: unknown dtype %d", dtype);
    SRX_REQUIRE(n_recordings >= 0 && n_samples >= 0 && n_blocks >= 0, "srx_endpoint_detect: negative size");
    SRX_REQUIRE(cfg->sample_rate > 0 && cfg->block_ms > 0, "srx_endpoint_detect: sample_rate and block_ms must be positive");
    const int64_t block = (int64_t)((double)cfg->sample_rate * cfg->block_ms / 1000.0);
    SRX_REQUIRE(block >= 1 && block < (1 << 24), "srx_endpoint_detect: block of %lld samples", (long long)block);
    SRX_REQUIRE(max_segments >= 1, "srx_endpoint_detect: max_segments must be >= 1");
    if (n_recordings == 0) return SRX_OK;
    SRX_REQUIRE(sample_offsets && block_offsets && out_segments && out_counts && out_thresholds,
                "srx_endpoint_detect: NULL buffer");
    SRX_REQUIRE(n_blocks == 0 || (audio && out_energy && out_zcr), "srx_endpoint_detect: NULL buffer");
    SRX_HIP(hipSetDevice(h->device));
    hipStream_t st = (hipStream_t)stream;
    if (n_blocks > 0) {
        int64_t grid = (n_blocks + 15) / 16;
        const int64_t cap = (int64_t)h->n_cu * 16;
        if (grid > cap) grid = cap;
        if (dtype == SRX_F64)
            hipLaunchKernelGGL(endpoint_blocks_kernel<double>, dim3((unsigned)grid), dim3(256), 0, st,
                               (const double*)audio, sample_offsets, block_offsets, n_recordings,
                               n_blocks, (int)block, out_energy, out_zcr);
        else
            hipLaunchKernelGGL(endpoint_blocks_kernel<float>, dim3((unsigned)grid), dim3(256), 0, st,
                               (const float*)audio, sample_offsets, block_offsets, n_recordings,
                               n_blocks, (int)block, out_energy, out_zcr);
    }
    const unsigned g2 = (unsigned)((n_recordings + 15) / 16);
    hipLaunchKernelGGL(endpoint_detect_kernel, dim3(g2), dim3(256), 0, st, out_energy, out_zcr,
                       block_offsets, n_recordings, *cfg, (int)block, (int)max_segments,
                       out_segments, out_counts, out_thresholds);
    SRX_HIP(hipGetLastError());
    return SRX_OK;
}

}  // extern "C"
