// Cepstra and log-mel rows of centred 2048-point frames (float32): 32 lanes per frame, the FFT in registers.
//
// A frame's 2048 real samples are packed as H = 1024 complex points z[n] and the H-point
// transform is split 32 x 32:
//   1. lane j of the frame's 32-lane group holds z[32 n1 + j], n1 < 32 -- the samples
//      64 n1 + 2j, +1, so each of the 32 loads of a group reads 256 contiguous bytes -- and does
//      a 32-point FFT over n1 in registers;
//   2. multiplies by the twiddles W_H^(j k1);
//   3. the group transposes through LDS ([k1][j], rows padded to 33) and lane k1 does the
//      32-point FFT over j: it ends up with Z[k1 + 32 m], m < 32;
//   4. the real-input spectrum is unpacked from Z[k], Z[H-k] -- a lane's own lower half and the
//      upper half of one partner lane, which is all that goes through LDS -- and squared into
//      pw[H+1] = |2 X|^2 in the frame's LDS tile (the factor 1/4, or 1/(4 N) for a spectrum
//      divided by N, is a power of two and sits in the mel weights);
//   5. mel energies: the filters' bin ranges, cut into quads of 4 bins (filter-major), are dealt
//      to the wave's 64 lanes in equal runs; a lane walks its run for both frames of the wave,
//      one 16-byte LDS read per frame and quad, and leaves one partial sum per (filter, lane)
//      in LDS; the partial sums of a filter (at most a few: it is wider than a run only at the
//      top of the scale) are added in a fixed order, floored, and 10 log10'd.
//   6. DCT-II rows of the UNCLAMPED log-mel row: lane (frame, coefficient, half of the filters)
//      sums its 64 filters from the row in LDS.
// A wave carries 2 frames at once, a workgroup of 8 waves 16, and a CU holds one workgroup
// (155 KB of LDS: 8.25 KB of transpose tile per frame).  The kernel writes the cepstra, the
// log-mel rows and each row's largest and smallest value.  The clamp to (utterance maximum -
// top_db) needs the maximum over the whole utterance: a second kernel finds it from the rows'
// maxima and redoes the DCT of exactly those rows whose minimum lies below the threshold (none
// in a recording without digital silence or a 80 dB range within one frame's spectrum).
//
// Rows live on the virtual sample axis of the generic kernel (mfcc.hip): utterance u starts at
// so[u] + u * hop there, frame t of it is row (so[u] + u * hop) / hop + t.  A small kernel first
// writes, for every row, where its frame starts and the utterance it belongs to.
#include "srx_internal.h"
#include "fft_regs.h"

#include <math.h>

#include <type_traits>

namespace {

constexpr int kWave = 64;
constexpr int kP = 32;                  // lanes per frame, complex points per lane
constexpr int kH = kP * kP, kN = 2 * kH;
constexpr int kFB = 16;                 // frames per workgroup step (2 per wave)
constexpr int kYS = kP * (kP + 1);      // float2 per frame in the transpose tile
constexpr int kPerFrame = 2 * kYS;      // floats
constexpr int kPwPad = kH + 4;          // power bins, padded to whole quads
constexpr int kMaxParts = 256;          // partial mel sums per frame, the spare one included
constexpr int kMelPad = 128;            // log-mel row in LDS (filters past n_mel: zeros)
constexpr int kDctStride = kMelPad + 4; // a DCT row in LDS (16 lanes read 16 rows: 16 groups of 4 banks)
constexpr int kMelBatch = 6;            // quads of the mel stage whose table entries are requested together
constexpr int kDctLdsRows = 15;         // rows that fit beside 16 frames in 160 KB

struct WideDev {
    int hop, n_mel, center;
    float log_mult, log_floor;
    const float* window;        // [2048] (the window sits in the middle of the transform)
    const float2* tw;           // [2048] exp(-2 pi i k / 2048)
    int lq;                     // quads per lane
    const float* w;             // [lq][64][4] filter weights of a quad
    const int32_t* meta;        // [lq][64] quad index | partial-sum slot << 16 | last of its filter << 31
    const int32_t* part_ptr;    // [n_mel + 1] partial sums of a filter
    int n_ceps;
    const float* dct;           // [n_ceps][128] DCT-II rows, zero past n_mel
    unsigned hop_m, hop_s;      // n / hop = umulhi(n, hop_m) >> hop_s for n < 2^31
};

// A 4-byte aligned pair of samples (one 8-byte load: the target allows unaligned access)
typedef float f2u __attribute__((ext_vector_type(2), aligned(4)));

// row_meta[row] = {first sample of the frame's 2048 points on the packed axis (it may lie before
// or reach behind its utterance), first sample of the utterance, its length, 1}; the table is
// zero-filled before (w = 0: no frame on this row).  One wave per utterance at a time.
__global__ __launch_bounds__(256) void row_meta_kernel(const int64_t* __restrict__ so, int64_t U,
                                                       unsigned hop, unsigned hop_m, unsigned hop_s,
                                                       int4* __restrict__ row_meta) {
    const int lane = threadIdx.x & 63;
    const int64_t n_waves = (int64_t)gridDim.x * (blockDim.x >> 6);
    for (int64_t u = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); u < U; u += n_waves) {
        const int64_t us = so[u];
        const unsigned n = (unsigned)(so[u + 1] - us);
        if (n == 0) continue;
        const unsigned T = (__umulhi(n, hop_m) >> hop_s) + 1u;
        const unsigned row0 = __umulhi((unsigned)us + (unsigned)u * hop, hop_m) >> hop_s;
        for (unsigned t = lane; t < T; t += 64)
            row_meta[row0 + t] = int4{(int)((unsigned)us + t * hop) - kH, (int)us, (int)n, 1};
    }
}

// CENTER: the padding rule (1 zeros, 2 reflection); PARTS: lanes that share a DCT coefficient
// (2 up to 16 cepstra, else 1) -- compile-time constants.
// (Plain (re, im) registers measured faster than packed pairs here: 1.86 against 1.94 ms per
// 770,048 frames, whole front-end pass.)
template <int CENTER, int PARTS>
__global__ __launch_bounds__(32 * kFB, 2) void cepstra_wide_kernel(
    const float* __restrict__ audio, const int4* __restrict__ row_meta, int64_t n_rows, WideDev c,
    float* __restrict__ logmel, float* __restrict__ rowmax, float* __restrict__ rowmin,
    float* __restrict__ cep) {
    typedef cf cpx;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    v2f* s_tw = (v2f*)smem_raw;                       // [H/2]: exp(-2 pi i k / N)
    float* s_win = (float*)(s_tw + kH / 2);           // [N]
    v2f* s_tw2 = (v2f*)(s_win + kN);                  // [32][32]: W_H^(j k1)
    float* s_frames = (float*)(s_tw2 + kP * kP);
    // the DCT rows, when they fit what the frames leave of the CU's LDS (n_ceps <= 15; from
    // global memory otherwise)
    float* s_dct = s_frames + (size_t)kFB * kPerFrame;
    const bool dct_lds = c.n_ceps <= kDctLdsRows;
    if (dct_lds)
        for (int i = threadIdx.x; i < c.n_ceps * kMelPad; i += blockDim.x)
            s_dct[(i / kMelPad) * kDctStride + i % kMelPad] = c.dct[i];
    for (int i = threadIdx.x; i < kH / 2; i += blockDim.x) s_tw[i] = v2f{c.tw[i].x, c.tw[i].y};
    for (int i = threadIdx.x; i < kN; i += blockDim.x) s_win[i] = c.window[i];
    for (int i = threadIdx.x; i < kP * kP; i += blockDim.x) {
        const float2 w = c.tw[2 * (((i & 31) * (i >> 5)) & (kH - 1))];
        s_tw2[i] = v2f{w.x, w.y};
    }
    __syncthreads();

    const int wave = threadIdx.x / kWave, lane = threadIdx.x % kWave;
    const int fs = lane >> 5, j = lane & 31;          // frame slot in the wave, lane in the group
    const int item = 2 * wave + fs;                   // frame slot in the workgroup
    // A frame's tile holds the transpose buffer first, then the upper halves of the spectrum;
    // once those have been read, the same floats hold pw [H+4] and the partial mel sums.
    float* fr = s_frames + (size_t)item * kPerFrame;
    v2f* Y = (v2f*)fr;
    float* pw = fr;
    float* pwA = s_frames + (size_t)(2 * wave) * kPerFrame;     // the wave's two frames
    float* pwB = pwA + kPerFrame;

    // A workgroup walks a CONTIGUOUS range of steps of kFB rows each.  A row's record is read
    // two steps ahead and the samples of step b+1 are requested before step b is transformed,
    // so neither load is waited for where it is issued.
    const int64_t n_chunks = (n_rows + kFB - 1) / kFB;
    const int64_t per_block = (n_chunks + gridDim.x - 1) / gridDim.x;
    const int64_t b_begin = (int64_t)blockIdx.x * per_block;
    const int64_t b_end = b_begin + per_block < n_chunks ? b_begin + per_block : n_chunks;
    if (b_begin >= b_end) return;
    auto row_of = [&](int64_t bq) { return bq * kFB + item; };
    auto load_meta = [&](int64_t bq) -> int4 {
        const int64_t r = row_of(bq);
        const int4 v = row_meta[r < n_rows ? r : n_rows - 1];
        return r < n_rows ? v : int4{0, 0, 0, 0};
    };
    // The samples of one frame.  A frame that lies inside its utterance is 32 contiguous loads
    // per lane, requested a step ahead; any other group (no frame, or a frame that reaches over
    // an end of its utterance) requests the buffer's first 2048 samples instead, so that the
    // requests are the same instructions in every step and land in the registers they are
    // used from.
    auto interior_of = [&](const int4& m) {
        const int rel = m.x - m.y;                  // first point, relative to the utterance
        return m.w != 0 && rel >= 0 && rel + kN <= m.z;
    };
    auto fetch = [&](const int4& m, float (&xr)[2 * kP]) {
        const float* p = audio + (interior_of(m) ? (int64_t)m.x : 0) + 2 * j;
#pragma unroll
        for (int n1 = 0; n1 < kP; ++n1) {
            const f2u v = *(const f2u*)(p + 64 * n1);
            xr[2 * n1] = v.x;
            xr[2 * n1 + 1] = v.y;
        }
    };
    // A frame that reaches over an end (the first and last two or three of an utterance) takes
    // every sample from where the padding rule puts it -- reflection about the first and the
    // last sample, then a clamp for frames that reach further than one utterance length; zeros
    // outside otherwise -- at the start of its own step; the wave's other frame comes along.
    // A group without a frame reads sample 0.
    auto gather = [&](const int4& m, float (&xr)[2 * kP]) {
        const int rel = m.x - m.y;
        const int n = m.z;
        const float* base = audio + m.y;
        if constexpr (CENTER == 2) {
#pragma unroll
            for (int i = 0; i < 2 * kP; ++i) {
                int q = rel + 64 * (i >> 1) + 2 * j + (i & 1);
                q = q < 0 ? -q : q;
                const int r = 2 * (n - 1) - q;
                q = r < q ? r : q;
                q = q < 0 ? 0 : q;
                xr[i] = base[q];
                // (eight at a time: the addresses of all 64 do not fit the registers)
                if ((i & 7) == 7) __builtin_amdgcn_sched_barrier(0);
            }
        } else {
#pragma unroll
            for (int i = 0; i < 2 * kP; ++i) {
                const int q = rel + 64 * (i >> 1) + 2 * j + (i & 1);
                const bool in = (unsigned)q < (unsigned)n;
                const float v = base[in ? q : 0];
                xr[i] = in ? v : 0.f;
                if ((i & 7) == 7) __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    // the lane's two filters (lane, lane + 64): first and end of their partial sums, a byte each
    unsigned ppk = 0;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        const int f = lane + 64 * half;
        if (f < c.n_mel) ppk |= ((unsigned)c.part_ptr[f] | ((unsigned)c.part_ptr[f + 1] << 8)) << (16 * half);
    }
    // kMelBatch steps of the lane's run of quads from s0 (lq is a multiple of two batches).  The
    // addresses are the tables' (uniform) bases plus a 32-bit byte offset per lane plus a constant.
    const unsigned lane_m = (unsigned)lane * 4u, lane_w = (unsigned)lane * 16u;
    auto mel_batch = [&](int s0, int32_t (&mt)[kMelBatch], f32x4 (&w)[kMelBatch]) {
        const unsigned om = lane_m + (unsigned)s0 * 256u, ow = lane_w + (unsigned)s0 * 1024u;
#pragma unroll
        for (int u = 0; u < kMelBatch; ++u) {
            mt[u] = *(const int32_t*)((const char*)c.meta + (size_t)om + u * 256);
            w[u] = *(const f32x4*)((const char*)c.w + (size_t)ow + u * 1024);
        }
    };
    // log_mult * log10(2) as a float and the part of it the float misses
    const double scale_d = (double)c.log_mult * 0.30102999566398120;
    const float log2_scale = (float)scale_d, log2_scale_lo = (float)(scale_d - (double)log2_scale);

    // First register stage of a step: the frames whose samples are in xc (record m) are windowed
    // into z1 (bit-reversed order) and transformed over n1.  The samples are dead once the
    // window has been applied: the step after's are requested into the same registers right
    // there, a whole step before they are used.
    auto head_window = [&](float (&xc)[2 * kP], cpx (&z1)[kP]) {
#pragma unroll
        for (int n1 = 0; n1 < kP; ++n1) {
            const float2 wv = *(const float2*)(s_win + 64 * n1 + 2 * j);
            z1[brev(n1, 5)] = cpx{xc[2 * n1] * wv.x, xc[2 * n1 + 1] * wv.y};
        }
    };
    auto head = [&](const int4& m, float (&xc)[2 * kP], const int4& m_next, cpx (&z1)[kP]) {
        if (__ballot(m.w != 0 && !interior_of(m)) != 0) gather(m, xc);
        head_window(xc, z1);
        fetch(m_next, xc);
        fft_regs<kP>(z1);
    };
    // The rest of a step (record m, row `row`, first stage in z1); at its end the first register
    // stage of the NEXT step (records m_n, then m_nn): the compiler may then start that stage's
    // arithmetic under the last LDS reads of this step.  (Interleaving the DCT's reads with the
    // next transform's stages by hand needs ~26 registers more than there are: 2.5 ms with the
    // spills against 1.73.)
    auto transform = [&](const int4& m, int64_t row, cpx (&z1)[kP], float (&xc)[2 * kP], const int4& m_n,
                         const int4& m_nn) {
        const bool have = m.w != 0;
        // (issue priority rising along the step, as in the 512 / 1024-point kernel: the wave
        // furthest into its step goes first; 1.725 -> 1.684 ms per pass on one box)
        __builtin_amdgcn_s_setprio(2);
        // ---- 2. twiddle W_H^(j k1) and 3. transpose through LDS: Y[k1][j]
#pragma unroll
        for (int k1 = 0; k1 < kP; ++k1) Y[k1 * (kP + 1) + j] = twiddled(s_tw2[k1 * kP + j], z1[k1], k1 == 0);
        wave_sync();
        cpx z2[kP];
#pragma unroll
        for (int n2 = 0; n2 < kP; ++n2) {
            const v2f y = Y[j * (kP + 1) + n2];
            z2[brev(n2, 5)] = cpx{y.x, y.y};
        }
        wave_sync();
        fft_regs<kP>(z2);
        __builtin_amdgcn_s_setprio(3);
        // ---- 4. real-input spectrum, power.  Bins k and H-k share their inputs.  Lane j now
        // holds Z[j + 32 m], m < 32.  Its bins k = j + 32 mm, mm < 16, are its own lower half;
        // their partners H - k = (32 - j) + 32 (31 - mm) are the UPPER half of lane
        // (32 - j) mod 32 -- for j = 0 (H - k = 32 (32 - mm)) one place further, Z[0] for
        // mm = 0.  So only half of the spectrum travels: every lane leaves its upper half in LDS
        // (slot u = m - 16; lanes 36 floats apart) and reads its partner's.
        constexpr int LT = kP + 4;
        float* slots = fr;
#pragma unroll
        for (int u = 0; u < kP / 2; u += 2) {
            const cpx a = z2[u + kP / 2], b = z2[u + 1 + kP / 2];
            *(f32x4*)(slots + j * LT + 2 * u) = f32x4{a.x, a.y, b.x, b.y};
        }
        wave_sync();
        cpx zcs[kP / 2];
        {
            const int jp = (kP - j) & (kP - 1);
            const float* sp = slots + jp * LT + (j == 0 ? 2 : 0);
#pragma unroll
            for (int mm = 0; mm < kP / 2; ++mm) {
                const v2f t = *(const v2f*)(sp + 2 * (kP / 2 - 1 - mm));
                zcs[mm] = cpx{t.x, t.y};
            }
            if (j == 0) zcs[0] = z2[0];
        }
        const cpx zh = z2[kP / 2];                // Z[H/2], lane 0's (the only lane that uses it)
        wave_sync();                              // the slots are in registers: pw may take their place
#pragma unroll
        for (int mm = 0; mm < kP / 2; ++mm) {
            const int k = j + kP * mm;
            power_pair(z2[mm], zcs[mm], s_tw[k], 1.f, pw[k], pw[kH - k]);
        }
        int32_t mt0[kMelBatch];
        f32x4 w0[kMelBatch];
        mel_batch(0, mt0, w0);                    // (the mel stage's first table entries)
        if (j == 0) pw[kH / 2] = (zh.x * zh.x + zh.y * zh.y) * 4.f;
        if (j < 3) pw[kH + 1 + j] = 0.f;          // padding that meets zero weights: keep it finite
        wave_sync();
        // ---- 5. mel energies: the lane's run of quads, for both frames of the wave
        float* partA = pwA + kPwPad;
        float* partB = pwB + kPwPad;
        float* lmA = partA + kMaxParts;
        float* lmB = partB + kMaxParts;
        {
            // (the tables come from global memory -- LDS is full -- six quads at a time and one
            // such batch ahead)
            float accA = 0.f, accB = 0.f;
            auto walk = [&](const int32_t (&mt)[kMelBatch], const f32x4 (&w)[kMelBatch]) {
#pragma unroll
                for (int u = 0; u < kMelBatch; ++u) {
                    const int q = (mt[u] & 0x3FF) * 4, slot = (mt[u] >> 16) & 0x3FF;
                    const f32x4 a = *(const f32x4*)(pwA + q), b = *(const f32x4*)(pwB + q);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        accA = fmaf(w[u][e], a[e], accA);
                        accB = fmaf(w[u][e], b[e], accB);
                    }
                    partA[slot] = accA;
                    partB[slot] = accB;
                    if (mt[u] < 0) accA = accB = 0.f;
                }
            };
            int32_t mt1[kMelBatch];
            f32x4 w1[kMelBatch];
            for (int s0 = 0; s0 < c.lq; s0 += 2 * kMelBatch) {
                mel_batch(s0 + kMelBatch, mt1, w1);
                walk(mt0, w0);
                if (s0 + 2 * kMelBatch < c.lq) mel_batch(s0 + 2 * kMelBatch, mt0, w0);
                walk(mt1, w1);
            }
        }
        wave_sync();
        // the partial sums of a filter add up; floor; log_mult * log10 on the base-2 logarithm
        // instruction (values below 2^-96, denormals among them, which the instruction does not
        // take, are scaled up first)
        {
            const int rowi = have ? (int)row : -1;
            const int rA = __builtin_amdgcn_readlane(rowi, 0), rB = __builtin_amdgcn_readlane(rowi, 32);
            float mxA = -INFINITY, mxB = -INFINITY, mnA = INFINITY, mnB = INFINITY;
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int f = lane + 64 * half;
                if (f < c.n_mel) {
                    const int p0 = (ppk >> (16 * half)) & 0xFF, p1 = (ppk >> (16 * half + 8)) & 0xFF;
                    float eA = partA[p0], eB = partB[p0];
                    for (int p = p0 + 1; p < p1; ++p) {
                        eA += partA[p];
                        eB += partB[p];
                    }
                    auto db = [&](float e) -> float {
                        float v;
                        if (c.log_floor > 0.f) v = e < c.log_floor ? c.log_floor : e;
                        else v = e == 0.f ? 2.220446049250313e-16f : e;
                        const bool tiny = v < 0x1p-96f;
                        const float l2 = __builtin_amdgcn_logf(tiny ? v * 0x1p32f : v) - (tiny ? 32.f : 0.f);
                        return fmaf(l2, log2_scale, l2 * log2_scale_lo);
                    };
                    const float lA = db(eA), lB = db(eB);
                    if (logmel) {       // (kept for the rows that the clamp changes)
                        if (rA >= 0) logmel[(int64_t)rA * c.n_mel + f] = lA;
                        if (rB >= 0) logmel[(int64_t)rB * c.n_mel + f] = lB;
                    }
                    lmA[f] = lA;
                    lmB[f] = lB;
                    mxA = lA > mxA ? lA : mxA;
                    mxB = lB > mxB ? lB : mxB;
                    mnA = lA < mnA ? lA : mnA;
                    mnB = lB < mnB ? lB : mnB;
                } else {
                    lmA[f] = 0.f;
                    lmB[f] = 0.f;
                }
            }
            // largest and smallest value of each row: lanes 0..31 take frame A's, lanes 32..63
            // frame B's (one swap of the halves), then five steps inside the 32 lanes
            {
                const unsigned smx = __float_as_uint(fs ? mxA : mxB), smn = __float_as_uint(fs ? mnA : mnB);
                const auto rx = __builtin_amdgcn_permlane32_swap(smx, smx, false, false);
                const auto rn = __builtin_amdgcn_permlane32_swap(smn, smn, false, false);
                // (r[0]: the lower half's value in both halves, r[1]: the upper half's)
                float mx = fmaxf(fs ? mxB : mxA, __uint_as_float(fs ? rx[0] : rx[1]));
                float mn = fminf(fs ? mnB : mnA, __uint_as_float(fs ? rn[0] : rn[1]));
                auto dpp = [](float v, auto ctrl) {
                    const int u = (int)__float_as_uint(v);
                    return __uint_as_float((unsigned)__builtin_amdgcn_update_dpp(u, u, decltype(ctrl)::value, 0xf, 0xf, false));
                };
                auto step = [&](auto ctrl) {
                    mx = fmaxf(mx, dpp(mx, ctrl));
                    mn = fminf(mn, dpp(mn, ctrl));
                };
                step(std::integral_constant<int, 0xB1>{});      // lanes ^ 1
                step(std::integral_constant<int, 0x4E>{});      // lanes ^ 2
                step(std::integral_constant<int, 0x141>{});     // mirror within 8
                step(std::integral_constant<int, 0x140>{});     // mirror within 16
                const unsigned ux = __float_as_uint(mx), un = __float_as_uint(mn);
                const auto qx = __builtin_amdgcn_permlane16_swap(ux, ux, false, false);
                const auto qn = __builtin_amdgcn_permlane16_swap(un, un, false, false);
                mx = fmaxf(__uint_as_float(qx[0]), __uint_as_float(qx[1]));
                mn = fminf(__uint_as_float(qn[0]), __uint_as_float(qn[1]));
                if (j == 0 && have) {
                    rowmax[row] = mx;
                    rowmin[row] = mn;
                }
            }
        }
        wave_sync();
        // ---- 6. DCT-II of the row as it is: lane (frame fs, coefficient k, part) sums every
        // filter of its part of the row (all lanes of a frame read the same LDS words)
        {
            constexpr int cnt = kMelPad / PARTS;
            const int k = j / PARTS, part = j % PARTS;
            const float* lm = (fs ? lmB : lmA) + part * cnt;
            const float* dr = (dct_lds ? s_dct + (k < c.n_ceps ? k : 0) * kDctStride
                                       : c.dct + (size_t)(k < c.n_ceps ? k : 0) * kMelPad) + part * cnt;
            float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
            // (32 filters at a time: 8 + 8 LDS reads, then 32 multiply-adds.  The scheduling
            // barriers keep the compiler from requesting everything first, which would not fit
            // the registers next to z1 and the samples.)
            f32x4 dv[8], lv[8];
            auto request = [&](int q) {
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    dv[u] = *(const f32x4*)(dr + q + 4 * u);
                    lv[u] = *(const f32x4*)(lm + q + 4 * u);
                }
            };
            auto sum_up = [&]() {
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    a0 = fmaf(dv[u][0], lv[u][0], a0);
                    a1 = fmaf(dv[u][1], lv[u][1], a1);
                    a2 = fmaf(dv[u][2], lv[u][2], a2);
                    a3 = fmaf(dv[u][3], lv[u][3], a3);
                }
            };
#pragma unroll 1
            for (int q = 0; q < cnt; q += 32) {
                request(q);
                sum_up();
            }
            float acc = (a0 + a1) + (a2 + a3);
            if (PARTS == 2) acc += __shfl_xor(acc, 1, kWave);
            if (have && k < c.n_ceps && part == 0) cep[row * c.n_ceps + k] = acc;
        }
        __builtin_amdgcn_s_setprio(0);
        head(m_n, xc, m_nn, z1);
        wave_sync();
    };
    float x[2 * kP];
#pragma unroll
    for (int i = 0; i < 2 * kP; ++i) x[i] = 0.f;
    int4 m0 = load_meta(b_begin), m1 = load_meta(b_begin + 1);
    fetch(m0, x);
    cf z1[kP];
    head(m0, x, b_begin + 1 < b_end ? m1 : m0, z1);
    for (int64_t b = b_begin; b < b_end; ++b) {
        // (past the range's end the last step is repeated: harmless, nothing of it is stored)
        const int4 m2 = b + 2 < b_end ? load_meta(b + 2) : (b + 1 < b_end ? m1 : m0);
        transform(m0, row_of(b), z1, x, b + 1 < b_end ? m1 : m0, m2);
        m0 = m1;
        m1 = m2;
    }
}

}  // namespace

int srx_mfcc_wide_cepstra(srx_context* h, srx_mfcc_plan* p, const float* audio, const int64_t* so,
                          int64_t U, int64_t total, float* logmel, float* cep, float** rowmax_out,
                          float** rowmin_out, hipStream_t st) {
    const srx_mfcc_cfg_t& c = p->cfg;
    *rowmax_out = *rowmin_out = nullptr;
    // (total >= 2048: the kernel requests the buffer's first 2048 samples where a step has no
    // frame that lies inside its utterance)
    if (!p->wd_ok || c.nfft != kN || !c.center || c.hop < 2 || total < kN) return SRX_EUNSUPPORTED;
    const int64_t v_total = total + U * (int64_t)c.hop;
    if (v_total >= 0x7FFFF000ll) return SRX_EUNSUPPORTED;
    WideDev d;
    d.hop = c.hop; d.n_mel = c.n_mel; d.center = c.center;
    d.log_mult = (float)c.log_mult;
    d.log_floor = (float)c.log_floor;
    d.window = (const float*)p->window[0];
    d.tw = (const float2*)p->tw_full;
    d.lq = p->wd_lq;
    d.w = (const float*)p->wd_w; d.meta = p->wd_meta; d.part_ptr = p->wd_ptr;
    d.n_ceps = c.n_ceps; d.dct = (const float*)p->wd_dct;
    {
        // l = ceil(log2 hop), m = ceil(2^(31+l) / hop) < 2^32: floor(n / hop) = (n m) >> (31 + l)
        int l = 0;
        while ((1ll << l) < c.hop) ++l;
        const unsigned __int128 one = 1;
        d.hop_m = (unsigned)(((one << (31 + l)) + c.hop - 1) / (unsigned)c.hop);
        d.hop_s = (unsigned)(l - 1);
    }
    const size_t lds = sizeof(float2) * (kH / 2) + 4 * (size_t)kN + sizeof(float2) * kP * kP +
                       4 * (size_t)kPerFrame * kFB + (c.n_ceps <= kDctLdsRows ? 4 * (size_t)c.n_ceps * kDctStride : 0);
    static_assert(kPwPad + kMaxParts + kMelPad <= kPerFrame,
                  "the partial mel sums and the log-mel row must fit the frame's tile");
    const bool two = c.n_ceps <= 16;
    auto kern = c.center == 2 ? (two ? cepstra_wide_kernel<2, 2> : cepstra_wide_kernel<2, 1>)
                              : (two ? cepstra_wide_kernel<1, 2> : cepstra_wide_kernel<1, 1>);
    SRX_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int64_t n_rows = v_total / c.hop + 1;
    // [n_rows] records, then [n_rows] row maxima and [n_rows] row minima
    void* misc;
    int rc = srx_ws_ensure(h, WS_MISC, (sizeof(int4) + 2 * sizeof(float)) * (size_t)n_rows, &misc);
    if (rc) return rc;
    int4* row_meta = (int4*)misc;
    float* rowmax = (float*)(row_meta + n_rows);
    float* rowmin = rowmax + n_rows;
    SRX_HIP(hipMemsetAsync(row_meta, 0, sizeof(int4) * (size_t)n_rows, st));
    {
        int64_t g = (U + 3) / 4;
        const int64_t cap = (int64_t)h->n_cu * 8;
        if (g > cap) g = cap;
        hipLaunchKernelGGL(row_meta_kernel, dim3((unsigned)g), dim3(256), 0, st, so, U, (unsigned)c.hop,
                           d.hop_m, d.hop_s, row_meta);
    }
    const int64_t n_chunks = (n_rows + kFB - 1) / kFB;
    int64_t grid = n_chunks < (int64_t)h->n_cu ? n_chunks : (int64_t)h->n_cu;
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(32 * kFB), lds, st, audio, (const int4*)row_meta,
                       n_rows, d, logmel, rowmax, rowmin, cep);
    SRX_HIP(hipGetLastError());
    *rowmax_out = rowmax;
    *rowmin_out = rowmin;
    return SRX_OK;
}
