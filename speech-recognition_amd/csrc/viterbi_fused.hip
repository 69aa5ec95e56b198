// Fused Viterbi decoding for models of at most 8 states, float32 (utils.py:243-266 of the
// reference with the emission of hmm.py:154-163 in front of it): the log-densities never
// go through global memory.
//
// A wave owns 16 utterances and walks them in chunks of 16 frames:
//   emission   one utterance at a time, its 16 frames as the columns of the matrix
//              instruction, the 8 state rows of the utterance's model as rows (split-bf16
//              products, the tables of score_split.h; the model tables of all models sit in
//              LDS), log-densities into the wave's LDS tile [utterance][frame][state];
//   recursion  four lanes per utterance (lane v + 16 r: states 2 r and 2 r + 1) run the 16
//              max-plus steps on registers: log A columns of the lane's two states, delta
//              exchanged between the four rows of the wave by v_permlane16_swap and
//              v_permlane32_swap, back-pointers of a frame packed 3 bits per state into one
//              word.  (16 rather than 32 utterances per wave: the LDS tile is what limits the
//              number of resident waves, and the emission needs three per SIMD to hide its
//              dependent matrix instructions and LDS reads.)
// The back-pointer words are the only intermediate that goes to memory (4 bytes per
// frame); lane v resolves them backwards at the end, sixteen frames per access.  Ties take the
// lowest state, as np.argmax does; delta is kept relative to its largest entry every four
// frames (the shift is carried in float64 to the log-probability).  The arithmetic is that of
// emission_utt_split_kernel + viterbi_lane_kernel term for term: same states, same
// log-probabilities, bit for bit.
#include "score_split.h"

namespace {

constexpr int kTcap = 256;                 // frames whose back-pointer words a wave of 8 utterances keeps in LDS
constexpr int kWordStride = kTcap + 4;      // words per utterance there (+4: bank spread)
constexpr int kTileStride = 16 * 8 + 4;     // floats per utterance in the tile (+4: bank spread)

__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// {value of lane l & 31, value of lane 32 + (l & 31)} in every lane
__device__ __forceinline__ void halves(float v, float& lo, float& hi) {
    const unsigned u = __float_as_uint(v);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    lo = __uint_as_float(r[0]);
    hi = __uint_as_float(r[1]);
}
// {value of the even row, value of the odd row} of this lane's pair of 16-lane rows
__device__ __forceinline__ void row_pair(float v, float& even, float& odd) {
    const unsigned u = __float_as_uint(v);
    const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    even = __uint_as_float(r[0]);
    odd = __uint_as_float(r[1]);
}
// the four rows' values of a lane column, in row order
__device__ __forceinline__ void rows4(float v, float (&o)[4]) {
    float e, d;
    row_pair(v, e, d);
    halves(e, o[0], o[2]);
    halves(d, o[1], o[3]);
}
__device__ __forceinline__ float rows_max(float v) {
    float a, b;
    row_pair(v, a, b);
    v = fmaxf(a, b);
    halves(v, a, b);
    return fmaxf(a, b);
}

template <int S, int RQ, int H, int UPW>
__global__ __launch_bounds__(1024) void viterbi_fused_kernel(
    const float* __restrict__ X, const int64_t* __restrict__ off, int64_t U,
    const int32_t* __restrict__ utt_model, int M, int D, SplitTables tb,
    const float* __restrict__ logA, const float* __restrict__ logpi, uint32_t* __restrict__ bpw,
    int32_t* __restrict__ path, double* __restrict__ out_lp) {
    static_assert(S >= 1 && S <= 8, "four lanes of two states each");
    static_assert(UPW == 16 || UPW == 8, "utterances per wave");
    // UPW = 8: half of the recursion's lanes idle, but the tile is half as large and the
    // back-pointer words of utterances of up to kTcap frames fit beside it in LDS -- no store
    // goes out between the loads of the feature stream (what even 4 bytes per frame of them
    // cost: tools/probes/chunk_stream.hip) and the backtrace reads no memory.
    constexpr int kWaveFloats = UPW * kTileStride + (UPW == 8 ? UPW * kWordStride : 0);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // The feature axis is cut as in the scoring kernel: RQ blocks of 16 features and, if H, a
    // last block of 8 whose operand carries hi and lo halves at once (a row of 33..40 features is
    // then read in exactly 160 bytes, not 192).
    // model tables: [m][q < RQ + H][part][g][row < 8][4 dwords] (a 16-lane read group then
    // meets 16 different 4-bank pieces), constants [m][8]
    constexpr int QT = RQ + H;
    struct item_t {
        u32x4 v[RQ ? RQ : 1];
        u32x2 h;
    };
    unsigned* s_w = (unsigned*)smem;
    const int w_dwords = M * QT * 2 * 4 * 8 * 4;
    float* s_cst = (float*)(s_w + w_dwords);
    float* s_tiles = s_cst + ((M * 8 + 3) & ~3);
    for (int i = threadIdx.x; i < w_dwords; i += blockDim.x) {
        const int dw = i & 3, row = (i >> 2) & 7, g = (i >> 5) & 3, part = (i >> 7) & 1;
        const int q = (i >> 8) % QT, m = (i >> 8) / QT;
        s_w[i] = tb.Wp[(((((size_t)(m * 16 + row)) * QT + q) * 2 + part) * 4 + g) * 4 + dw];
    }
    for (int i = threadIdx.x; i < M * 8; i += blockDim.x) s_cst[i] = tb.cst[(i >> 3) * 16 + (i & 7)];
    __syncthreads();

    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int g = lane >> 4, ul = lane & 15;         // emission: lane group, frame of the chunk
    const int h = lane >> 4, v = lane & 15;          // recursion: pair of states, utterance
    float* tile = s_tiles + wave * kWaveFloats;
    uint32_t* wl = (uint32_t*)(tile + UPW * kTileStride);     // [UPW][kWordStride] (UPW = 8)
    // groups of UPW utterances are dealt to the waves of the grid in turn
    const int64_t n_groups = (U + UPW - 1) / UPW;
    const int64_t n_waves = (int64_t)gridDim.x * (blockDim.x >> 6);
    for (int64_t grp = (int64_t)blockIdx.x * (blockDim.x >> 6) + wave; grp < n_groups; grp += n_waves) {
    const int64_t u = grp * UPW + v;
    const bool valid = v < UPW && u < U;
    const int64_t f0 = valid ? off[u] : 0;
    const int T = valid ? (int)(off[u + 1] - f0) : 0;
    const int m = valid && utt_model ? utt_model[u] : 0;
    int Tmax = T;
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) Tmax = max(Tmax, __shfl_xor(Tmax, o, 64));
    if (Tmax <= 0) {
        if (valid && h == 0) out_lp[u] = 0.0;
        continue;
    }
    const int n_chunks = (Tmax + 15) >> 4;
    const int n_items = n_chunks * UPW;
    const bool words_in_lds = UPW == 8 && Tmax <= kTcap;      // (the same for the whole wave)

    // log A columns of this lane's states (state 2 h + jj), -inf for states the model lacks
    float la[S][2], d[2];
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
        const int j = 2 * h + jj;
#pragma unroll
        for (int i = 0; i < S; ++i) la[i][jj] = j < S ? logA[((size_t)m * S + i) * S + j] : -INFINITY;
        d[jj] = -INFINITY;
    }
    const unsigned f0_lo = (unsigned)f0, f0_hi = (unsigned)((uint64_t)f0 >> 32);

    // ---- emission of item i = (chunk, utterance): loads, then products
    const unsigned lane_off = (unsigned)ul * (unsigned)D * 4u + 16u * g;
    const unsigned half_off = (unsigned)ul * (unsigned)D * 4u + 64u * RQ + 8u * g;
    auto issue = [&](int i, item_t& dst) {
        const int k = i / UPW, e = i % UPW;
        const int Te = __builtin_amdgcn_readlane(T, e);
        const uint64_t fe = ((uint64_t)(unsigned)__builtin_amdgcn_readlane((int)f0_hi, e) << 32) |
                            (unsigned)__builtin_amdgcn_readlane((int)f0_lo, e);
        const bool on = i < n_items && 16 * k < Te;
        // descriptor over [first frame of the chunk, end of the utterance): rows past it read zeros
        const int64_t rest = on ? (int64_t)(Te - 16 * k) * D * 4 : 0;
        const uintptr_t a = (uintptr_t)(X + (on ? ((int64_t)fe + 16 * k) * D : 0));
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a);
        const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
        const unsigned n = (unsigned)(rest < 0xFFFFFF00ll ? rest : 0xFFFFFF00ll);
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(((uintptr_t)hi << 32) | lo), 0, __builtin_amdgcn_readfirstlane(n), 0x00020000);
#pragma unroll
        for (int q = 0; q < RQ; ++q)
            dst.v[q] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane_off + 64u * q, 0, 0);
        if constexpr (H != 0) dst.h = __builtin_amdgcn_raw_buffer_load_b64(rsrc, half_off, 0, 0);
    };
    auto emit = [&](int k, int e, const item_t& x) {
        // (an utterance that has ended is skipped: in a ragged group only the recursion's
        // steps run on to the longest utterance's end)
        if (16 * k >= __builtin_amdgcn_readlane(T, e)) return;
        const int me = __builtin_amdgcn_readlane(m, e);
        // (rows 8..15 of the product are never stored: their lanes read rows 0..7 again)
        const unsigned* wp = s_w + (size_t)me * (QT * 256) + g * 32 + (ul & 7) * 4;
        // (one accumulator per kind of product: three chains of RQ dependent instructions
        // instead of one of 3 RQ)
        f32x4 acc = *(const f32x4*)(s_cst + me * 8 + 4 * (g & 1));
        f32x4 acc1 = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < RQ; ++q) {
            const u32x4 wh = *(const u32x4*)(wp + q * 256), wl = *(const u32x4*)(wp + q * 256 + 128);
            bf16x8 bh, bl;
            split_operand(x.v[q], bh, bl);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wh), bh, acc, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wh), bl, acc1, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wl), bh, acc2, 0, 0, 0);
        }
        if constexpr (H != 0) {
            // last block of 8 features, two per lane: hi halves in slots 0, 1 (x) and 4, 5 (x^2),
            // lo halves in slots 2, 3 and 6, 7 (the table repeats the hi weights there)
            const u32x4 wh = *(const u32x4*)(wp + RQ * 256), wl = *(const u32x4*)(wp + RQ * 256 + 128);
            const bf16x8 bb = split_half_operand(x.h);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wh), bb, acc1, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wl), bb, acc2, 0, 0, 0);
        }
        acc += acc1 + acc2;
        if (g < 2) *(f32x4*)(tile + e * kTileStride + ul * 8 + g * 4) = acc * (float)kLn2;
    };

    // ---- one max-plus step of this lane's two states; returns the frame's back-pointer word
    auto step = [&](const float (&row)[2], bool active) -> uint32_t {
        float da[8];
        {
            float a[4], b[4];
            rows4(d[0], a);
            rows4(d[1], b);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                da[2 * r] = a[r];
                da[2 * r + 1] = b[r];
            }
        }
        uint32_t bits = 0;
        float nd[2];
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            float best = -INFINITY;
            uint32_t arg = 0;
#pragma unroll
            for (int i = 0; i < S; ++i) {
                const float cand = da[i] + la[i][jj];
                arg = cand > best ? (uint32_t)i : arg;       // strict: the lowest index wins ties
                best = cand > best ? cand : best;
            }
            nd[jj] = best + row[jj];
            bits |= arg << (3 * jj);
        }
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) d[jj] = active ? nd[jj] : d[jj];
        // the word of the frame: 6 bits from each of the four rows, in row order
        const auto p = __builtin_amdgcn_permlane16_swap(bits, bits, false, false);
        const uint32_t pr = p[0] | (p[1] << 6);
        const auto r = __builtin_amdgcn_permlane32_swap(pr, pr, false, false);
        return r[0] | (r[1] << 12);
    };

    uint32_t* bw = bpw + f0;
    double shift = 0.0;
    item_t x0, x1, x2, x3;
    issue(0, x0);
    issue(1, x1);
    issue(2, x2);
    issue(3, x3);
    for (int k = 0; k < n_chunks; ++k) {
#pragma unroll 1
        for (int e4 = 0; e4 < UPW; e4 += 4) {
            const int i = k * UPW + e4;
            emit(k, e4, x0);
            issue(i + 4, x0);
            emit(k, e4 + 1, x1);
            issue(i + 5, x1);
            emit(k, e4 + 2, x2);
            issue(i + 6, x2);
            emit(k, e4 + 3, x3);
            issue(i + 7, x3);
        }
        wave_sync();
        const float* rows = tile + v * kTileStride + h * 2;
#pragma unroll
        for (int t4 = 0; t4 < 16; t4 += 4) {
            uint32_t w[4];
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) {
                const int t = 16 * k + t4 + tt;
                const float2 rw = *(const float2*)(rows + (t4 + tt) * 8);
                const float row[2] = {rw.x, rw.y};
                if (t4 + tt == 0 && k == 0) {
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj)
                        d[jj] = (2 * h + jj < S && T > 0) ? logpi[m * S + 2 * h + jj] + row[jj] : -INFINITY;
                    w[tt] = 0;
                } else {
                    w[tt] = step(row, t < T);
                }
                if (tt == 0 && (k > 0 || t4 > 0)) {
                    // delta relative to its largest entry after frames 4, 8, 12, ... (the path
                    // does not change; the schedule is viterbi_lane_kernel's)
                    const float top = rows_max(fmaxf(d[0], d[1]));
                    if (t < T && top > -INFINITY) {
                        shift += (double)top;
#pragma unroll
                        for (int jj = 0; jj < 2; ++jj) d[jj] -= top;
                    }
                }
            }
            const int t0 = 16 * k + t4;
            if (words_in_lds) {
                // (every utterance has its own kTcap words: what lies past its end is never read)
                if (h == 0 && v < UPW) *(u32x4*)(wl + v * kWordStride + t0) = u32x4{w[0], w[1], w[2], w[3]};
            } else if (h == 0) {
                if (t0 + 3 < T) {
                    struct __attribute__((packed, aligned(4))) Words4 { uint32_t v[4]; };
                    *(Words4*)(bw + t0) = Words4{{w[0], w[1], w[2], w[3]}};
                } else {
#pragma unroll
                    for (int tt = 0; tt < 4; ++tt)
                        if (t0 + tt < T) bw[t0 + tt] = w[tt];
                }
            }
        }
        wave_sync();
    }

    // ---- best final state (lowest index on ties), log-probability, backtrace by lane v
    float da[8];
    {
        float a[4], b[4];
        rows4(d[0], a);
        rows4(d[1], b);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            da[2 * r] = a[r];
            da[2 * r + 1] = b[r];
        }
    }
    if (!valid || h != 0) continue;
    if (T <= 0) {
        out_lp[u] = 0.0;
        continue;
    }
    float best = -INFINITY;
    int s = 0;
#pragma unroll
    for (int i = 0; i < S; ++i)
        if (da[i] > best) {
            best = da[i];
            s = i;
        }
    out_lp[u] = shift + (double)best;
    // (sixteen frames per piece, the next piece requested before this one is resolved: the
    // addresses do not depend on the states, so only the first piece's latency is waited for)
    struct __attribute__((packed, aligned(4))) Words16 { uint32_t v[16]; };
    int32_t* p = path + f0;
    p[T - 1] = s;
    auto backtrace = [&](const auto* words) {
        int t = T - 1;                       // frame whose word is looked up next
        Words16 cur, nxt;
        if (t >= 16) cur = *(const Words16*)(words + t - 15);
        for (; t >= 16; t -= 16) {
            if (t >= 32) nxt = *(const Words16*)(words + t - 31);   // words of frames t-31 .. t-16
            Words16 o;
#pragma unroll
            for (int k = 15; k >= 0; --k) {
                s = (int)((cur.v[k] >> (3 * s)) & 7u);            // cur: words of frames t-15 .. t
                o.v[k] = (uint32_t)s;                             // state of frame t-16+k
            }
            *(Words16*)(p + t - 16) = o;
            cur = nxt;
        }
        for (; t >= 1; --t) {
            s = (int)((words[t] >> (3 * s)) & 7u);
            p[t - 1] = s;
        }
    };
    if (words_in_lds) backtrace(wl + v * kWordStride);
    else backtrace(bw);
    }   // groups
}

}  // namespace

// Float32 Viterbi of every utterance under its own model without a log-density array
// (at most 8 states, at most 48 features, model tables that fit the LDS); SRX_EUNSUPPORTED
// for every other job.  logA / logpi: [M][S][S] / [M][S] float32 on the device.
int srx_fused_viterbi(srx_context* h, const void* X, const int64_t* off, int64_t U, int64_t n_frames,
                      const int32_t* utt_model, const srx_models_t* md, const float* logA,
                      const float* logpi, int32_t* path, double* lp, hipStream_t st) {
    const int M = md->n_models, S = md->n_states, D = md->n_features;
    if (h->exact_f32 || h->viterbi_path == SRX_VITERBI_SPLIT || S < 1 || S > 8 || D > 48 ||
        n_frames <= 0 || U <= 0)
        return SRX_EUNSUPPORTED;
    const int H = (D % 16 >= 1 && D % 16 <= 8) ? 1 : 0;
    const int RQ = H ? D / 16 : (D + 15) / 16, QT = RQ + H;
    const size_t w_bytes = (size_t)M * QT * 2 * 4 * 8 * 16 + 4 * (size_t)((M * 8 + 3) & ~3);
    // Utterances per wave: 8 with the back-pointer words in LDS when the utterances are short
    // enough on average for most groups to fit (a group with a longer one spills its words like
    // the 16-utterance variant does), else 16.
    const int UPW = n_frames / U <= kTcap * 3 / 4 ? 8 : 16;
    // waves per workgroup and workgroups per CU: as many resident waves as the LDS holds (the
    // model tables once per workgroup, a tile per wave), at most 16 (the kernel has 128 registers)
    const size_t tile_bytes = 4 * ((size_t)UPW * kTileStride + (UPW == 8 ? (size_t)UPW * kWordStride : 0));
    const size_t lds_cu = 160 * 1024;
    int wpb = 0, per_cu = 0;
    for (int w = 16; w >= 4; --w) {
        const size_t need = w_bytes + w * tile_bytes;
        if (need > lds_cu) continue;
        int b = (int)(lds_cu / need);
        if (b * w > 16) b = 16 / w;
        if (b * w > wpb * per_cu) {
            wpb = w;
            per_cu = b;
        }
    }
    if (!wpb) return SRX_EUNSUPPORTED;
    const size_t lds = w_bytes + wpb * tile_bytes;
    const int NT = M;
    const size_t n_wp = (size_t)NT * 16 * QT * 2 * 4 * 4, n_c = (size_t)NT * 16;
    unsigned* base;
    int rc = srx_ws_ensure(h, WS_MISC, 4 * (n_wp + n_c + (size_t)M * S * S + (size_t)M * S), (void**)&base);
    if (rc) return rc;
    uint32_t* bpw;
    rc = srx_ws_ensure(h, WS_BACKPTR, sizeof(uint32_t) * (size_t)n_frames, (void**)&bpw);
    if (rc) return rc;
    SplitTables tb;
    float* cst = (float*)(base + n_wp);
    float* A = cst + n_c;
    float* pi = A + (size_t)M * S * S;
    tb.Wp = base; tb.cst = cst; tb.A = A; tb.pi = pi; tb.order = nullptr;
    // one 16-row tile per model
    hipLaunchKernelGGL(split_prep_kernel, dim3(16), dim3(256), 0, st, *md, 16, NT, RQ, H, base, cst, A, pi);
    const int64_t groups = (U + UPW - 1) / UPW;
    if (h->viterbi_path == SRX_VITERBI_AUTO) {
        // A wave stays with its utterances from their first frame to their last, so the job
        // needs enough groups to fill the chip (two waves per SIMD), and a last round that
        // leaves most of the SIMDs idle costs more than the two-kernel path does.
        const int64_t slots = (int64_t)h->n_cu * wpb * per_cu;
        if (groups < (int64_t)h->n_cu * 8) return SRX_EUNSUPPORTED;
        if (groups > slots && groups * 5 < slots * 6) return SRX_EUNSUPPORTED;
    }
    int64_t blocks = (groups + wpb - 1) / wpb;
    if (blocks > (int64_t)h->n_cu * per_cu) blocks = (int64_t)h->n_cu * per_cu;
    const dim3 grid((unsigned)blocks);
    const float* Xf = (const float*)X;
#define SRX_CASE(s_, rq, hh, upw)                                                                \
    if (S == s_ && RQ == rq && H == hh && UPW == upw) {                                          \
        auto kern = viterbi_fused_kernel<s_, rq, hh, upw>;                                       \
        SRX_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                    (int)lds));                                                  \
        hipLaunchKernelGGL(kern, grid, dim3(64 * wpb), lds, st, Xf, off, U, utt_model, M, D, tb, \
                           logA, logpi, bpw, path, lp);                                          \
        SRX_HIP(hipGetLastError());                                                              \
        return SRX_OK;                                                                           \
    }
#define SRX_CASES(s_)                                                                            \
    SRX_CASE(s_, 0, 1, 8) SRX_CASE(s_, 1, 0, 8) SRX_CASE(s_, 1, 1, 8) SRX_CASE(s_, 2, 0, 8)      \
    SRX_CASE(s_, 2, 1, 8) SRX_CASE(s_, 3, 0, 8) SRX_CASE(s_, 0, 1, 16) SRX_CASE(s_, 1, 0, 16)    \
    SRX_CASE(s_, 1, 1, 16) SRX_CASE(s_, 2, 0, 16) SRX_CASE(s_, 2, 1, 16) SRX_CASE(s_, 3, 0, 16)
    SRX_CASES(1) SRX_CASES(2) SRX_CASES(3) SRX_CASES(4) SRX_CASES(5) SRX_CASES(6) SRX_CASES(7) SRX_CASES(8)
#undef SRX_CASES
#undef SRX_CASE
    return SRX_EUNSUPPORTED;
}
