// Fused Viterbi decoding for models of at most 8 states, float32 (utils.py:243-266 of the
// reference with the emission of hmm.py:154-163 in front of it): the log-densities never
// go through global memory.
//
// A wave owns 8 utterances and walks them in chunks of 16 frames:
//   emission   one utterance at a time, its 16 frames as the columns of the matrix
//              instruction, the 8 state rows of the utterance's model as rows (split-bf16
//              products, the tables of score_split.h; the model tables of all models sit in
//              LDS), log-densities into the wave's LDS tile [utterance][frame][state];
//   recursion  eight lanes per utterance, one state each (lane 8 v + j: state j of utterance
//              v), run the 16 max-plus steps on registers: the lane's column of log A, delta
//              of state i read from lane 8 v + i by a DPP row broadcast inside the add that
//              forms the candidate, back-pointers of a frame packed 3 bits per state into one
//              word by shifted ORs over the eight lanes.
// The back-pointer words of utterances of up to kTcap frames stay in LDS beside the tile (no
// store goes out between the loads of the feature stream -- what even 4 bytes per frame of them
// cost: tools/probes/chunk_stream.hip -- and the backtrace reads no memory); a group with a
// longer utterance spills its words to memory.  Lane 8 v resolves them backwards at the end,
// sixteen frames per access.  Ties take the lowest state, as np.argmax does; delta is kept
// relative to its largest entry every four frames (the shift is carried in float64 to the
// log-probability).  The arithmetic is that of emission_utt_split_kernel + viterbi_lane_kernel
// term for term: same states, same log-probabilities, bit for bit.
#include "score_split.h"

namespace {

constexpr int kUpw = 8;                     // utterances per wave
constexpr int kTcap = 256;                  // frames whose back-pointer words a wave keeps in LDS
constexpr int kWordStride = kTcap + 4;      // words per utterance there (+4: bank spread)
constexpr int kTileStride = 16 * 8 + 8;     // floats per utterance in the tile (+8: 32 lanes read 32 banks)
constexpr int kWaveFloats = kUpw * kTileStride + kUpw * kWordStride;

__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// d of lane 8 v + I (I < 8) of this lane's group of eight, plus la: two DPP adds, each writing
// one half of every 16-lane row (bank_mask) from that half's own broadcast lane.  A DPP read
// needs two wait states after a vector write of the same register, and the hazard recogniser
// does not look inside inline asm, so every pair brings its own s_nop: whatever the compiler
// places in front of it (a copy or a split of d or la, the select that rewrites d) has retired
// by the time the first add reads its operands.
template <int I>
__device__ __forceinline__ float bcast_add(float d, float la) {
    float c;
    asm volatile("s_nop 1\n\t"
                 "v_add_f32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0x3\n\t"
                 "v_add_f32_dpp %0, %1, %2 row_newbcast:%4 row_mask:0xf bank_mask:0xc"
                 : "=&v"(c) : "v"(d), "v"(la), "n"(I), "n"(I + 8));
    return c;
}
// value of lane L + N of the row (zero past the row's end)
template <int N>
__device__ __forceinline__ unsigned shl_lanes(unsigned v) {
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x100 + N, 0xF, 0xF, true);
}
// OR / maximum over the eight lanes L .. L + 7 of a row, valid in lanes 0 and 8 of the row
__device__ __forceinline__ unsigned or8(unsigned v) {
    v |= shl_lanes<1>(v);
    v |= shl_lanes<2>(v);
    v |= shl_lanes<4>(v);
    return v;
}
__device__ __forceinline__ float max8(float v) {
    // (-inf, not the shift's zero fill, is what lies past the row's end: bound_ctrl off keeps
    // the lane's own value there, which a maximum ignores)
    v = fmaxf(v, __uint_as_float((unsigned)__builtin_amdgcn_update_dpp((int)__float_as_uint(v), (int)__float_as_uint(v), 0x101, 0xF, 0xF, false)));
    v = fmaxf(v, __uint_as_float((unsigned)__builtin_amdgcn_update_dpp((int)__float_as_uint(v), (int)__float_as_uint(v), 0x102, 0xF, 0xF, false)));
    v = fmaxf(v, __uint_as_float((unsigned)__builtin_amdgcn_update_dpp((int)__float_as_uint(v), (int)__float_as_uint(v), 0x104, 0xF, 0xF, false)));
    return v;
}
// value of lane 8 v of this lane's group of eight
__device__ __forceinline__ float first_of8(float v) {
    int r = __builtin_amdgcn_update_dpp(0, (int)__float_as_uint(v), 0x150, 0xF, 0x3, false);
    r = __builtin_amdgcn_update_dpp(r, (int)__float_as_uint(v), 0x158, 0xF, 0xC, false);
    return __uint_as_float((unsigned)r);
}

template <int S, int RQ, int H>
__global__ __launch_bounds__(1024) void viterbi_fused_kernel(
    const float* __restrict__ X, const int64_t* __restrict__ off, int64_t U,
    const int32_t* __restrict__ utt_model, int M, int D, SplitTables tb,
    const float* __restrict__ logA, const float* __restrict__ logpi, uint32_t* __restrict__ bpw,
    int32_t* __restrict__ path, double* __restrict__ out_lp) {
    static_assert(S >= 1 && S <= 8, "eight lanes per utterance, one state each");
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
    const int v = lane >> 3, j = lane & 7;           // recursion: utterance, state
    float* tile = s_tiles + wave * kWaveFloats;
    uint32_t* wl = (uint32_t*)(tile + kUpw * kTileStride);     // [kUpw][kWordStride]
    // groups of 8 utterances are dealt to the waves of the grid in turn
    const int64_t n_groups = (U + kUpw - 1) / kUpw;
    const int64_t n_waves = (int64_t)gridDim.x * (blockDim.x >> 6);
    for (int64_t grp = (int64_t)blockIdx.x * (blockDim.x >> 6) + wave; grp < n_groups; grp += n_waves) {
    const int64_t u = grp * kUpw + v;
    const bool valid = u < U;
    const int64_t f0 = valid ? off[u] : 0;
    const int T = valid ? (int)(off[u + 1] - f0) : 0;
    const int m = valid && utt_model ? utt_model[u] : 0;
    int Tmax = T;
#pragma unroll
    for (int o = 32; o >= 8; o >>= 1) Tmax = max(Tmax, __shfl_xor(Tmax, o, 64));
    Tmax = __builtin_amdgcn_readfirstlane(Tmax);     // (loop bounds in scalar registers)
    if (Tmax <= 0) {
        if (valid && j == 0) out_lp[u] = 0.0;
        continue;
    }
    const int n_chunks = (Tmax + 15) >> 4;
    const int n_items = n_chunks * kUpw;
    const bool words_in_lds = Tmax <= kTcap;      // (the same for the whole wave)

    // column j of log A (-inf for a state the model lacks), delta of state j
    float la[S];
#pragma unroll
    for (int i = 0; i < S; ++i) la[i] = j < S ? logA[((size_t)m * S + i) * S + j] : -INFINITY;
    float d = -INFINITY;
    const unsigned f0_lo = (unsigned)f0, f0_hi = (unsigned)((uint64_t)f0 >> 32);

    // ---- emission of item i = (chunk, utterance): loads, then products
    const unsigned lane_off = (unsigned)ul * (unsigned)D * 4u + 16u * g;
    const unsigned half_off = (unsigned)ul * (unsigned)D * 4u + 64u * RQ + 8u * g;
    auto issue = [&](int i, item_t& dst) {
        const int k = i / kUpw, e = i % kUpw;
        const int Te = __builtin_amdgcn_readlane(T, 8 * e);
        const uint64_t fe = ((uint64_t)(unsigned)__builtin_amdgcn_readlane((int)f0_hi, 8 * e) << 32) |
                            (unsigned)__builtin_amdgcn_readlane((int)f0_lo, 8 * e);
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
        if (16 * k >= __builtin_amdgcn_readlane(T, 8 * e)) return;
        const int me = __builtin_amdgcn_readlane(m, 8 * e);
        // (rows 8..15 of the product are never stored: their lanes read rows 0..7 again)
        const unsigned* wp = s_w + (size_t)me * (QT * 256) + g * 32 + (ul & 7) * 4;
        // (one accumulator per kind of product: three chains of RQ dependent instructions
        // instead of one of 3 RQ)
        f32x4 acc = *(const f32x4*)(s_cst + me * 8 + 4 * (g & 1));
        f32x4 acc1 = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < RQ; ++q) {
            const u32x4 wh = *(const u32x4*)(wp + q * 256), wl_ = *(const u32x4*)(wp + q * 256 + 128);
            bf16x8 bh, bl;
            split_operand(x.v[q], bh, bl);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wh), bh, acc, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wh), bl, acc1, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wl_), bh, acc2, 0, 0, 0);
        }
        if constexpr (H != 0) {
            // last block of 8 features, two per lane: hi halves in slots 0, 1 (x) and 4, 5 (x^2),
            // lo halves in slots 2, 3 and 6, 7 (the table repeats the hi weights there)
            const u32x4 wh = *(const u32x4*)(wp + RQ * 256), wl_ = *(const u32x4*)(wp + RQ * 256 + 128);
            const bf16x8 bb = split_half_operand(x.h);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wh), bb, acc1, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wl_), bb, acc2, 0, 0, 0);
        }
        acc += acc1 + acc2;
        if (g < 2) *(f32x4*)(tile + e * kTileStride + ul * 8 + g * 4) = acc * (float)kLn2;
    };

    // ---- one max-plus step of this lane's state; returns the frame's back-pointer word (in
    // lane 8 v): candidates in state order, a later one wins only if strictly larger
    auto step = [&](float row, bool active) -> uint32_t {
        float best = -INFINITY;
        uint32_t arg = 0;
        auto take = [&](float cand, uint32_t i) {
            arg = cand > best ? i : arg;
            best = cand > best ? cand : best;
        };
        // (a chain of eight dependent compare-selects; as a tree of three levels, with all
        // eight candidates formed first, the kernel measured a third slower)
        if constexpr (S > 0) take(bcast_add<0>(d, la[0]), 0);
        if constexpr (S > 1) take(bcast_add<1>(d, la[1]), 1);
        if constexpr (S > 2) take(bcast_add<2>(d, la[2]), 2);
        if constexpr (S > 3) take(bcast_add<3>(d, la[3]), 3);
        if constexpr (S > 4) take(bcast_add<4>(d, la[4]), 4);
        if constexpr (S > 5) take(bcast_add<5>(d, la[5]), 5);
        if constexpr (S > 6) take(bcast_add<6>(d, la[6]), 6);
        if constexpr (S > 7) take(bcast_add<7>(d, la[7]), 7);
        const float nd = best + row;
        d = active ? nd : d;
        return or8(arg << (3 * j));
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
        for (int e4 = 0; e4 < kUpw; e4 += 4) {
            const int i = k * kUpw + e4;
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
        // (the recursion is a chain of dependent steps, the emission above it bulk work: the wave
        // inside its chain issues first; 0.487 -> 0.471 ms per 13.1 M frames on one box)
        __builtin_amdgcn_s_setprio(2);
        const float* rows = tile + v * kTileStride + j;
#pragma unroll
        for (int t4 = 0; t4 < 16; t4 += 4) {
            uint32_t w[4];
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) {
                const int t = 16 * k + t4 + tt;
                const float row = rows[(t4 + tt) * 8];
                if (t4 + tt == 0 && k == 0) {
                    d = (j < S && T > 0) ? logpi[m * S + j] + row : -INFINITY;
                    w[tt] = 0;
                } else {
                    w[tt] = step(row, t < T);
                }
                if (tt == 0 && (k > 0 || t4 > 0)) {
                    // delta relative to its largest entry after frames 4, 8, 12, ... (the path
                    // does not change; the schedule is viterbi_lane_kernel's)
                    const float top = first_of8(max8(d));
                    if (t < T && top > -INFINITY) {
                        shift += (double)top;
                        d -= top;
                    }
                }
            }
            const int t0 = 16 * k + t4;
            if (j == 0) {
                if (words_in_lds) {
                    // (every utterance has its own kTcap words: what lies past its end is never read)
                    *(u32x4*)(wl + v * kWordStride + t0) = u32x4{w[0], w[1], w[2], w[3]};
                } else if (t0 + 3 < T) {
                    struct __attribute__((packed, aligned(4))) Words4 { uint32_t v[4]; };
                    *(Words4*)(bw + t0) = Words4{{w[0], w[1], w[2], w[3]}};
                } else {
#pragma unroll
                    for (int tt = 0; tt < 4; ++tt)
                        if (t0 + tt < T) bw[t0 + tt] = w[tt];
                }
            }
        }
        __builtin_amdgcn_s_setprio(0);
        wave_sync();
    }

    // ---- best final state (lowest index on ties), log-probability, backtrace by lane 8 v
    float best = -INFINITY;
    int s = 0;
    {
        auto take = [&](float cand, int i) {
            if (cand > best) {
                best = cand;
                s = i;
            }
        };
        if constexpr (S > 0) take(bcast_add<0>(d, 0.f), 0);
        if constexpr (S > 1) take(bcast_add<1>(d, 0.f), 1);
        if constexpr (S > 2) take(bcast_add<2>(d, 0.f), 2);
        if constexpr (S > 3) take(bcast_add<3>(d, 0.f), 3);
        if constexpr (S > 4) take(bcast_add<4>(d, 0.f), 4);
        if constexpr (S > 5) take(bcast_add<5>(d, 0.f), 5);
        if constexpr (S > 6) take(bcast_add<6>(d, 0.f), 6);
        if constexpr (S > 7) take(bcast_add<7>(d, 0.f), 7);
    }
    if (!valid || j != 0) continue;
    if (T <= 0) {
        out_lp[u] = 0.0;
        continue;
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
    // waves per workgroup and workgroups per CU: as many resident waves as the LDS holds (the
    // model tables once per workgroup, a tile and a word array per wave), at most 16 (the
    // kernel is compiled for 128 registers)
    const size_t tile_bytes = 4 * (size_t)kWaveFloats, lds_cu = 160 * 1024;
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
    // (a bank whose tables leave room for fewer than two waves per SIMD -- more than about 20
    // models of 39 features -- is better served by the two-kernel path unless asked otherwise)
    if (h->viterbi_path == SRX_VITERBI_AUTO && wpb * per_cu < 8) return SRX_EUNSUPPORTED;
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
    const int64_t groups = (U + kUpw - 1) / kUpw;
    int64_t blocks = (groups + wpb - 1) / wpb;
    if (blocks > (int64_t)h->n_cu * per_cu) blocks = (int64_t)h->n_cu * per_cu;
    const dim3 grid((unsigned)blocks);
    const float* Xf = (const float*)X;
#define SRX_CASE(s_, rq, hh)                                                                     \
    if (S == s_ && RQ == rq && H == hh) {                                                        \
        auto kern = viterbi_fused_kernel<s_, rq, hh>;                                            \
        SRX_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                    (int)lds));                                                  \
        hipLaunchKernelGGL(kern, grid, dim3(64 * wpb), lds, st, Xf, off, U, utt_model, M, D, tb, \
                           logA, logpi, bpw, path, lp);                                          \
        SRX_HIP(hipGetLastError());                                                              \
        return SRX_OK;                                                                           \
    }
#define SRX_CASES(s_)                                                                            \
    SRX_CASE(s_, 0, 1) SRX_CASE(s_, 1, 0) SRX_CASE(s_, 1, 1) SRX_CASE(s_, 2, 0) SRX_CASE(s_, 2, 1) \
    SRX_CASE(s_, 3, 0)
    SRX_CASES(1) SRX_CASES(2) SRX_CASES(3) SRX_CASES(4) SRX_CASES(5) SRX_CASES(6) SRX_CASES(7) SRX_CASES(8)
#undef SRX_CASES
#undef SRX_CASE
    return SRX_EUNSUPPORTED;
}
