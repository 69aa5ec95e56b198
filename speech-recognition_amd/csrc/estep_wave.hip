// Baum-Welch E-step of SMALL jobs (the reference's own: a word model of 3 states trained on
// 25 utterances of 60 to 240 frames, my_hmmlearn/hmm.py:419-433): one WAVE per utterance, and
// the recursions over time cut into 64 pieces that run side by side.
//
// The scaled forward recursion alpha_t ~ (alpha_{t-1} A) o b_t (utils.py:132-150) is a product
// of the S x S matrices F_t = A diag(b_t), and so is the backward one, beta_{t-1} ~ F_t beta_t
// (utils.py:169-183).  Only the DIRECTION of a forward vector matters to what follows it -- the
// scaling recursion renormalises at every frame, gamma_t and the xi of a transition are
// normalised to sum 1 (hmm.py:325-328; utils.py:201-218 sums terms that add up to 1 per
// transition) -- so the work splits: lane k owns the frames [k C, (k+1) C), C = ceil(T / 64):
//   1. emission: log-densities of its frames (hmm.py:154-163, direct form), b~ = exp(l - max l);
//   2. its piece's product P_k = F_t ... F_t' (every row renormalised with a binary exponent
//      of its own at every factor, see Mat; the scale of a row is of no consequence in itself,
//      the scales of the rows relative to each other are);
//   3. an inclusive prefix scan and a suffix scan of the P_k over the 64 lanes (6 steps each,
//      matrices passed by lane shuffles): what comes into the piece from the left as a
//      forward vector, what comes in from the right as a backward vector;
//   4. the scaled forward recursion over its own frames from the incoming vector (this gives
//      the scale factors c_t, hence the log-probability, frame by frame exactly as the
//      sequential recursion would), then the backward recursion with gamma, xi and the
//      per-state sums;
//   5. the sums over the features, gamma^T [x | x^2] (hmm.py:243-245), lanes over feature
//      columns, frames in order.
// A 240-frame utterance is then 4 dependent steps per lane and 12 shuffle steps instead of
// 480 dependent steps in one lane.  Per-frame arrays (b~, alpha^, gamma) live in a global
// scratch area; each lane reads and writes its own frames only, except phase 5.
// Every utterance leaves one record [nobs, logprob, start, trans, post, obs, obs2] (float64);
// a second kernel adds the records of a model's utterances in utterance order.
// Float32 and float64 data run the same code (the arithmetic is the data's dtype; sums that
// cross lanes or utterances are float64).
#include "srx_internal.h"

#include <math.h>

#include <type_traits>

namespace {

template <typename R> struct Ops;
template <> struct Ops<float> {
    static __device__ __forceinline__ float exp(float x) { return __expf(x); }
    static __device__ __forceinline__ float log(float x) { return __logf(x); }
    static __device__ __forceinline__ float rcp(float x) { return __builtin_amdgcn_rcpf(x); }
    // x = m 2^e with m in [0.5, 1)
    static __device__ __forceinline__ int expo(float x) { return __builtin_amdgcn_frexp_expf(x); }
    static __device__ __forceinline__ float scale(float x, int n) { return ldexpf(x, n); }
    static __device__ __forceinline__ float finite(float x) { return fminf(x, 3.4028234664e38f); }
};
template <> struct Ops<double> {
    static __device__ __forceinline__ double exp(double x) { return ::exp(x); }
    static __device__ __forceinline__ double log(double x) { return ::log(x); }
    static __device__ __forceinline__ double rcp(double x) { return 1.0 / x; }
    static __device__ __forceinline__ int expo(double x) { return __builtin_amdgcn_frexp_exp(x); }
    static __device__ __forceinline__ double scale(double x, int n) { return ldexp(x, n); }
    static __device__ __forceinline__ double finite(double x) { return fmin(x, 1.7976931348623157e308); }
};

__device__ __forceinline__ void wave_mem_sync() {
    // orders this wave's global stores before its later loads of them from other lanes (one
    // wave = one workgroup here)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// A transfer matrix with ONE BINARY EXPONENT PER ROW: entry (i, j) = v[i][j] 2^e[i], the largest
// v of a row in [0.5, 1) (kNone for a row of zeros).  Row i of a product of F_t is the
// likelihood of the piece's frames from start state i; between start states that the model
// does not really offer (a structural zero of A, a state the past has ruled out) and the one it
// does, that likelihood differs by e^50 per frame at 39 features -- far more than one exponent
// for the whole matrix could hold in the data's dtype, and the row that matters may be the small
// one.  With its own exponent no row is ever lost to another; inside a row, terms that fall
// below the dtype's range relative to the row's largest are the ones the sequential recursion
// loses too.
constexpr int kNone = -(1 << 28);
template <typename R, int S>
struct Mat {
    R v[S][S];
    int e[S];
};

// row i to its largest entry in [0.5, 1), exponent adjusted
template <typename R, int S>
__device__ __forceinline__ void renorm_row(Mat<R, S>& m, int i, int base) {
    R mx = 0;
#pragma unroll
    for (int j = 0; j < S; ++j) mx = m.v[i][j] > mx ? m.v[i][j] : mx;
    const int ex = mx > 0 ? Ops<R>::expo(mx) : 0;
#pragma unroll
    for (int j = 0; j < S; ++j) m.v[i][j] = Ops<R>::scale(m.v[i][j], -ex);
    m.e[i] = mx > 0 ? base + ex : kNone;
}

template <typename R, int S>
__device__ __forceinline__ Mat<R, S> matmul(const Mat<R, S>& a, const Mat<R, S>& b) {
    Mat<R, S> c;
#pragma unroll
    for (int i = 0; i < S; ++i) {
        // term k of row i is a.v[i][k] 2^(a.e[i] + b.e[k]) b.v[k][:]: bring the terms to the
        // largest of their exponents
        int ek[S], er = kNone;
#pragma unroll
        for (int k = 0; k < S; ++k) {
            ek[k] = a.v[i][k] > 0 && b.e[k] != kNone ? b.e[k] + Ops<R>::expo(a.v[i][k]) : kNone;
            er = ek[k] > er ? ek[k] : er;
        }
        R w[S];
#pragma unroll
        for (int k = 0; k < S; ++k) {
            int sh = b.e[k] - er + 1;                 // (a.v <= 1: the largest term stays below 2)
            sh = sh < -4096 ? -4096 : sh;
            w[k] = ek[k] != kNone ? Ops<R>::scale(a.v[i][k], sh) : (R)0;
        }
#pragma unroll
        for (int j = 0; j < S; ++j) {
            R acc = 0;
#pragma unroll
            for (int k = 0; k < S; ++k) acc += w[k] * b.v[k][j];
            c.v[i][j] = acc;
        }
        renorm_row(c, i, a.e[i] == kNone || er == kNone ? kNone : a.e[i] + er - 1);
        if (a.e[i] == kNone || er == kNone) c.e[i] = kNone;
    }
    return c;
}

template <typename R, int S>
__device__ __forceinline__ Mat<R, S> shfl_mat(const Mat<R, S>& m, int src_lane) {
    Mat<R, S> o;
#pragma unroll
    for (int i = 0; i < S; ++i) {
        o.e[i] = __shfl(m.e[i], src_lane, 64);
#pragma unroll
        for (int j = 0; j < S; ++j) o.v[i][j] = __shfl(m.v[i][j], src_lane, 64);
    }
    return o;
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// scratch per frame: b~ [S], alpha^ [S], gamma [S] and the row maximum, in the data's dtype
template <int S> constexpr int kFrameWords = 3 * S + 1;

template <typename R, int S>
__global__ __launch_bounds__(64) void estep_wave_kernel(
    const R* __restrict__ X, const int64_t* __restrict__ off, int64_t U,
    const int32_t* __restrict__ utt_model, srx_models_t md, int with_obs, R* __restrict__ scratch,
    double* __restrict__ records, int rec_size, int lds_frames, int red_offset) {
    const int D = md.n_features;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // [S][DP]{mean, -0.5 / var}: DP = D rounded up to whole blocks of kBlk features, the padding
    // with weight zero; then the S constants
    constexpr int kBlk = 16;
    const int DP = (D + kBlk - 1) / kBlk * kBlk;
    R* s_tab = (R*)smem;
    R* s_cst = s_tab + 2 * S * DP;
    const int lane = threadIdx.x;
    const int64_t u = blockIdx.x;
    const srx_stats_layout L(S, D);
    double* rec = records + (size_t)u * rec_size;
    const int m = utt_model ? utt_model[u] : 0;
    const int64_t f0 = off[u];
    const int T = (int)(off[u + 1] - f0);
    for (int i = lane; i < rec_size; i += 64) rec[i] = 0.0;
    wave_mem_sync();                    // (the zeros land before the entries written below)
    if (T <= 0) return;                 // (wave-uniform: an empty utterance adds nothing, not even a count)

    // ---- model tables (hmm.py:154-163: variances floored at the smallest normal double)
    for (int i = lane; i < S * DP; i += 64) {
        const int s = i / DP, d = i % DP;
        const size_t k = ((size_t)m * S + s) * D + d;
        const double var = d < D ? fmax(md.covars[k], 2.2250738585072014e-308) : 1.0;
        s_tab[2 * i] = d < D ? (R)md.means[k] : (R)0;
        s_tab[2 * i + 1] = d < D ? (R)(-0.5 / var) : (R)0;
    }
#pragma unroll
    for (int s = 0; s < S; ++s) {
        double acc = 0.0;
        for (int d = lane; d < D; d += 64)
            acc += ::log(fmax(md.covars[((size_t)m * S + s) * D + d], 2.2250738585072014e-308));
        acc = wave_sum(acc);
        if (lane == 0) s_cst[s] = (R)(-0.5 * (D * 1.8378770664093453 + acc));
    }
    R A[S][S], pi[S];
#pragma unroll
    for (int i = 0; i < S; ++i) {
        pi[i] = (R)md.startprob[(size_t)m * S + i];
#pragma unroll
        for (int j = 0; j < S; ++j) A[i][j] = (R)md.transmat[((size_t)m * S + i) * S + j];
    }
    __syncthreads();

    const int C = (T + 63) / 64;
    const int t_begin = lane * C < T ? lane * C : T;
    const int t_end = t_begin + C < T ? t_begin + C : T;
    // The per-frame arrays of an utterance of at most lds_frames frames stay in LDS (behind the
    // tables): every one of them is written and read back within this wave, and a round trip to
    // memory per phase is most of what so small a job waits for.  Longer utterances use the
    // global scratch area.
    R* fr = T <= lds_frames ? s_cst + S : scratch + (size_t)f0 * kFrameWords<S>;
    // (behind the frames: the area the cross-lane sums go through)
    constexpr int kRedStride = 65;       // doubles per quantity: lanes reading different rows hit different banks
    double* red = (double*)(smem + red_offset);
    auto bt_at = [&](int t) { return fr + (size_t)t * kFrameWords<S>; };             // b~[S]
    auto al_at = [&](int t) { return fr + (size_t)t * kFrameWords<S> + S; };         // alpha^[S]
    auto ga_at = [&](int t) { return fr + (size_t)t * kFrameWords<S> + 2 * S; };     // gamma[S]
    auto mx_at = [&](int t) { return fr + (size_t)t * kFrameWords<S> + 3 * S; };     // row maximum

    // ---- 1. emission of this lane's frames; 2. its piece's product of F_t = A diag(b~_t), t >= 1
    Mat<R, S> P;                                           // the identity
#pragma unroll
    for (int i = 0; i < S; ++i) {
        P.e[i] = 1;
#pragma unroll
        for (int j = 0; j < S; ++j) P.v[i][j] = i == j ? (R)0.5 : (R)0;
    }
    {
        // (A block of features is requested at once -- every lane reads its own row, so a load is
        // a round trip to the cache for 64 different lines -- and the block after it, of this
        // frame or the next, is requested before this one is used: the loads are issued whatever
        // the lane's range is, from a clamped frame, because a load under a branch is waited for
        // where the branch ends.  The padding repeats the last feature under a zero weight.)
        auto load_blk = [&](int t, int d0, R (&xv)[kBlk]) {
            const R* x = X + (size_t)(f0 + (t < T ? t : T - 1)) * D;
#pragma unroll
            for (int k = 0; k < kBlk; ++k) xv[k] = x[d0 + k < D ? d0 + k : D - 1];
        };
        R cur[kBlk], nxt[kBlk], ll[S];
#pragma unroll
        for (int s = 0; s < S; ++s) ll[s] = 0;
        int t = t_begin, d0 = 0;
        load_blk(t, 0, cur);
        while (t < t_end) {
            int nt = t, nd = d0 + kBlk;
            if (nd >= DP) {
                nd = 0;
                nt = t + 1;
            }
            load_blk(nt, nd, nxt);
#pragma unroll
            for (int s = 0; s < S; ++s) {
                const R* tb = s_tab + 2 * (s * DP + d0);
#pragma unroll
                for (int k = 0; k < kBlk; ++k) {
                    const R c = cur[k] - tb[2 * k];
                    ll[s] += tb[2 * k + 1] * c * c;
                }
            }
            if (nd == 0) {                                 // the frame's last block
                R mx = ll[0] + s_cst[0];
#pragma unroll
                for (int s = 0; s < S; ++s) {
                    ll[s] += s_cst[s];
                    mx = ll[s] > mx ? ll[s] : mx;
                }
                R b[S];
#pragma unroll
                for (int s = 0; s < S; ++s) {
                    b[s] = Ops<R>::exp(ll[s] - mx);
                    bt_at(t)[s] = b[s];
                    ll[s] = 0;
                }
                *mx_at(t) = mx;
                if (t >= 1) {
                    Mat<R, S> F;
#pragma unroll
                    for (int i = 0; i < S; ++i) {
#pragma unroll
                        for (int j = 0; j < S; ++j) F.v[i][j] = A[i][j] * b[j];
                        renorm_row(F, i, 0);
                    }
                    P = matmul(P, F);
                }
            }
#pragma unroll
            for (int k = 0; k < kBlk; ++k) cur[k] = nxt[k];
            t = nt;
            d0 = nd;
        }
    }

    // ---- 3. prefix and suffix products over the lanes (Kogge-Stone; identity beyond the ends)
    Mat<R, S> pre = P, suf = P;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const Mat<R, S> lft = shfl_mat(pre, lane >= d ? lane - d : lane);
        const Mat<R, S> rgt = shfl_mat(suf, lane + d < 64 ? lane + d : lane);
        if (lane >= d) pre = matmul(lft, pre);
        if (lane + d < 64) suf = matmul(suf, rgt);
    }
    // exclusive: what lies left of this lane's first frame / right of its last one
    const Mat<R, S> pre_x = shfl_mat(pre, lane >= 1 ? lane - 1 : 0);
    const Mat<R, S> suf_x = shfl_mat(suf, lane < 63 ? lane + 1 : 63);

    // ---- 4a. forward over this lane's frames.  alpha^_0 = pi o b~_0 normalised (computed by
    // every lane from frame 0's b~: one broadcast load), the incoming vector alpha^_0 pre_x.
    R v[S];            // alpha^ of the frame before the current one
    double lp = 0.0;
    {
        wave_mem_sync();                                   // frame 0's b~ was written by lane 0
        R a0[S], c0 = 0;
#pragma unroll
        for (int s = 0; s < S; ++s) {
            a0[s] = pi[s] * bt_at(0)[s];
            c0 += a0[s];
        }
        const R ci = Ops<R>::rcp(c0);
#pragma unroll
        for (int s = 0; s < S; ++s) a0[s] *= ci;
        if (lane == 0) {
            lp += (double)(*mx_at(0) + Ops<R>::log(c0));
#pragma unroll
            for (int s = 0; s < S; ++s) {
                v[s] = a0[s];
                al_at(0)[s] = a0[s];
            }
        } else {
            // alpha^_0 pre_x, the rows brought to the largest exponent among those alpha^_0 uses
            int er = kNone;
#pragma unroll
            for (int i = 0; i < S; ++i) {
                const int ei = a0[i] > 0 && pre_x.e[i] != kNone ? pre_x.e[i] + Ops<R>::expo(a0[i]) : kNone;
                er = ei > er ? ei : er;
            }
            R w0[S];
#pragma unroll
            for (int i = 0; i < S; ++i) {
                int sh = pre_x.e[i] - er;
                sh = sh < -4096 ? -4096 : sh;
                w0[i] = a0[i] > 0 && pre_x.e[i] != kNone ? Ops<R>::scale(a0[i], sh) : (R)0;
            }
            R sum = 0;
#pragma unroll
            for (int j = 0; j < S; ++j) {
                R acc = 0;
#pragma unroll
                for (int i = 0; i < S; ++i) acc += w0[i] * pre_x.v[i][j];
                v[j] = acc;
                sum += acc;
            }
            const R si = sum > 0 ? Ops<R>::rcp(sum) : (R)0;
#pragma unroll
            for (int j = 0; j < S; ++j) v[j] *= si;
        }
    }
    R v_in[S];         // alpha^ of frame t_begin - 1 (lanes > 0), needed again by the backward pass
#pragma unroll
    for (int s = 0; s < S; ++s) v_in[s] = v[s];
    for (int t = t_begin > 1 ? t_begin : 1; t < t_end; ++t) {
        R a[S], c = 0;
#pragma unroll
        for (int j = 0; j < S; ++j) {
            R acc = 0;
#pragma unroll
            for (int i = 0; i < S; ++i) acc += v[i] * A[i][j];
            a[j] = acc * bt_at(t)[j];
            c += a[j];
        }
        const R ci = Ops<R>::rcp(c);
        lp += (double)(*mx_at(t) + Ops<R>::log(c));
#pragma unroll
        for (int j = 0; j < S; ++j) {
            v[j] = a[j] * ci;
            al_at(t)[j] = v[j];
        }
    }

    // ---- 4b. backward over this lane's frames: beta of its last frame = suf_x 1
    // beta is carried on the scale of the sequential recursion, sum_i alpha^_t(i) beta_t(i) = 1:
    // the states that hold the posterior mass then have a beta of order 1 / alpha^, whatever the
    // others' are (a state the past has all but ruled out and the future favours by e^100 has a
    // large beta and a tiny alpha; scaled to the largest beta instead, the states that matter
    // would be the ones to underflow).  Entries beyond the dtype's range are held at its largest
    // finite value: their alpha^ is below its smallest, and 0 x finite is the 0 the sequential
    // recursion gets there.  v is alpha^ of this lane's last frame.
    R beta[S];
    {
        int eb = kNone;
#pragma unroll
        for (int i = 0; i < S; ++i) {
            const int ei = v[i] > 0 && suf_x.e[i] != kNone ? suf_x.e[i] + Ops<R>::expo(v[i]) : kNone;
            eb = ei > eb ? ei : eb;
        }
        R tot = 0, braw[S];
#pragma unroll
        for (int i = 0; i < S; ++i) {
            R acc = 0;
#pragma unroll
            for (int j = 0; j < S; ++j) acc += suf_x.v[i][j];
            int sh = eb != kNone ? suf_x.e[i] - eb : 0;
            sh = sh < -4096 ? -4096 : sh > 4096 ? 4096 : sh;
            braw[i] = suf_x.e[i] != kNone ? Ops<R>::finite(Ops<R>::scale(acc, sh)) : (R)0;
            tot += v[i] * braw[i];
        }
        const R ti = tot > 0 ? Ops<R>::rcp(tot) : (R)1;
#pragma unroll
        for (int i = 0; i < S; ++i)
            beta[i] = (lane == 63 || (lane + 1) * C >= T) ? (R)1 : Ops<R>::finite(braw[i] * ti);
    }
    R xi[S][S], post[S], start[S];
#pragma unroll
    for (int i = 0; i < S; ++i) {
        post[i] = start[i] = 0;
#pragma unroll
        for (int j = 0; j < S; ++j) xi[i][j] = 0;
    }
    for (int t = t_end - 1; t >= t_begin; --t) {
        R al[S], g[S], gs = 0;
#pragma unroll
        for (int i = 0; i < S; ++i) {
            al[i] = al_at(t)[i];
            g[i] = al[i] * beta[i];
            gs += g[i];
        }
        const R gi = gs > 0 ? Ops<R>::rcp(gs) : (R)1;       // utils.normalize: zero rows divide by 1
#pragma unroll
        for (int i = 0; i < S; ++i) {
            g[i] *= gi;
            ga_at(t)[i] = g[i];
            post[i] += g[i];
            if (t == 0) start[i] = g[i];
        }
        if (t >= 1) {
            R w[S], ap[S];
#pragma unroll
            for (int j = 0; j < S; ++j) w[j] = bt_at(t)[j] * beta[j];
#pragma unroll
            for (int i = 0; i < S; ++i) ap[i] = t - 1 >= t_begin ? al_at(t - 1)[i] : v_in[i];
            R e[S][S], tot = 0, nb[S], bm = 0;
#pragma unroll
            for (int i = 0; i < S; ++i) {
                R acc = 0;
#pragma unroll
                for (int j = 0; j < S; ++j) {
                    const R aw = A[i][j] * w[j];
                    acc += aw;
                    e[i][j] = ap[i] * aw;
                    tot += e[i][j];
                }
                nb[i] = acc;
                bm = acc > bm ? acc : bm;
            }
            const R ti = tot > 0 ? Ops<R>::rcp(tot) : (R)0;
            const R bi = bm > 0 ? Ops<R>::rcp(bm) : (R)1;
#pragma unroll
            for (int i = 0; i < S; ++i) {
                // (tot = sum_i alpha^_{t-1}(i) nb(i): the scale on which alpha^ beta sums to 1)
                beta[i] = tot > 0 ? Ops<R>::finite(nb[i] * ti) : nb[i] * bi;
#pragma unroll
                for (int j = 0; j < S; ++j) xi[i][j] += e[i][j] * ti;
            }
        }
    }

    // ---- sums over the lanes (float64, fixed order) into the utterance's record: every lane
    // leaves its terms in LDS, lane q adds the 64 terms of quantity q (four running sums over
    // the lanes 4k + j, then (s0 + s1) + (s2 + s3)) -- a sixth of the time of butterfly sums
    // of 64-bit values through lane shuffles
    {
        constexpr int NQ = 2 * S + S * S + 1;
        auto put = [&](int q, double v) { red[q * kRedStride + lane] = v; };
#pragma unroll
        for (int i = 0; i < S; ++i) {
            put(i, (double)post[i]);
            put(S + i, (double)start[i]);
#pragma unroll
            for (int j = 0; j < S; ++j) put(2 * S + i * S + j, (double)xi[i][j]);
        }
        put(NQ - 1, lp);
        wave_mem_sync();
        if (lane < NQ) {
            const double* row = red + lane * kRedStride;
            double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
#pragma unroll 4
            for (int k = 0; k < 64; k += 4) {
                a0 += row[k];
                a1 += row[k + 1];
                a2 += row[k + 2];
                a3 += row[k + 3];
            }
            const double tot = (a0 + a1) + (a2 + a3);
            const int q = lane;
            if (q < S) rec[L.post + q] = with_obs ? tot : 0.0;
            else if (q < 2 * S) rec[L.start + (q - S)] = tot;
            else if (q < NQ - 1) rec[L.trans + (q - 2 * S)] = tot;
            else rec[L.logprob] = tot;
        }
        if (lane == 0) rec[L.nobs] = 1.0;
    }
    if (!with_obs) return;

    // ---- 5. gamma^T [x | x^2]: lane = feature column.  With at most 32 features the wave holds
    // several columns' worth of lanes: lane (part, d) adds the frames of its part, the parts are
    // then added in order through LDS.  Wider rows: lanes d, d + 64, ..., all frames each.
    wave_mem_sync();                                       // gamma of every frame is in the scratch
    auto column = [&](int d, int ta, int tb, R (&o1)[S], R (&o2)[S]) {
#pragma unroll
        for (int s = 0; s < S; ++s) o1[s] = o2[s] = 0;
        const R* x = X + (size_t)f0 * D + d;
        int t = ta;
        for (; t + 8 <= tb; t += 8) {
            R xv[8], gv[8][S];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                xv[k] = x[(size_t)(t + k) * D];
#pragma unroll
                for (int s = 0; s < S; ++s) gv[k][s] = ga_at(t + k)[s];
            }
#pragma unroll
            for (int k = 0; k < 8; ++k)
#pragma unroll
                for (int s = 0; s < S; ++s) {
                    const R gx = gv[k][s] * xv[k];
                    o1[s] += gx;
                    o2[s] += gx * xv[k];
                }
        }
        for (; t < tb; ++t) {
            const R xv = x[(size_t)t * D];
#pragma unroll
            for (int s = 0; s < S; ++s) {
                const R gx = ga_at(t)[s] * xv;
                o1[s] += gx;
                o2[s] += gx * xv;
            }
        }
    };
    const int parts = D <= 32 ? 64 / D : 1;
    if (parts > 1) {
        const int np = parts < 8 ? parts : 8;              // (8 x 2S x 8 columns at most: fits `red`)
        const int part = lane / D, d = lane % D;
        const int per = (T + np - 1) / np;
        R o1[S], o2[S];
        const bool mine = part < np;
        column(d, mine ? (part * per < T ? part * per : T) : 0, mine ? ((part + 1) * per < T ? (part + 1) * per : T) : 0,
               o1, o2);
        wave_mem_sync();                                   // (the sums above have been read out of `red`)
        if (mine) {
#pragma unroll
            for (int s = 0; s < S; ++s) {
                red[((part * 2 * S) + s) * D + d] = (double)o1[s];
                red[((part * 2 * S) + S + s) * D + d] = (double)o2[s];
            }
        }
        wave_mem_sync();
        if (part == 0) {
#pragma unroll
            for (int s = 0; s < S; ++s) {
                double a = 0.0, b = 0.0;
                for (int q = 0; q < np; ++q) {
                    a += red[((q * 2 * S) + s) * D + d];
                    b += red[((q * 2 * S) + S + s) * D + d];
                }
                rec[L.obs + s * D + d] = a;
                rec[L.obs2 + s * D + d] = b;
            }
        }
    } else {
        for (int d = lane; d < D; d += 64) {
            R o1[S], o2[S];
            column(d, 0, T, o1, o2);
#pragma unroll
            for (int s = 0; s < S; ++s) {
                rec[L.obs + s * D + d] = (double)o1[s];
                rec[L.obs2 + s * D + d] = (double)o2[s];
            }
        }
    }
}

// out[m][r] = sum of records[u][r] over the utterances of model m.  One wave per run of 64
// consecutive entries r of one model: lane = entry, utterances in order, four loads in flight
// (a fixed order of additions: the same sums on every run).
__global__ __launch_bounds__(64) void wave_reduce_kernel(const double* __restrict__ records, int64_t U,
                                                         const int32_t* __restrict__ utt_model, int M,
                                                         int rec_size, double* __restrict__ out) {
    const int runs = (rec_size + 63) / 64;
    const int m = blockIdx.x / runs, r = (blockIdx.x % runs) * 64 + threadIdx.x;
    if (r >= rec_size) return;
    double s = 0.0;
    int64_t u = 0;
    for (; u + 4 <= U; u += 4) {
        double v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            // (every record is read, a select keeps the model's own: no load under a branch)
            const double x = records[(size_t)(u + k) * rec_size + r];
            v[k] = (utt_model ? utt_model[u + k] : 0) == m ? x : 0.0;
        }
        s += v[0];
        s += v[1];
        s += v[2];
        s += v[3];
    }
    for (; u < U; ++u)
        if ((utt_model ? utt_model[u] : 0) == m) s += records[(size_t)u * rec_size + r];
    out[(size_t)m * rec_size + r] = s;
}

template <typename R>
int launch_wave(srx_context* h, const void* X, const int64_t* off, int64_t U, int64_t n_frames,
                const int32_t* utt_model, const srx_models_t* md, double* out_stats, hipStream_t st) {
    const int M = md->n_models, S = md->n_states, D = md->n_features;
    const int rec_size = (int)srx_stats_layout(S, D).size;
    R* scratch;
    double* records;
    int rc;
    if ((rc = srx_ws_ensure(h, WS_LATTICE, sizeof(R) * (size_t)n_frames * (3 * S + 1), (void**)&scratch))) return rc;
    if ((rc = srx_ws_ensure(h, WS_PARTIAL, sizeof(double) * (size_t)U * rec_size, (void**)&records))) return rc;
    const size_t lds_tab = sizeof(R) * ((size_t)2 * S * ((D + 15) / 16 * 16) + S);
    // (frames of the longest utterance this call can hold, up to 48 KB: three workgroups per CU)
    const size_t per_frame = sizeof(R) * (3 * (size_t)S + 1);
    // (behind them 65 doubles per quantity summed over the lanes: at most 25 x 65 x 8 = 13 KB)
    const size_t red_bytes = sizeof(double) * 65 * (2 * (size_t)S + (size_t)S * S + 1);
    size_t room = (size_t)44 * 1024 > lds_tab ? (size_t)44 * 1024 - lds_tab : 0;
    if ((size_t)n_frames * per_frame < room) room = (size_t)n_frames * per_frame;
    const int lds_frames = (int)(room / per_frame);
    const int red_offset = (int)((lds_tab + (size_t)lds_frames * per_frame + 15) / 16 * 16);
    const size_t lds = (size_t)red_offset + red_bytes;
    const int with_obs = h->estep_stats != SRX_STATS_TRANSITIONS;
#define SRX_CASE(s_)                                                                               \
    if (S == s_)                                                                                   \
        hipLaunchKernelGGL((estep_wave_kernel<R, s_>), dim3((unsigned)U), dim3(64), lds, st, (const R*)X, \
                           off, U, utt_model, *md, with_obs, scratch, records, rec_size, lds_frames, red_offset);
    SRX_CASE(1) SRX_CASE(2) SRX_CASE(3) SRX_CASE(4)
#undef SRX_CASE
    hipLaunchKernelGGL(wave_reduce_kernel, dim3((unsigned)(M * ((rec_size + 63) / 64))), dim3(64), 0, st,
                       (const double*)records, U, utt_model, M, rec_size, out_stats);
    SRX_HIP(hipGetLastError());
    return SRX_OK;
}

}  // namespace

// Small jobs only: at most kMaxUtts utterances (one wave each; beyond that the kernels that put
// several utterances into a wave are the faster ones), at most 4 states.  SRX_EUNSUPPORTED
// otherwise, and the caller goes on to its other paths.
int srx_wave_estep(srx_context* h, const void* X, int dtype, const int64_t* off, int64_t U, int64_t n_frames,
                   const int32_t* utt_model, const srx_models_t* md, double* out_stats, hipStream_t st) {
    constexpr int64_t kMaxUtts = 2048;
    const int S = md->n_states, D = md->n_features;
    // (n_frames: the row exponents of a product over T frames stay far inside 32 bits)
    if (S > 4 || D > 256 || U <= 0 || U > kMaxUtts || n_frames <= 0 || n_frames > (1 << 17)) return SRX_EUNSUPPORTED;
    return dtype == SRX_F64 ? launch_wave<double>(h, X, off, U, n_frames, utt_model, md, out_stats, st)
                            : launch_wave<float>(h, X, off, U, n_frames, utt_model, md, out_stats, st);
}
