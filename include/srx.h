/*
 * srx -- MI355X (gfx950) kernels for the MFCC + diagonal-Gaussian HMM hot path.
 *
 * C ABI of libsrx.so.  The reference (Tejashreenk/Speech-Recognition) is pure
 * Python and has no FFI of its own for this path; each entry point below names
 * the reference function(s) whose arithmetic it replaces.  A Python binding is
 * a ctypes stub over exactly these symbols.
 *
 * Conventions
 *   - every pointer argument except `cfg`, `models`, `priors` and out-handles
 *     is a DEVICE pointer owned by the caller;
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream); all
 *     entry points enqueue work on it and return without synchronising;
 *   - return value: 0 on success, negative srx_status on failure, the message
 *     is available from srx_last_error() (thread-local);
 *   - dtype selects the arithmetic and the element type of X / audio / feature
 *     buffers: SRX_F32 is the fast path, SRX_F64 the validation ("parity")
 *     path.  Model parameters, scores, statistics are always double.
 *   - utterances are ragged: `utt_offsets` holds U+1 int64 frame (or sample)
 *     offsets into the packed buffer, offsets[0] == 0 and offsets[U] ==
 *     n_frames (n_samples), which the caller also passes by value so that the
 *     library can size its scratch without reading device memory.
 */
#ifndef SRX_H
#define SRX_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct srx_context* srx_handle_t;
typedef struct srx_mfcc_plan* srx_mfcc_plan_t;

enum srx_dtype { SRX_F32 = 0, SRX_F64 = 1 };

enum srx_status {
    SRX_OK = 0,
    SRX_EINVAL = -1,       /* bad shape / argument (the reference raises ValueError) */
    SRX_EHIP = -2,         /* a HIP runtime call failed */
    SRX_ENOMEM = -3,       /* workspace allocation failed */
    SRX_EUNSUPPORTED = -4  /* shape outside what the kernels are built for */
};

enum srx_mstep_mode {
    SRX_MSTEP_FULL = 0,             /* do_mstep, my_hmmlearn/hmm.py:165-184 */
    SRX_MSTEP_TRANSITIONS_ONLY = 1  /* do_em_mstep only (what fit() calls, hmm.py:213) */
};

/* A bank of n_models HMMs that share n_states and n_features.  Device pointers,
 * C-contiguous doubles: attributes startprob_ / transmat_ / means_ / _covars_
 * of MyGaussianHMM (my_hmmlearn/hmm.py:116-134). */
typedef struct {
    int32_t n_models;
    int32_t n_states;
    int32_t n_features;
    double* startprob; /* [n_models, S]    */
    double* transmat;  /* [n_models, S, S] */
    double* means;     /* [n_models, S, D] */
    double* covars;    /* [n_models, S, D] */
} srx_models_t;

/* Constructor defaults of MyGaussianHMM, my_hmmlearn/hmm.py:75-82. */
typedef struct {
    double startprob_prior; /* 1.0  */
    double transmat_prior;  /* 1.0  */
    double means_prior;     /* 0.0  */
    double means_weight;    /* 0.0  */
    double covars_prior;    /* 1e-2 */
    double covars_weight;   /* 1.0  */
} srx_priors_t;

/* Front-end constants.  The reference fixes nfft=1024, n_mel=26,
 * mel_sample_rate=16000 (mfcc.py:25,40,47), n_ceps=13 (mfcc.py:78),
 * delta_order=1, delta_n=2 (mfcc.py:88,165), log_mult=20 (mfcc.py:44) and
 * z-scores every utterance (HMM_final.py:22). frame_len/hop are
 * int(0.025*sr) / int(0.01*sr) (mfcc.py:18-19). */
typedef struct {
    int32_t sample_rate;
    int32_t frame_len;
    int32_t hop;
    int32_t nfft;            /* power of two, 64..4096 */
    int32_t n_mel;           /* <= 128 */
    int32_t mel_sample_rate;
    int32_t n_ceps;          /* <= n_mel, <= 32 */
    int32_t delta_order;     /* 0, 1 or 2 */
    int32_t delta_n;         /* regression half-width, reference: 2 */
    int32_t zscore;          /* 0 / 1 */
    double log_mult;         /* multiplies log10(mel energy) */
    /* The fields below select the conventions of librosa.feature.mfcc, the
     * third-party front-end of extract_mfcc (HMM_lib.py:11-15, HMM_final.py:11-15);
     * all zero = the reference's own mfcc.py. */
    int32_t window;          /* 0: symmetric Hamming of frame_len (mfcc.py:21);
                                1: periodic Hann of frame_len, centred in the nfft points */
    int32_t center;          /* 0: frame t starts at sample t*hop, tail dropped (mfcc.py:19-20);
                                1: frame t is the nfft samples centred on t*hop, zeros outside the
                                   signal, 1 + n/hop frames; 2: the same with reflected samples */
    int32_t mel_style;       /* 0: HTK mel, triangles between floored bins (mfcc.py:47-61);
                                1: Slaney mel, triangles over bin frequencies, area-normalised */
    int32_t power_norm;      /* 0: |X|^2 / nfft (mfcc.py:38); 1: |X|^2 */
    double log_floor;        /* 0: exact zeros become eps (mfcc.py:43); > 0: energies below it
                                are raised to it before the logarithm */
    double top_db;           /* 0: off; > 0: log-mel values of an utterance are raised to
                                (their maximum - top_db) before the DCT */
} srx_mfcc_cfg_t;

/* ---- library / context ------------------------------------------------- */
int srx_version(void);
const char* srx_last_error(void);
/* One context per GPU / rank; owns the scratch workspace the kernels need. */
int srx_create(int device, srx_handle_t* out);
int srx_destroy(srx_handle_t h);
/* Context options.  SRX_OPT_EXACT_F32 = 1 keeps the SRX_F32 scoring path on
 * exact-f32 matrix instructions; the default (0) evaluates the emission
 * contraction with split-bf16 operands (three partial products, error ~1e-6
 * relative on a log-likelihood), which is several times faster.
 * SRX_OPT_ESTEP_PATH chooses the kernels of the E-step: SRX_ESTEP_AUTO (0, default) takes
 * one WAVE per utterance for the smallest jobs (at most 2048 utterances and 4 states, either
 * dtype: the reference's own training runs; SRX_ESTEP_WAVE (3) asks for it wherever those
 * limits allow, the other values never take it), then for SRX_F32
 * one lane per utterance for small jobs (at most 8 states and, by the
 * number of states, 3072 to 8192 utterances) and the fused matrix-core kernel otherwise; SRX_ESTEP_FUSED (1) and
 * SRX_ESTEP_LANE (2) ask for one of the two wherever its limits allow (tests, timing).
 * The results agree to float32 rounding.  SRX_F64 data: small jobs take one lane per
 * utterance too (at most 4 states, 4096 utterances); SRX_ESTEP_FUSED keeps them on the
 * sub-wave chain kernels every other float64 job runs on.
 * SRX_OPT_VITERBI_PATH chooses the kernels of SRX_F32 srx_viterbi for models of at most 8 states
 * and at most 48 features: SRX_VITERBI_AUTO (0, default) and SRX_VITERBI_FUSED (1) decode in
 * one kernel that keeps the log-densities on the chip (it measured faster than the other path
 * at every batch size from 128 utterances up), SRX_VITERBI_SPLIT (2) in two (log-densities to
 * memory, then the recursion; tests, timing).  Same states and log-probabilities, bit for bit. */
enum srx_option { SRX_OPT_EXACT_F32 = 1, SRX_OPT_ESTEP_PATH = 2, SRX_OPT_VITERBI_PATH = 3 };
/* Which statistics a caller of srx_estep_stats_select will use: SRX_STATS_ALL (0) or
 * SRX_STATS_TRANSITIONS (1) -- nobs, logprob, start and trans only, what the reference's own
 * fit() re-estimates from (do_em_mstep, my_hmmlearn/hmm.py:435-451; srx_mstep in
 * SRX_MSTEP_TRANSITIONS_ONLY mode).  The post, obs and obs2 entries of the records are then
 * all zero: the kernels leave out their pass over the features. */
enum srx_estep_stats { SRX_STATS_ALL = 0, SRX_STATS_TRANSITIONS = 1 };
enum srx_estep_path { SRX_ESTEP_AUTO = 0, SRX_ESTEP_FUSED = 1, SRX_ESTEP_LANE = 2, SRX_ESTEP_WAVE = 3 };
enum srx_viterbi_path { SRX_VITERBI_AUTO = 0, SRX_VITERBI_FUSED = 1, SRX_VITERBI_SPLIT = 2 };
int srx_set_option(srx_handle_t h, int option, int value);
/* Bytes of scratch currently held by the context (diagnostics). */
int64_t srx_workspace_bytes(srx_handle_t h);

/* Doubles per model in a packed statistics record:
 * [nobs, logprob, start(S), trans(S*S), post(S), obs(S*D), obs2(S*D)]
 * = the dict of initialize_sufficient_statistics (hmm.py:224-235) plus the
 * summed log-probability do_estep returns (hmm.py:433). */
int64_t srx_stats_size(int32_t n_states, int32_t n_features);

/* ---- front-end: replaces mfcc.compute_mfcc / delta / the concat in
 * calculate_mfcc_for_onerecording (mfcc.py:82-96, :141-169) and the
 * StandardScaler of extract_mfcc_calc (HMM_final.py:18-23) ---------------- */
int srx_mfcc_plan_create(srx_handle_t h, const srx_mfcc_cfg_t* cfg, srx_mfcc_plan_t* out);
int srx_mfcc_plan_destroy(srx_mfcc_plan_t plan);
/* Frames produced for an utterance of n_samples (after the delta trim). */
int64_t srx_mfcc_out_frames(const srx_mfcc_cfg_t* cfg, int64_t n_samples);
/* audio: packed samples (dtype), sample_offsets[U+1]; out_feats: packed
 * [sum frames, n_ceps*(1+delta_order)] (dtype), frame_offsets[U+1] must hold
 * the cumulative srx_mfcc_out_frames of every utterance. */
int srx_mfcc_batch(srx_handle_t h, srx_mfcc_plan_t plan, const void* audio, int dtype,
                   const int64_t* sample_offsets, int64_t n_utts, int64_t n_samples,
                   void* out_feats, const int64_t* frame_offsets, void* stream);
/* The dense (n_mel, nfft/2+1) filter matrix the plan is built from, written to
 * HOST memory: mel_filterbank, mfcc.py:47-61 (diagnostic, no GPU needed). */
int srx_mfcc_filterbank_host(const srx_mfcc_cfg_t* cfg, double* out_dense);

/* ---- end-pointing in front of the front-end: AudioMonitor.start_saved_audio with
 * compute_short_time_energy, compute_zero_crossing_rate_sum, detect_if_noise,
 * detect_speech_start_2, calc_N1_prime, calc_N2_prime
 * (new_speech_detect.py:44-61, :80-89, :197-334), for many recordings per call. */
typedef struct {
    int32_t sample_rate;     /* Hz */
    int32_t block_ms;        /* block length, reference: 10 */
    int32_t zcr_cap;         /* upper bound of the zero-crossing threshold, reference: 25 */
    int32_t reserved;
    double noise_peak;       /* a recording whose peak block energy is below is noise, reference: 4.5 */
} srx_endpoint_cfg_t;
/* Blocks of a recording of n_samples: ceil(n_samples / int(sample_rate * block_ms / 1000)). */
int64_t srx_endpoint_blocks(const srx_endpoint_cfg_t* cfg, int64_t n_samples);
/* audio: packed samples (dtype), sample_offsets[R+1]; block_offsets[R+1]: cumulative
 * srx_endpoint_blocks of every recording (device, int64), n_blocks their total.
 * out_energy[n_blocks] (double) = sum |x| per block, out_zcr[n_blocks] = sign changes
 * inside the block.  Per recording: out_counts[r] segments found (may exceed
 * max_segments; the first max_segments are stored), out_segments[r][k] =
 * {N1', N1, N2, N2'} in blocks, out_thresholds[r] = {ITU, ITL, IZCT}, or
 * {peak energy, 0, 0} when nothing is found. */
int srx_endpoint_detect(srx_handle_t h, const void* audio, int dtype, const int64_t* sample_offsets,
                        int64_t n_recordings, int64_t n_samples, const srx_endpoint_cfg_t* cfg,
                        const int64_t* block_offsets, int64_t n_blocks, double* out_energy,
                        int32_t* out_zcr, int32_t* out_segments, int32_t* out_counts,
                        double* out_thresholds, int32_t max_segments, void* stream);

/* ---- emissions: _log_multivariate_normal_density_diag, hmm.py:154-163 ----
 * out[f, m*S + s] for every frame and every state of every model (dtype). */
int srx_emission_loglik(srx_handle_t h, const void* X, int dtype, int64_t n_frames,
                        const srx_models_t* models, void* out, void* stream);

/* ---- scoring: score / score_scaling + utils.forward_scaling,
 * hmm.py:186-196, :308-323, utils.py:132-150; the per-word loop of recognize
 * (HMM_final.py:49-52) is the model axis.  out_scores[u, m] (double) is the
 * forward log-probability of utterance u under model m. */
int srx_forward_score(srx_handle_t h, const void* X, int dtype, const int64_t* utt_offsets,
                      int64_t n_utts, int64_t n_frames, const srx_models_t* models,
                      double* out_scores, void* stream);
/* The same with a visiting order (a permutation of 0 .. n_utts-1 on the device): the SRX_F32
 * matrix-core kernel scores 16 consecutive entries of the order in one wave, step by step
 * until the longest of them ends, so an order that puts utterances of similar length side by
 * side (descending length buckets) keeps its matrix columns busy on ragged batches --
 * the batched form of recognize's loop over files (HMM_final.py:134-165), where the files
 * come in any order.  Results are those of srx_forward_score, out_scores[u, m] indexed by
 * utterance, not by position.  The order is used while X is smaller than 4 GiB (one buffer
 * descriptor must reach all of it) and ignored otherwise, for SRX_F64, and with NULL. */
int srx_forward_score_ordered(srx_handle_t h, const void* X, int dtype, const int64_t* utt_offsets,
                              int64_t n_utts, int64_t n_frames, const int64_t* utt_order,
                              const srx_models_t* models, double* out_scores, void* stream);
/* Same contract, always through the shape-generic kernels (any S <= 64, any D);
 * srx_forward_score uses them only for shapes without a matrix-core kernel. */
int srx_forward_score_generic(srx_handle_t h, const void* X, int dtype,
                              const int64_t* utt_offsets, int64_t n_utts, int64_t n_frames,
                              const srx_models_t* models, double* out_scores, void* stream);

/* ---- classification: the label step of recognize, max(scores, key=scores.get)
 * (HMM_final.py:51-52; recognize_lib, HMM_final.py:54-57, HMM_lib.py:50-53), for every row of a
 * score matrix on the device: out_labels[u] (int32) is the first column of scores[u, 0..n_models)
 * whose value is larger than every earlier column's -- on ties the first model in the
 * caller's order wins, as the first key of the reference's dict does. */
int srx_argmax_scores(srx_handle_t h, const double* scores, int64_t n_utts, int32_t n_models,
                      int32_t* out_labels, void* stream);

/* Allocation-free variant of srx_forward_score (same arguments, same results, SRX_F32
 * only).  It has no counterpart in the reference: it is for callers that cannot allocate
 * device memory while they run -- code under HIP graph capture, tight serving loops,
 * harnesses that forbid hipMalloc.  The prepared model tables, which srx_forward_score
 * keeps in the context's hipMalloc'ed workspace, live in a fixed device array of
 * srx_static_scratch_bytes() (256 KiB) that every GPU the library runs on holds until
 * the process ends; no hipMalloc / hipFree is made.  Limits: n_states <= 16,
 * n_features <= 48, default (split-bf16) arithmetic, and a bank whose tables fit the
 * array (about 40 sixteen-row tiles: 80 models of 8 states, 40 of 16 states at 39
 * features); anything else returns SRX_EUNSUPPORTED or, for a bank that is too large,
 * SRX_EINVAL with the sizes in srx_last_error().  All contexts of one GPU share the
 * array.  Calls of one process are put in order by the library: a call on another stream
 * than the previous one makes its stream wait for the previous call's kernels (one event per
 * GPU; the array is looked up with hipGetSymbolAddress in srx_create or, failing that, at the
 * first call, which then returns SRX_EHIP if the lookup fails again).  Not ordered: calls
 * made while their stream is under graph capture, and therefore the launches of such graphs
 * (their owner keeps two of them from overlapping on one GPU), and calls of other processes
 * (each process has its own array). */
int srx_forward_score_static(srx_handle_t h, const void* X, int dtype, const int64_t* utt_offsets,
                             int64_t n_utts, int64_t n_frames, const srx_models_t* models,
                             double* out_scores, void* stream);
int64_t srx_static_scratch_bytes(void);

/* ---- decoding: utils.viterbi, utils.py:243-266 (decode, hmm.py:357-373, is
 * broken in the reference).  utt_model[u] (int32, may be NULL = model 0)
 * selects the model; out_path[f] int32 state per frame, out_logprob[u]. */
int srx_viterbi(srx_handle_t h, const void* X, int dtype, const int64_t* utt_offsets,
                int64_t n_utts, int64_t n_frames, const int32_t* utt_model,
                const srx_models_t* models, int32_t* out_path, double* out_logprob,
                void* stream);

/* ---- posteriors: score_scaling(compute_posteriors=True) +
 * utils.backward_scaling + compute_posteriors_scaling + utils.normalize,
 * hmm.py:308-328, utils.py:169-183, :6-30.  out_posteriors[f, s] (dtype) is the
 * row-normalised product of the scaled forward and backward lattices under the
 * utterance's model utt_model[u] (NULL = model 0); out_logprob[u] the forward
 * log-probability.  out_posteriors also serves as the kernels' scratch while the call's
 * stream work runs (forward vectors are parked in its rows before the posteriors replace
 * them): read it only after that work is done.  SRX_F32, at most 8 states: an array that is
 * 16-byte aligned and smaller than 4 GiB takes the faster kernel; same results otherwise. */
int srx_posteriors(srx_handle_t h, const void* X, int dtype, const int64_t* utt_offsets,
                   int64_t n_utts, int64_t n_frames, const int32_t* utt_model,
                   const srx_models_t* models, void* out_posteriors, double* out_logprob,
                   void* stream);

/* ---- large emission model: S states, each a mixture of n_mix diagonal
 * Gaussians (the per-component density is hmm.py:154-163; the mixture,
 * ln sum_m w_m N_m, is hmmlearn's GMMHMM emission -- the reference itself has
 * single Gaussians only, which is the n_mix = 1, weight 1 case).  All arrays
 * are device pointers, double.  out[f, s] (dtype).  The float32 path runs on the
 * matrix cores for n_mix <= 16 and n_features <= 48. */
typedef struct {
    int32_t n_states, n_mix, n_features;
    const double* weights;   /* [n_states, n_mix] */
    const double* means;     /* [n_states, n_mix, n_features] */
    const double* covars;    /* [n_states, n_mix, n_features] */
} srx_gmm_t;
int srx_gmm_emission_loglik(srx_handle_t h, const void* X, int dtype, int64_t n_frames,
                            const srx_gmm_t* gmm, void* out, void* stream);

/* ---- decoding from given log-densities over a dense transition matrix:
 * utils.viterbi(log_startprob, log_transmat, framelogprob), utils.py:243-266,
 * for models too large for srx_viterbi (n_states <= 1024).  loglik[f, s] (dtype),
 * startprob[n_states] and transmat[n_states, n_states] are probabilities
 * (double, device); zero entries are structural.  Ties take the lowest state. */
int srx_viterbi_dense(srx_handle_t h, const void* loglik, int dtype, const int64_t* utt_offsets,
                      int64_t n_utts, int64_t n_frames, int32_t n_states,
                      const double* startprob, const double* transmat, int32_t* out_path,
                      double* out_logprob, void* stream);

/* ---- Baum-Welch E-step: do_estep + fit_scaling +
 * accumulate_sufficient_statistics(_scaling), hmm.py:419-433, :330-354,
 * :237-253, utils.py:132-218.  Every utterance contributes to the record of
 * its model utt_model[u] (NULL = model 0).  utt_order is the order utterances
 * are visited in; a permutation that groups the utterances of one model (and,
 * inside a model, of similar length) together makes the accumulation cheaper
 * and changes nothing but the summation order.  With utt_order NULL the
 * float32 kernels build that order on the device themselves (a counting sort
 * by model and length bucket; nothing for the caller to do), the float64
 * kernels visit 0..U-1.  out_stats[n_models, stats_size] is overwritten. */
int srx_estep_stats(srx_handle_t h, const void* X, int dtype, const int64_t* utt_offsets,
                    int64_t n_utts, int64_t n_frames, const int32_t* utt_model,
                    const int64_t* utt_order, const srx_models_t* models, double* out_stats,
                    void* stream);

/* The same with the statistics the caller will read named per call (enum srx_estep_stats):
 * srx_estep_stats is this with SRX_STATS_ALL. */
int srx_estep_stats_select(srx_handle_t h, const void* X, int dtype, const int64_t* utt_offsets,
                           int64_t n_utts, int64_t n_frames, const int32_t* utt_model,
                           const int64_t* utt_order, const srx_models_t* models, int which_stats,
                           double* out_stats, void* stream);

/* Same contract, always through the shape-generic kernels. */
int srx_estep_stats_generic(srx_handle_t h, const void* X, int dtype, const int64_t* utt_offsets,
                            int64_t n_utts, int64_t n_frames, const int32_t* utt_model,
                            const int64_t* utt_order, const srx_models_t* models,
                            double* out_stats, void* stream);

/* ---- multi-GPU: the one exchange step of the path.  do_estep sums the statistics of
 * all utterances (hmm.py:427-432); with the utterances sharded over ranks (one process
 * per GPU) that sum is completed by ONE all-reduce of the packed float64 records of all
 * models per EM iteration, after which srx_mstep runs replicated on identical sums.  The
 * reference has no distributed code.  `nccl_comm` is the caller's ncclComm_t (RCCL);
 * the library binds to the RCCL already loaded in the process (else librccl.so.1) at the
 * first call and enqueues ncclAllReduce(stats, stats, count, ncclDouble, ncclSum) on the
 * stream.  In place; `count` doubles (n_models * srx_stats_size). */
int srx_allreduce_stats(srx_handle_t h, void* nccl_comm, double* stats, int64_t count, void* stream);

/* ---- set-up of a fit: what MyGaussianHMM.init takes from the training frames besides
 * the rows it picks as means (my_hmmlearn/hmm.py:130-132): per feature column the mean and
 * the variance with one degree of freedom taken off (the diagonal of np.cov(X.T)), in float64
 * whatever the dtype of X[n_frames, n_features].  out_mean / out_var: n_features doubles. */
int srx_column_moments(srx_handle_t h, const void* X, int dtype, int64_t n_frames, int32_t n_features,
                       double* out_mean, double* out_var, void* stream);

/* ---- M-step: do_em_mstep (hmm.py:435-451) and do_mstep (hmm.py:165-184),
 * applied in place to every model of the bank from its statistics record. */
int srx_mstep(srx_handle_t h, const double* stats, const srx_priors_t* priors, int mode,
              const srx_models_t* models, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SRX_H */
