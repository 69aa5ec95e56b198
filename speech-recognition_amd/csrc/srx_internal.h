// Internal declarations shared by the libsrx translation units (not installed).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

#include "srx.h"

// Scratch buffers owned by a context; grown on demand, never shrunk.
enum srx_ws_slot {
    WS_PARAMS = 0,   // prepared model tables in the compute dtype
    WS_LOGLIK,       // [n_frames, n_cols] emission log-densities
    WS_LATTICE,      // [n_frames, S] scaled forward lattice, then posteriors
    WS_SCALE,        // [n_frames] per-frame scaling constants
    WS_PARTIAL,      // per-workgroup partial statistics
    WS_BACKPTR,      // [n_frames, S] Viterbi back-pointers (uint8)
    WS_CEPSTRA,      // [raw frames, n_ceps] cepstra before the delta stage
    WS_MISC,
    WS_NUM
};

struct srx_context {
    int device;
    int n_cu;
    int exact_f32;   // SRX_OPT_EXACT_F32: keep the float32 paths on exact-f32 matrix instructions
    int estep_path;  // SRX_OPT_ESTEP_PATH
    int viterbi_path;  // SRX_OPT_VITERBI_PATH
    int estep_stats;   // enum srx_estep_stats of the E-step call being queued
    void* ws[WS_NUM];
    size_t ws_bytes[WS_NUM];
};

struct srx_mfcc_plan {
    srx_context* ctx;
    srx_mfcc_cfg_t cfg;
    int log2n;             // log2(nfft)
    // device tables (both precisions are kept so either dtype can run)
    void* window[2];       // [frame_len]
    void* twiddle[2];      // [nfft/2] complex (cos, -sin)
    void* mel_w[2];        // [nnz] filter weights, filter-major
    int32_t* mel_bin;      // [nnz] spectrum bin of each weight
    int32_t* mel_ptr;      // [n_mel+1] CSR row pointers
    void* dct[2];          // [n_ceps, n_mel]
    int nnz;
    // extras of the register-FFT kernel (float32)
    void* tw_full;         // [nfft] complex exp(-2 pi i k / nfft)
    int32_t* mel_start;    // [n_mel] first bin of each filter
    int fast_ok;           // every filter is one contiguous run of bins
    // mel / DCT stages of that kernel as 4x4 matrix-core blocks (see mfcc_fast.hip)
    void* mm_w;            // [mm_nq][64][4] mel weights per (quad of bins, lane)
    void* mm_dct;          // [64][mm_kk] DCT coefficients per lane
    int32_t* mm_meta;      // [16] first bin of a block, [16] first block and [16] block count of a filter group
    int mm_nq, mm_kk, mm_groups, mm_ok;
    // mel stage of the 2048-point register-FFT kernel (see mfcc_wide.hip)
    void* wd_w;            // [wd_lq][64][4] filter weights of a quad of bins
    int32_t* wd_meta;      // [wd_lq][64] quad index | partial-sum slot << 16 | last quad of its filter << 31
    int32_t* wd_ptr;       // [n_mel + 1] partial sums of a filter
    void* wd_dct;          // [n_ceps][128] DCT-II rows, zero past n_mel
    int wd_lq, wd_ok;
};

void srx_set_error(const char* fmt, ...);
int srx_ws_ensure(srx_context* h, int slot, size_t bytes, void** out);

#define SRX_HIP(call)                                                            \
    do {                                                                         \
        hipError_t e__ = (call);                                                 \
        if (e__ != hipSuccess) {                                                 \
            srx_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), \
                          __FILE__, __LINE__);                                   \
            return SRX_EHIP;                                                     \
        }                                                                        \
    } while (0)

#define SRX_REQUIRE(cond, ...)           \
    do {                                 \
        if (!(cond)) {                   \
            srx_set_error(__VA_ARGS__);  \
            return SRX_EINVAL;           \
        }                                \
    } while (0)

// Layout of one packed statistics record (doubles).
struct srx_stats_layout {
    int64_t nobs, logprob, start, trans, post, obs, obs2, size;
    __host__ __device__ srx_stats_layout(int S, int D) {
        nobs = 0;
        logprob = 1;
        start = 2;
        trans = start + S;
        post = trans + (int64_t)S * S;
        obs = post + S;
        obs2 = obs + (int64_t)S * D;
        size = obs2 + (int64_t)S * D;
    }
};

// Implemented in hmm_generic.hip
int srx_generic_emission(srx_context* h, const void* X, int dtype, int64_t n_frames,
                         const srx_models_t* models, void* out, hipStream_t st);
int srx_generic_score(srx_context* h, const void* X, int dtype, const int64_t* off,
                      int64_t U, int64_t n_frames, const srx_models_t* models,
                      double* out, hipStream_t st);
int srx_generic_viterbi(srx_context* h, const void* X, int dtype, const int64_t* off,
                        int64_t U, int64_t n_frames, const int32_t* utt_model,
                        const srx_models_t* models, int32_t* path, double* logprob,
                        hipStream_t st);
int srx_generic_posteriors(srx_context* h, const void* X, int dtype, const int64_t* off, int64_t U,
                           int64_t n_frames, const int32_t* utt_model, const srx_models_t* md,
                           void* gamma, double* lp, hipStream_t st);
int srx_generic_estep(srx_context* h, const void* X, int dtype, const int64_t* off,
                      int64_t U, int64_t n_frames, const int32_t* utt_model,
                      const int64_t* utt_order, const srx_models_t* models,
                      double* out_stats, hipStream_t st);
int srx_generic_mstep(srx_context* h, const double* stats, const srx_priors_t* pr,
                      int mode, const srx_models_t* models, hipStream_t st);
// Float32 E-step of small jobs (at most 8 states, a few thousand utterances, or any number
// when the context's SRX_OPT_ESTEP_PATH asks for it): one lane per utterance;
// SRX_EUNSUPPORTED for every other job.
int srx_lane_estep(srx_context* h, const void* X, const int64_t* off, int64_t U, int64_t n_frames,
                   const int32_t* utt_model, const int64_t* utt_order, const srx_models_t* md,
                   double* out_stats, hipStream_t st);
// E-step of the smallest jobs (at most 2048 utterances, at most 4 states, either dtype): one
// wave per utterance, the recursions cut into 64 pieces joined by a scan of transfer-matrix
// products (estep_wave.hip); SRX_EUNSUPPORTED for every other job.
int srx_wave_estep(srx_context* h, const void* X, int dtype, const int64_t* off, int64_t U, int64_t n_frames,
                   const int32_t* utt_model, const srx_models_t* md, double* out_stats, hipStream_t st);
// The same for float64 data (b~ of every frame exponentiated in a pass of its own).
int srx_lane_estep_f64(srx_context* h, const void* X, const int64_t* off, int64_t U, int64_t n_frames,
                       const int32_t* utt_model, const int64_t* utt_order, const srx_models_t* md,
                       double* out_stats, hipStream_t st);

// Implemented in score_mfma.hip: returns SRX_EUNSUPPORTED when the shape has no
// matrix-core kernel, so the caller can take the generic path.
// float32 emission log-densities on the matrix cores (split-bf16); SRX_EUNSUPPORTED
// when the shape or the exact-f32 option rules it out.
int srx_mfma_emission(srx_context* h, const void* X, int64_t n_frames, const srx_models_t* md,
                      float* out, hipStream_t st);
// The same for each utterance's own model only: out[frame][state] (S <= 16).  Its
// tables live in the WS_MISC slot.
int srx_mfma_emission_utt(srx_context* h, const void* X, const int64_t* off, int64_t U,
                          int64_t n_frames, const int32_t* utt_model, const srx_models_t* md,
                          float* out, hipStream_t st);
// Same output, plain float32 products (no split-bf16).
int srx_mfma_emission_utt_f32(srx_context* h, const void* X, const int64_t* off, int64_t U,
                          int64_t n_frames, const int32_t* utt_model, const srx_models_t* md,
                          float* out, hipStream_t st);
int srx_mfma_score(srx_context* h, const void* X, const int64_t* off, int64_t U,
                   int64_t n_frames, const srx_models_t* models, double* out,
                   hipStream_t st);
// The same with the prepared tables in memory the caller brings (`tables`, at least
// `tables_bytes` long, 16-byte aligned) instead of the context's workspace: nothing is
// allocated.  Only the split-bf16 kernels; SRX_EINVAL when the tables do not fit.
int srx_mfma_score_tables(srx_context* h, const void* X, const int64_t* off, int64_t U,
                          int64_t n_frames, const srx_models_t* models, double* out, void* tables,
                          size_t tables_bytes, hipStream_t st, const int64_t* order = nullptr);

// Implemented in score_f64.hip: float64 scoring on the f64 matrix instruction (<= 8 states,
// <= 48 features), else SRX_EUNSUPPORTED.
int srx_mfma_score_f64(srx_context* h, const void* X, const int64_t* off, int64_t U,
                       int64_t n_frames, const srx_models_t* models, double* out, hipStream_t st);

// Float64 log-densities of each utterance under its own model on the f64 matrix
// instruction, out[frame][state] (<= 16 states, <= 48 features), else SRX_EUNSUPPORTED.
int srx_mfma_emission_utt_f64(srx_context* h, const void* X, const int64_t* off, int64_t U,
                              int64_t n_frames, const int32_t* utt_model, const srx_models_t* md,
                              double* out, hipStream_t st);

// The same for all state columns of all models, out[frame][m * S + s].
int srx_mfma_emission_f64(srx_context* h, const void* X, int64_t n_frames, const srx_models_t* md,
                          double* out, hipStream_t st);

// Observation sums of the float64 E-step as f64 matrix products into
// partial[n_slices][M][S + 2 S D] (the layout of the generic kernel), else SRX_EUNSUPPORTED.
int srx_mfma_obs_stats_f64(srx_context* h, const void* X, const void* gamma, const int64_t* off,
                           int64_t U, const int32_t* utt_model, const int64_t* utt_order,
                           const srx_models_t* md, double* partial, int n_slices, hipStream_t st);

// Implemented in large_model.hip.
int srx_large_gmm_emission(srx_context* h, const void* X, int dtype, int64_t n_frames,
                           const srx_gmm_t* gm, void* out, hipStream_t st);
int srx_large_viterbi_dense(srx_context* h, const void* ll, int dtype, const int64_t* off, int64_t U,
                            int64_t n_frames, int S, const double* pi, const double* A,
                            int32_t* path, double* lp, hipStream_t st);

// Implemented in estep_fused.hip (float32, S <= 16, D <= 40), else SRX_EUNSUPPORTED.
int srx_fused_estep(srx_context* h, const void* X, const int64_t* off, int64_t U, int64_t n_frames,
                    const int32_t* utt_model, const int64_t* utt_order, const srx_models_t* models,
                    double* out_stats, hipStream_t st);
int srx_fused_posteriors(srx_context* h, const void* X, const int64_t* off, int64_t U,
                         int64_t n_frames, const int32_t* utt_model, const srx_models_t* md,
                         float* out_post, double* out_lp, hipStream_t st);

// Implemented in viterbi_fused.hip (float32, at most 8 states, D <= 48): emission and max-plus
// recursion in one kernel, else SRX_EUNSUPPORTED.  logA / logpi: float32 device tables.
int srx_fused_viterbi(srx_context* h, const void* X, const int64_t* off, int64_t U, int64_t n_frames,
                      const int32_t* utt_model, const srx_models_t* md, const float* logA,
                      const float* logpi, int32_t* path, double* lp, hipStream_t st);

// Implemented in mfcc_wide.hip (float32, centred 2048-point frames): log-mel rows
// [rows, n_mel] and the cepstra of the unclamped rows [rows, n_ceps] on the virtual sample axis
// and, in the WS_MISC slot, each row's largest and smallest log-mel value; else
// SRX_EUNSUPPORTED.
int srx_mfcc_wide_cepstra(srx_context* h, srx_mfcc_plan* plan, const float* audio,
                          const int64_t* sample_offsets, int64_t U, int64_t total, float* logmel,
                          float* cep, float** rowmax_out, float** rowmin_out, hipStream_t st);

// Implemented in mfcc_fast.hip (float32, nfft 512 / 1024), else SRX_EUNSUPPORTED.
int srx_mfcc_fast_cepstra(srx_context* h, srx_mfcc_plan* plan, const float* audio,
                          const int64_t* sample_offsets, int64_t U, int64_t total, float* cep,
                          hipStream_t st);
