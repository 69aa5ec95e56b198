// C ABI of libsrx: context, argument checks and dispatch to the kernels.
#include "srx_internal.h"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <mutex>

namespace {
thread_local char g_err[512] = "";

// Table memory of srx_forward_score_static: a fixed device array instead of the
// context's hipMalloc'ed workspace, for callers that must not allocate.  Every GPU
// holds its own copy at its own address; the address is looked up at the first static
// call on a device (the array's name is only a host-side handle for that lookup and is
// never passed to a kernel or a memory routine).  All contexts and streams of a GPU share
// the array, so its users are put in order: a call on another stream than the last one
// first makes its stream wait for an event recorded behind the last call's kernels.
constexpr size_t kStaticScratchBytes = 256 * 1024;
constexpr int kMaxDevices = 64;
static __device__ __align__(256) __attribute__((used)) unsigned char scratch[kStaticScratchBytes];
struct StaticScratch {
    void* ptr = nullptr;
    hipEvent_t done = nullptr;       // behind the kernels of the last call that was not captured
    hipStream_t last = nullptr;
    bool recorded = false;
};
StaticScratch g_scratch[kMaxDevices];
std::mutex g_scratch_mutex;

// With `device` current and g_scratch_mutex held: this device's entry, its address looked up.
int static_scratch_for(int device, StaticScratch** out) {
    SRX_REQUIRE(device >= 0 && device < kMaxDevices, "device %d: at most %d devices are supported",
                device, kMaxDevices);
    StaticScratch& s = g_scratch[device];
    if (!s.ptr) {
        void* p = nullptr;
        SRX_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
        SRX_REQUIRE(p != nullptr, "hipGetSymbolAddress returned a NULL address for device %d", device);
        SRX_HIP(hipEventCreateWithFlags(&s.done, hipEventDisableTiming));
        s.ptr = p;
    }
    *out = &s;
    return SRX_OK;
}
}

void srx_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int srx_ws_ensure(srx_context* h, int slot, size_t bytes, void** out) {
    if (bytes == 0) bytes = 16;
    if (h->ws_bytes[slot] < bytes) {
        if (h->ws[slot]) {
            // hipFree synchronises the device, so queued kernels that still
            // read the old buffer have finished.
            SRX_HIP(hipFree(h->ws[slot]));
            h->ws[slot] = nullptr;
            h->ws_bytes[slot] = 0;
        }
        const size_t want = bytes + bytes / 8 + 256;
        hipError_t e = hipMalloc(&h->ws[slot], want);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            srx_set_error("workspace slot %d: hipMalloc(%zu) failed: %s", slot, want,
                          hipGetErrorString(e));
            return SRX_ENOMEM;
        }
        h->ws_bytes[slot] = want;
    }
    *out = h->ws[slot];
    return SRX_OK;
}

namespace {

int check_models(const srx_models_t* m, const char* who) {
    SRX_REQUIRE(m != nullptr, "%s: models is NULL", who);
    SRX_REQUIRE(m->n_models >= 1 && m->n_states >= 1 && m->n_features >= 1,
                "%s: Shape mismatch (n_models=%d n_states=%d n_features=%d)", who, m->n_models,
                m->n_states, m->n_features);
    SRX_REQUIRE(m->n_states <= 64, "%s: n_states %d > 64 is not supported", who, m->n_states);
    SRX_REQUIRE(m->startprob && m->transmat && m->means && m->covars,
                "%s: NULL model parameter array", who);
    return SRX_OK;
}

int check_common(srx_handle_t h, const void* X, int dtype, int64_t n_utts, int64_t n_frames,
                 const void* off, const char* who) {
    SRX_REQUIRE(h != nullptr, "%s: NULL context", who);
    SRX_REQUIRE(dtype == SRX_F32 || dtype == SRX_F64, "%s: unknown dtype %d", who, dtype);
    SRX_REQUIRE(n_utts >= 0 && n_frames >= 0, "%s: negative size", who);
    SRX_REQUIRE(n_utts == 0 || off != nullptr, "%s: NULL utterance offsets", who);
    SRX_REQUIRE(n_frames == 0 || X != nullptr, "%s: NULL X", who);
    return SRX_OK;
}

}  // namespace

extern "C" {

int srx_version(void) { return 100; }

const char* srx_last_error(void) { return g_err; }

int srx_create(int device, srx_handle_t* out) {
    SRX_REQUIRE(out != nullptr, "srx_create: NULL out");
    int n = 0;
    SRX_HIP(hipGetDeviceCount(&n));
    SRX_REQUIRE(device >= 0 && device < n, "srx_create: device %d out of range (%d visible)",
                device, n);
    SRX_HIP(hipSetDevice(device));
    hipDeviceProp_t prop;
    SRX_HIP(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        srx_set_error("srx_create: device %d is %s; this library is built for gfx950 only", device,
                      prop.gcnArchName);
        return SRX_EUNSUPPORTED;
    }
    {
        // The fixed table array of srx_forward_score_static is looked up here when it can be,
        // so that the first static call (possibly under graph capture) finds it done; a
        // failure is not this call's: that entry point tries again and reports it.
        std::lock_guard<std::mutex> lock(g_scratch_mutex);
        StaticScratch* ss = nullptr;
        if (static_scratch_for(device, &ss) != SRX_OK) {
            (void)hipGetLastError();
            g_err[0] = 0;
        }
    }
    srx_context* h = new srx_context();
    memset(h, 0, sizeof(*h));
    h->device = device;
    h->n_cu = prop.multiProcessorCount;
    *out = h;
    return SRX_OK;
}

int srx_destroy(srx_handle_t h) {
    if (!h) return SRX_OK;
    (void)hipSetDevice(h->device);
    for (int i = 0; i < WS_NUM; ++i)
        if (h->ws[i]) (void)hipFree(h->ws[i]);
    delete h;
    return SRX_OK;
}

int srx_set_option(srx_handle_t h, int option, int value) {
    SRX_REQUIRE(h != nullptr, "srx_set_option: NULL context");
    switch (option) {
        case SRX_OPT_EXACT_F32: h->exact_f32 = value != 0; return SRX_OK;
        case SRX_OPT_ESTEP_PATH:
            SRX_REQUIRE(value >= SRX_ESTEP_AUTO && value <= SRX_ESTEP_WAVE, "srx_set_option: unknown E-step path %d", value);
            h->estep_path = value;
            return SRX_OK;
        case SRX_OPT_VITERBI_PATH:
            SRX_REQUIRE(value >= SRX_VITERBI_AUTO && value <= SRX_VITERBI_SPLIT, "srx_set_option: unknown Viterbi path %d", value);
            h->viterbi_path = value;
            return SRX_OK;
        default: srx_set_error("srx_set_option: unknown option %d", option); return SRX_EINVAL;
    }
}

int64_t srx_workspace_bytes(srx_handle_t h) {
    if (!h) return 0;
    int64_t s = 0;
    for (int i = 0; i < WS_NUM; ++i) s += (int64_t)h->ws_bytes[i];
    return s;
}

// RCCL is bound at run time: a process that already holds one (torch ships its own copy)
// must not get a second, and a single-GPU caller needs none at all.
namespace {
typedef int (*nccl_allreduce_fn)(const void*, void*, size_t, int, int, void*, hipStream_t);
typedef const char* (*nccl_errstr_fn)(int);
nccl_allreduce_fn g_allreduce = nullptr;
nccl_errstr_fn g_errstr = nullptr;

std::once_flag g_rccl_once;
int g_rccl_rc = SRX_OK;
char g_rccl_msg[256] = "";

int bind_rccl_once();
int bind_rccl() {
    std::call_once(g_rccl_once, [] { g_rccl_rc = bind_rccl_once(); });
    if (g_rccl_rc) srx_set_error("%s", g_rccl_msg);
    return g_rccl_rc;
}

int bind_rccl_once() {
    void* lib = nullptr;
    const char* names[] = {"librccl.so", "librccl.so.1"};
    for (const char* n : names)
        if (!lib) lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD);      // the copy already in the process
    if (!lib) lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!lib) {
        snprintf(g_rccl_msg, sizeof(g_rccl_msg), "srx_allreduce_stats: no RCCL library (%s)", dlerror());
        return SRX_EUNSUPPORTED;
    }
    g_errstr = (nccl_errstr_fn)dlsym(lib, "ncclGetErrorString");
    nccl_allreduce_fn f = (nccl_allreduce_fn)dlsym(lib, "ncclAllReduce");
    if (!f) {
        snprintf(g_rccl_msg, sizeof(g_rccl_msg),
                 "srx_allreduce_stats: ncclAllReduce not found in the RCCL library");
        return SRX_EUNSUPPORTED;
    }
    g_allreduce = f;
    return SRX_OK;
}
}  // namespace

int srx_allreduce_stats(srx_handle_t h, void* nccl_comm, double* stats, int64_t count, void* stream) {
    SRX_REQUIRE(h != nullptr, "srx_allreduce_stats: NULL context");
    SRX_REQUIRE(count >= 0, "srx_allreduce_stats: negative count");
    if (count == 0) return SRX_OK;
    SRX_REQUIRE(nccl_comm != nullptr && stats != nullptr, "srx_allreduce_stats: NULL communicator or buffer");
    int rc = bind_rccl();
    if (rc) return rc;
    SRX_HIP(hipSetDevice(h->device));
    // the codes of the RCCL this image ships (rccl/rccl.h; only the two enumerators are
    // taken from the header, the library itself is bound at run time)
    const int kNcclDouble = (int)ncclFloat64, kNcclSum = (int)ncclSum;
    const int res = g_allreduce(stats, stats, (size_t)count, kNcclDouble, kNcclSum, nccl_comm,
                                (hipStream_t)stream);
    if (res != 0) {
        srx_set_error("srx_allreduce_stats: ncclAllReduce failed: %s", g_errstr ? g_errstr(res) : "?");
        return SRX_EHIP;
    }
    return SRX_OK;
}

int64_t srx_stats_size(int32_t n_states, int32_t n_features) {
    return srx_stats_layout(n_states, n_features).size;
}

int srx_emission_loglik(srx_handle_t h, const void* X, int dtype, int64_t n_frames,
                        const srx_models_t* models, void* out, void* stream) {
    int rc = check_common(h, X, dtype, 0, n_frames, nullptr, "srx_emission_loglik");
    if (rc) return rc;
    if ((rc = check_models(models, "srx_emission_loglik"))) return rc;
    if (n_frames == 0) return SRX_OK;
    SRX_REQUIRE(out != nullptr, "srx_emission_loglik: NULL out");
    SRX_HIP(hipSetDevice(h->device));
    if (dtype == SRX_F32) {
        rc = srx_mfma_emission(h, X, n_frames, models, (float*)out, (hipStream_t)stream);
        if (rc != SRX_EUNSUPPORTED) return rc;
    } else {
        rc = srx_mfma_emission_f64(h, X, n_frames, models, (double*)out, (hipStream_t)stream);
        if (rc != SRX_EUNSUPPORTED) return rc;
    }
    return srx_generic_emission(h, X, dtype, n_frames, models, out, (hipStream_t)stream);
}

int srx_forward_score(srx_handle_t h, const void* X, int dtype, const int64_t* utt_offsets,
                      int64_t n_utts, int64_t n_frames, const srx_models_t* models,
                      double* out_scores, void* stream) {
    int rc = check_common(h, X, dtype, n_utts, n_frames, utt_offsets, "srx_forward_score");
    if (rc) return rc;
    if ((rc = check_models(models, "srx_forward_score"))) return rc;
    if (n_utts == 0) return SRX_OK;
    SRX_REQUIRE(out_scores != nullptr, "srx_forward_score: NULL out_scores");
    SRX_HIP(hipSetDevice(h->device));
    if (dtype == SRX_F32) {
        rc = srx_mfma_score(h, X, utt_offsets, n_utts, n_frames, models, out_scores,
                            (hipStream_t)stream);
        if (rc != SRX_EUNSUPPORTED) return rc;
    } else {
        rc = srx_mfma_score_f64(h, X, utt_offsets, n_utts, n_frames, models, out_scores,
                                (hipStream_t)stream);
        if (rc != SRX_EUNSUPPORTED) return rc;
    }
    return srx_generic_score(h, X, dtype, utt_offsets, n_utts, n_frames, models, out_scores,
                             (hipStream_t)stream);
}

int srx_forward_score_ordered(srx_handle_t h, const void* X, int dtype, const int64_t* utt_offsets,
                              int64_t n_utts, int64_t n_frames, const int64_t* utt_order,
                              const srx_models_t* models, double* out_scores, void* stream) {
    if (utt_order == nullptr || dtype != SRX_F32)
        return srx_forward_score(h, X, dtype, utt_offsets, n_utts, n_frames, models, out_scores, stream);
    int rc = check_common(h, X, dtype, n_utts, n_frames, utt_offsets, "srx_forward_score_ordered");
    if (rc) return rc;
    if ((rc = check_models(models, "srx_forward_score_ordered"))) return rc;
    if (n_utts == 0) return SRX_OK;
    SRX_REQUIRE(out_scores != nullptr, "srx_forward_score_ordered: NULL out_scores");
    SRX_HIP(hipSetDevice(h->device));
    rc = srx_mfma_score_tables(h, X, utt_offsets, n_utts, n_frames, models, out_scores, nullptr, 0,
                               (hipStream_t)stream, utt_order);
    if (rc != SRX_EUNSUPPORTED) return rc;
    return srx_generic_score(h, X, dtype, utt_offsets, n_utts, n_frames, models, out_scores,
                             (hipStream_t)stream);
}

int64_t srx_static_scratch_bytes(void) { return (int64_t)kStaticScratchBytes; }

int srx_forward_score_static(srx_handle_t h, const void* X, int dtype, const int64_t* utt_offsets,
                             int64_t n_utts, int64_t n_frames, const srx_models_t* models,
                             double* out_scores, void* stream) {
    int rc = check_common(h, X, dtype, n_utts, n_frames, utt_offsets, "srx_forward_score_static");
    if (rc) return rc;
    if ((rc = check_models(models, "srx_forward_score_static"))) return rc;
    if (dtype != SRX_F32 || models->n_states > 16 || models->n_features > 48 || h->exact_f32) {
        srx_set_error("srx_forward_score_static: only SRX_F32 with n_states <= 16, n_features <= 48 and "
                      "the default (split-bf16) arithmetic has an allocation-free kernel");
        return SRX_EUNSUPPORTED;
    }
    if (n_utts == 0) return SRX_OK;
    SRX_REQUIRE(out_scores != nullptr, "srx_forward_score_static: NULL out_scores");
    SRX_REQUIRE(n_frames > 0, "srx_forward_score_static: no frames");
    SRX_HIP(hipSetDevice(h->device));
    // One call at a time per process queues work on the array; across streams the order on
    // the GPU follows from the event.  A stream under capture is left alone (an event
    // recorded outside the capture cannot be waited for inside it, and one recorded inside
    // says nothing about the graph's later launches): graphs that use this entry point on
    // one GPU have to be ordered by their owner.
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    StaticScratch* ss = nullptr;
    if ((rc = static_scratch_for(h->device, &ss))) return rc;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    SRX_HIP(hipStreamIsCapturing((hipStream_t)stream, &cap));
    const bool eager = cap == hipStreamCaptureStatusNone;
    if (eager && ss->recorded && ss->last != (hipStream_t)stream)
        SRX_HIP(hipStreamWaitEvent((hipStream_t)stream, ss->done, 0));
    rc = srx_mfma_score_tables(h, X, utt_offsets, n_utts, n_frames, models, out_scores,
                               ss->ptr, kStaticScratchBytes, (hipStream_t)stream);
    if (eager && rc == SRX_OK) {
        SRX_HIP(hipEventRecord(ss->done, (hipStream_t)stream));
        ss->last = (hipStream_t)stream;
        ss->recorded = true;
    }
    if (rc == SRX_EUNSUPPORTED)
        srx_set_error("srx_forward_score_static: no allocation-free kernel for n_models=%d n_states=%d "
                      "n_features=%d", models->n_models, models->n_states, models->n_features);
    return rc;
}

int srx_forward_score_generic(srx_handle_t h, const void* X, int dtype,
                              const int64_t* utt_offsets, int64_t n_utts, int64_t n_frames,
                              const srx_models_t* models, double* out_scores, void* stream) {
    int rc = check_common(h, X, dtype, n_utts, n_frames, utt_offsets, "srx_forward_score_generic");
    if (rc) return rc;
    if ((rc = check_models(models, "srx_forward_score_generic"))) return rc;
    if (n_utts == 0) return SRX_OK;
    SRX_REQUIRE(out_scores != nullptr, "srx_forward_score_generic: NULL out_scores");
    SRX_HIP(hipSetDevice(h->device));
    return srx_generic_score(h, X, dtype, utt_offsets, n_utts, n_frames, models, out_scores,
                             (hipStream_t)stream);
}

int srx_viterbi(srx_handle_t h, const void* X, int dtype, const int64_t* utt_offsets,
                int64_t n_utts, int64_t n_frames, const int32_t* utt_model,
                const srx_models_t* models, int32_t* out_path, double* out_logprob,
                void* stream) {
    int rc = check_common(h, X, dtype, n_utts, n_frames, utt_offsets, "srx_viterbi");
    if (rc) return rc;
    if ((rc = check_models(models, "srx_viterbi"))) return rc;
    if (n_utts == 0) return SRX_OK;
    SRX_REQUIRE(out_logprob != nullptr && (n_frames == 0 || out_path != nullptr),
                "srx_viterbi: NULL output");
    SRX_HIP(hipSetDevice(h->device));
    return srx_generic_viterbi(h, X, dtype, utt_offsets, n_utts, n_frames, utt_model, models,
                               out_path, out_logprob, (hipStream_t)stream);
}

int srx_posteriors(srx_handle_t h, const void* X, int dtype, const int64_t* utt_offsets,
                   int64_t n_utts, int64_t n_frames, const int32_t* utt_model,
                   const srx_models_t* models, void* out_posteriors, double* out_logprob,
                   void* stream) {
    int rc = check_common(h, X, dtype, n_utts, n_frames, utt_offsets, "srx_posteriors");
    if (rc) return rc;
    if ((rc = check_models(models, "srx_posteriors"))) return rc;
    if (n_utts == 0) return SRX_OK;
    SRX_REQUIRE(out_logprob != nullptr && (n_frames == 0 || out_posteriors != nullptr),
                "srx_posteriors: NULL output");
    SRX_HIP(hipSetDevice(h->device));
    // (models of at most 8 states: the lane-per-utterance kernel behind
    // srx_generic_posteriors is the faster one)
    if (dtype == SRX_F32 && n_frames > 0 && models->n_states > 8) {
        rc = srx_fused_posteriors(h, X, utt_offsets, n_utts, n_frames, utt_model, models,
                                  (float*)out_posteriors, out_logprob, (hipStream_t)stream);
        if (rc != SRX_EUNSUPPORTED) return rc;
    }
    return srx_generic_posteriors(h, X, dtype, utt_offsets, n_utts, n_frames, utt_model, models,
                                  out_posteriors, out_logprob, (hipStream_t)stream);
}

int srx_gmm_emission_loglik(srx_handle_t h, const void* X, int dtype, int64_t n_frames,
                            const srx_gmm_t* gmm, void* out, void* stream) {
    int rc = check_common(h, X, dtype, 0, n_frames, nullptr, "srx_gmm_emission_loglik");
    if (rc) return rc;
    SRX_REQUIRE(gmm != nullptr, "srx_gmm_emission_loglik: gmm is NULL");
    SRX_REQUIRE(gmm->n_states >= 1 && gmm->n_mix >= 1 && gmm->n_features >= 1,
                "srx_gmm_emission_loglik: Shape mismatch (n_states=%d n_mix=%d n_features=%d)",
                gmm->n_states, gmm->n_mix, gmm->n_features);
    SRX_REQUIRE(gmm->weights && gmm->means && gmm->covars,
                "srx_gmm_emission_loglik: NULL parameter array");
    if (n_frames == 0) return SRX_OK;
    SRX_REQUIRE(out != nullptr, "srx_gmm_emission_loglik: NULL out");
    SRX_HIP(hipSetDevice(h->device));
    return srx_large_gmm_emission(h, X, dtype, n_frames, gmm, out, (hipStream_t)stream);
}

int srx_viterbi_dense(srx_handle_t h, const void* loglik, int dtype, const int64_t* utt_offsets,
                      int64_t n_utts, int64_t n_frames, int32_t n_states,
                      const double* startprob, const double* transmat, int32_t* out_path,
                      double* out_logprob, void* stream) {
    int rc = check_common(h, loglik, dtype, n_utts, n_frames, utt_offsets, "srx_viterbi_dense");
    if (rc) return rc;
    SRX_REQUIRE(n_states >= 1 && n_states <= 1024,
                "srx_viterbi_dense: n_states %d outside [1, 1024]", n_states);
    SRX_REQUIRE(startprob && transmat, "srx_viterbi_dense: NULL model parameter array");
    if (n_utts == 0) return SRX_OK;
    SRX_REQUIRE(out_logprob != nullptr && (n_frames == 0 || out_path != nullptr),
                "srx_viterbi_dense: NULL output");
    SRX_HIP(hipSetDevice(h->device));
    rc = srx_large_viterbi_dense(h, loglik, dtype, utt_offsets, n_utts, n_frames, n_states,
                                 startprob, transmat, out_path, out_logprob, (hipStream_t)stream);
    if (rc == SRX_EUNSUPPORTED) srx_set_error("srx_viterbi_dense: shape not supported");
    return rc;
}

static int estep_common(srx_handle_t h, const void* X, int dtype, const int64_t* utt_offsets,
                        int64_t n_utts, int64_t n_frames, const int32_t* utt_model,
                        const int64_t* utt_order, const srx_models_t* models, double* out_stats,
                        void* stream, bool allow_fused, int which_stats, const char* who) {
    int rc = check_common(h, X, dtype, n_utts, n_frames, utt_offsets, who);
    if (rc) return rc;
    if ((rc = check_models(models, who))) return rc;
    SRX_REQUIRE(out_stats != nullptr, "%s: NULL out_stats", who);
    SRX_REQUIRE(which_stats == SRX_STATS_ALL || which_stats == SRX_STATS_TRANSITIONS,
                "%s: unknown statistics selection %d", who, which_stats);
    SRX_HIP(hipSetDevice(h->device));
    // (a property of this call, not of the context: every E-step call writes it before any
    // kernel of that call is chosen, and nothing else reads it)
    h->estep_stats = which_stats;
    if (allow_fused && (h->estep_path == SRX_ESTEP_AUTO || h->estep_path == SRX_ESTEP_WAVE)) {
        // the smallest jobs (the reference's own size): one wave per utterance
        rc = srx_wave_estep(h, X, dtype, utt_offsets, n_utts, n_frames, utt_model, models, out_stats,
                            (hipStream_t)stream);
        if (rc != SRX_EUNSUPPORTED) return rc;
    }
    if (allow_fused && dtype == SRX_F32) {
        // small jobs first: one lane per utterance beats a matrix-core wave they cannot fill
        if (h->estep_path != SRX_ESTEP_FUSED) {
            rc = srx_lane_estep(h, X, utt_offsets, n_utts, n_frames, utt_model, utt_order, models, out_stats,
                                (hipStream_t)stream);
            if (rc != SRX_EUNSUPPORTED) return rc;
        }
        rc = srx_fused_estep(h, X, utt_offsets, n_utts, n_frames, utt_model, utt_order, models,
                             out_stats, (hipStream_t)stream);
        if (rc != SRX_EUNSUPPORTED) return rc;
    }
    if (allow_fused && dtype == SRX_F64 && h->estep_path != SRX_ESTEP_FUSED) {
        rc = srx_lane_estep_f64(h, X, utt_offsets, n_utts, n_frames, utt_model, utt_order, models, out_stats,
                                (hipStream_t)stream);
        if (rc != SRX_EUNSUPPORTED) return rc;
    }
    return srx_generic_estep(h, X, dtype, utt_offsets, n_utts, n_frames, utt_model, utt_order,
                             models, out_stats, (hipStream_t)stream);
}

int srx_estep_stats(srx_handle_t h, const void* X, int dtype, const int64_t* utt_offsets,
                    int64_t n_utts, int64_t n_frames, const int32_t* utt_model,
                    const int64_t* utt_order, const srx_models_t* models, double* out_stats,
                    void* stream) {
    return estep_common(h, X, dtype, utt_offsets, n_utts, n_frames, utt_model, utt_order, models,
                        out_stats, stream, true, SRX_STATS_ALL, "srx_estep_stats");
}

int srx_estep_stats_select(srx_handle_t h, const void* X, int dtype, const int64_t* utt_offsets,
                           int64_t n_utts, int64_t n_frames, const int32_t* utt_model,
                           const int64_t* utt_order, const srx_models_t* models, int which_stats,
                           double* out_stats, void* stream) {
    return estep_common(h, X, dtype, utt_offsets, n_utts, n_frames, utt_model, utt_order, models,
                        out_stats, stream, true, which_stats, "srx_estep_stats_select");
}

int srx_estep_stats_generic(srx_handle_t h, const void* X, int dtype, const int64_t* utt_offsets,
                            int64_t n_utts, int64_t n_frames, const int32_t* utt_model,
                            const int64_t* utt_order, const srx_models_t* models,
                            double* out_stats, void* stream) {
    return estep_common(h, X, dtype, utt_offsets, n_utts, n_frames, utt_model, utt_order, models,
                        out_stats, stream, false, SRX_STATS_ALL, "srx_estep_stats_generic");
}

int srx_mstep(srx_handle_t h, const double* stats, const srx_priors_t* priors, int mode,
              const srx_models_t* models, void* stream) {
    SRX_REQUIRE(h != nullptr && stats != nullptr && priors != nullptr, "srx_mstep: NULL argument");
    SRX_REQUIRE(mode == SRX_MSTEP_FULL || mode == SRX_MSTEP_TRANSITIONS_ONLY,
                "srx_mstep: unknown mode %d", mode);
    int rc = check_models(models, "srx_mstep");
    if (rc) return rc;
    SRX_HIP(hipSetDevice(h->device));
    return srx_generic_mstep(h, stats, priors, mode, models, (hipStream_t)stream);
}

}  // extern "C"
