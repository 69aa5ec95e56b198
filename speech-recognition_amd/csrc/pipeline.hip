// The two small steps that let the audio -> features -> scores -> label chain and the
// set-up of a fit stay on the device: the arg max over the word models' scores
// (HMM_final.py:51-52) and the per-feature mean / variance that MyGaussianHMM.init takes
// from the training frames (my_hmmlearn/hmm.py:130-132).
#include "srx_internal.h"

namespace {

// max(scores, key=scores.get): the first column in order whose value is larger than every
// earlier one (a later equal value does not replace it; a NaN never wins a comparison).
__global__ __launch_bounds__(256) void argmax_rows_kernel(const double* __restrict__ scores, int64_t U,
                                                          int M, int32_t* __restrict__ out) {
    const int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= U) return;
    const double* row = scores + u * M;
    double best = row[0];
    int arg = 0;
    for (int m = 1; m < M; ++m) {
        const double v = row[m];
        if (v > best) {
            best = v;
            arg = m;
        }
    }
    out[u] = arg;
}

// One workgroup per feature column: the mean in a first pass over the rows, the sum of
// squared deviations in a second (float64 throughout, a fixed order of additions: thread t
// adds rows t, t + 256, ..., then a tree over the threads).
template <typename T>
__global__ __launch_bounds__(256) void column_moments_kernel(const T* __restrict__ X, int64_t n, int D,
                                                             double* __restrict__ mean,
                                                             double* __restrict__ var) {
    __shared__ double part[256];
    const int d = blockIdx.x, tid = threadIdx.x;
    auto block_sum = [&](double v) {
        part[tid] = v;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {
            if (tid < o) part[tid] += part[tid + o];
            __syncthreads();
        }
        const double s = part[0];
        __syncthreads();
        return s;
    };
    double s = 0.0;
    for (int64_t i = tid; i < n; i += 256) s += (double)X[i * D + d];
    const double mu = block_sum(s) / (double)n;
    double q = 0.0;
    for (int64_t i = tid; i < n; i += 256) {
        const double c = (double)X[i * D + d] - mu;
        q += c * c;
    }
    const double ss = block_sum(q);
    if (tid == 0) {
        mean[d] = mu;
        var[d] = ss / (double)(n - 1);       // np.cov: one degree of freedom taken off
    }
}

}  // namespace

extern "C" {

int srx_argmax_scores(srx_handle_t h, const double* scores, int64_t n_utts, int32_t n_models,
                      int32_t* out_labels, void* stream) {
    SRX_REQUIRE(h != nullptr, "srx_argmax_scores: NULL context");
    SRX_REQUIRE(n_utts >= 0 && n_models >= 1, "srx_argmax_scores: Shape mismatch (n_utts=%lld n_models=%d)",
                (long long)n_utts, n_models);
    if (n_utts == 0) return SRX_OK;
    SRX_REQUIRE(scores != nullptr && out_labels != nullptr, "srx_argmax_scores: NULL argument");
    SRX_HIP(hipSetDevice(h->device));
    hipLaunchKernelGGL(argmax_rows_kernel, dim3((unsigned)((n_utts + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, scores, n_utts, (int)n_models, out_labels);
    SRX_HIP(hipGetLastError());
    return SRX_OK;
}

int srx_column_moments(srx_handle_t h, const void* X, int dtype, int64_t n_frames, int32_t n_features,
                       double* out_mean, double* out_var, void* stream) {
    SRX_REQUIRE(h != nullptr, "srx_column_moments: NULL context");
    SRX_REQUIRE(dtype == SRX_F32 || dtype == SRX_F64, "srx_column_moments: unknown dtype %d", dtype);
    SRX_REQUIRE(n_frames >= 1 && n_features >= 1 && n_features <= 65535,
                "srx_column_moments: Shape mismatch (n_frames=%lld n_features=%d)", (long long)n_frames,
                n_features);
    SRX_REQUIRE(X != nullptr && out_mean != nullptr && out_var != nullptr, "srx_column_moments: NULL argument");
    SRX_HIP(hipSetDevice(h->device));
    if (dtype == SRX_F32)
        hipLaunchKernelGGL(column_moments_kernel<float>, dim3((unsigned)n_features), dim3(256), 0,
                           (hipStream_t)stream, (const float*)X, n_frames, (int)n_features, out_mean, out_var);
    else
        hipLaunchKernelGGL(column_moments_kernel<double>, dim3((unsigned)n_features), dim3(256), 0,
                           (hipStream_t)stream, (const double*)X, n_frames, (int)n_features, out_mean, out_var);
    SRX_HIP(hipGetLastError());
    return SRX_OK;
}

}  // extern "C"
