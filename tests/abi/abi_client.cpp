// A client of libsrx that knows nothing but include/srx.h and the HIP runtime: it
// reads a case file written by the test (model parameters, features, offsets, expected
// scores), calls srx_forward_score in float64 on device buffers it allocates itself,
// and reports the largest relative error.  Exit status 0 when it is below 1e-10.
//
// Case file (little endian): int64 M, S, D, U, n_frames; double pi[M*S], A[M*S*S],
// mu[M*S*D], var[M*S*D], X[n_frames*D]; int64 off[U+1]; double expected[U*M].
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <vector>

#include "srx.h"

#define HIP_OK(call)                                                              \
    do {                                                                          \
        hipError_t e_ = (call);                                                   \
        if (e_ != hipSuccess) {                                                   \
            std::fprintf(stderr, "%s: %s\n", #call, hipGetErrorString(e_));       \
            return 2;                                                             \
        }                                                                         \
    } while (0)

template <typename T>
static bool read_vec(std::FILE* f, std::vector<T>& v, size_t n) {
    v.resize(n);
    return n == 0 || std::fread(v.data(), sizeof(T), n, f) == n;
}

template <typename T>
static int to_device(const std::vector<T>& v, T** out) {
    HIP_OK(hipMalloc((void**)out, sizeof(T) * (v.size() ? v.size() : 1)));
    HIP_OK(hipMemcpy(*out, v.data(), sizeof(T) * v.size(), hipMemcpyHostToDevice));
    return 0;
}

int main(int argc, char** argv) {
    if (argc < 2) {
        std::fprintf(stderr, "usage: abi_client <case file>\n");
        return 2;
    }
    std::FILE* f = std::fopen(argv[1], "rb");
    if (!f) {
        std::perror(argv[1]);
        return 2;
    }
    int64_t hdr[5];
    if (std::fread(hdr, sizeof(int64_t), 5, f) != 5) return 2;
    const int64_t M = hdr[0], S = hdr[1], D = hdr[2], U = hdr[3], n_frames = hdr[4];
    std::vector<double> pi, A, mu, var, X, expected;
    std::vector<int64_t> off;
    if (!read_vec(f, pi, M * S) || !read_vec(f, A, M * S * S) || !read_vec(f, mu, M * S * D) ||
        !read_vec(f, var, M * S * D) || !read_vec(f, X, n_frames * D) || !read_vec(f, off, U + 1) ||
        !read_vec(f, expected, U * M)) {
        std::fprintf(stderr, "short case file\n");
        return 2;
    }
    std::fclose(f);

    srx_handle_t h = nullptr;
    if (srx_create(0, &h) != 0) {
        std::fprintf(stderr, "srx_create: %s\n", srx_last_error());
        return 2;
    }
    double *d_pi, *d_A, *d_mu, *d_var, *d_X, *d_out;
    int64_t* d_off;
    if (to_device(pi, &d_pi) || to_device(A, &d_A) || to_device(mu, &d_mu) || to_device(var, &d_var) ||
        to_device(X, &d_X) || to_device(off, &d_off))
        return 2;
    HIP_OK(hipMalloc((void**)&d_out, sizeof(double) * U * M));
    srx_models_t md;
    md.n_models = (int32_t)M;
    md.n_states = (int32_t)S;
    md.n_features = (int32_t)D;
    md.startprob = d_pi;
    md.transmat = d_A;
    md.means = d_mu;
    md.covars = d_var;
    hipStream_t st;
    HIP_OK(hipStreamCreate(&st));
    const int rc = srx_forward_score(h, d_X, SRX_F64, d_off, U, n_frames, &md, d_out, st);
    if (rc != 0) {
        std::fprintf(stderr, "srx_forward_score: %d %s\n", rc, srx_last_error());
        return 2;
    }
    HIP_OK(hipStreamSynchronize(st));
    std::vector<double> got(U * M);
    HIP_OK(hipMemcpy(got.data(), d_out, sizeof(double) * got.size(), hipMemcpyDeviceToHost));
    double worst = 0.0;
    for (size_t i = 0; i < got.size(); ++i)
        worst = std::fmax(worst, std::fabs(got[i] - expected[i]) / std::fabs(expected[i]));
    // an error path: a NULL output is refused with SRX_EINVAL and a message
    const int bad = srx_forward_score(h, d_X, SRX_F64, d_off, U, n_frames, &md, nullptr, st);
    std::printf("scores %zu max_rel_err %.3e version %d refused_null %d (%s)\n", got.size(), worst,
                srx_version(), bad, srx_last_error());
    srx_destroy(h);
    return worst < 1e-10 && bad != 0 ? 0 : 1;
}
