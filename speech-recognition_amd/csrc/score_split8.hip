// Instantiations of the split-bf16 scoring kernel for models padded to 8 states.
#include "score_split.h"

int srx_split_score_sp8(const float* X, const int64_t* off, int64_t U, int64_t n_frames, int M,
                        int S, int D, int NT, int chunks, int RQ, int H, const unsigned* Wp, const float* cst,
                        const float* A, const float* pi, double* out, hipStream_t st, const int64_t* order) {
    SplitTables tb;
    tb.Wp = Wp; tb.cst = cst; tb.A = A; tb.pi = pi; tb.order = order;
#define SRX_CASE(nt, rq, hh)                                                       \
    if (NT == nt && RQ == rq && H == hh) {                                         \
        launch_split<8, nt, rq, hh>(X, off, U, n_frames, M, S, D, chunks, tb, out, st); \
        SRX_HIP(hipGetLastError());                                                \
        return SRX_OK;                                                             \
    }
#define SRX_CASES(nt)                                                              \
    SRX_CASE(nt, 0, 1) SRX_CASE(nt, 1, 0) SRX_CASE(nt, 1, 1)                       \
    SRX_CASE(nt, 2, 0) SRX_CASE(nt, 2, 1) SRX_CASE(nt, 3, 0)
    SRX_CASES(3) SRX_CASES(2) SRX_CASES(1)
#undef SRX_CASES
#undef SRX_CASE
    return SRX_EUNSUPPORTED;
}
