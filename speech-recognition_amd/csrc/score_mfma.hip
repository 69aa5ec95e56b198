// Matrix-core scoring path (placeholder until the kernel lands).
#include "srx_internal.h"

int srx_mfma_score(srx_context*, const void*, const int64_t*, int64_t, int64_t,
                   const srx_models_t*, double*, hipStream_t) {
    return SRX_EUNSUPPORTED;
}
