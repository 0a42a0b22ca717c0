#include <hip/hip_runtime.h>
#include "common.h"
using namespace sq;
extern "C" int sq_tile_minmax(const void *const *, const void *, int64_t, int32_t, int32_t, int32_t, int32_t, int32_t,
                              uint32_t *, void *) {
    return fail(SQ_ERR_UNSUPPORTED, "sq_tile_minmax: not built yet");
}
extern "C" int64_t sq_register_workspace_bytes(int32_t, int32_t, int32_t, int32_t) { return 0; }
extern "C" int sq_register_pairs(const sq_register_args *, void *) {
    return fail(SQ_ERR_UNSUPPORTED, "sq_register_pairs: not built yet");
}
