// Device-side family DAG build (mseq ctor + reduce_edges) -- see below.
#include "common.h"
#include "ctx.h"
using namespace sina_hip;
extern "C" int sina_hip_align_families(sina_hip_ctx *c, const uint32_t *, const uint64_t *, uint32_t,
                                       const uint8_t *, const uint64_t *, const sina_hip_align_params *,
                                       sina_hip_align_out *, uint32_t *) {
    SH_FAIL("align_families: not built yet");
}
