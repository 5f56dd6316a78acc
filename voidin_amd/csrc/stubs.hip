// stubs.hip — entry points not implemented yet return VD_ERR_INVALID_ARG with a message.
#include "vd_common.hpp"
#define STUB(ctx, name) do { if (!(ctx)) return VD_ERR_INVALID_ARG; VD_FAIL(ctx, VD_ERR_INVALID_ARG, name ": not implemented yet"); } while (0)
extern "C" {
int vd_bvh_build(VdCtx* c, const float*, uint32_t, uint32_t*, uint32_t, VdBvhNode*, uint32_t, uint32_t*) { STUB(c, "vd_bvh_build"); }
int vd_bvh_build_dev(VdCtx* c, const float*, uint32_t, uint32_t*, uint32_t, VdBvhNode*, uint32_t, uint32_t*) { STUB(c, "vd_bvh_build_dev"); }
}
