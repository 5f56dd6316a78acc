// stubs.hip — entry points not implemented yet return VD_ERR_INVALID_ARG with a message.
#include "vd_common.hpp"
#define STUB(ctx, name) do { if (!(ctx)) return VD_ERR_INVALID_ARG; VD_FAIL(ctx, VD_ERR_INVALID_ARG, name ": not implemented yet"); } while (0)
extern "C" {
int vd_bvh_build(VdCtx* c, const float*, uint32_t, uint32_t*, uint32_t, VdBvhNode*, uint32_t, uint32_t*) { STUB(c, "vd_bvh_build"); }
int vd_bvh_build_dev(VdCtx* c, const float*, uint32_t, uint32_t*, uint32_t, VdBvhNode*, uint32_t, uint32_t*) { STUB(c, "vd_bvh_build_dev"); }
int vd_tlas_build(VdCtx* c, const VdInstance*, uint32_t, const VdMeshInfo*, uint32_t, VdTlasNode*) { STUB(c, "vd_tlas_build"); }
int vd_tlas_build_dev(VdCtx* c, const VdInstance*, uint32_t, const VdMeshInfo*, uint32_t, VdTlasNode*) { STUB(c, "vd_tlas_build_dev"); }
int vd_tlas_build_wide(VdCtx* c, const VdInstance*, uint32_t, const VdMeshInfo*, uint32_t, VdTlasNodeWide*) { STUB(c, "vd_tlas_build_wide"); }
int vd_tlas_build_wide_dev(VdCtx* c, const VdInstance*, uint32_t, const VdMeshInfo*, uint32_t, VdTlasNodeWide*) { STUB(c, "vd_tlas_build_wide_dev"); }
int vd_tlas_refit(VdCtx* c, const VdInstance*, uint32_t, const VdMeshInfo*, uint32_t, VdTlasNode*) { STUB(c, "vd_tlas_refit"); }
int vd_tlas_refit_dev(VdCtx* c, const VdInstance*, uint32_t, const VdMeshInfo*, uint32_t, VdTlasNode*) { STUB(c, "vd_tlas_refit_dev"); }
int vd_tlas_refit_wide_dev(VdCtx* c, const VdInstance*, uint32_t, const VdMeshInfo*, uint32_t, VdTlasNodeWide*) { STUB(c, "vd_tlas_refit_wide_dev"); }
int vd_trace(VdCtx* c, const VdTraceScene*, const VdRay*, uint32_t, VdHit*) { STUB(c, "vd_trace"); }
int vd_trace_dev(VdCtx* c, const VdTraceScene*, const VdRay*, uint32_t, VdHit*) { STUB(c, "vd_trace_dev"); }
}
