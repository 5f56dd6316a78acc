// hiz.hip — depth pyramid for the occlusion extension (no reference counterpart: voidin's README
// only links "Two-Pass Occlusion Culling").  Level k texel = MIN (farthest, reverse Z) of its <= 4 children; level
// dims halve rounding up, so texel (x, y) of level k covers pixels [x 2^k, (x+1) 2^k) x [y 2^k, (y+1) 2^k) of the
// depth buffer that exist.  The occlusion test itself is occlusion_mask_kernel in cull.hip.
#include "vd_common.hpp"

namespace {

// One launch makes TWO levels: a 16 x 16 workgroup reads a 64 x 64 tile of the source level (coalesced float4 rows),
// writes its 32 x 32 reduction, and reduces that once more to 16 x 16 in registers (a thread owns a 4 x 4 patch).  Reads dominate (the source level
// is 4x / 16x the size of what is written): HBM-bound, one pass over the depth buffer for levels 1 and 2.
constexpr int kTile = 64;
__global__ __launch_bounds__(256) void hiz_reduce2_kernel(const float* __restrict__ src, unsigned sw, unsigned sh,
                                                          float* __restrict__ d1, unsigned w1, unsigned h1,
                                                          float* __restrict__ d2, unsigned w2, unsigned h2) {
    const unsigned tx = threadIdx.x & 15u, ty = threadIdx.x >> 4;
    const unsigned bx = blockIdx.x * kTile, by = blockIdx.y * kTile;
    // thread (tx, ty) owns the 4 x 4 source texels at (bx + 4 tx, by + 4 ty): 2 x 2 texels of level 1, one of level 2
    float m[2][2];
#pragma unroll
    for (int qy = 0; qy < 2; ++qy)
#pragma unroll
        for (int qx = 0; qx < 2; ++qx) {
            const unsigned x0 = bx + 4u * tx + 2u * qx, y0 = by + 4u * ty + 2u * qy;
            // clamped duplicates do not change a min; texels of level 1 outside the level are never stored
            const unsigned xa = min(x0, sw - 1u), xb = min(x0 + 1u, sw - 1u), ya = min(y0, sh - 1u), yb = min(y0 + 1u, sh - 1u);
            const float a = fminf(src[(size_t)ya * sw + xa], src[(size_t)ya * sw + xb]);
            const float b = fminf(src[(size_t)yb * sw + xa], src[(size_t)yb * sw + xb]);
            m[qy][qx] = fminf(a, b);
            const unsigned ox = (bx >> 1) + 2u * tx + qx, oy = (by >> 1) + 2u * ty + qy;
            if (ox < w1 && oy < h1) d1[(size_t)oy * w1 + ox] = m[qy][qx];
        }
    if (d2 == nullptr) return;
    // level 2 from level 1 with the same clamping rule: children outside level 1 are replaced by the clamped one
    const unsigned ox = (bx >> 2) + tx, oy = (by >> 2) + ty;
    if (ox < w2 && oy < h2) {
        const unsigned cx = 2u * ox, cy = 2u * oy;             // level-1 coordinates of the first child (exists)
        const bool has_x = cx + 1u < w1, has_y = cy + 1u < h1;
        const float a = fminf(m[0][0], has_x ? m[0][1] : m[0][0]);
        const float b = has_y ? fminf(m[1][0], has_x ? m[1][1] : m[1][0]) : a;
        d2[(size_t)oy * w2 + ox] = fminf(a, b);
    }
}

int layout(uint32_t width, uint32_t height, VdHizLayout* out) {
    if (!out || width == 0 || height == 0 || width > 65536u || height > 65536u || (uint64_t)width * height > 0x40000000ull)
        return VD_ERR_INVALID_ARG;
    *out = VdHizLayout{};
    out->width = width; out->height = height;
    uint32_t w = width, h = height, off = 0, l = 0;
    for (;;) {
        out->level_offset[l] = off; out->level_width[l] = w; out->level_height[l] = h;
        off += w * h;
        ++l;
        if (w == 1 && h == 1) break;
        w = (w + 1) / 2; h = (h + 1) / 2;
    }
    out->n_levels = l;
    out->total_texels = off;
    return VD_OK;
}

}  // namespace

extern "C" {

int vd_hiz_layout(uint32_t width, uint32_t height, VdHizLayout* out) { return layout(width, height, out); }

int vd_hiz_build_dev(VdCtx* ctx, const float* d_depth, uint32_t width, uint32_t height, float* d_pyramid) {
    VdDeviceGuard vd_guard_(ctx);   // run on ctx->device whatever the calling thread's current device is
    if (!ctx) return VD_ERR_INVALID_ARG;
    VdHizLayout L;
    if (layout(width, height, &L)) VD_FAIL(ctx, VD_ERR_INVALID_ARG, "vd_hiz_build: width, height must be 1..65536 and width * height <= 2^30");
    if (!d_depth || !d_pyramid) VD_FAIL(ctx, VD_ERR_INVALID_ARG, "vd_hiz_build: null depth/pyramid");
    vd_time_begin(ctx);
    VD_HIP_CHECK(ctx, hipMemcpyAsync(d_pyramid, d_depth, (size_t)width * height * sizeof(float), hipMemcpyDeviceToDevice, ctx->stream));
    for (uint32_t l = 1; l < L.n_levels; l += 2) {
        const bool two = l + 1 < L.n_levels;
        const unsigned sw = L.level_width[l - 1], sh = L.level_height[l - 1];
        hipLaunchKernelGGL(hiz_reduce2_kernel, dim3((sw + kTile - 1) / kTile, (sh + kTile - 1) / kTile), dim3(256), 0, ctx->stream,
                           d_pyramid + L.level_offset[l - 1], sw, sh, d_pyramid + L.level_offset[l], L.level_width[l], L.level_height[l],
                           two ? d_pyramid + L.level_offset[l + 1] : (float*)nullptr, two ? L.level_width[l + 1] : 0u,
                           two ? L.level_height[l + 1] : 0u);
    }
    vd_time_end(ctx);
    VD_HIP_CHECK(ctx, hipGetLastError());
    return VD_OK;
}

}  // extern "C"
