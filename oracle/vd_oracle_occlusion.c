/* vd_oracle_occlusion.c - CPU twin of the HiZ occlusion extension.  TEST INFRASTRUCTURE ONLY.
 *
 * The reference has NO occlusion culling (README.md:33 links "Two-Pass Occlusion Culling", nothing in the source), so
 * there is nothing to restate: this file is the independent CPU statement of the extension's own definition
 * (include/voidin_abi.h, "Occlusion culling"), written to be compared bit for bit with the HIP kernels.  Strict fp32,
 * no FMA contraction. */
#include "vd_oracle.h"
#include "vd_oracle_math.h"

#include <math.h>
#include <string.h>

int vd_ref_hiz_layout(uint32_t width, uint32_t height, VdHizLayout* out) {
    /* <= 2^30 texels in level 0 keeps every offset (total < 4/3 * 2^30 + 17) in 32 bits */
    if (!out || width == 0 || height == 0 || width > 65536u || height > 65536u || (uint64_t)width * height > 0x40000000ull)
        return VD_ERR_INVALID_ARG;
    memset(out, 0, sizeof(*out));
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

int vd_ref_hiz_build(const float* depth, uint32_t width, uint32_t height, float* pyramid) {
    VdHizLayout L;
    int rc = vd_ref_hiz_layout(width, height, &L);
    if (rc) return rc;
    if (!depth || !pyramid) return VD_ERR_INVALID_ARG;
    memcpy(pyramid, depth, (size_t)width * height * sizeof(float));
    for (uint32_t l = 1; l < L.n_levels; ++l) {
        const float* src = pyramid + L.level_offset[l - 1];
        float* dst = pyramid + L.level_offset[l];
        const uint32_t sw = L.level_width[l - 1], sh = L.level_height[l - 1], dw = L.level_width[l], dh = L.level_height[l];
        for (uint32_t y = 0; y < dh; ++y)
            for (uint32_t x = 0; x < dw; ++x) {
                const uint32_t x0 = 2 * x, y0 = 2 * y, x1 = x0 + 1 < sw ? x0 + 1 : sw - 1, y1 = y0 + 1 < sh ? y0 + 1 : sh - 1;
                const float a = fminf(src[(size_t)y0 * sw + x0], src[(size_t)y0 * sw + x1]);
                const float b = fminf(src[(size_t)y1 * sw + x0], src[(size_t)y1 * sw + x1]);
                dst[(size_t)y * dw + x] = fminf(a, b);
            }
    }
    return VD_OK;
}

static uint32_t bits_of(uint32_t v) { uint32_t b = 0; while (v) { ++b; v >>= 1; } return b; }

/* 1 = hidden behind the depth pyramid, 0 = keep.  See the definition in include/voidin_abi.h. */
static int occluded(const VdCameraUniform* cam, const VdMeshInfo* m, const float* T, const float* pyr, const VdHizLayout* L) {
    const float* V = cam->view;
    const float* P = cam->projection;
    v3 c0 = v3_make((m->max[0] + m->min[0]) / 2.0f, (m->max[1] + m->min[1]) / 2.0f, (m->max[2] + m->min[2]) / 2.0f);
    float c[3];
    for (int r = 0; r < 3; ++r) {
        float col[4];
        for (int j = 0; j < 4; ++j)
            col[j] = ((V[r] * T[4 * j] + V[4 + r] * T[4 * j + 1]) + V[8 + r] * T[4 * j + 2]) + V[12 + r] * T[4 * j + 3];
        c[r] = ((col[0] * c0.x + col[1] * c0.y) + col[2] * c0.z) + col[3] * 1.0f;
    }
    const float sx = v3_length(v3_make(T[0], T[1], T[2])), sy = v3_length(v3_make(T[4], T[5], T[6])), sz = v3_length(v3_make(T[8], T[9], T[10]));
    const float max_scale = fmaxf(fmaxf(fabsf(sx), fabsf(sy)), fabsf(sz));
    const float r = (v3_length(v3_sub(v3_load(m->max), v3_load(m->min))) * 0.5f) * max_scale;
    const float d = -c[2];
    const float dn = d - r;
    if (!(dn > cam->znear)) return 0;
    const float rr = r * r, dd = d * d, rd = r * d;
    const float tx = sqrtf((c[0] * c[0] + dd) - rr), ty = sqrtf((c[1] * c[1] + dd) - rr);
    const float dxm = d * tx + c[0] * r, dxp = d * tx - c[0] * r, dym = d * ty + c[1] * r, dyp = d * ty - c[1] * r;
    if (!(dxm > 0.0f && dxp > 0.0f && dym > 0.0f && dyp > 0.0f)) return 0;
    const float sx0 = (c[0] * tx - rd) / dxm, sx1 = (c[0] * tx + rd) / dxp;
    const float sy0 = (c[1] * ty - rd) / dym, sy1 = (c[1] * ty + rd) / dyp;
    const float nxa = P[0] * sx0 - P[8], nxb = P[0] * sx1 - P[8], nya = P[5] * sy0 - P[9], nyb = P[5] * sy1 - P[9];
    const float nx_lo = fminf(nxa, nxb), nx_hi = fmaxf(nxa, nxb), ny_lo = fminf(nya, nyb), ny_hi = fmaxf(nya, nyb);
    const float W = (float)L->width, H = (float)L->height;
    const float u0 = (nx_lo * 0.5f + 0.5f) * W - 0.5f, u1 = (nx_hi * 0.5f + 0.5f) * W + 0.5f;
    const float v0 = (0.5f - ny_hi * 0.5f) * H - 0.5f, v1 = (0.5f - ny_lo * 0.5f) * H + 0.5f;
    if (!(u1 >= 0.0f && v1 >= 0.0f && u0 < W && v0 < H)) return 0;
    const uint32_t x0 = (uint32_t)floorf(fmaxf(u0, 0.0f)), x1 = (uint32_t)floorf(fminf(u1, W - 1.0f));
    const uint32_t y0 = (uint32_t)floorf(fmaxf(v0, 0.0f)), y1 = (uint32_t)floorf(fminf(v1, H - 1.0f));
    uint32_t lvl = bits_of((x1 - x0) > (y1 - y0) ? (x1 - x0) : (y1 - y0));
    if (lvl > L->n_levels - 1) lvl = L->n_levels - 1;
    const float* t = pyr + L->level_offset[lvl];
    const uint32_t lw = L->level_width[lvl];
    const uint32_t ax = x0 >> lvl, bx = x1 >> lvl, ay = y0 >> lvl, by = y1 >> lvl;
    const float h0 = fminf(t[(size_t)ay * lw + ax], t[(size_t)ay * lw + bx]);
    const float h1 = fminf(t[(size_t)by * lw + ax], t[(size_t)by * lw + bx]);
    const float hmin = fminf(h0, h1);
    const float depth = (P[14] - P[10] * dn) / dn;
    return depth < hmin;
}

int vd_ref_occlusion_mask(const VdCameraUniform* camera, const VdMeshInfo* meshes, uint32_t n_mesh, const VdInstance* instances,
                          uint32_t n_inst, const float* pyramid, uint32_t width, uint32_t height, const uint64_t* mask_in,
                          uint64_t* mask_out) {
    VdHizLayout L;
    int rc = vd_ref_hiz_layout(width, height, &L);
    if (rc) return rc;
    if (!camera || !meshes || n_mesh == 0 || !pyramid || (n_inst && (!instances || !mask_in || !mask_out))) return VD_ERR_INVALID_ARG;
    if (!(camera->projection[11] == -1.0f && camera->projection[15] == 0.0f)) return VD_ERR_INVALID_ARG;
    const uint32_t n_words = (n_inst + 63u) / 64u;
    for (uint32_t w = 0; w < n_words; ++w) {
        uint64_t in = mask_in[w], out = 0;
        for (uint32_t b = 0; b < 64u; ++b) {
            const uint32_t i = 64u * w + b;
            if (!((in >> b) & 1u) || i >= n_inst) continue;
            const uint32_t mid = instances[i].mesh < n_mesh ? instances[i].mesh : n_mesh - 1;
            if (!occluded(camera, &meshes[mid], instances[i].transform, pyramid, &L)) out |= 1ull << b;
        }
        mask_out[w] = out;
    }
    return VD_OK;
}
