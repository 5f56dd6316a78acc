/*
 * vd_oracle_cull.c — CPU restatement of voidin's emit_draws compute shader.
 * TEST INFRASTRUCTURE ONLY (see vd_oracle.h; parity unpinned).
 *
 * Follows shaders/emit_draws.wgsl:13-64, shaders/utils/math.wgsl:67-73 and the struct
 * layouts of shaders/shared.wgsl:13-75.  Floating-point evaluation order is the spec
 * decision of SURVEY.md §8a C2':
 *   mat*mat column j = ((A.c0*b0 + A.c1*b1) + A.c2*b2) + A.c3*b3,  mat*vec likewise,
 *   length(v) = sqrt((x*x + y*y) + z*z), distance(a,b) = length(a-b), no FMA, IEEE sqrt.
 */
#include "vd_oracle.h"
#include "vd_oracle_math.h"

#include <pthread.h>
#include <stdlib.h>

typedef struct {
    v3 center;      /* view-space centre                                      */
    float radius;
    float lhs_x;    /* center.z*fr.y - |center.x|*fr.x                        */
    float lhs_y;    /* center.z*fr.w - |center.y|*fr.z                        */
} cull_terms;

/* emit_draws.wgsl:13-19 + math.wgsl:67-73 */
static inline cull_terms cull_eval(const VdCameraUniform* cam, const VdMeshInfo* m,
                                   const float* T /* column-major mat4 */) {
    const float* V = cam->view;
    /* emit_draws.wgsl:14   var center = (mesh.max + mesh.min) / 2.; */
    v3 c0 = v3_make((m->max[0] + m->min[0]) / 2.0f, (m->max[1] + m->min[1]) / 2.0f,
                    (m->max[2] + m->min[2]) / 2.0f);
    /* emit_draws.wgsl:15   (camera.view * transform * vec4(center, 1.0)).xyz
     * WGSL `*` is left-associative: the full mat4*mat4 product comes first.  Only rows 0..2
     * of the product reach .xyz. */
    float VT[4][3]; /* [column][row] */
    for (int j = 0; j < 4; ++j)
        for (int r = 0; r < 3; ++r)
            VT[j][r] = ((V[0 + r] * T[4 * j + 0] + V[4 + r] * T[4 * j + 1]) +
                        V[8 + r] * T[4 * j + 2]) +
                       V[12 + r] * T[4 * j + 3];
    float c[3];
    for (int r = 0; r < 3; ++r)
        c[r] = ((VT[0][r] * c0.x + VT[1][r] * c0.y) + VT[2][r] * c0.z) + VT[3][r] * 1.0f;

    /* math.wgsl:67-73 extract_scale; emit_draws.wgsl:17-18 */
    float sx = v3_length(v3_make(T[0], T[1], T[2]));
    float sy = v3_length(v3_make(T[4], T[5], T[6]));
    float sz = v3_length(v3_make(T[8], T[9], T[10]));
    float max_scale = fmaxf(fmaxf(fabsf(sx), fabsf(sy)), fabsf(sz));

    /* emit_draws.wgsl:19 — `center` is already view-space while mesh.min/max are
     * object-space: bug-compatible, kept. */
    cull_terms t;
    t.center = v3_make(c[0], c[1], c[2]);
    float d0 = v3_length(v3_sub(v3_load(m->min), t.center));
    float d1 = v3_length(v3_sub(v3_load(m->max), t.center));
    t.radius = fmaxf(d0, d1) * max_scale;
    const float* fr = cam->frustum;
    t.lhs_x = c[2] * fr[1] - fabsf(c[0]) * fr[0];
    t.lhs_y = c[2] * fr[3] - fabsf(c[1]) * fr[2];
    return t;
}

/* emit_draws.wgsl:21-32 */
static inline int cull_visible(const VdCameraUniform* cam, const cull_terms* t) {
    if (t->lhs_x < -t->radius) return 0;
    if (t->lhs_y < -t->radius) return 0;
    if (t->center.z + t->radius > cam->znear && t->center.z - t->radius > cam->zfar) return 0;
    return 1;
}

static void cull_range(const VdCameraUniform* cam, const VdMeshInfo* meshes, uint32_t n_mesh,
                       const VdInstance* inst, uint32_t begin, uint32_t end,
                       VdDrawIndexedIndirect* out) {
    for (uint32_t i = begin; i < end; ++i) {
        /* emit_draws.wgsl:43-45; out-of-range mesh ids are clamped (reference: undefined) */
        uint32_t mid = inst[i].mesh < n_mesh ? inst[i].mesh : n_mesh - 1;
        const VdMeshInfo* m = &meshes[mid];
        cull_terms t = cull_eval(cam, m, inst[i].transform);
        /* emit_draws.wgsl:49-63 — every slot is written */
        out[i].vertex_count = m->index_count;
        out[i].instance_count = cull_visible(cam, &t) ? 1u : 0u;
        out[i].base_index = m->base_index;
        out[i].vertex_offset = m->vertex_offset;
        out[i].base_instance = i;
    }
}

typedef struct {
    const VdCameraUniform* cam; const VdMeshInfo* meshes; uint32_t n_mesh;
    const VdInstance* inst; uint32_t begin, end; VdDrawIndexedIndirect* out;
} cull_job;

static void* cull_thread(void* p) {
    cull_job* j = (cull_job*)p;
    cull_range(j->cam, j->meshes, j->n_mesh, j->inst, j->begin, j->end, j->out);
    return NULL;
}

int vd_ref_cull_emit(const VdCameraUniform* camera, const VdMeshInfo* meshes, uint32_t n_mesh,
                     const VdInstance* instances, uint32_t n_inst, VdDrawIndexedIndirect* out,
                     int threads) {
    if (!camera || !meshes || n_mesh == 0 || (n_inst && (!instances || !out)))
        return VD_ERR_INVALID_ARG;
    if (threads <= 1 || n_inst < 4096) {
        cull_range(camera, meshes, n_mesh, instances, 0, n_inst, out);
        return VD_OK;
    }
    if (threads > 256) threads = 256;
    pthread_t tid[256];
    cull_job job[256];
    for (int t = 0; t < threads; ++t) {
        job[t].cam = camera; job[t].meshes = meshes; job[t].n_mesh = n_mesh;
        job[t].inst = instances; job[t].out = out;
        job[t].begin = (uint32_t)((uint64_t)n_inst * t / threads);
        job[t].end = (uint32_t)((uint64_t)n_inst * (t + 1) / threads);
        if (pthread_create(&tid[t], NULL, cull_thread, &job[t]) != 0) {
            cull_thread(&job[t]);
            tid[t] = 0;
        }
    }
    for (int t = 0; t < threads; ++t)
        if (tid[t]) pthread_join(tid[t], NULL);
    return VD_OK;
}

int vd_ref_cull_margins(const VdCameraUniform* camera, const VdMeshInfo* meshes, uint32_t n_mesh,
                        const VdInstance* instances, uint32_t n_inst, float* out_margin_x,
                        float* out_margin_y, float* out_radius) {
    if (!camera || !meshes || n_mesh == 0 || (n_inst && !instances)) return VD_ERR_INVALID_ARG;
    for (uint32_t i = 0; i < n_inst; ++i) {
        uint32_t mid = instances[i].mesh < n_mesh ? instances[i].mesh : n_mesh - 1;
        cull_terms t = cull_eval(camera, &meshes[mid], instances[i].transform);
        if (out_margin_x) out_margin_x[i] = t.lhs_x + t.radius;
        if (out_margin_y) out_margin_y[i] = t.lhs_y + t.radius;
        if (out_radius) out_radius[i] = t.radius;
    }
    return VD_OK;
}

/* S = [i : in[i].instance_count == 1] ascending; out[k] = in[S[k]]. */
int vd_ref_compact(const VdDrawIndexedIndirect* in, uint32_t n, VdDrawIndexedIndirect* out,
                   uint32_t* out_count, int pad_tail) {
    if ((n && (!in || !out)) || !out_count) return VD_ERR_INVALID_ARG;
    uint32_t k = 0;
    for (uint32_t i = 0; i < n; ++i)
        if (in[i].instance_count == 1u) out[k++] = in[i];
    *out_count = k;
    if (pad_tail)
        for (uint32_t i = k; i < n; ++i) memset(&out[i], 0, sizeof(out[i]));
    return VD_OK;
}

/* shaders/compute_update.wgsl:10-28.  rotz = from_rotation_z(angle) (utils/math.wgsl:55-63):
 * columns (c, s, 0, 0), (-s, c, 0, 0), (0, 0, 1, 0), (0, 0, 0, 1); product column j =
 * ((R.c0*T[j].x + R.c1*T[j].y) + R.c2*T[j].z) + R.c3*T[j].w (spec decision C2'). */
static void mat_mul_cm(const float* A, const float* B, float* out) { /* out = A * B, column-major */
    float r[16];
    for (int j = 0; j < 4; ++j)
        for (int i = 0; i < 4; ++i)
            r[4 * j + i] = ((A[i] * B[4 * j] + A[4 + i] * B[4 * j + 1]) + A[8 + i] * B[4 * j + 2]) + A[12 + i] * B[4 * j + 3];
    memcpy(out, r, sizeof(r));
}

int vd_ref_compute_update(const uint32_t* indices, uint32_t n_indices, VdInstance* instances,
                          uint32_t n_instances, float time, float dt, int fix_inverse) {
    if ((n_indices && !indices) || !instances) return VD_ERR_INVALID_ARG;
    const float speed0 = 2.0f * sinf(time * 0.5f);
    for (uint32_t k = 0; k < n_indices; ++k) {
        const uint32_t idx = indices[k];
        if (idx >= n_instances) continue;               /* robust-buffer-access: out of range is dropped */
        float* T = instances[idx].transform;
        float speed = speed0;
        if (T[14] > -15.0f) speed *= 1.0f; else speed *= -1.0f;   /* transform[3][2] */
        const float ang = speed * dt;
        const float c = cosf(ang), s = sinf(ang);
        const float R[16] = {c, s, 0, 0, -s, c, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
        mat_mul_cm(R, T, T);
        if (fix_inverse) {
            const float Ri[16] = {c, -s, 0, 0, s, c, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};   /* rotz(-ang) */
            mat_mul_cm(instances[idx].inv_transform, Ri, instances[idx].inv_transform);
        }
    }
    return VD_OK;
}

const char* vd_ref_version(void) { return "vd_oracle 0.1 (voidin v0.69.0 restatement; parity unpinned)"; }
