/*
 * TEST INFRASTRUCTURE — NOT PRODUCT CODE.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may build, load or call
 * this file.  The product (topo4d_amd/) never links or falls back to it.
 *
 * raster_oracle.c — plain-C, fp32, CPU restatement of the differentiable Gaussian-splatting
 * rasterizer that Topo4D invokes at /root/reference train.py:307,388,463,484
 * (`Renderer(raster_settings=cam)(**rendervar)`; imports train.py:19, helpers.py:18-19), with the
 * boundary semantics of helpers.py:63-112 (transposed 4x4 matrices consumed flat/column-major,
 * (r,x,y,z) quaternions normalised by the caller, sigmoid/exp activations applied by the caller).
 *
 * PARITY UNPINNED.  The rasterizer package itself (ashawkey/diff-gaussian-rasterization, unpinned
 * HEAD, README.md:22-24) is NOT vendored under /root/reference (`diff-gaussian-rasterization-w-depth/`
 * is empty) and the reference holds no tests or golden vectors for it.  This file therefore restates
 * the library's published algorithm: preprocess -> duplicate-with-keys ->
 * stable radix sort on (tile | depth bits) -> tile ranges -> per-tile front-to-back alpha blend with
 * colour, depth and alpha outputs; backward = per-pixel back-to-front replay + conic/cov2D, projection,
 * cov3D and SH backward.  Every constant comes from include/t4d_config.h.  The backward formulas are
 * checked against torch.autograd over oracle/torch_oracle.py (float64) by tests/test_oracle.py.
 *
 * Quirks of the published library that are reproduced on purpose:
 *   - alpha = min(0.99, opacity*G) back-propagates as if unclamped;
 *   - the splat that would push T below 1e-4 is NOT blended, and stops the pixel;
 *   - n_contrib = 1-based position of the last splat that was blended;
 *   - the conic backward divides by (det^2 + 1e-7);
 *   - the frustum clamp zeroes dL/dt.x (dL/dt.y) when active; quaternions are not normalised here.
 *
 * Build:  gcc -O3 -march=native -ffp-contract=off -fopenmp -shared -fPIC raster_oracle.c -o libraster_oracle.so -lm
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "../include/t4d_config.h"

#define BX T4D_TILE_X
#define BY T4D_TILE_Y

typedef struct OrcState {
    int P, H, W, gx, gy, M, deg, use_sh;
    float tanfovx, tanfovy, scale_modifier;
    float view[16], proj[16], campos[3], bg[3];
    float *depth, *xy, *conic_opacity, *rgb, *cov3D;
    int *radii;
    uint32_t *tiles_touched, *offsets;
    uint8_t *clamped;
    uint64_t R;
    uint64_t *keys;
    uint32_t *point_list;
    uint32_t *ranges;           /* [gx*gy][2] */
    float *final_T;
    uint32_t *n_contrib;
} OrcState;

static inline float ndc2pix(float v, int S) { return ((v + 1.0f) * S - 1.0f) * 0.5f; }

static inline void get_rect(float px, float py, int r, int gx, int gy, int *x0, int *y0, int *x1, int *y1)
{
    int a;
    a = (int)((px - r) / BX);           *x0 = a < 0 ? 0 : (a > gx ? gx : a);
    a = (int)((py - r) / BY);           *y0 = a < 0 ? 0 : (a > gy ? gy : a);
    a = (int)((px + r + BX - 1) / BX);  *x1 = a < 0 ? 0 : (a > gx ? gx : a);
    a = (int)((py + r + BY - 1) / BY);  *y1 = a < 0 ? 0 : (a > gy ? gy : a);
}

static inline void xform4x3(const float *p, const float *m, float *o)
{
    o[0] = m[0] * p[0] + m[4] * p[1] + m[8] * p[2] + m[12];
    o[1] = m[1] * p[0] + m[5] * p[1] + m[9] * p[2] + m[13];
    o[2] = m[2] * p[0] + m[6] * p[1] + m[10] * p[2] + m[14];
}

static inline void xform4x4(const float *p, const float *m, float *o)
{
    o[0] = m[0] * p[0] + m[4] * p[1] + m[8] * p[2] + m[12];
    o[1] = m[1] * p[0] + m[5] * p[1] + m[9] * p[2] + m[13];
    o[2] = m[2] * p[0] + m[6] * p[1] + m[10] * p[2] + m[14];
    o[3] = m[3] * p[0] + m[7] * p[1] + m[11] * p[2] + m[15];
}

/* standard rotation matrix (row-major) of the un-normalised quaternion (r,x,y,z); external.py:26-43 */
static inline void quat_rot(const float *q, float R[9])
{
    float r = q[0], x = q[1], y = q[2], z = q[3];
    R[0] = 1.f - 2.f * (y * y + z * z); R[1] = 2.f * (x * y - r * z);       R[2] = 2.f * (x * z + r * y);
    R[3] = 2.f * (x * y + r * z);       R[4] = 1.f - 2.f * (x * x + z * z); R[5] = 2.f * (y * z - r * x);
    R[6] = 2.f * (x * z - r * y);       R[7] = 2.f * (y * z + r * x);       R[8] = 1.f - 2.f * (x * x + y * y);
}

/* Sigma = R S S^T R^T, six unique terms (xx xy xz yy yz zz) */
static void compute_cov3D(const float *scale, float mod, const float *q, float *cov)
{
    float R[9], M[9];
    quat_rot(q, R);
    float s[3] = { mod * scale[0], mod * scale[1], mod * scale[2] };
    for (int i = 0; i < 3; i++) for (int k = 0; k < 3; k++) M[i * 3 + k] = R[i * 3 + k] * s[k];
    cov[0] = M[0] * M[0] + M[1] * M[1] + M[2] * M[2];
    cov[1] = M[0] * M[3] + M[1] * M[4] + M[2] * M[5];
    cov[2] = M[0] * M[6] + M[1] * M[7] + M[2] * M[8];
    cov[3] = M[3] * M[3] + M[4] * M[4] + M[5] * M[5];
    cov[4] = M[3] * M[6] + M[4] * M[7] + M[5] * M[8];
    cov[5] = M[6] * M[6] + M[7] * M[7] + M[8] * M[8];
}

/* rows T0,T1 of T = J*W (EWA), also returns the clamp flags and the clamped t */
static void ewa_T(const float *mean, const float *view, float fx, float fy, float tanx, float tany,
                  float T0[3], float T1[3], float t[3], int *inx, int *iny)
{
    xform4x3(mean, view, t);
    float limx = T4D_FRUSTUM_CLAMP * tanx, limy = T4D_FRUSTUM_CLAMP * tany;
    float txtz = t[0] / t[2], tytz = t[1] / t[2];
    *inx = !(txtz < -limx || txtz > limx);
    *iny = !(tytz < -limy || tytz > limy);
    t[0] = fminf(limx, fmaxf(-limx, txtz)) * t[2];
    t[1] = fminf(limy, fmaxf(-limy, tytz)) * t[2];
    float J00 = fx / t[2], J02 = -(fx * t[0]) / (t[2] * t[2]);
    float J11 = fy / t[2], J12 = -(fy * t[1]) / (t[2] * t[2]);
    /* W rows: W[i][j] = view[j*4+i] */
    for (int j = 0; j < 3; j++) {
        T0[j] = J00 * view[j * 4 + 0] + J02 * view[j * 4 + 2];
        T1[j] = J11 * view[j * 4 + 1] + J12 * view[j * 4 + 2];
    }
}

static inline void sym_mul(const float *c, const float v[3], float o[3])
{
    o[0] = c[0] * v[0] + c[1] * v[1] + c[2] * v[2];
    o[1] = c[1] * v[0] + c[3] * v[1] + c[4] * v[2];
    o[2] = c[2] * v[0] + c[4] * v[1] + c[5] * v[2];
}

static void sh_basis(int deg, const float d[3], float *b)
{
    float x = d[0], y = d[1], z = d[2];
    b[0] = T4D_SH_C0;
    if (deg > 0) {
        b[1] = -T4D_SH_C1 * y; b[2] = T4D_SH_C1 * z; b[3] = -T4D_SH_C1 * x;
        if (deg > 1) {
            float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
            b[4] = T4D_SH_C2_0 * xy; b[5] = T4D_SH_C2_1 * yz; b[6] = T4D_SH_C2_2 * (2.f * zz - xx - yy);
            b[7] = T4D_SH_C2_3 * xz; b[8] = T4D_SH_C2_4 * (xx - yy);
            if (deg > 2) {
                b[9] = T4D_SH_C3_0 * y * (3.f * xx - yy);
                b[10] = T4D_SH_C3_1 * xy * z;
                b[11] = T4D_SH_C3_2 * y * (4.f * zz - xx - yy);
                b[12] = T4D_SH_C3_3 * z * (2.f * zz - 3.f * xx - 3.f * yy);
                b[13] = T4D_SH_C3_4 * x * (4.f * zz - xx - yy);
                b[14] = T4D_SH_C3_5 * z * (xx - yy);
                b[15] = T4D_SH_C3_6 * x * (xx - 3.f * yy);
            }
        }
    }
}

/* d(basis_k)/d(dir) */
static void sh_basis_grad(int deg, const float d[3], float *bx, float *by, float *bz)
{
    float x = d[0], y = d[1], z = d[2];
    int K = (deg + 1) * (deg + 1);
    for (int k = 0; k < K; k++) bx[k] = by[k] = bz[k] = 0.f;
    if (deg > 0) {
        by[1] = -T4D_SH_C1; bz[2] = T4D_SH_C1; bx[3] = -T4D_SH_C1;
        if (deg > 1) {
            float xx = x * x, yy = y * y, zz = z * z;
            bx[4] = T4D_SH_C2_0 * y; by[4] = T4D_SH_C2_0 * x;
            by[5] = T4D_SH_C2_1 * z; bz[5] = T4D_SH_C2_1 * y;
            bx[6] = T4D_SH_C2_2 * -2.f * x; by[6] = T4D_SH_C2_2 * -2.f * y; bz[6] = T4D_SH_C2_2 * 4.f * z;
            bx[7] = T4D_SH_C2_3 * z; bz[7] = T4D_SH_C2_3 * x;
            bx[8] = T4D_SH_C2_4 * 2.f * x; by[8] = T4D_SH_C2_4 * -2.f * y;
            if (deg > 2) {
                bx[9] = T4D_SH_C3_0 * 6.f * x * y;          by[9] = T4D_SH_C3_0 * (3.f * xx - 3.f * yy);
                bx[10] = T4D_SH_C3_1 * y * z;               by[10] = T4D_SH_C3_1 * x * z;  bz[10] = T4D_SH_C3_1 * x * y;
                bx[11] = T4D_SH_C3_2 * -2.f * x * y;        by[11] = T4D_SH_C3_2 * (4.f * zz - xx - 3.f * yy);
                bz[11] = T4D_SH_C3_2 * 8.f * y * z;
                bx[12] = T4D_SH_C3_3 * -6.f * x * z;        by[12] = T4D_SH_C3_3 * -6.f * y * z;
                bz[12] = T4D_SH_C3_3 * (6.f * zz - 3.f * xx - 3.f * yy);
                bx[13] = T4D_SH_C3_4 * (4.f * zz - 3.f * xx - yy); by[13] = T4D_SH_C3_4 * -2.f * x * y;
                bz[13] = T4D_SH_C3_4 * 8.f * x * z;
                bx[14] = T4D_SH_C3_5 * 2.f * x * z;         by[14] = T4D_SH_C3_5 * -2.f * y * z;
                bz[14] = T4D_SH_C3_5 * (xx - yy);
                bx[15] = T4D_SH_C3_6 * (3.f * xx - 3.f * yy); by[15] = T4D_SH_C3_6 * -6.f * x * y;
            }
        }
    }
}

/* stable LSD radix sort of (key,val) pairs, 16-bit digits */
static void radix_sort_pairs(uint64_t *keys, uint32_t *vals, uint64_t n)
{
    if (n < 2) return;
    uint64_t *k2 = (uint64_t *)malloc(n * sizeof(uint64_t));
    uint32_t *v2 = (uint32_t *)malloc(n * sizeof(uint32_t));
    uint64_t *cnt = (uint64_t *)malloc(65536 * sizeof(uint64_t));
    for (int pass = 0; pass < 4; pass++) {
        int sh = pass * 16;
        memset(cnt, 0, 65536 * sizeof(uint64_t));
        for (uint64_t i = 0; i < n; i++) cnt[(keys[i] >> sh) & 0xffff]++;
        uint64_t run = 0;
        for (int d = 0; d < 65536; d++) { uint64_t c = cnt[d]; cnt[d] = run; run += c; }
        for (uint64_t i = 0; i < n; i++) {
            uint64_t pos = cnt[(keys[i] >> sh) & 0xffff]++;
            k2[pos] = keys[i]; v2[pos] = vals[i];
        }
        uint64_t *tk = keys; keys = k2; k2 = tk;
        uint32_t *tv = vals; vals = v2; v2 = tv;
    }
    /* four passes: data is back in the caller's arrays */
    free(k2); free(v2); free(cnt);
}

void orc_free(void *h)
{
    OrcState *s = (OrcState *)h;
    if (!s) return;
    free(s->depth); free(s->xy); free(s->conic_opacity); free(s->rgb); free(s->cov3D); free(s->radii);
    free(s->tiles_touched); free(s->offsets); free(s->clamped); free(s->keys); free(s->point_list);
    free(s->ranges); free(s->final_T); free(s->n_contrib);
    free(s);
}

void *orc_forward(int P, int H, int W, int deg, int M, float tanfovx, float tanfovy, float scale_modifier,
                  const float *bg, const float *viewmatrix, const float *projmatrix, const float *campos,
                  const float *means3D, const float *shs, const float *colors_precomp, const float *opacities,
                  const float *scales, const float *rotations, const float *cov3D_precomp, int prefiltered,
                  float *out_color, float *out_depth, float *out_alpha, int *out_radii)
{
    (void)prefiltered;
    OrcState *s = (OrcState *)calloc(1, sizeof(OrcState));
    s->P = P; s->H = H; s->W = W; s->deg = deg; s->M = M; s->use_sh = (colors_precomp == NULL);
    s->gx = (W + BX - 1) / BX; s->gy = (H + BY - 1) / BY;
    s->tanfovx = tanfovx; s->tanfovy = tanfovy; s->scale_modifier = scale_modifier;
    memcpy(s->view, viewmatrix, 64); memcpy(s->proj, projmatrix, 64);
    memcpy(s->campos, campos, 12); memcpy(s->bg, bg, 12);
    size_t Pn = P > 0 ? (size_t)P : 1;
    s->depth = (float *)calloc(Pn, 4); s->xy = (float *)calloc(Pn * 2, 4);
    s->conic_opacity = (float *)calloc(Pn * 4, 4); s->rgb = (float *)calloc(Pn * 3, 4);
    s->cov3D = (float *)calloc(Pn * 6, 4); s->radii = (int *)calloc(Pn, 4);
    s->tiles_touched = (uint32_t *)calloc(Pn, 4); s->offsets = (uint32_t *)calloc(Pn, 4);
    s->clamped = (uint8_t *)calloc(Pn * 3, 1);
    const float focal_x = W / (2.0f * tanfovx), focal_y = H / (2.0f * tanfovy);
    const int gx = s->gx, gy = s->gy;

    /* ---- A.1 preprocess ---- */
#pragma omp parallel for schedule(static)
    for (int i = 0; i < P; i++) {
        const float *p = means3D + 3 * i;
        float pv[3], ph[4];
        xform4x3(p, viewmatrix, pv);
        if (pv[2] <= T4D_NEAR_CULL_Z) continue;
        xform4x4(p, projmatrix, ph);
        float pw = 1.0f / (ph[3] + T4D_HOM_W_EPS);
        float pp[3] = { ph[0] * pw, ph[1] * pw, ph[2] * pw };
        float *cov3 = s->cov3D + 6 * i;
        if (cov3D_precomp) memcpy(cov3, cov3D_precomp + 6 * i, 24);
        else compute_cov3D(scales + 3 * i, scale_modifier, rotations + 4 * i, cov3);
        float T0[3], T1[3], t[3]; int inx, iny;
        ewa_T(p, viewmatrix, focal_x, focal_y, tanfovx, tanfovy, T0, T1, t, &inx, &iny);
        float v0[3], v1[3];
        sym_mul(cov3, T0, v0); sym_mul(cov3, T1, v1);
        float a = T0[0] * v0[0] + T0[1] * v0[1] + T0[2] * v0[2] + T4D_COV2D_DILATION;
        float b = T0[0] * v1[0] + T0[1] * v1[1] + T0[2] * v1[2];
        float c = T1[0] * v1[0] + T1[1] * v1[1] + T1[2] * v1[2] + T4D_COV2D_DILATION;
        float det = a * c - b * b;
        if (det == 0.0f) continue;
        float det_inv = 1.f / det;
        float conic[3] = { c * det_inv, -b * det_inv, a * det_inv };
        float mid = 0.5f * (a + c);
        float l1 = mid + sqrtf(fmaxf(T4D_EIGEN_FLOOR, mid * mid - det));
        float l2 = mid - sqrtf(fmaxf(T4D_EIGEN_FLOOR, mid * mid - det));
        float my_radius = ceilf(T4D_RADIUS_SIGMAS * sqrtf(fmaxf(l1, l2)));
        float px = ndc2pix(pp[0], W), py = ndc2pix(pp[1], H);
        int x0, y0, x1, y1;
        get_rect(px, py, (int)my_radius, gx, gy, &x0, &y0, &x1, &y1);
        if ((x1 - x0) * (y1 - y0) == 0) continue;
        if (s->use_sh) {
            float d[3] = { p[0] - campos[0], p[1] - campos[1], p[2] - campos[2] };
            float len = sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
            d[0] /= len; d[1] /= len; d[2] /= len;
            float bas[16];
            sh_basis(deg, d, bas);
            int K = (deg + 1) * (deg + 1);
            for (int ch = 0; ch < 3; ch++) {
                float r = 0.f;
                for (int k = 0; k < K; k++) r += bas[k] * shs[((size_t)i * M + k) * 3 + ch];
                r += 0.5f;
                s->clamped[3 * i + ch] = r < 0.f;
                s->rgb[3 * i + ch] = fmaxf(r, 0.f);
            }
        } else {
            for (int ch = 0; ch < 3; ch++) s->rgb[3 * i + ch] = colors_precomp[3 * i + ch];
        }
        s->depth[i] = pv[2];
        s->radii[i] = (int)my_radius;
        s->xy[2 * i] = px; s->xy[2 * i + 1] = py;
        s->conic_opacity[4 * i] = conic[0]; s->conic_opacity[4 * i + 1] = conic[1];
        s->conic_opacity[4 * i + 2] = conic[2]; s->conic_opacity[4 * i + 3] = opacities[i];
        s->tiles_touched[i] = (uint32_t)((x1 - x0) * (y1 - y0));
    }
    if (out_radii) memcpy(out_radii, s->radii, (size_t)P * 4);

    /* ---- A.2 binning: inclusive scan, duplicate with keys, stable sort, ranges ---- */
    uint64_t run = 0;
    for (int i = 0; i < P; i++) { run += s->tiles_touched[i]; s->offsets[i] = (uint32_t)run; }
    s->R = run;
    s->keys = (uint64_t *)malloc((run ? run : 1) * 8);
    s->point_list = (uint32_t *)malloc((run ? run : 1) * 4);
    for (int i = 0; i < P; i++) {
        if (s->radii[i] <= 0) continue;
        uint64_t off = i == 0 ? 0 : s->offsets[i - 1];
        int x0, y0, x1, y1;
        get_rect(s->xy[2 * i], s->xy[2 * i + 1], s->radii[i], gx, gy, &x0, &y0, &x1, &y1);
        uint32_t dbits; memcpy(&dbits, &s->depth[i], 4);
        for (int y = y0; y < y1; y++)
            for (int x = x0; x < x1; x++) {
                uint64_t key = (uint64_t)(y * gx + x);
                key = (key << 32) | dbits;
                s->keys[off] = key; s->point_list[off] = (uint32_t)i; off++;
            }
    }
    radix_sort_pairs(s->keys, s->point_list, run);
    s->ranges = (uint32_t *)calloc((size_t)gx * gy * 2 + 2, 4);
    for (uint64_t k = 0; k < run; k++) {
        uint32_t tile = (uint32_t)(s->keys[k] >> 32);
        if (k == 0) s->ranges[2 * tile] = 0;
        else {
            uint32_t prev = (uint32_t)(s->keys[k - 1] >> 32);
            if (prev != tile) { s->ranges[2 * prev + 1] = (uint32_t)k; s->ranges[2 * tile] = (uint32_t)k; }
        }
        if (k == run - 1) s->ranges[2 * tile + 1] = (uint32_t)run;
    }

    /* ---- A.3 per-tile front-to-back alpha blend ---- */
    s->final_T = (float *)malloc((size_t)H * W * 4);
    s->n_contrib = (uint32_t *)malloc((size_t)H * W * 4);
    const size_t HW = (size_t)H * W;
#pragma omp parallel for schedule(dynamic, 4)
    for (int tile = 0; tile < gx * gy; tile++) {
        int ty = tile / gx, tx = tile % gx;
        uint32_t r0 = s->ranges[2 * tile], r1 = s->ranges[2 * tile + 1];
        for (int ly = 0; ly < BY; ly++) for (int lx = 0; lx < BX; lx++) {
            int x = tx * BX + lx, y = ty * BY + ly;
            if (x >= W || y >= H) continue;
            float pxf = (float)x, pyf = (float)y;
            float T = 1.0f, C[3] = { 0, 0, 0 }, weight = 0.f, D = 0.f;
            uint32_t contributor = 0, last_contributor = 0;
            for (uint32_t k = r0; k < r1; k++) {
                contributor++;
                uint32_t g = s->point_list[k];
                float dx = s->xy[2 * g] - pxf, dy = s->xy[2 * g + 1] - pyf;
                const float *co = s->conic_opacity + 4 * g;
                float power = -0.5f * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
                if (power > 0.0f) continue;
                float alpha = fminf(T4D_ALPHA_MAX, co[3] * expf(power));
                if (alpha < T4D_ALPHA_MIN) continue;
                float test_T = T * (1 - alpha);
                if (test_T < T4D_T_STOP) break;
                for (int ch = 0; ch < 3; ch++) C[ch] += s->rgb[3 * g + ch] * alpha * T;
                weight += alpha * T;
                D += s->depth[g] * alpha * T;
                T = test_T;
                last_contributor = contributor;
            }
            size_t pix = (size_t)y * W + x;
            s->final_T[pix] = T; s->n_contrib[pix] = last_contributor;
            for (int ch = 0; ch < 3; ch++) out_color[ch * HW + pix] = C[ch] + T * bg[ch];
            out_depth[pix] = D;
            out_alpha[pix] = weight;
        }
    }
    return s;
}

static inline void atomic_addf(float *p, float v)
{
#pragma omp atomic
    *p += v;
}

void orc_backward(void *h, const float *means3D, const float *shs, const float *scales, const float *rotations,
                  const float *cov3D_precomp,
                  const float *dL_dcolor, const float *dL_ddepth_px, const float *dL_dalpha_px,
                  float *dL_dmeans3D, float *dL_dmeans2D, float *dL_dcolors, float *dL_dshs, float *dL_dopacity,
                  float *dL_dscales, float *dL_drotations, float *dL_dcov3D)
{
    OrcState *s = (OrcState *)h;
    const int P = s->P, H = s->H, W = s->W, gx = s->gx, gy = s->gy, M = s->M, deg = s->deg;
    const size_t HW = (size_t)H * W;
    size_t Pn = P > 0 ? (size_t)P : 1;
    float *g_mean2D = (float *)calloc(Pn * 2, 4);   /* d/d(ndc xy): pixel gradient * 0.5*(W,H) */
    float *g_conic = (float *)calloc(Pn * 3, 4);    /* true d/d(A,B,C) */
    float *g_rgb = (float *)calloc(Pn * 3, 4);
    float *g_depth = (float *)calloc(Pn, 4);
    memset(dL_dopacity, 0, (size_t)P * 4);

    /* ---- A.4 per-pixel back-to-front replay ---- */
#pragma omp parallel for schedule(dynamic, 4)
    for (int tile = 0; tile < gx * gy; tile++) {
        int ty = tile / gx, tx = tile % gx;
        uint32_t r0 = s->ranges[2 * tile], r1 = s->ranges[2 * tile + 1];
        if (r1 <= r0) continue;
        for (int ly = 0; ly < BY; ly++) for (int lx = 0; lx < BX; lx++) {
            int x = tx * BX + lx, y = ty * BY + ly;
            if (x >= W || y >= H) continue;
            size_t pix = (size_t)y * W + x;
            float pxf = (float)x, pyf = (float)y;
            const float T_final = s->final_T[pix];
            float T = T_final;
            uint32_t last_contributor = s->n_contrib[pix];
            float dpix[3] = { dL_dcolor[pix], dL_dcolor[HW + pix], dL_dcolor[2 * HW + pix] };
            float ddep = dL_ddepth_px ? dL_ddepth_px[pix] : 0.f;
            float dalp = dL_dalpha_px ? dL_dalpha_px[pix] : 0.f;
            float accum_rec[3] = { 0, 0, 0 }, last_color[3] = { 0, 0, 0 };
            float accum_depth_rec = 0.f, last_depth = 0.f, accum_alpha_rec = 0.f, last_alpha = 0.f;
            const float ddelx_dx = 0.5f * W, ddely_dy = 0.5f * H;
            float bg_dot = s->bg[0] * dpix[0] + s->bg[1] * dpix[1] + s->bg[2] * dpix[2];
            for (uint32_t k = r0 + last_contributor; k-- > r0;) {
                uint32_t g = s->point_list[k];
                float dx = s->xy[2 * g] - pxf, dy = s->xy[2 * g + 1] - pyf;
                const float *co = s->conic_opacity + 4 * g;
                float power = -0.5f * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
                if (power > 0.0f) continue;
                float G = expf(power);
                float alpha = fminf(T4D_ALPHA_MAX, co[3] * G);
                if (alpha < T4D_ALPHA_MIN) continue;
                T = T / (1.f - alpha);
                float dchannel_dcolor = alpha * T;
                float dL_dalpha = 0.f;
                for (int ch = 0; ch < 3; ch++) {
                    float c = s->rgb[3 * g + ch];
                    accum_rec[ch] = last_alpha * last_color[ch] + (1.f - last_alpha) * accum_rec[ch];
                    last_color[ch] = c;
                    dL_dalpha += (c - accum_rec[ch]) * dpix[ch];
                    atomic_addf(&g_rgb[3 * g + ch], dchannel_dcolor * dpix[ch]);
                }
                float c_d = s->depth[g];
                accum_depth_rec = last_alpha * last_depth + (1.f - last_alpha) * accum_depth_rec;
                last_depth = c_d;
                dL_dalpha += (c_d - accum_depth_rec) * ddep;
                atomic_addf(&g_depth[g], dchannel_dcolor * ddep);
                accum_alpha_rec = last_alpha + (1.f - last_alpha) * accum_alpha_rec;
                dL_dalpha += (1.f - accum_alpha_rec) * dalp;
                dL_dalpha *= T;
                last_alpha = alpha;
                dL_dalpha += (-T_final / (1.f - alpha)) * bg_dot;
                float dL_dG = co[3] * dL_dalpha;
                float gdx = G * dx, gdy = G * dy;
                float dG_ddelx = -gdx * co[0] - gdy * co[1];
                float dG_ddely = -gdy * co[2] - gdx * co[1];
                atomic_addf(&g_mean2D[2 * g], dL_dG * dG_ddelx * ddelx_dx);
                atomic_addf(&g_mean2D[2 * g + 1], dL_dG * dG_ddely * ddely_dy);
                atomic_addf(&g_conic[3 * g], -0.5f * gdx * dx * dL_dG);
                atomic_addf(&g_conic[3 * g + 1], -gdx * dy * dL_dG);
                atomic_addf(&g_conic[3 * g + 2], -0.5f * gdy * dy * dL_dG);
                atomic_addf(&dL_dopacity[g], G * dL_dalpha);
            }
        }
    }

    /* ---- A.5 per-Gaussian backward ---- */
    const float focal_x = W / (2.0f * s->tanfovx), focal_y = H / (2.0f * s->tanfovy);
#pragma omp parallel for schedule(static)
    for (int i = 0; i < P; i++) {
        float *gm = dL_dmeans3D + 3 * i;
        gm[0] = gm[1] = gm[2] = 0.f;
        dL_dmeans2D[3 * i] = dL_dmeans2D[3 * i + 1] = dL_dmeans2D[3 * i + 2] = 0.f;
        if (dL_dcolors) dL_dcolors[3 * i] = dL_dcolors[3 * i + 1] = dL_dcolors[3 * i + 2] = 0.f;
        if (dL_dshs) memset(dL_dshs + (size_t)i * M * 3, 0, (size_t)M * 12);
        if (dL_dscales) dL_dscales[3 * i] = dL_dscales[3 * i + 1] = dL_dscales[3 * i + 2] = 0.f;
        if (dL_drotations) memset(dL_drotations + 4 * i, 0, 16);
        if (dL_dcov3D) memset(dL_dcov3D + 6 * i, 0, 24);
        if (s->radii[i] <= 0) continue;
        const float *p = means3D + 3 * i;
        const float *view = s->view, *proj = s->proj;
        const float *cov3 = s->cov3D + 6 * i;

        /* conic -> cov2D */
        float T0[3], T1[3], t[3]; int inx, iny;
        ewa_T(p, view, focal_x, focal_y, s->tanfovx, s->tanfovy, T0, T1, t, &inx, &iny);
        float v0[3], v1[3];
        sym_mul(cov3, T0, v0); sym_mul(cov3, T1, v1);
        float a = T0[0] * v0[0] + T0[1] * v0[1] + T0[2] * v0[2] + T4D_COV2D_DILATION;
        float b = T0[0] * v1[0] + T0[1] * v1[1] + T0[2] * v1[2];
        float c = T1[0] * v1[0] + T1[1] * v1[1] + T1[2] * v1[2] + T4D_COV2D_DILATION;
        float denom = a * c - b * b;
        float d2inv = 1.f / ((denom * denom) + T4D_CONIC_BWD_EPS);
        float X = g_conic[3 * i], Y = g_conic[3 * i + 1], Z = g_conic[3 * i + 2];
        float dL_da = 0.f, dL_db = 0.f, dL_dc = 0.f;
        float gcov[6] = { 0, 0, 0, 0, 0, 0 };
        if (d2inv != 0.f) {
            dL_da = d2inv * (-c * c * X + b * c * Y + (denom - a * c) * Z);
            dL_dc = d2inv * (-a * a * Z + a * b * Y + (denom - a * c) * X);
            dL_db = d2inv * (2.f * b * c * X - (denom + 2.f * b * b) * Y + 2.f * a * b * Z);
            gcov[0] = T0[0] * T0[0] * dL_da + T0[0] * T1[0] * dL_db + T1[0] * T1[0] * dL_dc;
            gcov[3] = T0[1] * T0[1] * dL_da + T0[1] * T1[1] * dL_db + T1[1] * T1[1] * dL_dc;
            gcov[5] = T0[2] * T0[2] * dL_da + T0[2] * T1[2] * dL_db + T1[2] * T1[2] * dL_dc;
            gcov[1] = 2.f * T0[0] * T0[1] * dL_da + (T0[0] * T1[1] + T0[1] * T1[0]) * dL_db + 2.f * T1[0] * T1[1] * dL_dc;
            gcov[2] = 2.f * T0[0] * T0[2] * dL_da + (T0[0] * T1[2] + T0[2] * T1[0]) * dL_db + 2.f * T1[0] * T1[2] * dL_dc;
            gcov[4] = 2.f * T0[2] * T0[1] * dL_da + (T0[1] * T1[2] + T0[2] * T1[1]) * dL_db + 2.f * T1[1] * T1[2] * dL_dc;
        }
        float dT0[3], dT1[3];
        for (int j = 0; j < 3; j++) {
            dT0[j] = 2.f * v0[j] * dL_da + v1[j] * dL_db;
            dT1[j] = 2.f * v1[j] * dL_dc + v0[j] * dL_db;
        }
        float dJ00 = 0, dJ02 = 0, dJ11 = 0, dJ12 = 0;
        for (int j = 0; j < 3; j++) {
            dJ00 += view[j * 4 + 0] * dT0[j]; dJ02 += view[j * 4 + 2] * dT0[j];
            dJ11 += view[j * 4 + 1] * dT1[j]; dJ12 += view[j * 4 + 2] * dT1[j];
        }
        float tz = 1.f / t[2], tz2 = tz * tz, tz3 = tz2 * tz;
        float dtx = (inx ? 1.f : 0.f) * -focal_x * tz2 * dJ02;
        float dty = (iny ? 1.f : 0.f) * -focal_y * tz2 * dJ12;
        float dtz = -focal_x * tz2 * dJ00 - focal_y * tz2 * dJ11 + (2.f * focal_x * t[0]) * tz3 * dJ02
                    + (2.f * focal_y * t[1]) * tz3 * dJ12;
        gm[0] = view[0] * dtx + view[1] * dty + view[2] * dtz;
        gm[1] = view[4] * dtx + view[5] * dty + view[6] * dtz;
        gm[2] = view[8] * dtx + view[9] * dty + view[10] * dtz;

        /* screen position -> mean (perspective divide) */
        float ph[4];
        xform4x4(p, proj, ph);
        float mw = 1.0f / (ph[3] + T4D_HOM_W_EPS);
        float mul1 = ph[0] * mw * mw, mul2 = ph[1] * mw * mw;
        float g2x = g_mean2D[2 * i], g2y = g_mean2D[2 * i + 1];
        gm[0] += (proj[0] * mw - proj[3] * mul1) * g2x + (proj[1] * mw - proj[3] * mul2) * g2y;
        gm[1] += (proj[4] * mw - proj[7] * mul1) * g2x + (proj[5] * mw - proj[7] * mul2) * g2y;
        gm[2] += (proj[8] * mw - proj[11] * mul1) * g2x + (proj[9] * mw - proj[11] * mul2) * g2y;
        dL_dmeans2D[3 * i] = g2x; dL_dmeans2D[3 * i + 1] = g2y;

        /* view depth -> mean */
        gm[0] += view[2] * g_depth[i]; gm[1] += view[6] * g_depth[i]; gm[2] += view[10] * g_depth[i];

        /* colour */
        if (s->use_sh) {
            float d0[3] = { p[0] - s->campos[0], p[1] - s->campos[1], p[2] - s->campos[2] };
            float len = sqrtf(d0[0] * d0[0] + d0[1] * d0[1] + d0[2] * d0[2]);
            float d[3] = { d0[0] / len, d0[1] / len, d0[2] / len };
            float bas[16], bx[16], by[16], bz[16];
            int K = (deg + 1) * (deg + 1);
            sh_basis(deg, d, bas); sh_basis_grad(deg, d, bx, by, bz);
            float gd[3] = { 0, 0, 0 };
            for (int ch = 0; ch < 3; ch++) {
                float gc = s->clamped[3 * i + ch] ? 0.f : g_rgb[3 * i + ch];
                for (int k = 0; k < K; k++) {
                    float sh = shs[((size_t)i * M + k) * 3 + ch];
                    if (dL_dshs) dL_dshs[((size_t)i * M + k) * 3 + ch] = bas[k] * gc;
                    gd[0] += bx[k] * sh * gc; gd[1] += by[k] * sh * gc; gd[2] += bz[k] * sh * gc;
                }
            }
            float dot = d[0] * gd[0] + d[1] * gd[1] + d[2] * gd[2];
            for (int j = 0; j < 3; j++) gm[j] += (gd[j] - d[j] * dot) / len;
        } else if (dL_dcolors) {
            for (int ch = 0; ch < 3; ch++) dL_dcolors[3 * i + ch] = g_rgb[3 * i + ch];
        }

        /* cov3D -> scale, rotation */
        if (cov3D_precomp) {
            if (dL_dcov3D) memcpy(dL_dcov3D + 6 * i, gcov, 24);
        } else {
            float R[9];
            const float *q = rotations + 4 * i;
            quat_rot(q, R);
            float sc[3] = { s->scale_modifier * scales[3 * i], s->scale_modifier * scales[3 * i + 1],
                            s->scale_modifier * scales[3 * i + 2] };
            float Gs[9] = { gcov[0], 0.5f * gcov[1], 0.5f * gcov[2], 0.5f * gcov[1], gcov[3], 0.5f * gcov[4],
                            0.5f * gcov[2], 0.5f * gcov[4], gcov[5] };
            float Mp[9], dM[9], D[9];
            for (int r = 0; r < 3; r++) for (int k = 0; k < 3; k++) Mp[r * 3 + k] = R[r * 3 + k] * sc[k];
            for (int r = 0; r < 3; r++) for (int k = 0; k < 3; k++)
                dM[r * 3 + k] = 2.f * (Gs[r * 3] * Mp[k] + Gs[r * 3 + 1] * Mp[3 + k] + Gs[r * 3 + 2] * Mp[6 + k]);
            for (int k = 0; k < 3; k++) {
                float ds = dM[k] * R[k] + dM[3 + k] * R[3 + k] + dM[6 + k] * R[6 + k];
                if (dL_dscales) dL_dscales[3 * i + k] = s->scale_modifier * ds;
                for (int r = 0; r < 3; r++) D[r * 3 + k] = dM[r * 3 + k] * sc[k];
            }
            float r = q[0], x = q[1], y = q[2], z = q[3];
            if (dL_drotations) {
                float *gq = dL_drotations + 4 * i;
                gq[0] = 2.f * z * (D[3] - D[1]) + 2.f * y * (D[2] - D[6]) + 2.f * x * (D[7] - D[5]);
                gq[1] = 2.f * y * (D[1] + D[3]) + 2.f * z * (D[2] + D[6]) + 2.f * r * (D[7] - D[5]) - 4.f * x * (D[4] + D[8]);
                gq[2] = 2.f * x * (D[1] + D[3]) + 2.f * r * (D[2] - D[6]) + 2.f * z * (D[5] + D[7]) - 4.f * y * (D[0] + D[8]);
                gq[3] = 2.f * r * (D[3] - D[1]) + 2.f * x * (D[2] + D[6]) + 2.f * y * (D[5] + D[7]) - 4.f * z * (D[0] + D[4]);
            }
        }
    }
    free(g_mean2D); free(g_conic); free(g_rgb); free(g_depth);
}

/* ---- introspection for tests: sizes and raw state ---- */
uint64_t orc_num_rendered(void *h) { return ((OrcState *)h)->R; }
const float *orc_xy(void *h) { return ((OrcState *)h)->xy; }
const float *orc_depth(void *h) { return ((OrcState *)h)->depth; }
const float *orc_conic_opacity(void *h) { return ((OrcState *)h)->conic_opacity; }
const float *orc_rgb(void *h) { return ((OrcState *)h)->rgb; }
const float *orc_cov3D(void *h) { return ((OrcState *)h)->cov3D; }
const uint32_t *orc_point_list(void *h) { return ((OrcState *)h)->point_list; }
const uint32_t *orc_ranges(void *h) { return ((OrcState *)h)->ranges; }
const float *orc_final_T(void *h) { return ((OrcState *)h)->final_T; }
const uint32_t *orc_n_contrib(void *h) { return ((OrcState *)h)->n_contrib; }
const uint32_t *orc_tiles_touched(void *h) { return ((OrcState *)h)->tiles_touched; }

/* markVisible of the published API: frustum test only */
void orc_mark_visible(int P, const float *means3D, const float *viewmatrix, uint8_t *present)
{
    for (int i = 0; i < P; i++) {
        float pv[3];
        xform4x3(means3D + 3 * i, viewmatrix, pv);
        present[i] = pv[2] > T4D_NEAR_CULL_Z;
    }
}
