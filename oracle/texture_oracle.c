/*
 * TEST INFRASTRUCTURE — NOT PRODUCT CODE.
 * Plain-C restatement of the reference's CPU UV-space triangle rasterizer
 *   /root/reference/face3d/mesh/cython/mesh_core.cpp:169-234  `_render_colors_core`
 *   (+ :25-52 `isPointInTri`, :55-84 `get_point_weight`),
 * reached from Topo4D's texture bake at helpers.py:953-960 via face3d/mesh/render.py:52-86.
 * Pinned: bit-for-bit against the REAL reference code compiled from its own source (oracle/_ref, oracle/build_ref.py)
 * by tests/test_texture_oracle.py, and against tests/golden/g5_render_colors.npz (outputs of that real library).
 *
 * Semantics restated: triangles in index order; pixel bbox = [ceil(min), floor(max)] clipped to the image; a pixel is
 * drawn when it is inside the triangle OR lies in the 2-pixel border ring of the image (mesh_core.cpp:211 — Topo4D's
 * dilation quirk, with extrapolated barycentrics); depth test is strict `>` so on equal depth the FIRST triangle wins.
 * Build: gcc -O2 -ffp-contract=off -shared -fPIC texture_oracle.c -o libtexture_oracle.so -lm
 */
#include <math.h>

typedef struct { float x, y; } pt;
static inline float dot2(pt a, pt b) { return a.x * b.x + a.y * b.y; }
static inline pt sub2(pt a, pt b) { pt r = { a.x - b.x, a.y - b.y }; return r; }

static void uv_of(pt p, pt p0, pt p1, pt p2, float *u, float *v)
{
    pt v0 = sub2(p2, p0), v1 = sub2(p1, p0), v2 = sub2(p, p0);
    float dot00 = dot2(v0, v0), dot01 = dot2(v0, v1), dot02 = dot2(v0, v2), dot11 = dot2(v1, v1), dot12 = dot2(v1, v2);
    float inverDeno;
    if (dot00 * dot11 - dot01 * dot01 == 0) inverDeno = 0;
    else inverDeno = 1 / (dot00 * dot11 - dot01 * dot01);
    *u = (dot11 * dot02 - dot01 * dot12) * inverDeno;
    *v = (dot00 * dot12 - dot01 * dot02) * inverDeno;
}

void tex_render_colors(float *image, const float *vertices, const int *triangles, const float *colors, float *depth_buffer,
                       int nver, int ntri, int h, int w, int c)
{
    (void)nver;
    for (int i = 0; i < ntri; i++) {
        int i0 = triangles[3 * i], i1 = triangles[3 * i + 1], i2 = triangles[3 * i + 2];
        pt p0 = { vertices[3 * i0], vertices[3 * i0 + 1] }, p1 = { vertices[3 * i1], vertices[3 * i1 + 1] },
           p2 = { vertices[3 * i2], vertices[3 * i2 + 1] };
        float d0 = vertices[3 * i0 + 2], d1 = vertices[3 * i1 + 2], d2 = vertices[3 * i2 + 2];
        int x_min = (int)ceilf(fminf(p0.x, fminf(p1.x, p2.x))); if (x_min < 0) x_min = 0;
        int x_max = (int)floorf(fmaxf(p0.x, fmaxf(p1.x, p2.x))); if (x_max > w - 1) x_max = w - 1;
        int y_min = (int)ceilf(fminf(p0.y, fminf(p1.y, p2.y))); if (y_min < 0) y_min = 0;
        int y_max = (int)floorf(fmaxf(p0.y, fmaxf(p1.y, p2.y))); if (y_max > h - 1) y_max = h - 1;
        if (x_max < x_min || y_max < y_min) continue;
        for (int y = y_min; y <= y_max; y++)
            for (int x = x_min; x <= x_max; x++) {
                pt p = { (float)x, (float)y };
                float u, v;
                uv_of(p, p0, p1, p2, &u, &v);
                int inside = (u >= 0) && (v >= 0) && (u + v < 1);
                if (p.x < 2 || p.x > w - 3 || p.y < 2 || p.y > h - 3 || inside) {
                    float w0 = 1 - u - v, w1 = v, w2 = u;
                    float pd = w0 * d0 + w1 * d1 + w2 * d2;
                    if (pd > depth_buffer[y * w + x]) {
                        for (int k = 0; k < c; k++)
                            image[(y * w + x) * c + k] = w0 * colors[c * i0 + k] + w1 * colors[c * i1 + k] + w2 * colors[c * i2 + k];
                        depth_buffer[y * w + x] = pd;
                    }
                }
            }
    }
}
