/*
 * t4d_config.h — every constant of the Gaussian-splatting rasterizer hot path, in one place.
 *
 * Provenance. Topo4D imports the rasterizer at /root/reference train.py:19 and helpers.py:18-19
 * (`diff_gaussian_rasterization`), but the package source is NOT vendored in the reference tree
 * (`diff-gaussian-rasterization-w-depth/` is an empty directory; README.md:22-24 installs
 * ashawkey/diff-gaussian-rasterization at an unpinned HEAD).  The values below restate the
 * published behaviour of that library; none of them can carry a
 * reference file:line.  The CPU oracle (oracle/raster_oracle.c, oracle/torch_oracle.py) and
 * the HIP kernels (topo4d_amd/csrc) all read THIS header, so a constant can only be changed
 * for every implementation at once.
 */
#ifndef T4D_CONFIG_H
#define T4D_CONFIG_H

/* screen-space tiling: one tile = one 256-thread workgroup = 4 wave64 */
#define T4D_TILE_X 16
#define T4D_TILE_Y 16
#define T4D_TILE_PIXELS (T4D_TILE_X * T4D_TILE_Y)

/* preprocess (Appendix A.1) */
#define T4D_NEAR_CULL_Z      0.2f        /* cull when view-space z <= this                      */
#define T4D_HOM_W_EPS        0.0000001f  /* p_w = 1 / (p_hom.w + eps)                           */
#define T4D_FRUSTUM_CLAMP    1.3f        /* clamp t.x/t.z, t.y/t.z to +-1.3*tanfov in the EWA J */
#define T4D_COV2D_DILATION   0.3f        /* low-pass added to both diagonal terms of cov2D      */
#define T4D_EIGEN_FLOOR      0.1f        /* sqrt(max(floor, mid^2 - det))                       */
#define T4D_RADIUS_SIGMAS    3.0f        /* radius = ceil(3 * sqrt(lambda_max))                 */
#define T4D_CONIC_BWD_EPS    0.0000001f  /* 1 / (det^2 + eps) in the conic backward             */

/* alpha blend (Appendix A.3 / A.4) */
#define T4D_ALPHA_MAX        0.99f
#define T4D_ALPHA_MIN        (1.0f / 255.0f)
#define T4D_T_STOP           0.0001f

/* real spherical-harmonic basis constants (same values as reference helpers.py:836-863) */
#define T4D_SH_C0   0.28209479177387814f
#define T4D_SH_C1   0.4886025119029199f
#define T4D_SH_C2_0 1.0925484305920792f
#define T4D_SH_C2_1 -1.0925484305920792f
#define T4D_SH_C2_2 0.31539156525252005f
#define T4D_SH_C2_3 -1.0925484305920792f
#define T4D_SH_C2_4 0.5462742152960396f
#define T4D_SH_C3_0 -0.5900435899266435f
#define T4D_SH_C3_1 2.890611442640554f
#define T4D_SH_C3_2 -0.4570457994644658f
#define T4D_SH_C3_3 0.3731763325901154f
#define T4D_SH_C3_4 -0.4570457994644658f
#define T4D_SH_C3_5 1.445305721320277f
#define T4D_SH_C3_6 -0.5900435899266435f

#endif /* T4D_CONFIG_H */
