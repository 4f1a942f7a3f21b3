// t4d_raster_gaussian_bwd.h - part of the translation unit t4d_raster.hip (included there, inside its anonymous namespace; not a
// stand-alone header).  A.5: per-Gaussian backward (pair gather + chain rule), SH backward kernels, per-view dot, mark_visible.
// See t4d_raster.hip for the overview, the constants, the state layout and the kernel parameter block.
// ---------------------------------------------------------------------------------------------------------
// A.5 per-Gaussian backward: gather pair records, then the chain rule down to the inputs
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void preprocess_bwd(const KP &kp)
{
    // One launch index, decoded by block_and_view: the V workgroups that read the same 256 Gaussians run back to back on one XCD
    // (round 2's view-fastest order: config 4 516 -> 484 us; all views on ONE XCD: another 1 %).  The same order makes
    // k_preprocess SLOWER (210 -> 290 us, measured in rounds 2 and 3) and is not used there.  The spare workgroups of the
    // per-view dot sit behind all of them.
    const uint32_t nblocks = (uint32_t)(kp.P + kBlock - 1) / kBlock;
    const uint32_t n_pv = gaussian_grid(kp.P, kp.V);
    const bool spare = blockIdx.x >= n_pv;
    uint32_t gb = 0, vb = 0;
    if (!spare && !block_and_view(blockIdx.x, (uint32_t)kp.V, nblocks, gb, vb)) return;
    const int v = spare ? (int)(blockIdx.x - n_pv) : (int)vb;
    const int g = (int)gb * kBlock + threadIdx.x;
    if (spare) {
        // one spare workgroup per view: the view's <outputs, cotangents> = sum of its tiles' dots, in a fixed order
        __shared__ float s_w[4];
        const float4 *td = reinterpret_cast<const float4 *>(kp.tile_dot) + (size_t)v * kp.T;      // one float per wave of the tile
        const float *vb = kp.views + (size_t)v * T4D_VIEW_FLOATS + 35;
        const bool black = vb[0] == 0.f && vb[1] == 0.f && vb[2] == 0.f;        // then nobody wrote the empty tiles' entries
        const uint32_t *tc = kp.tile_count + (size_t)v * kp.T;
        float a = 0.f;
        for (int t = threadIdx.x; t < kp.T; t += kBlock) {
            if (tc[t] == 0u && black) continue;
            const float4 d4 = td[t];
            a += (d4.x + d4.y) + (d4.z + d4.w);
        }
        a = wave_sum_to_lane63(a);
        if ((threadIdx.x & 63) == 63) s_w[threadIdx.x >> 6] = a;
        __syncthreads();
        if (threadIdx.x == 0) kp.cotangent_dot[v] = kp.status->overflow != 0u ? 0.f : (s_w[0] + s_w[1]) + (s_w[2] + s_w[3]);
        return;
    }
    if (g >= kp.P) return;
    const size_t vg = (size_t)v * kp.P + g;
    const size_t gp = param_row0(kp, v) + (size_t)g;          // row of this Gaussian in its view's parameter set
    const ViewRecord vrec = load_view_record(kp.views, v);
    const float *view = vrec.view, *proj = vrec.proj;
    // Everything that depends on (view, Gaussian) alone is requested HERE, before any of it is used: as the kernel was written
    // (radius, then centre, then pair slot, then the records, then conic and mean, then scale and rotation) a thread went through
    // eight dependent round trips, and the kernel is made of those (a quarter of the vector ALUs busy).
    const uint32_t flag = kp.status->overflow;
    const int radius_in = kp.radii[vg];
    const float2 p2 = kp.xy[vg];
    const uint32_t base = kp.pair_off[vg];
    const float4 cq = kp.conic_opacity[vg];
    const float mean[3] = { kp.means3D[3 * gp], kp.means3D[3 * gp + 1], kp.means3D[3 * gp + 2] };
    float cov3_in[6] = { 0.f, 0.f, 0.f, 0.f, 0.f, 0.f };
    float4 q = make_float4(1.f, 0.f, 0.f, 0.f);
    float sc[3] = { 0.f, 0.f, 0.f };
    if (kp.cov3D_precomp) {
#pragma unroll
        for (int k = 0; k < 6; k++) cov3_in[k] = kp.cov3D_precomp[6 * gp + k];
    } else {
        sc[0] = kp.scales[3 * gp]; sc[1] = kp.scales[3 * gp + 1]; sc[2] = kp.scales[3 * gp + 2];
        q = reinterpret_cast<const float4 *>(kp.rotations)[gp];
    }
    const float4 q_raw = q;
    if (kp.raw_params && !kp.cov3D_precomp) {            // T4D_FLAG_RAW_PARAMS: activate as the forward did
        sc[0] = t4d_act_exp(sc[0]); sc[1] = t4d_act_exp(sc[1]); sc[2] = t4d_act_exp(sc[2]);
        q = t4d_act_normalize(q);
    }
    // A forward whose pair arena overflowed (possible only without T4D_FLAG_CHECKED) left tile lists truncated and pair
    // records unwritten: its backward returns ZERO gradients for every view instead of sums over uninitialised scratch.
    const bool truncated = flag != 0u;
    const int radius = truncated ? 0 : radius_in;

    float gm[3] = { 0.f, 0.f, 0.f }, g2x = 0.f, g2y = 0.f, gop = 0.f;
    float grgb[3] = { 0.f, 0.f, 0.f }, gsc[3] = { 0.f, 0.f, 0.f }, gq[4] = { 0.f, 0.f, 0.f, 0.f };
    float gcov[6] = { 0.f, 0.f, 0.f, 0.f, 0.f, 0.f };

    if (radius > 0) {
        // ---- gather the partial gradients of this Gaussian's tiles ----
        int x0, y0, x1, y1;
        tile_rect(p2.x, p2.y, radius, kp.gx, kp.gy, x0, y0, x1, y1);
        const uint32_t npairs = (uint32_t)((x1 - x0) * (y1 - y0));
        const float2 *gp = reinterpret_cast<const float2 *>(kp.grad_pair) + (size_t)v * kp.cap * (kGP / 2);
        float S0 = 0.f, S1 = 0.f, S2 = 0.f, S3 = 0.f, S4 = 0.f, S5 = 0.f, gdep = 0.f;
        for (uint32_t k = 0; k < npairs; k++) {
            const uint32_t pr = base + k;
            if (pr >= kp.cap) break;
            const float2 *rec = gp + (size_t)pr * (kGP / 2);           // 40-byte records: 8-byte aligned
            const float2 a0 = rec[0], a1 = rec[1], a2 = rec[2], a3 = rec[3], a4 = rec[4];
            S0 += a0.x; S1 += a0.y; S2 += a1.x; S3 += a1.y;
            S4 += a2.x; S5 += a2.y; grgb[0] += a3.x; grgb[1] += a3.y;
            grgb[2] += a4.x; gdep += a4.y;
        }
        // per-splat constants applied once (see k_render_bwd): dL/dG = opacity * dL/dalpha, dG/dd = -G * conic * d
        gop = S0;
        g2x = -cq.w * (cq.x * S1 + cq.y * S2) * (0.5f * kp.W);
        g2y = -cq.w * (cq.z * S2 + cq.y * S1) * (0.5f * kp.H);
        const float X = -0.5f * cq.w * S3, Y = -cq.w * S4, Z = -0.5f * cq.w * S5;      // true d/d(conic A, B, C)

        float cov3[6];
        if (kp.cov3D_precomp) {
#pragma unroll
            for (int k = 0; k < 6; k++) cov3[k] = cov3_in[k];
        } else {
            cov3d_from_scale_rot(sc, kp.scale_modifier, q, cov3);
        }
        const float tanx = vrec.tanx, tany = vrec.tany;
        const float fx = kp.W / (2.0f * tanx), fy = kp.H / (2.0f * tany);
        float T0[3], T1[3], t[3];
        bool inx, iny;
        ewa_rows(mean, view, fx, fy, tanx, tany, T0, T1, t, inx, iny);
        float v0[3], v1[3];
        sym3_mul(cov3, T0, v0);
        sym3_mul(cov3, T1, v1);
        const float a = T0[0] * v0[0] + T0[1] * v0[1] + T0[2] * v0[2] + T4D_COV2D_DILATION;
        const float b = T0[0] * v1[0] + T0[1] * v1[1] + T0[2] * v1[2];
        const float c = T1[0] * v1[0] + T1[1] * v1[1] + T1[2] * v1[2] + T4D_COV2D_DILATION;
        const float denom = a * c - b * b;
        const float d2inv = 1.f / ((denom * denom) + T4D_CONIC_BWD_EPS);
        float dL_da = 0.f, dL_db = 0.f, dL_dc = 0.f;
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
        float dJ00 = 0.f, dJ02 = 0.f, dJ11 = 0.f, dJ12 = 0.f;
#pragma unroll
        for (int j = 0; j < 3; j++) {
            const float dT0 = 2.f * v0[j] * dL_da + v1[j] * dL_db;
            const float dT1 = 2.f * v1[j] * dL_dc + v0[j] * dL_db;
            dJ00 += view[j * 4 + 0] * dT0; dJ02 += view[j * 4 + 2] * dT0;
            dJ11 += view[j * 4 + 1] * dT1; dJ12 += view[j * 4 + 2] * dT1;
        }
        const float tz = 1.f / t[2], tz2 = tz * tz, tz3 = tz2 * tz;
        const float dtx = (inx ? 1.f : 0.f) * -fx * tz2 * dJ02;
        const float dty = (iny ? 1.f : 0.f) * -fy * tz2 * dJ12;
        const float dtz = -fx * tz2 * dJ00 - fy * tz2 * dJ11 + (2.f * fx * t[0]) * tz3 * dJ02 + (2.f * fy * t[1]) * tz3 * dJ12;
        gm[0] = view[0] * dtx + view[1] * dty + view[2] * dtz;
        gm[1] = view[4] * dtx + view[5] * dty + view[6] * dtz;
        gm[2] = view[8] * dtx + view[9] * dty + view[10] * dtz;

        // screen position -> mean (perspective divide)
        const float hx = proj[0] * mean[0] + proj[4] * mean[1] + proj[8] * mean[2] + proj[12];
        const float hy = proj[1] * mean[0] + proj[5] * mean[1] + proj[9] * mean[2] + proj[13];
        const float hw = proj[3] * mean[0] + proj[7] * mean[1] + proj[11] * mean[2] + proj[15];
        const float mw = 1.0f / (hw + T4D_HOM_W_EPS);
        const float mul1 = hx * mw * mw, mul2 = hy * mw * mw;
        gm[0] += (proj[0] * mw - proj[3] * mul1) * g2x + (proj[1] * mw - proj[3] * mul2) * g2y;
        gm[1] += (proj[4] * mw - proj[7] * mul1) * g2x + (proj[5] * mw - proj[7] * mul2) * g2y;
        gm[2] += (proj[8] * mw - proj[11] * mul1) * g2x + (proj[9] * mw - proj[11] * mul2) * g2y;
        // view depth -> mean
        gm[0] += view[2] * gdep; gm[1] += view[6] * gdep; gm[2] += view[10] * gdep;

        // colour: with precomputed RGB the pair sums ARE dL/dcolour; with SH colours they go to k_sh_bwd through a scratch
        // array (kp.dL_dcolors points at it), which also adds the view-direction term to dL/dmeans3D
        // cov3D -> scale, rotation
        if (!kp.cov3D_precomp) {
            float R[9];
            quat_rot(q, R);
            const float s[3] = { kp.scale_modifier * sc[0], kp.scale_modifier * sc[1], kp.scale_modifier * sc[2] };
            const float Gs[9] = { gcov[0], 0.5f * gcov[1], 0.5f * gcov[2], 0.5f * gcov[1], gcov[3], 0.5f * gcov[4],
                                  0.5f * gcov[2], 0.5f * gcov[4], gcov[5] };
            float Mp[9], D[9];
#pragma unroll
            for (int r = 0; r < 3; r++)
#pragma unroll
                for (int k = 0; k < 3; k++) Mp[r * 3 + k] = R[r * 3 + k] * s[k];
#pragma unroll
            for (int k = 0; k < 3; k++) {
                float dM[3];
#pragma unroll
                for (int r = 0; r < 3; r++)
                    dM[r] = 2.f * (Gs[r * 3] * Mp[k] + Gs[r * 3 + 1] * Mp[3 + k] + Gs[r * 3 + 2] * Mp[6 + k]);
                gsc[k] = kp.scale_modifier * (dM[0] * R[k] + dM[1] * R[3 + k] + dM[2] * R[6 + k]);
#pragma unroll
                for (int r = 0; r < 3; r++) D[r * 3 + k] = dM[r] * s[k];
            }
            const float r = q.x, x = q.y, y = q.z, z = q.w;
            gq[0] = 2.f * z * (D[3] - D[1]) + 2.f * y * (D[2] - D[6]) + 2.f * x * (D[7] - D[5]);
            gq[1] = 2.f * y * (D[1] + D[3]) + 2.f * z * (D[2] + D[6]) + 2.f * r * (D[7] - D[5]) - 4.f * x * (D[4] + D[8]);
            gq[2] = 2.f * x * (D[1] + D[3]) + 2.f * r * (D[2] - D[6]) + 2.f * z * (D[5] + D[7]) - 4.f * y * (D[0] + D[8]);
            gq[3] = 2.f * r * (D[3] - D[1]) + 2.f * x * (D[2] + D[6]) + 2.f * y * (D[5] + D[7]) - 4.f * z * (D[0] + D[4]);
        }
    }

    if (kp.raw_params && radius > 0) {                   // ... and return the gradients of the raw parameters (t4d_activate_backward's
        // arithmetic; an invisible Gaussian keeps its zeros - the forward stored nothing for it)
        gop = t4d_act_sigmoid_bwd(cq.w, gop);            // (conic_opacity.w is the activated opacity the forward stored)
        if (!kp.cov3D_precomp) {
#pragma unroll
            for (int k = 0; k < 3; k++) gsc[k] = t4d_act_exp_bwd(sc[k], gsc[k]);
            const float4 gr = t4d_act_normalize_bwd(q_raw, make_float4(gq[0], gq[1], gq[2], gq[3]));
            gq[0] = gr.x; gq[1] = gr.y; gq[2] = gr.z; gq[3] = gr.w;
        }
    }
    kp.dL_dmeans3D[vg * 3] = gm[0]; kp.dL_dmeans3D[vg * 3 + 1] = gm[1]; kp.dL_dmeans3D[vg * 3 + 2] = gm[2];
    kp.dL_dmeans2D[vg * 3] = g2x; kp.dL_dmeans2D[vg * 3 + 1] = g2y; kp.dL_dmeans2D[vg * 3 + 2] = 0.f;
    kp.dL_dopacities[vg] = gop;
    if (kp.dL_dcolors) { kp.dL_dcolors[vg * 3] = grgb[0]; kp.dL_dcolors[vg * 3 + 1] = grgb[1]; kp.dL_dcolors[vg * 3 + 2] = grgb[2]; }
    if (kp.cov3D_precomp) {
#pragma unroll
        for (int k = 0; k < 6; k++) kp.dL_dcov3D[vg * 6 + k] = gcov[k];
    } else {
        kp.dL_dscales[vg * 3] = gsc[0]; kp.dL_dscales[vg * 3 + 1] = gsc[1]; kp.dL_dscales[vg * 3 + 2] = gsc[2];
        reinterpret_cast<float4 *>(kp.dL_drotations)[vg] = make_float4(gq[0], gq[1], gq[2], gq[3]);
    }
}

#ifndef T4D_PBWD_WAVES
#define T4D_PBWD_WAVES 7         // 72 registers: seven waves per SIMD without a spill (config 2: 32.7 -> 31.3 us; eight spills: 38.0)
#endif
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(T4D_PBWD_WAVES, T4D_PBWD_WAVES))) void k_preprocess_bwd(const KP kp)
{
    preprocess_bwd(kp);
}

// SH colours (BASELINE config 4): dL/dshs and the view-direction term of dL/dmeans3D, AFTER k_preprocess_bwd.  Rounds 1-2 did
// this inside the per-Gaussian kernel: 48 coefficients AND 48 gradients per thread in registers took it to 137 registers
// (three waves per SIMD), both with 192-byte lane strides - 484 us at config 4, a quarter of the vector ALUs busy.  Here:
//   1. one thread per (view, Gaussian): direction, basis, masked dL/dcolour; the coefficients stream through (they are only
//      needed for the gradient of the view direction, sum_k grad(basis_k) * (sh_k . dL/dcolour), which goes to dL/dmeans3D);
//      basis and dL/dcolour go to LDS;
//   2. the workgroup writes dL/dshs[k][c] = basis_k * dL/dcolour_c of its 256 Gaussians as ONE contiguous 48 KiB stream,
//      16 bytes per lane - 553 MB per step at config 4, the bulk of this kernel's traffic.
// Launch index view-fastest, like k_preprocess_bwd: the V workgroups that read the same coefficient rows run back to back.
__global__ __launch_bounds__(kBlock) void k_sh_bwd(const KP kp)
{
    __shared__ float s_bas[kBlock][17];                  // basis (odd pitch: one row per lane without bank conflicts)
    __shared__ float s_gc[kBlock][4];                    // masked dL/dcolour (zero for an invisible Gaussian)
    const int tid = threadIdx.x;
    const uint32_t nblocks = (uint32_t)(kp.P + kBlock - 1) / kBlock;
    uint32_t pblock, vb;
    if (!block_and_view(blockIdx.x, (uint32_t)kp.V, nblocks, pblock, vb)) return;
    const int v = (int)vb;
    const int g0 = (int)pblock * kBlock;
    const int n = min(kBlock, kp.P - g0);                // Gaussians of this workgroup
    const int M3 = kp.M * 3;
    // ---- 1. per Gaussian
    if (tid < n) {
        const int g = g0 + tid;
        const size_t vg = (size_t)v * kp.P + g;
        const size_t gp = param_row0(kp, v) + (size_t)g;
        // a truncated forward (arena overflow without T4D_FLAG_CHECKED) returns zero gradients everywhere
        const bool vis = kp.status->overflow == 0u && kp.radii[vg] > 0;
        float gc[3] = { 0.f, 0.f, 0.f };
        float bas[16];
#pragma unroll
        for (int i = 0; i < 16; i++) bas[i] = 0.f;
        if (vis) {
            const float *vr = kp.views + (size_t)v * T4D_VIEW_FLOATS;
            const float d0[3] = { kp.means3D[3 * gp] - vr[32], kp.means3D[3 * gp + 1] - vr[33], kp.means3D[3 * gp + 2] - vr[34] };
            const float len = sqrtf(d0[0] * d0[0] + d0[1] * d0[1] + d0[2] * d0[2]);
            const float d[3] = { d0[0] / len, d0[1] / len, d0[2] / len };
            float bx[16], by[16], bz[16];
            sh_basis(kp.deg, d, bas);
            sh_basis_grad(kp.deg, d, bx, by, bz);
            const uint32_t cl = kp.clamped[vg];                                   // channels the forward clamped at zero carry no gradient
            const float *grgb = kp.dL_dcolors + vg * 3;                           // the pair sums, left here by k_preprocess_bwd
            gc[0] = (cl & 1u) ? 0.f : grgb[0]; gc[1] = (cl & 2u) ? 0.f : grgb[1]; gc[2] = (cl & 4u) ? 0.f : grgb[2];
            const int K = (kp.deg + 1) * (kp.deg + 1);
            float gd[3] = { 0.f, 0.f, 0.f };
            const float *sh = kp.shs + gp * M3;
            if ((kp.M & 3) == 0 && K == 16) {            // degree 3, 16-byte aligned rows: twelve 16-byte loads, consumed as they come
                const float4 *sh4 = reinterpret_cast<const float4 *>(sh);
                float c[48];
#pragma unroll
                for (int i = 0; i < 12; i++) { const float4 t4 = sh4[i]; c[4 * i] = t4.x; c[4 * i + 1] = t4.y; c[4 * i + 2] = t4.z; c[4 * i + 3] = t4.w; }
#pragma unroll
                for (int k = 0; k < 16; k++) {
                    const float t = c[3 * k] * gc[0] + c[3 * k + 1] * gc[1] + c[3 * k + 2] * gc[2];
                    gd[0] += bx[k] * t; gd[1] += by[k] * t; gd[2] += bz[k] * t;
                }
            } else {
#pragma unroll
                for (int k = 0; k < 16; k++) {
                    if (k < K) {
                        const float t = sh[k * 3] * gc[0] + sh[k * 3 + 1] * gc[1] + sh[k * 3 + 2] * gc[2];
                        gd[0] += bx[k] * t; gd[1] += by[k] * t; gd[2] += bz[k] * t;
                    } else {
                        bas[k] = 0.f;
                    }
                }
            }
            const float dot = d[0] * gd[0] + d[1] * gd[1] + d[2] * gd[2];         // the direction was normalised: project its gradient
            float *gm = kp.dL_dmeans3D + vg * 3;
#pragma unroll
            for (int jj = 0; jj < 3; jj++) gm[jj] += (gd[jj] - d[jj] * dot) / len;
        }
#pragma unroll
        for (int i = 0; i < 16; i++) s_bas[tid][i] = bas[i];
        s_gc[tid][0] = gc[0]; s_gc[tid][1] = gc[1]; s_gc[tid][2] = gc[2];
    }
    __syncthreads();
    // ---- 2. dL/dshs, as one contiguous stream
    float *out = kp.dL_dshs + ((size_t)v * kp.P + g0) * M3;
    if (kp.M == 16) {
        float4 *out4 = reinterpret_cast<float4 *>(out);
        for (int i = tid; i < n * 12; i += kBlock) {
            const int slot = i / 12, e0 = (i % 12) * 4;
            float o[4];
#pragma unroll
            for (int e = 0; e < 4; e++) o[e] = s_bas[slot][(e0 + e) / 3] * s_gc[slot][(e0 + e) % 3];
            out4[i] = make_float4(o[0], o[1], o[2], o[3]);
        }
    } else {
        // (coefficients beyond degree 3 - M > 16 - have no basis function: zero gradient)
        for (int i = tid; i < n * M3; i += kBlock) {
            const int slot = i / M3, e = i % M3, k = e / 3;
            out[i] = k < 16 ? s_bas[slot][k] * s_gc[slot][e % 3] : 0.f;
        }
    }
}

// The same for the case that matters (degree 3, M = 16: BASELINE config 4), built around how the coefficient rows travel.  Above,
// every lane fetches its own 192-byte row with twelve 16-byte loads: one load instruction touches 64 different cache lines, the
// rows of all resident waves (240 KB per CU) do not survive in the 32 KB L1 from one load to the next, and every view fetches
// them again.  Here a workgroup takes its 256 rows ONCE, as one contiguous 48 KiB stream (16 bytes per lane, consecutive lanes
// consecutive addresses), turns them through LDS into one row per lane held in registers, and then serves T4D_SHB_VIEWS views
// from them; the staging area is reused for the basis / dL/dcolour exchange of the write-out.
#ifndef T4D_SHB_VIEWS
#define T4D_SHB_VIEWS 8
#endif
__global__ __launch_bounds__(kBlock) void k_sh_bwd16(const KP kp)
{
    constexpr int kPitch = 52;                           // floats per staged row: 16-byte aligned, 13 (odd) 16-byte words -> no bank conflicts
    __shared__ __attribute__((aligned(16))) float s_raw[kBlock * kPitch];
    float (*s_bas)[17] = reinterpret_cast<float (*)[17]>(s_raw);                   // after the staging: basis (odd pitch) ...
    float (*s_gc)[4] = reinterpret_cast<float (*)[4]>(s_raw + kBlock * 17);        // ... and masked dL/dcolour
    static_assert(kBlock * 17 + kBlock * 4 <= kBlock * kPitch && (kBlock * 17) % 4 == 0, "the exchange arrays live inside the staging area");
    const int tid = threadIdx.x;
    const uint32_t nblocks = (uint32_t)(kp.P + kBlock - 1) / kBlock;
    const uint32_t ngroups = (uint32_t)(kp.V + T4D_SHB_VIEWS - 1) / T4D_SHB_VIEWS;
    uint32_t pblock, vgrp;
    if (!block_and_view(blockIdx.x, ngroups, nblocks, pblock, vgrp)) return;
    const int g0 = (int)pblock * kBlock;
    const int n = min(kBlock, kp.P - g0);                // Gaussians of this workgroup
    // (several parameter sets per launch: the host takes this kernel only when a set's views fill whole groups, so the group's
    // first view names the set of all of them)
    const size_t row0 = param_row0(kp, (int)vgrp * T4D_SHB_VIEWS);
    {
        const float4 *src = reinterpret_cast<const float4 *>(kp.shs + (row0 + (size_t)g0) * 48);
        for (int i = tid; i < n * 12; i += kBlock) {
            const int r = i / 12, part = i - r * 12;
            *reinterpret_cast<float4 *>(s_raw + r * kPitch + part * 4) = src[i];
        }
    }
    __syncthreads();
    float c[48];
#pragma unroll
    for (int i = 0; i < 12; i++) {
        const float4 t4 = *reinterpret_cast<const float4 *>(s_raw + min(tid, n - 1) * kPitch + 4 * i);
        c[4 * i] = t4.x; c[4 * i + 1] = t4.y; c[4 * i + 2] = t4.z; c[4 * i + 3] = t4.w;
    }
    __syncthreads();                                     // the rows are in registers: the staging area is free
    const int g = g0 + min(tid, n - 1);
    const float mean[3] = { kp.means3D[3 * (row0 + (size_t)g)], kp.means3D[3 * (row0 + (size_t)g) + 1], kp.means3D[3 * (row0 + (size_t)g) + 2] };
    const bool truncated = kp.status->overflow != 0u;   // a truncated forward (arena overflow without T4D_FLAG_CHECKED): zero gradients
    const int v_end = min(kp.V, (int)(vgrp + 1u) * T4D_SHB_VIEWS);
    for (int v = (int)vgrp * T4D_SHB_VIEWS; v < v_end; v++) {
        // ---- 1. per Gaussian: direction, basis, masked dL/dcolour, the view-direction term of dL/dmeans3D
        if (tid < n) {
            const size_t vg = (size_t)v * kp.P + g;
            const bool vis = !truncated && kp.radii[vg] > 0;
            float gc[3] = { 0.f, 0.f, 0.f };
            float bas[16];
#pragma unroll
            for (int i = 0; i < 16; i++) bas[i] = 0.f;
            if (vis) {
                const float *vr = kp.views + (size_t)v * T4D_VIEW_FLOATS;
                const float d0[3] = { mean[0] - vr[32], mean[1] - vr[33], mean[2] - vr[34] };
                const float len = sqrtf(d0[0] * d0[0] + d0[1] * d0[1] + d0[2] * d0[2]);
                const float d[3] = { d0[0] / len, d0[1] / len, d0[2] / len };
                float bx[16], by[16], bz[16];
                sh_basis(3, d, bas);
                sh_basis_grad(3, d, bx, by, bz);
                const uint32_t cl = kp.clamped[vg];                               // channels the forward clamped at zero carry no gradient
                const float *grgb = kp.dL_dcolors + vg * 3;                       // the pair sums, left here by k_preprocess_bwd
                gc[0] = (cl & 1u) ? 0.f : grgb[0]; gc[1] = (cl & 2u) ? 0.f : grgb[1]; gc[2] = (cl & 4u) ? 0.f : grgb[2];
                float gd[3] = { 0.f, 0.f, 0.f };
#pragma unroll
                for (int k = 0; k < 16; k++) {
                    const float t = c[3 * k] * gc[0] + c[3 * k + 1] * gc[1] + c[3 * k + 2] * gc[2];
                    gd[0] += bx[k] * t; gd[1] += by[k] * t; gd[2] += bz[k] * t;
                }
                const float dot = d[0] * gd[0] + d[1] * gd[1] + d[2] * gd[2];     // the direction was normalised: project its gradient
                float *gm = kp.dL_dmeans3D + vg * 3;
#pragma unroll
                for (int jj = 0; jj < 3; jj++) gm[jj] += (gd[jj] - d[jj] * dot) / len;
            }
#pragma unroll
            for (int i = 0; i < 16; i++) s_bas[tid][i] = bas[i];
            s_gc[tid][0] = gc[0]; s_gc[tid][1] = gc[1]; s_gc[tid][2] = gc[2];
        }
        __syncthreads();
        // ---- 2. dL/dshs of this view, as one contiguous stream
        float4 *out4 = reinterpret_cast<float4 *>(kp.dL_dshs + ((size_t)v * kp.P + g0) * 48);
        for (int i = tid; i < n * 12; i += kBlock) {
            const int slot = i / 12, e0 = (i - slot * 12) * 4;
            float o[4];
#pragma unroll
            for (int e = 0; e < 4; e++) o[e] = s_bas[slot][(e0 + e) / 3] * s_gc[slot][(e0 + e) % 3];
            out4[i] = make_float4(o[0], o[1], o[2], o[3]);
        }
        __syncthreads();                                 // the exchange arrays are rewritten by the next view
    }
}

// ---------------------------------------------------------------------------------------------------------
// per-view scalar <a, b> (e.g. the loss term sum(colour * dL/dcolour) each rank contributes to the loss gather):
// one pass over both images, deterministic two-level sum.
// ---------------------------------------------------------------------------------------------------------
constexpr int kDotBlocks = 64;

__global__ __launch_bounds__(kBlock) void k_view_dot_partial(const float *a, const float *b, size_t n, float *partial)
{
    __shared__ float s_w[4];
    const int v = blockIdx.y, blk = blockIdx.x, tid = threadIdx.x;
    const float *pa = a + (size_t)v * n, *pb = b + (size_t)v * n;
    const size_t per = (((n + kDotBlocks - 1) / kDotBlocks) + 3) & ~(size_t)3;     // multiple of 4
    const size_t lo = min(n, (size_t)blk * per), hi = min(n, lo + per);
    float acc = 0.f;
    if ((n & 3) == 0 && (((uintptr_t)a | (uintptr_t)b) & 15) == 0) {               // 16-byte loads
        for (size_t i = lo + (size_t)tid * 4; i < hi; i += (size_t)kBlock * 4) {
            const float4 x = *reinterpret_cast<const float4 *>(pa + i), y = *reinterpret_cast<const float4 *>(pb + i);
            acc += (x.x * y.x + x.y * y.y) + (x.z * y.z + x.w * y.w);
        }
    } else {
        for (size_t k = lo + tid; k < hi; k += kBlock) acc += pa[k] * pb[k];
    }
    acc = wave_sum_to_lane63(acc);
    if ((tid & 63) == 63) s_w[tid >> 6] = acc;
    __syncthreads();
    if (tid == 0) partial[(size_t)v * kDotBlocks + blk] = (s_w[0] + s_w[1]) + (s_w[2] + s_w[3]);
}

__global__ __launch_bounds__(64) void k_view_dot_final(const float *partial, float *out)
{
    const int v = blockIdx.x;
    float x = partial[(size_t)v * kDotBlocks + threadIdx.x];
    x = wave_sum_to_lane63(x);
    if (threadIdx.x == 63) out[v] = x;
}

__global__ __launch_bounds__(kBlock) void k_mark_visible(int P, const float *means3D, const float *view, uint8_t *present)
{
#pragma clang fp contract(off)
    const int g = blockIdx.x * kBlock + threadIdx.x;
    if (g >= P) return;
    const float z = view[2] * means3D[3 * (size_t)g] + view[6] * means3D[3 * (size_t)g + 1] +
                    view[10] * means3D[3 * (size_t)g + 2] + view[14];
    present[g] = z > T4D_NEAR_CULL_Z ? 1 : 0;
}

