// gauss_bwd.hip -- per-Gaussian backward: one thread per Gaussian.
//
// Fuses, in one streaming pass, what the reference runs as two kernels plus eleven zero-fills:
//   * the segmented sum of the Gaussian's per-instance gradient slots written by the backward
//     blend (replaces the 11 global atomicAdd targets of DGR backward.cu:554-601),
//   * computeCov2DCUDA (backward.cu:144-274): dL/dconic -> dL/dcov2D -> dL/dcov3D, dL/dmean (cov path),
//   * preprocessCUDA bwd (backward.cu:346-406): projection term, depth term, SH bwd (:20-139),
//     computeCov3D bwd (:278-341) -> dL/dscale, dL/drot (w.r.t. the un-normalised quaternion).
// Every output row is written (exact zeros for radii <= 0, backward.cu:156,369), so the caller's
// gradient tensors need no memset (the reference zero-fills 116 B/Gaussian, rasterize_points.cu:160-170).
// Built with -ffp-contract=off: same evaluation order and rounding as oracle/gs_oracle.c.
//
// HBM traffic per Gaussian: read 48 B x traversed instances (the slots the backward blend actually wrote) + inputs;
// write 12 + 12 + 4 + 4 + 12 + 12 + 16 = 72 B.
#include "gsr_math.h"

// DGR backward.cu:20-139 computeColorFromSH (bwd): returns dL/dmean contribution, writes dL/dsh.
__device__ __forceinline__ float3 gsr_sh_backward(int idx, int deg, int M, float3 pos, const GsrCam& cam,
                                                  const float* __restrict__ shs, const uint8_t* __restrict__ clamped,
                                                  const float dcol[3], float* __restrict__ dL_dsh)
{
    const float dox = pos.x - cam.campos[0], doy = pos.y - cam.campos[1], doz = pos.z - cam.campos[2];
    const float len = sqrtf(dox * dox + doy * doy + doz * doz);
    const float x = dox / len, y = doy / len, z = doz / len;
    const float* sh = shs + (size_t)idx * M * 3;
    float* dsh = dL_dsh + (size_t)idx * M * 3;
    float dRGB[3], ddx[3] = { 0, 0, 0 }, ddy[3] = { 0, 0, 0 }, ddz[3] = { 0, 0, 0 };
#pragma unroll
    for (int k = 0; k < 3; k++) dRGB[k] = dcol[k] * (clamped[3 * idx + k] ? 0.f : 1.f);
#define SH(i, k) sh[(i) * 3 + (k)]
#define DSH(i, k) dsh[(i) * 3 + (k)]
#pragma unroll
    for (int k = 0; k < 3; k++) DSH(0, k) = GSR_SH_C0 * dRGB[k];
    if (deg > 0) {
        const float d1 = -GSR_SH_C1 * y, d2 = GSR_SH_C1 * z, d3 = -GSR_SH_C1 * x;
#pragma unroll
        for (int k = 0; k < 3; k++) {
            DSH(1, k) = d1 * dRGB[k]; DSH(2, k) = d2 * dRGB[k]; DSH(3, k) = d3 * dRGB[k];
            ddx[k] = -GSR_SH_C1 * SH(3, k); ddy[k] = -GSR_SH_C1 * SH(1, k); ddz[k] = GSR_SH_C1 * SH(2, k);
        }
        if (deg > 1) {
            const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
            const float d4 = GSR_SH_C2[0] * xy, d5 = GSR_SH_C2[1] * yz, d6 = GSR_SH_C2[2] * (2.f * zz - xx - yy);
            const float d7 = GSR_SH_C2[3] * xz, d8 = GSR_SH_C2[4] * (xx - yy);
#pragma unroll
            for (int k = 0; k < 3; k++) {
                DSH(4, k) = d4 * dRGB[k]; DSH(5, k) = d5 * dRGB[k]; DSH(6, k) = d6 * dRGB[k];
                DSH(7, k) = d7 * dRGB[k]; DSH(8, k) = d8 * dRGB[k];
                ddx[k] += GSR_SH_C2[0] * y * SH(4, k) + GSR_SH_C2[2] * 2.f * -x * SH(6, k) + GSR_SH_C2[3] * z * SH(7, k)
                          + GSR_SH_C2[4] * 2.f * x * SH(8, k);
                ddy[k] += GSR_SH_C2[0] * x * SH(4, k) + GSR_SH_C2[1] * z * SH(5, k) + GSR_SH_C2[2] * 2.f * -y * SH(6, k)
                          + GSR_SH_C2[4] * 2.f * -y * SH(8, k);
                ddz[k] += GSR_SH_C2[1] * y * SH(5, k) + GSR_SH_C2[2] * 2.f * 2.f * z * SH(6, k) + GSR_SH_C2[3] * x * SH(7, k);
            }
            if (deg > 2) {
                const float d9 = GSR_SH_C3[0] * y * (3.f * xx - yy), d10 = GSR_SH_C3[1] * xy * z;
                const float d11 = GSR_SH_C3[2] * y * (4.f * zz - xx - yy);
                const float d12 = GSR_SH_C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy);
                const float d13 = GSR_SH_C3[4] * x * (4.f * zz - xx - yy), d14 = GSR_SH_C3[5] * z * (xx - yy);
                const float d15 = GSR_SH_C3[6] * x * (xx - 3.f * yy);
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    DSH(9, k) = d9 * dRGB[k]; DSH(10, k) = d10 * dRGB[k]; DSH(11, k) = d11 * dRGB[k];
                    DSH(12, k) = d12 * dRGB[k]; DSH(13, k) = d13 * dRGB[k]; DSH(14, k) = d14 * dRGB[k];
                    DSH(15, k) = d15 * dRGB[k];
                    ddx[k] += (GSR_SH_C3[0] * SH(9, k) * 3.f * 2.f * xy + GSR_SH_C3[1] * SH(10, k) * yz
                               + GSR_SH_C3[2] * SH(11, k) * -2.f * xy + GSR_SH_C3[3] * SH(12, k) * -3.f * 2.f * xz
                               + GSR_SH_C3[4] * SH(13, k) * (-3.f * xx + 4.f * zz - yy) + GSR_SH_C3[5] * SH(14, k) * 2.f * xz
                               + GSR_SH_C3[6] * SH(15, k) * 3.f * (xx - yy));
                    ddy[k] += (GSR_SH_C3[0] * SH(9, k) * 3.f * (xx - yy) + GSR_SH_C3[1] * SH(10, k) * xz
                               + GSR_SH_C3[2] * SH(11, k) * (-3.f * yy + 4.f * zz - xx) + GSR_SH_C3[3] * SH(12, k) * -3.f * 2.f * yz
                               + GSR_SH_C3[4] * SH(13, k) * -2.f * xy + GSR_SH_C3[5] * SH(14, k) * -2.f * yz
                               + GSR_SH_C3[6] * SH(15, k) * -3.f * 2.f * xy);
                    ddz[k] += (GSR_SH_C3[1] * SH(10, k) * xy + GSR_SH_C3[2] * SH(11, k) * 4.f * 2.f * yz
                               + GSR_SH_C3[3] * SH(12, k) * 3.f * (2.f * zz - xx - yy) + GSR_SH_C3[4] * SH(13, k) * 4.f * 2.f * xz
                               + GSR_SH_C3[5] * SH(14, k) * (xx - yy));
                }
            }
        }
    }
#undef SH
#undef DSH
    // coefficients beyond the active degree get no gradient: exact zeros, as in the reference's zero-filled dL_dsh
    // (rasterize_points.cu:168; the usual 3DGS degree ramp runs with M = 16 and sh_degree 0..2)
    for (int i = (deg + 1) * (deg + 1) * 3; i < M * 3; i++) dsh[i] = 0.f;
    const float gx_ = ddx[0] * dRGB[0] + ddx[1] * dRGB[1] + ddx[2] * dRGB[2];
    const float gy_ = ddy[0] * dRGB[0] + ddy[1] * dRGB[1] + ddy[2] * dRGB[2];
    const float gz_ = ddz[0] * dRGB[0] + ddz[1] * dRGB[1] + ddz[2] * dRGB[2];
    // DGR auxiliary.h:107-118 dnormvdv
    const float sum2 = dox * dox + doy * doy + doz * doz;
    const float inv = 1.0f / sqrtf(sum2 * sum2 * sum2);
    float3 r;
    r.x = ((+sum2 - dox * dox) * gx_ - doy * dox * gy_ - doz * dox * gz_) * inv;
    r.y = (-dox * doy * gx_ + (sum2 - doy * doy) * gy_ - doz * doy * gz_) * inv;
    r.z = (-dox * doz * gx_ - doy * doz * gy_ + (sum2 - doz * doz) * gz_) * inv;
    return r;
}

// Everything the two kernels below share: the caller's arrays.
struct GsrGaussArgs {
    int P, D, M;
    GsrCam cam;
    const float* means3D; const int32_t* radii; const float* shs; const uint8_t* clamped; const float* scales;
    const float* rotations; const float* cov3D_precomp; const uint32_t* offsets; const uint32_t* tiles; int num_slots;
    const float4* slots; uint8_t* slot_written;
    uint32_t* heavy;  // [0] = number of heavy groups, [1] = groups fetched dynamically (both reset by the backward blend), [16 ..] = indices
    uint32_t* heavy_seen;  // host-mapped word (may be null), the host's hint for its NEXT backward: the one-wave kernel sets it to 1 at any
                           // heavy group, the heavy kernel (when launched) to 2 + the number of groups it was given
    int inline_heavy;      // 1 = no heavy kernel follows this launch: the one-wave kernel does the heavy groups itself
    float *dL_dmeans2D, *dL_dcolors, *dL_dopacity, *dL_dfeatures, *dL_dmeans3D, *dL_dcov3D, *dL_dsh, *dL_dscales, *dL_drotations;
};

// Groups of 64 consecutive Gaussians that own more than this many gradient slots go to the cooperative kernel (a few
// large splats: the bench scene has none, a scene of large splats has them in ~1 % of its groups -- and they were its tail).
#ifndef GSR_K7_HEAVY_SLOTS
#define GSR_K7_HEAVY_SLOTS 1024
#endif
// Sum of n consecutive staged entries of one field, in ascending order, eight LDS reads in flight at a time (the adds stay
// in order -- same bits as a plain loop --, only the read latency leaves the dependency chain: a plain loop took ~100
// cycles per entry, and a splat that owns a whole 512-entry round made that the launch's critical path).
__device__ __forceinline__ void gsr_sum_entries(double& acc, const float* sf, uint32_t n)
{
    uint32_t e = 0;
    for (; e + 8 <= n; e += 8) {
        float x[8];
#pragma unroll
        for (int j = 0; j < 8; j++) x[j] = sf[(e + j) * 12];
#pragma unroll
        for (int j = 0; j < 8; j++) acc += x[j];
    }
    for (; e < n; e++) acc += sf[e * 12];
}

// Second half of the per-Gaussian backward: from the 11 summed slot fields of Gaussian `idx` (acc) to its output rows.
__device__ __forceinline__ void gsr_gauss_finish(const GsrGaussArgs& A, const int idx, const bool vis, const double (&acc)[11],
                                                 const float3 m, const float3 sc, const float4 q)
{
    const GsrCam& cam = A.cam;
    const int D = A.D, M = A.M;
    const float* __restrict__ shs = A.shs;
    const uint8_t* __restrict__ clamped = A.clamped;
    const float* __restrict__ cov3D_precomp = A.cov3D_precomp;
    float* __restrict__ dL_dmeans2D = A.dL_dmeans2D; float* __restrict__ dL_dcolors = A.dL_dcolors;
    float* __restrict__ dL_dopacity = A.dL_dopacity; float* __restrict__ dL_dfeatures = A.dL_dfeatures;
    float* __restrict__ dL_dmeans3D = A.dL_dmeans3D; float* __restrict__ dL_dcov3D = A.dL_dcov3D;
    float* __restrict__ dL_dsh = A.dL_dsh; float* __restrict__ dL_dscales = A.dL_dscales; float* __restrict__ dL_drotations = A.dL_drotations;
    float gcol[3] = { (float)acc[0], (float)acc[1], (float)acc[2] }, gdepth = (float)acc[3], gfeat = (float)acc[4];
    float gm2x = (float)acc[5], gm2y = (float)acc[6], gcx = (float)acc[7], gcy = (float)acc[8], gcw = (float)acc[9];
    float gop = (float)acc[10];
    float dmean[3] = { 0, 0, 0 }, dcov[6] = { 0, 0, 0, 0, 0, 0 }, dscale[3] = { 0, 0, 0 }, dq[4] = { 0, 0, 0, 0 };

    if (vis) {
        float cov3D[6];
        if (cov3D_precomp) {
#pragma unroll
            for (int i = 0; i < 6; i++) cov3D[i] = cov3D_precomp[6 * idx + i];
        } else {
            gsr_cov3d(sc, cam.scale_modifier, q, cov3D);
        }
        GsrCov2D c2;
        gsr_cov2d(m, cam, cov3D, c2);
        const float* vm = cam.view;
        const float* proj = cam.proj;
        const float h_x = cam.focal_x, h_y = cam.focal_y;

        // ---- computeCov2DCUDA, backward.cu:175-273 ----
        const float x_grad_mul = (c2.txtz < -c2.limx || c2.txtz > c2.limx) ? 0.f : 1.f;
        const float y_grad_mul = (c2.tytz < -c2.limy || c2.tytz > c2.limy) ? 0.f : 1.f;
        const float a = c2.a, b = c2.b, c = c2.c;
        const float* T0 = c2.A0;
        const float* T1 = c2.A1;
        const float V[3][3] = { { cov3D[0], cov3D[1], cov3D[2] }, { cov3D[1], cov3D[3], cov3D[4] }, { cov3D[2], cov3D[4], cov3D[5] } };
        const float denom = a * c - b * b;
        float dL_da = 0, dL_db = 0, dL_dc = 0;
        const float denom2inv = 1.0f / ((denom * denom) + 0.0000001f);
        if (denom2inv != 0) {
            dL_da = denom2inv * (-c * c * gcx + 2 * b * c * gcy + (denom - a * c) * gcw);
            dL_dc = denom2inv * (-a * a * gcw + 2 * a * b * gcy + (denom - a * c) * gcx);
            dL_db = denom2inv * 2 * (b * c * gcx - (denom + 2 * b * b) * gcy + a * b * gcw);
            dcov[0] = (T0[0] * T0[0] * dL_da + T0[0] * T1[0] * dL_db + T1[0] * T1[0] * dL_dc);
            dcov[3] = (T0[1] * T0[1] * dL_da + T0[1] * T1[1] * dL_db + T1[1] * T1[1] * dL_dc);
            dcov[5] = (T0[2] * T0[2] * dL_da + T0[2] * T1[2] * dL_db + T1[2] * T1[2] * dL_dc);
            dcov[1] = 2 * T0[0] * T0[1] * dL_da + (T0[0] * T1[1] + T0[1] * T1[0]) * dL_db + 2 * T1[0] * T1[1] * dL_dc;
            dcov[2] = 2 * T0[0] * T0[2] * dL_da + (T0[0] * T1[2] + T0[2] * T1[0]) * dL_db + 2 * T1[0] * T1[2] * dL_dc;
            dcov[4] = 2 * T0[2] * T0[1] * dL_da + (T0[1] * T1[2] + T0[2] * T1[1]) * dL_db + 2 * T1[1] * T1[2] * dL_dc;
        }
        const float dL_dT00 = 2 * (T0[0] * V[0][0] + T0[1] * V[0][1] + T0[2] * V[0][2]) * dL_da + (T1[0] * V[0][0] + T1[1] * V[0][1] + T1[2] * V[0][2]) * dL_db;
        const float dL_dT01 = 2 * (T0[0] * V[1][0] + T0[1] * V[1][1] + T0[2] * V[1][2]) * dL_da + (T1[0] * V[1][0] + T1[1] * V[1][1] + T1[2] * V[1][2]) * dL_db;
        const float dL_dT02 = 2 * (T0[0] * V[2][0] + T0[1] * V[2][1] + T0[2] * V[2][2]) * dL_da + (T1[0] * V[2][0] + T1[1] * V[2][1] + T1[2] * V[2][2]) * dL_db;
        const float dL_dT10 = 2 * (T1[0] * V[0][0] + T1[1] * V[0][1] + T1[2] * V[0][2]) * dL_dc + (T0[0] * V[0][0] + T0[1] * V[0][1] + T0[2] * V[0][2]) * dL_db;
        const float dL_dT11 = 2 * (T1[0] * V[1][0] + T1[1] * V[1][1] + T1[2] * V[1][2]) * dL_dc + (T0[0] * V[1][0] + T0[1] * V[1][1] + T0[2] * V[1][2]) * dL_db;
        const float dL_dT12 = 2 * (T1[0] * V[2][0] + T1[1] * V[2][1] + T1[2] * V[2][2]) * dL_dc + (T0[0] * V[2][0] + T0[1] * V[2][1] + T0[2] * V[2][2]) * dL_db;
        const float dL_dJ00 = vm[0] * dL_dT00 + vm[4] * dL_dT01 + vm[8] * dL_dT02;
        const float dL_dJ02 = vm[2] * dL_dT00 + vm[6] * dL_dT01 + vm[10] * dL_dT02;
        const float dL_dJ11 = vm[1] * dL_dT10 + vm[5] * dL_dT11 + vm[9] * dL_dT12;
        const float dL_dJ12 = vm[2] * dL_dT10 + vm[6] * dL_dT11 + vm[10] * dL_dT12;
        const float tz = 1.f / c2.tz, tz2 = tz * tz, tz3 = tz2 * tz;
        const float dL_dtx = x_grad_mul * -h_x * tz2 * dL_dJ02;
        const float dL_dty = y_grad_mul * -h_y * tz2 * dL_dJ12;
        const float dL_dtz = -h_x * tz2 * dL_dJ00 - h_y * tz2 * dL_dJ11 + (2 * h_x * c2.tx) * tz3 * dL_dJ02 + (2 * h_y * c2.ty) * tz3 * dL_dJ12;
        dmean[0] = vm[0] * dL_dtx + vm[1] * dL_dty + vm[2] * dL_dtz;
        dmean[1] = vm[4] * dL_dtx + vm[5] * dL_dty + vm[6] * dL_dtz;
        dmean[2] = vm[8] * dL_dtx + vm[9] * dL_dty + vm[10] * dL_dtz;

        // ---- preprocessCUDA (bwd), backward.cu:371-396 ----
        const float hw = proj[3] * m.x + proj[7] * m.y + proj[11] * m.z + proj[15];
        const float m_w = 1.0f / (hw + 0.0000001f);
        const float mul1 = (proj[0] * m.x + proj[4] * m.y + proj[8] * m.z + proj[12]) * m_w * m_w;
        const float mul2 = (proj[1] * m.x + proj[5] * m.y + proj[9] * m.z + proj[13]) * m_w * m_w;
        const float ax = (proj[0] * m_w - proj[3] * mul1) * gm2x + (proj[1] * m_w - proj[3] * mul2) * gm2y;
        const float ay = (proj[4] * m_w - proj[7] * mul1) * gm2x + (proj[5] * m_w - proj[7] * mul2) * gm2y;
        const float az = (proj[8] * m_w - proj[11] * mul1) * gm2x + (proj[9] * m_w - proj[11] * mul2) * gm2y;
        dmean[0] += ax; dmean[1] += ay; dmean[2] += az;
        dmean[0] += vm[2] * gdepth; dmean[1] += vm[6] * gdepth; dmean[2] += vm[10] * gdepth;

        if (shs) {
            const float3 e = gsr_sh_backward(idx, D, M, m, cam, shs, clamped, gcol, dL_dsh);
            dmean[0] += e.x; dmean[1] += e.y; dmean[2] += e.z;
        }

        if (!cov3D_precomp) {
            // ---- computeCov3D (bwd), backward.cu:278-341 ----
            float R[3][3];
            gsr_quat_to_R(q, R);
            const float r = q.x, x = q.y, y = q.z, z = q.w;
            const float s[3] = { cam.scale_modifier * sc.x, cam.scale_modifier * sc.y, cam.scale_modifier * sc.z };
            float M2[3][3];
#pragma unroll
            for (int cc = 0; cc < 3; cc++)
#pragma unroll
                for (int rr = 0; rr < 3; rr++) M2[cc][rr] = (s[rr] * R[cc][rr]) * 2.0f;
            const float Dm[3][3] = { { dcov[0], 0.5f * dcov[1], 0.5f * dcov[2] }, { 0.5f * dcov[1], dcov[3], 0.5f * dcov[4] }, { 0.5f * dcov[2], 0.5f * dcov[4], dcov[5] } };
            float dMt[3][3];  // dMt[c][r] = dL_dM[r][c],  dL_dM[c][r] = sum_k M2[k][r] * Dm[c][k]
#pragma unroll
            for (int cc = 0; cc < 3; cc++)
#pragma unroll
                for (int rr = 0; rr < 3; rr++) dMt[rr][cc] = M2[0][rr] * Dm[cc][0] + M2[1][rr] * Dm[cc][1] + M2[2][rr] * Dm[cc][2];
#pragma unroll
            for (int i = 0; i < 3; i++) dscale[i] = R[0][i] * dMt[i][0] + R[1][i] * dMt[i][1] + R[2][i] * dMt[i][2];
#pragma unroll
            for (int i = 0; i < 3; i++)
#pragma unroll
                for (int rr = 0; rr < 3; rr++) dMt[i][rr] *= s[i];
            dq[0] = 2 * z * (dMt[0][1] - dMt[1][0]) + 2 * y * (dMt[2][0] - dMt[0][2]) + 2 * x * (dMt[1][2] - dMt[2][1]);
            dq[1] = 2 * y * (dMt[1][0] + dMt[0][1]) + 2 * z * (dMt[2][0] + dMt[0][2]) + 2 * r * (dMt[1][2] - dMt[2][1]) - 4 * x * (dMt[2][2] + dMt[1][1]);
            dq[2] = 2 * x * (dMt[1][0] + dMt[0][1]) + 2 * r * (dMt[2][0] - dMt[0][2]) + 2 * z * (dMt[1][2] + dMt[2][1]) - 4 * y * (dMt[2][2] + dMt[0][0]);
            dq[3] = 2 * r * (dMt[0][1] - dMt[1][0]) + 2 * x * (dMt[2][0] + dMt[0][2]) + 2 * y * (dMt[1][2] + dMt[2][1]) - 4 * z * (dMt[1][1] + dMt[0][0]);
        }
    } else if (shs && dL_dsh) {
        for (int i = 0; i < M * 3; i++) dL_dsh[(size_t)idx * M * 3 + i] = 0.f;
    }

    dL_dmeans2D[3 * idx] = gm2x; dL_dmeans2D[3 * idx + 1] = gm2y; dL_dmeans2D[3 * idx + 2] = 0.f;
    dL_dcolors[3 * idx] = gcol[0]; dL_dcolors[3 * idx + 1] = gcol[1]; dL_dcolors[3 * idx + 2] = gcol[2];
    dL_dopacity[idx] = gop;
    dL_dfeatures[idx] = gfeat;
    dL_dmeans3D[3 * idx] = dmean[0]; dL_dmeans3D[3 * idx + 1] = dmean[1]; dL_dmeans3D[3 * idx + 2] = dmean[2];
    if (dL_dcov3D) {
#pragma unroll
        for (int i = 0; i < 6; i++) dL_dcov3D[6 * idx + i] = dcov[i];
    }
    if (dL_dscales) { dL_dscales[3 * idx] = dscale[0]; dL_dscales[3 * idx + 1] = dscale[1]; dL_dscales[3 * idx + 2] = dscale[2]; }
    if (dL_drotations) reinterpret_cast<float4*>(dL_drotations)[idx] = make_float4(dq[0], dq[1], dq[2], dq[3]);
}

#ifndef GSR_K7_WAVES
#define GSR_K7_WAVES 4  // minimum waves per SIMD the register allocation has to leave room for
#endif
// LIGHT groups (at most GSR_K7_HEAVY_SLOTS gradient slots in the group's range): one wavefront per group.
__global__ void __launch_bounds__(64, GSR_K7_WAVES) gsr_gauss_bwd_kernel(const GsrGaussArgs A)
{
    // One wave per block owns 64 consecutive Gaussians; their gradient slots are one contiguous range [S0, S1) of
    // `slots` (offsets[] is the exclusive scan of the per-Gaussian slot counts).  The backward blend writes a slot only
    // for instances it traversed (slot_written[s] = 1; cleared by the forward's tile sort and left set here: which slots a backward
    // writes depends on the forward state only, so a second backward over the same forward sets exactly the same flags): on the bench scene three quarters of the
    // instances lie behind the depth where their tile saturates.  The wave therefore
    //   1. reads the flag bytes of its range (coalesced), turns them into 64-bit masks + running counts (ballot),
    //   2. builds the compact list of written slots and fetches only those, 3 lanes per 48-byte slot, into LDS,
    //   3. lets every lane sum its own written slots out of LDS in ascending slot order (double accumulators).
    // Three dependent memory round trips per wave; the per-Gaussian inputs of the second half are requested before any of it.
    constexpr int BS = 64;
    constexpr int FCH = 512;  // flags per pass (per-wave ranges average ~170 slots)
    constexpr int WCH = 128;  // written slots staged per sub-pass (48 B each)
    __shared__ float4 stage[WCH * 3];
    __shared__ uint16_t wl[FCH];
#ifdef GSR_K7_BYTE_FLAGS
    __shared__ unsigned long long gmask[FCH / 64];
    __shared__ uint32_t gbase[FCH / 64 + 1];
#endif
    const int P = A.P, num_slots = A.num_slots;
    const uint32_t* __restrict__ offsets = A.offsets;
    const float4* __restrict__ slots = A.slots;
    const uint8_t* __restrict__ slot_written = A.slot_written;
    const int lane = threadIdx.x;
    const int g0 = blockIdx.x * BS;
    const int idx = g0 + lane;
    const bool live = idx < P;
    const uint32_t cnt = live ? A.tiles[idx] : 0u;
    // clamped to the slot count: a forward that binned nothing (num_slots = 0) need not have produced offsets
    const uint32_t S1 = min((g0 + BS < P) ? offsets[g0 + BS] : (uint32_t)num_slots, (uint32_t)num_slots);
    const uint32_t S0 = min(offsets[g0], S1);
    if (S1 - S0 > GSR_K7_HEAVY_SLOTS) {  // a group with large splats (wave-uniform)
        if (lane == 0 && A.heavy_seen) *A.heavy_seen = 1u;  // plain store, every such group writes the same value
        if (!A.inline_heavy) {                              // listed for gsr_gauss_bwd_heavy_kernel
            if (lane == 0) A.heavy[16 + atomicAdd(&A.heavy[0], 1u)] = (uint32_t)blockIdx.x;
            return;
        }
        // no heavy kernel was launched (the host had seen no heavy group in its previous backward): done here, window by
        // window on one wave -- same bits, slow, and only until the next call, which the store above switches over
    }
    const bool vis = live && A.radii[idx] > 0;
    const uint32_t off = live ? offsets[idx] : 0u;
    // early requests for the per-Gaussian half (independent of the slot phase)
    float3 m = make_float3(0.f, 0.f, 0.f), sc = make_float3(0.f, 0.f, 0.f);
    float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
    if (vis) {
        m = make_float3(A.means3D[3 * idx], A.means3D[3 * idx + 1], A.means3D[3 * idx + 2]);
        if (!A.cov3D_precomp) {
            sc = make_float3(A.scales[3 * idx], A.scales[3 * idx + 1], A.scales[3 * idx + 2]);
            q = reinterpret_cast<const float4*>(A.rotations)[idx];
        }
    }
    double acc[11] = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };  // the per-tile partials are summed in double, rounded once
    uint32_t any_written = 0;                              // (wave-uniform) some slot of the group was written
    // Waves that hold a Gaussian with more than 24 slots spread the summation over the lanes as (Gaussian, field) tasks
    // instead of letting that Gaussian's lane walk all its slots 11 fields at a time while 63 lanes wait: 11 consecutive
    // lanes share a Gaussian, each sums one field in the same ascending order (so both schemes give the same bits).
    // Wave-uniform choice from the slot counts.  The per-lane task sums stay in registers, acc[k] for task k * 64 + lane;
    // one transposition through LDS -- aliased onto `stage`, which is free by then -- hands every Gaussian's 11 sums to
    // its own lane.
    double* accG = reinterpret_cast<double*>(stage);
    static_assert(sizeof(double) * BS * 11 <= sizeof(float4) * WCH * 3, "accG must fit into the staging buffer");
    __shared__ uint32_t sCi0[BS], sCi1[BS];
    uint32_t cmax = cnt;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) cmax = max(cmax, (uint32_t)__shfl_xor((int)cmax, d, 64));
    const bool spread = cmax > 24u;
#ifndef GSR_K7_BYTE_FLAGS
    for (uint32_t base = S0, nf; base < S1; base += nf) {
        // One 8-byte load per lane covers the 512 flag bytes from the 8-aligned address below `base` (every pass but the
        // first starts aligned); lane l holds the flags of slots base - shift + 8 l + k, k = 0..7, as bit k of m.
        const uint32_t shift = base & 7u;
        nf = min(S1 - base, (uint32_t)FCH - shift);
        uint2 w8 = make_uint2(0u, 0u);
        if (8u * lane < shift + nf) w8 = *reinterpret_cast<const uint2*>(slot_written + (base - shift) + 8u * lane);
        // bytes are 0 / 1: a multiplication gathers the four low bits of a word's bytes in its top byte
        uint32_t m = (((w8.x & 0x01010101u) * 0x01020408u) >> 24) | ((((w8.y & 0x01010101u) * 0x01020408u) >> 24) << 4);
        {
            const int lo_k = min(max((int)shift - 8 * lane, 0), 8), hi_k = min(max((int)(shift + nf) - 8 * lane, 0), 8);
            m &= ((1u << hi_k) - 1u) & ~((1u << lo_k) - 1u);  // slots of the neighbouring groups / behind the range
        }
        const uint32_t c = (uint32_t)__popc(m), incl = gsr_wave_scan_add(c), excl = incl - c;
        const uint32_t run = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
#pragma unroll
        for (int k = 0; k < 8; k++)
            if ((m >> k) & 1u) wl[excl + (uint32_t)__popc(m & ((1u << k) - 1u))] = (uint16_t)(8 * lane + k - (int)shift);
        __syncthreads();
        // A lane's written slots are consecutive entries [ci0, ci1) of the compact list: ci = number of written slots
        // of the pass below the lane's first / past its last slot (the owner lane's running count + the set bits below the
        // position, fetched with one cross-lane read each).  No flag test is left in the summation loop.
        const uint32_t lo = min(max(off, base), base + nf) - base, hi = min(max(off + cnt, base), base + nf) - base;
        const uint32_t pk = (excl << 8) | m;
        uint32_t ci0, ci1;
        {
            const uint32_t a0 = lo + shift, a1 = hi + shift;
            const uint32_t p0 = (uint32_t)__shfl((int)pk, (int)min(a0 >> 3, 63u), 64), p1 = (uint32_t)__shfl((int)pk, (int)min(a1 >> 3, 63u), 64);
            ci0 = lo < nf ? (p0 >> 8) + (uint32_t)__popc(p0 & 0xffu & ((1u << (a0 & 7u)) - 1u)) : run;
            ci1 = hi < nf ? (p1 >> 8) + (uint32_t)__popc(p1 & 0xffu & ((1u << (a1 & 7u)) - 1u)) : run;
        }
#else
    for (uint32_t base = S0, nf; base < S1; base += nf) {
        nf = min(S1 - base, (uint32_t)FCH);
        uint8_t f[FCH / 64];
#pragma unroll
        for (int k = 0; k < FCH / 64; k++) {
            const uint32_t i = k * 64 + lane;
            f[k] = i < nf ? slot_written[base + i] : (uint8_t)0;
        }
        uint32_t run = 0;
#pragma unroll
        for (int k = 0; k < FCH / 64; k++) {
            const unsigned long long mk = __ballot(f[k] != 0);
            if (f[k]) wl[run + __builtin_amdgcn_mbcnt_hi((uint32_t)(mk >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mk, 0u))] = (uint16_t)(k * 64 + lane);
            if (lane == 0) { gmask[k] = mk; gbase[k] = run; }
            run += (uint32_t)__popcll(mk);
        }
        __syncthreads();
        const uint32_t lo = min(max(off, base), base + nf) - base, hi = min(max(off + cnt, base), base + nf) - base;
        uint32_t ci0 = run, ci1 = run;
        if (lo < nf) ci0 = gbase[lo >> 6] + (uint32_t)__popcll(gmask[lo >> 6] & ((1ull << (lo & 63)) - 1ull));
        if (hi < nf) ci1 = gbase[hi >> 6] + (uint32_t)__popcll(gmask[hi >> 6] & ((1ull << (hi & 63)) - 1ull));
#endif
        any_written |= run;
        for (uint32_t w0 = 0; w0 < run; w0 += WCH) {
            const uint32_t nw = min(run - w0, (uint32_t)WCH);
            float4 v[WCH * 3 / 64];
#pragma unroll
            for (int k = 0; k < WCH * 3 / 64; k++) {
                const uint32_t i = lane + k * 64;
                v[k] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (i < nw * 3) v[k] = slots[(size_t)(base + wl[w0 + i / 3]) * 3 + i % 3];
            }
#pragma unroll
            for (int k = 0; k < WCH * 3 / 64; k++) stage[lane + k * 64] = v[k];
            __syncthreads();
            if (spread) {
                if (w0 == 0) { sCi0[lane] = ci0; sCi1[lane] = ci1; }
                __syncthreads();
                const float* sf = reinterpret_cast<const float*>(stage);
#pragma unroll
                for (int k = 0; k < 11; k++) {
                    const int task = k * BS + lane, g = task / 11, v = task - g * 11;
                    const uint32_t e0 = max(sCi0[g], w0), e1 = min(sCi1[g], w0 + nw);
                    if (e1 > e0) gsr_sum_entries(acc[k], sf + (e0 - w0) * 12 + v, e1 - e0);
                }
            } else {
                const uint32_t e0 = max(ci0, w0), e1 = min(ci1, w0 + nw);
                for (uint32_t e = e0; e < e1; e++) {
                    const float4 a = stage[(e - w0) * 3], b = stage[(e - w0) * 3 + 1], c = stage[(e - w0) * 3 + 2];
                    acc[0] += a.x; acc[1] += a.y; acc[2] += a.z; acc[3] += a.w;
                    acc[4] += b.x; acc[5] += b.y; acc[6] += b.z; acc[7] += b.w;
                    acc[8] += c.x; acc[9] += c.y; acc[10] += c.z;
                }
            }
            __syncthreads();
        }
        __syncthreads();
    }
    if (spread) {  // transpose: task sums (k * 64 + lane) -> the 11 sums of Gaussian `lane`
#pragma unroll
        for (int k = 0; k < 11; k++) accG[k * BS + lane] = acc[k];
        __syncthreads();
#pragma unroll
        for (int v = 0; v < 11; v++) acc[v] = accG[lane * 11 + v];
    }
    if (!live) return;
    // A group none of whose slots was written (wave-uniform: every Gaussian of it culled, or behind the depth its tiles are walked to):
    // all eleven sums of all its lanes are zero and every row is an exact zero -- the second half is skipped.  (Round 6: in storage order
    // a trained scene keeps the offsets of an anchor, and mostly neighbouring anchors, next to each other, so whole groups are hidden
    // together; the synthetic bench cloud is stored in random order and has no such group: see DESIGN 11 for the compaction
    // experiments that tried to get the same saving there.)
    gsr_gauss_finish(A, idx, vis && any_written != 0u, acc, m, sc, q);
}

// HEAVY groups (more than GSR_K7_HEAVY_SLOTS gradient slots: a splat can cover thousands of tiles): four wavefronts per group.  A lone wave has ~6 KB of slot data in flight per memory round trip and walks a dense range of
// thousands of slots window by window (measured on a scene of large splats: the launch lasted 347 us, of which ~35 us'
// worth was arithmetic -- the rest was the few waves that own the big front splats).  Here the four waves scan 2048 flags
// per pass, gather four 128-slot windows per round trip and sum them as (Gaussian, field) tasks spread over all 256
// threads, each task in ascending slot order -- the same order, hence the same bits, as the one-wave scheme.  Wave 0
// then runs the second half for the 64 Gaussians.
__global__ void __launch_bounds__(256) gsr_gauss_bwd_heavy_kernel(const GsrGaussArgs A)
{
    constexpr int BS = 64, NW = 4, NT = 256;
    constexpr int FCH = 2048;      // flags per pass: 8 per thread
    constexpr int NG = FCH / 64;   // 64-flag groups per pass
    constexpr int WCH = 128;       // written slots per wave and round
    __shared__ float4 stage[NW * WCH * 3];  // 24 KiB: four windows, contiguous
    __shared__ uint16_t wl[FCH];
    __shared__ unsigned long long gmask[NG];
    __shared__ uint32_t gcnt[NG], gbase[NG + 1];
    __shared__ uint32_t sCi0[BS], sCi1[BS];
    const int P = A.P, num_slots = A.num_slots;
    const uint32_t* __restrict__ offsets = A.offsets;
    const float4* __restrict__ slots = A.slots;
    uint8_t* __restrict__ slot_written = A.slot_written;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const uint32_t nheavy = A.heavy[0];  // the groups gsr_gauss_bwd_kernel listed (any order: groups are independent)
    if (blockIdx.x == 0 && t == 0 && A.heavy_seen) *A.heavy_seen = 2u + nheavy;  // (after every store of the first kernel: stream order)
    __shared__ uint32_t s_next;
    // the first gridDim.x groups are dealt by block index, the rest is fetched from a counter as workgroups come free
    // (groups differ in size by an order of magnitude; a few hundred fetches per launch: the counter is not a bottleneck)
    for (uint32_t it = blockIdx.x; it < nheavy;) {
    const int g0 = (int)A.heavy[16 + it] * BS;
    const int idx = g0 + lane;  // every wave looks at the same 64 Gaussians
    const bool live = idx < P;
    const uint32_t S1 = min((g0 + BS < P) ? offsets[g0 + BS] : (uint32_t)num_slots, (uint32_t)num_slots);
    const uint32_t S0 = min(offsets[g0], S1);
    const uint32_t cnt = live ? A.tiles[idx] : 0u;
    const bool vis = live && A.radii[idx] > 0;
    const uint32_t off = live ? offsets[idx] : 0u;
    float3 m = make_float3(0.f, 0.f, 0.f), sc = make_float3(0.f, 0.f, 0.f);
    float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
    if (vis && wave == 0) {
        m = make_float3(A.means3D[3 * idx], A.means3D[3 * idx + 1], A.means3D[3 * idx + 2]);
        if (!A.cov3D_precomp) {
            sc = make_float3(A.scales[3 * idx], A.scales[3 * idx + 1], A.scales[3 * idx + 2]);
            q = reinterpret_cast<const float4*>(A.rotations)[idx];
        }
    }
    // (Gaussian, field) tasks: task k * 256 + t -> Gaussian task / 11, field task % 11; 704 tasks over 256 threads
    double tacc[3] = { 0, 0, 0 };
    int tg[3], tv[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const int task = k * NT + t;
        tg[k] = task < BS * 11 ? task / 11 : -1;
        tv[k] = task - (task / 11) * 11;
    }
    for (uint32_t base = S0; base < S1; base += FCH) {
        const uint32_t nf = min(S1 - base, (uint32_t)FCH);
        uint8_t f[FCH / NT];
#pragma unroll
        for (int k = 0; k < FCH / NT; k++) {
            const uint32_t i = k * NT + t;
            f[k] = i < nf ? slot_written[base + i] : (uint8_t)0;
        }
        unsigned long long mk[FCH / NT];
#pragma unroll
        for (int k = 0; k < FCH / NT; k++) {
            mk[k] = __ballot(f[k] != 0);
            if (lane == 0) { gmask[k * NW + wave] = mk[k]; gcnt[k * NW + wave] = (uint32_t)__popcll(mk[k]); }  // group = 64 consecutive flags
        }
        __syncthreads();
        if (t <= NG) {  // exclusive prefix of the 32 group counts (ascending slot order); gbase[NG] = written slots of the pass
            uint32_t s = 0;
            for (int g = 0; g < t; g++) s += gcnt[g];
            gbase[t] = s;
        }
        __syncthreads();
        const uint32_t run = gbase[NG];
#pragma unroll
        for (int k = 0; k < FCH / NT; k++)
            if (f[k]) wl[gbase[k * NW + wave] + __builtin_amdgcn_mbcnt_hi((uint32_t)(mk[k] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mk[k], 0u))] = (uint16_t)(k * NT + t);
        if (wave == 0) {  // a Gaussian's written slots = entries [ci0, ci1) of the pass's compact list
            const uint32_t lo = min(max(off, base), base + nf) - base, hi = min(max(off + cnt, base), base + nf) - base;
            uint32_t ci0 = run, ci1 = run;
            if (lo < nf) ci0 = gbase[lo >> 6] + (uint32_t)__popcll(gmask[lo >> 6] & ((1ull << (lo & 63)) - 1ull));
            if (hi < nf) ci1 = gbase[hi >> 6] + (uint32_t)__popcll(gmask[hi >> 6] & ((1ull << (hi & 63)) - 1ull));
            sCi0[lane] = ci0; sCi1[lane] = ci1;
        }
        __syncthreads();
        for (uint32_t w0 = 0; w0 < run; w0 += NW * WCH) {
            const uint32_t nall = min(run - w0, (uint32_t)(NW * WCH));   // entries of this round
            const uint32_t wb = (uint32_t)wave * WCH;                    // this wave's window inside the round
            const uint32_t nw = wb < nall ? min(nall - wb, (uint32_t)WCH) : 0u;
            float4 v[WCH * 3 / 64];
#pragma unroll
            for (int k = 0; k < WCH * 3 / 64; k++) {
                const uint32_t i = lane + k * 64;
                v[k] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (i < nw * 3) v[k] = slots[(size_t)(base + wl[w0 + wb + i / 3]) * 3 + i % 3];
            }
#pragma unroll
            for (int k = 0; k < WCH * 3 / 64; k++) stage[wb * 3 + lane + k * 64] = v[k];
            __syncthreads();
            const float* sf = reinterpret_cast<const float*>(stage);
#pragma unroll
            for (int k = 0; k < 3; k++) {
                if (tg[k] < 0) continue;
                const uint32_t e0 = max(sCi0[tg[k]], w0), e1 = min(sCi1[tg[k]], w0 + nall);
                if (e1 > e0) gsr_sum_entries(tacc[k], sf + (e0 - w0) * 12 + tv[k], e1 - e0);
            }
            __syncthreads();
        }
        __syncthreads();
    }
    // task sums -> the 11 sums of Gaussian `lane` (through LDS, aliased onto the staging buffer, which is free now)
    double* accG = reinterpret_cast<double*>(stage);
    static_assert(sizeof(double) * BS * 11 <= sizeof(float4) * NW * WCH * 3, "accG must fit into the staging buffer");
#pragma unroll
    for (int k = 0; k < 3; k++)
        if (tg[k] >= 0) accG[k * NT + t] = tacc[k];
    __syncthreads();
    if (wave == 0 && live) {
        double acc[11];
#pragma unroll
        for (int v = 0; v < 11; v++) acc[v] = accG[lane * 11 + v];
        gsr_gauss_finish(A, idx, vis, acc, m, sc, q);
    }
    if (t == 0) s_next = gridDim.x + atomicAdd(&A.heavy[1], 1u);
    __syncthreads();  // also: the next group's staging must not overtake wave 0's reads of accG
    it = s_next;
    __syncthreads();
    }
}

hipError_t gsr_launch_gauss_backward(int P, int D, int M, const GsrCam& cam, const float* means3D, const int32_t* radii,
                                     const float* shs, const float* scales, const float* rotations,
                                     const float* cov3D_precomp, const GsrGeom& geom, const float* slots,
                                     uint8_t* slot_written, uint32_t* heavy_groups, uint32_t* heavy_seen_mapped, bool heavy_expected,
                                     int num_slots, float* dL_dmeans2D, float* dL_dcolors,
                                     float* dL_dopacity, float* dL_dfeatures, float* dL_dmeans3D, float* dL_dcov3D, float* dL_dsh,
                                     float* dL_dscales, float* dL_drotations, hipStream_t stream)
{
    if (P <= 0) return hipSuccess;
    GsrGaussArgs A;
    A.P = P; A.D = D; A.M = M; A.cam = cam;
    A.means3D = means3D; A.radii = radii; A.shs = shs; A.clamped = geom.clamped; A.scales = scales; A.rotations = rotations;
    A.cov3D_precomp = cov3D_precomp; A.offsets = geom.offsets; A.tiles = geom.tiles; A.num_slots = num_slots;
    A.slots = reinterpret_cast<const float4*>(slots); A.slot_written = slot_written; A.heavy = heavy_groups;
    A.dL_dmeans2D = dL_dmeans2D; A.dL_dcolors = dL_dcolors; A.dL_dopacity = dL_dopacity; A.dL_dfeatures = dL_dfeatures;
    A.dL_dmeans3D = dL_dmeans3D; A.dL_dcov3D = dL_dcov3D; A.dL_dsh = dL_dsh; A.dL_dscales = dL_dscales; A.dL_drotations = dL_drotations;
    const dim3 grid((P + 63) / 64);
    // every group of 64 Gaussians is done by exactly one kernel: the first one does the light groups and lists the heavy
    // ones, a small grid of 256-thread workgroups then walks that list.  On a scene without large splats that second launch
    // found nothing to do and still cost 4.5 us, and with a hundred heavy groups it costs more (~22 us, a chain of dependent
    // round trips however few the groups) than the first kernel loses by doing them itself (config 4: 112 groups, 24.5 us against
    // +11 us), so it is only made when the caller's PREVIOUS backwards met enough of them (heavy_expected, api.hip: from the
    // host-mapped word both kernels report to); otherwise the first kernel does whatever heavy groups turn up itself -- same
    // bits either way.
    A.heavy_seen = heavy_seen_mapped; A.inline_heavy = heavy_expected ? 0 : 1;
    hipLaunchKernelGGL(gsr_gauss_bwd_kernel, grid, dim3(64), 0, stream, A);
    if (heavy_expected)
        hipLaunchKernelGGL(gsr_gauss_bwd_heavy_kernel, dim3(min((P + 63) / 64, 768)), dim3(256), 0, stream, A);  // 3 workgroups per CU are resident
    return hipGetLastError();
}
