// blend.hip -- tile-based forward / backward alpha blend of RGB + depth + one feature channel.
//
// Replaces DGR forward.cu:441-568 renderCUDA (fwd) and backward.cu:409-604 renderCUDA (bwd).
//
// Mapping onto CDNA4.  One workgroup per 16x16 screen tile.  PPT = pixels per thread (1, 2 or 4):
// the workgroup has 256/PPT threads = 4/2/1 wavefronts of 64; lane l of wave w owns column
// (l & 15) and rows  4w + (l >> 4) + k * (16/PPT),  k < PPT.  So one DPP "row" of 16 lanes is one
// pixel row of the tile, and a wavefront owns whole pixel rows.
//   * Tile lists are consumed in batches of 256 instances.  Each thread gathers the 64-byte record
//     of its instance(s) with three (fwd) / four (bwd) 16-byte loads and parks it in LDS as three
//     float4 SoA arrays; the inner loop reads them back as same-address (broadcast) ds_read_b128,
//     conflict-free.  The reference re-reads colour and depth from global memory for every
//     (pixel, Gaussian) contribution (forward.cu:545-546).
//   * Early-out is wave-granular: a wave leaves the batch loop when all of its pixels are done
//     (__all), and in the backward a wave skips gradient math + reduction for an instance none of
//     its pixels sees (__any); the block leaves when every wave is done (__syncthreads_and).
//   * Backward gradient scatter: the reference issues 11 global atomicAdd per (pixel, Gaussian)
//     contribution (backward.cu:554-601).  Here the 11 partials are summed over the PPT pixels of a
//     lane in registers, over the 16 lanes of a row with 4 DPP adds, and the 4 row leaders of a wave
//     add into an LDS accumulator [256][12] (ds_add_f32).  After the batch every thread stores the
//     12 floats of its instance with three plain 16-byte stores into that instance's private
//     gradient slot (slot = Gaussian's scan offset + tile position inside its rectangle).  The
//     per-Gaussian kernel (gauss_bwd.hip) then sums each Gaussian's contiguous slots in a fixed
//     order: no global atomics at all, and gradients are bit-reproducible.
//   * XCD awareness: workgroup b runs on XCD b % 8 (observed dispatch rule); the block->tile map
//     hands each XCD a contiguous band of tile rows so neighbouring tiles, which share most of their
//     Gaussians, hit the same 4 MiB L2.  Pure speed: any placement gives the same result.
#include "gsr_common.h"

#define GSR_BATCH 256

// Fast-math knobs of the blend inner loops.  Default: exp via v_exp_f32 (exp2(x*log2e), ~2 ulp) and
// T/(1-alpha) via v_rcp_f32 (1 ulp).  -DGSR_PRECISE_MATH selects libm-accurate expf and IEEE division
// (diagnostic build, used to attribute parity differences; not shipped).
#ifdef GSR_PRECISE_MATH
#define GSR_EXP(x) expf(x)
#define GSR_RCP(x) (1.0f / (x))
#else
#define GSR_EXP(x) __expf(x)
#define GSR_RCP(x) __builtin_amdgcn_rcpf(x)
#endif

__device__ __forceinline__ int gsr_tile_of_block(int b, int T)
{
    const int xcd = b & 7, i = b >> 3, q = T >> 3, r = T & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + i;
}

template <int CTRL>
__device__ __forceinline__ float gsr_dpp_add(float v)
{
    const int o = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true);
    return v + __int_as_float(o);
}
// Sum over the 16 lanes of a DPP row; every lane of the row ends up with the row total.
__device__ __forceinline__ float gsr_row_sum16(float v)
{
    v = gsr_dpp_add<0xB1>(v);   // quad_perm:[1,0,3,2]
    v = gsr_dpp_add<0x4E>(v);   // quad_perm:[2,3,0,1]
    v = gsr_dpp_add<0x141>(v);  // row_half_mirror
    v = gsr_dpp_add<0x140>(v);  // row_mirror
    return v;
}

// ---------------------------------------------------------------------------------------------
// Forward
// ---------------------------------------------------------------------------------------------
template <int PPT>
__global__ void __launch_bounds__(256 / PPT) gsr_blend_fwd_kernel(
    const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list, const GsrRec* __restrict__ rec, int W,
    int H, int gx, int T, const float* __restrict__ bg, float* __restrict__ out_color, float* __restrict__ out_depth,
    float* __restrict__ out_feature, float* __restrict__ final_T, uint32_t* __restrict__ n_contrib)
{
    constexpr int NT = 256 / PPT, ROWS = 16 / PPT;
    __shared__ float4 sA[GSR_BATCH], sB[GSR_BATCH], sC[GSR_BATCH];

    const int tile = gsr_tile_of_block(blockIdx.x, T);
    const int tx = tile % gx, ty = tile / gx;
    const int t = threadIdx.x;
    const int px = tx * 16 + (t & 15);
    const float pxf = (float)px;
    const uint2 rg = ranges[tile];
    const int n = (int)(rg.y - rg.x);

    int py[PPT];
    float pyf[PPT], Tr[PPT], C0[PPT], C1[PPT], C2[PPT], Dp[PPT], Uf[PPT];
    uint32_t last[PPT];
    bool done[PPT], inside[PPT];
#pragma unroll
    for (int k = 0; k < PPT; k++) {
        py[k] = ty * 16 + (t >> 4) + k * ROWS;
        pyf[k] = (float)py[k];
        inside[k] = px < W && py[k] < H;
        done[k] = !inside[k];
        Tr[k] = 1.0f; C0[k] = C1[k] = C2[k] = Dp[k] = Uf[k] = 0.f; last[k] = 0;
    }

    for (int base = 0; base < n; base += GSR_BATCH) {
        bool all_done = true;
#pragma unroll
        for (int k = 0; k < PPT; k++) all_done = all_done && done[k];
        if (__syncthreads_and(all_done)) break;  // also fences the previous batch's LDS reads

        for (int i = t; i < GSR_BATCH; i += NT) {
            const int p = base + i;
            if (p < n) {
                const float4* r = reinterpret_cast<const float4*>(rec + point_list[rg.x + p]);
                sA[i] = r[0]; sB[i] = r[1]; sC[i] = r[2];
            }
        }
        __syncthreads();

        const int cnt = min(GSR_BATCH, n - base);
        for (int j = 0; j < cnt; j++) {
            if (__all(all_done)) break;  // wave-uniform
            const float4 A = sA[j], B = sB[j], C = sC[j];
            all_done = true;
#pragma unroll
            for (int k = 0; k < PPT; k++) {
                const float dx = A.x - pxf, dy = A.y - pyf[k];
                const float power = -0.5f * (A.z * dx * dx + B.x * dy * dy) - A.w * dx * dy;
                const float alpha = fminf(0.99f, B.y * GSR_EXP(power));
                bool ok = !done[k] && power <= 0.0f && alpha >= (1.0f / 255.0f);
                const float test_T = Tr[k] * (1.0f - alpha);
                const bool stop = ok && test_T < 0.0001f;
                done[k] = done[k] || stop;
                ok = ok && !stop;
                const float w = ok ? alpha * Tr[k] : 0.0f;
                C0[k] += C.x * w; C1[k] += C.y * w; C2[k] += C.z * w;
                Dp[k] += B.z * w; Uf[k] += B.w * w;
                Tr[k] = ok ? test_T : Tr[k];
                last[k] = ok ? (uint32_t)(base + j + 1) : last[k];
                all_done = all_done && done[k];
            }
        }
    }

    const float bg0 = bg[0], bg1 = bg[1], bg2 = bg[2];
    const size_t HW = (size_t)H * W;
#pragma unroll
    for (int k = 0; k < PPT; k++) {
        if (inside[k]) {
            const size_t pid = (size_t)py[k] * W + px;
            final_T[pid] = Tr[k];
            n_contrib[pid] = last[k];
            out_color[pid] = C0[k] + Tr[k] * bg0;
            out_color[HW + pid] = C1[k] + Tr[k] * bg1;
            out_color[2 * HW + pid] = C2[k] + Tr[k] * bg2;
            out_depth[pid] = Dp[k];
            out_feature[pid] = Uf[k];
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Backward
// ---------------------------------------------------------------------------------------------
template <int PPT>
__global__ void __launch_bounds__(256 / PPT) gsr_blend_bwd_kernel(
    const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list, const GsrRec* __restrict__ rec, int W,
    int H, int gx, int T, const float* __restrict__ bg, const float* __restrict__ final_T,
    const uint32_t* __restrict__ n_contrib, const float* __restrict__ dL_dcolor, const float* __restrict__ dL_ddepth,
    const float* __restrict__ dL_dfeature, float4* __restrict__ slots)
{
    constexpr int NT = 256 / PPT, ROWS = 16 / PPT;
    __shared__ float4 sA[GSR_BATCH], sB[GSR_BATCH], sC[GSR_BATCH];
    __shared__ __attribute__((aligned(16))) float acc[GSR_BATCH * GSR_SLOT_FLOATS];
    __shared__ uint32_t sSlot[GSR_BATCH];
    __shared__ int sMax;

    const int tile = gsr_tile_of_block(blockIdx.x, T);
    const int tx = tile % gx, ty = tile / gx;
    const int t = threadIdx.x;
    const uint2 rg = ranges[tile];
    const int n = (int)(rg.y - rg.x);
    if (n == 0) return;

    const int px = tx * 16 + (t & 15);
    const float pxf = (float)px;
    const float ddelx_dx = 0.5f * W, ddely_dy = 0.5f * H;
    const float bg0 = bg[0], bg1 = bg[1], bg2 = bg[2];
    const size_t HW = (size_t)H * W;

    float pyf[PPT], Tf[PPT], Tr[PPT], g0[PPT], g1[PPT], g2[PPT], gd[PPT], gu[PPT], bgdot[PPT];
    float ar0[PPT], ar1[PPT], ar2[PPT], ard[PPT], aru[PPT], la[PPT], lc0[PPT], lc1[PPT], lc2[PPT], lcd[PPT], lcu[PPT];
    int lastc[PPT];
    int mymax = 0;
#pragma unroll
    for (int k = 0; k < PPT; k++) {
        const int py = ty * 16 + (t >> 4) + k * ROWS;
        pyf[k] = (float)py;
        const bool inside = px < W && py < H;
        const size_t pid = inside ? (size_t)py * W + px : 0;
        Tf[k] = inside ? final_T[pid] : 0.f;
        Tr[k] = Tf[k];
        lastc[k] = inside ? (int)n_contrib[pid] : 0;
        g0[k] = inside ? dL_dcolor[pid] : 0.f;
        g1[k] = inside ? dL_dcolor[HW + pid] : 0.f;
        g2[k] = inside ? dL_dcolor[2 * HW + pid] : 0.f;
        gd[k] = inside ? dL_ddepth[pid] : 0.f;
        gu[k] = inside ? dL_dfeature[pid] : 0.f;
        bgdot[k] = bg0 * g0[k] + bg1 * g1[k] + bg2 * g2[k];
        ar0[k] = ar1[k] = ar2[k] = ard[k] = aru[k] = 0.f;
        la[k] = lc0[k] = lc1[k] = lc2[k] = lcd[k] = lcu[k] = 0.f;
        mymax = max(mymax, lastc[k]);
    }
    if (t == 0) sMax = 0;
    for (int i = t; i < GSR_BATCH * GSR_SLOT_FLOATS; i += NT) acc[i] = 0.f;
    __syncthreads();
    atomicMax(&sMax, mymax);
    __syncthreads();
    const int nproc = min(n, sMax);  // instances at positions >= nproc contributed to no pixel of this tile

    // back to front, in batches of 256 instances; local j = 0 is the backmost instance of the batch
    for (int hi = n; hi > 0; hi -= GSR_BATCH) {
        const int lo = max(0, hi - GSR_BATCH), cnt = hi - lo;
        const bool active = lo < nproc;
        for (int i = t; i < cnt; i += NT) {
            const GsrRec* r = rec + point_list[rg.x + (hi - 1 - i)];
            const uint4 d = r->d;
            const int x0 = d.y & 0xffff, x1 = d.y >> 16, y0 = d.z & 0xffff;
            sSlot[i] = d.x + (uint32_t)((ty - y0) * (x1 - x0) + (tx - x0));
            if (active) { sA[i] = r->a; sB[i] = r->b; sC[i] = r->c; }
        }
        __syncthreads();

        if (active) {
            for (int j = 0; j < cnt; j++) {
                const int p = hi - 1 - j;
                if (p >= nproc) continue;  // block-uniform
                const float4 A = sA[j], B = sB[j], C = sC[j];
                float dx[PPT], dy[PPT], G[PPT], alpha[PPT];
                bool ok[PPT], any_ok = false;
#pragma unroll
                for (int k = 0; k < PPT; k++) {
                    dx[k] = A.x - pxf; dy[k] = A.y - pyf[k];
                    const float power = -0.5f * (A.z * dx[k] * dx[k] + B.x * dy[k] * dy[k]) - A.w * dx[k] * dy[k];
                    G[k] = GSR_EXP(power);
                    alpha[k] = fminf(0.99f, B.y * G[k]);
                    ok[k] = p < lastc[k] && power <= 0.0f && alpha[k] >= (1.0f / 255.0f);
                    any_ok = any_ok || ok[k];
                }
                if (!__any(any_ok)) continue;  // wave-uniform: no pixel of this wave sees the instance

                float s[11];
#pragma unroll
                for (int v = 0; v < 11; v++) s[v] = 0.f;
#pragma unroll
                for (int k = 0; k < PPT; k++) {
                    if (ok[k]) {  // divergent: executed under the EXEC mask of the lanes that blend
                        const float rinv = GSR_RCP(1.0f - alpha[k]);
                        const float Tn = Tr[k] * rinv;  // T / (1 - alpha)
                        const float w = alpha[k] * Tn;
                        const float oml = 1.0f - la[k];
                        ar0[k] = la[k] * lc0[k] + oml * ar0[k]; ar1[k] = la[k] * lc1[k] + oml * ar1[k];
                        ar2[k] = la[k] * lc2[k] + oml * ar2[k]; ard[k] = la[k] * lcd[k] + oml * ard[k];
                        aru[k] = la[k] * lcu[k] + oml * aru[k];
                        float dL_dalpha = (C.x - ar0[k]) * g0[k] + (C.y - ar1[k]) * g1[k] + (C.z - ar2[k]) * g2[k]
                                        + (B.z - ard[k]) * gd[k] + (B.w - aru[k]) * gu[k];
                        dL_dalpha *= Tn;
                        dL_dalpha += (-Tf[k] * rinv) * bgdot[k];
                        const float dL_dG = B.y * dL_dalpha;
                        const float gdx = G[k] * dx[k], gdy = G[k] * dy[k];
                        const float dG_ddelx = -gdx * A.z - gdy * A.w;
                        const float dG_ddely = -gdy * B.x - gdx * A.w;
                        s[0] += w * g0[k]; s[1] += w * g1[k]; s[2] += w * g2[k];
                        s[3] += w * gd[k]; s[4] += w * gu[k];
                        s[5] += dL_dG * dG_ddelx * ddelx_dx; s[6] += dL_dG * dG_ddely * ddely_dy;
                        s[7] += -0.5f * gdx * dx[k] * dL_dG; s[8] += -0.5f * gdx * dy[k] * dL_dG;
                        s[9] += -0.5f * gdy * dy[k] * dL_dG;
                        s[10] += G[k] * dL_dalpha;
                        Tr[k] = Tn; la[k] = alpha[k];
                        lc0[k] = C.x; lc1[k] = C.y; lc2[k] = C.z; lcd[k] = B.z; lcu[k] = B.w;
                    }
                }
#pragma unroll
                for (int v = 0; v < 11; v++) s[v] = gsr_row_sum16(s[v]);
                if ((t & 15) == 0) {
                    float* a = acc + j * GSR_SLOT_FLOATS;
#pragma unroll
                    for (int v = 0; v < 11; v++) atomicAdd(a + v, s[v]);
                }
            }
        }
        __syncthreads();
        for (int i = t; i < cnt; i += NT) {
            float4* a4 = reinterpret_cast<float4*>(acc + i * GSR_SLOT_FLOATS);
            float4* dst = slots + (size_t)sSlot[i] * 3;
            dst[0] = a4[0]; dst[1] = a4[1]; dst[2] = a4[2];
            if (active) { a4[0] = a4[1] = a4[2] = make_float4(0.f, 0.f, 0.f, 0.f); }
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
hipError_t gsr_launch_blend_forward(int W, int H, int gx, int T, const float* bg, const GsrGeom& geom,
                                    const GsrImage& image, const GsrBinning& bin, float* out_color, float* out_depth,
                                    float* out_feature, int ppt, hipStream_t stream)
{
    if (T <= 0) return hipSuccess;
#define GSR_FWD(PPT_)                                                                                               \
    hipLaunchKernelGGL(gsr_blend_fwd_kernel<PPT_>, dim3(T), dim3(256 / PPT_), 0, stream, image.ranges, bin.point_list, \
                       geom.rec, W, H, gx, T, bg, out_color, out_depth, out_feature, image.final_T, image.n_contrib)
    if (ppt == 1) GSR_FWD(1);
    else if (ppt == 4) GSR_FWD(4);
    else GSR_FWD(2);
#undef GSR_FWD
    return hipGetLastError();
}

hipError_t gsr_launch_blend_backward(int W, int H, int gx, int T, const float* bg, const GsrGeom& geom,
                                     const GsrImage& image, const GsrBinning& bin, const float* dL_dcolor,
                                     const float* dL_ddepth, const float* dL_dfeature, float* slots, int ppt,
                                     hipStream_t stream)
{
    if (T <= 0) return hipSuccess;
#define GSR_BWD(PPT_)                                                                                               \
    hipLaunchKernelGGL(gsr_blend_bwd_kernel<PPT_>, dim3(T), dim3(256 / PPT_), 0, stream, image.ranges, bin.point_list, \
                       geom.rec, W, H, gx, T, bg, image.final_T, image.n_contrib, dL_dcolor, dL_ddepth, dL_dfeature,  \
                       reinterpret_cast<float4*>(slots))
    if (ppt == 1) GSR_BWD(1);
    else if (ppt == 4) GSR_BWD(4);
    else GSR_BWD(2);
#undef GSR_BWD
    return hipGetLastError();
}
