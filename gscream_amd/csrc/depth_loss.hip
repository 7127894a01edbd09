// depth_loss.hip -- the depth terms of GScream's image-space loss (second half), value + gradient.
//
// Replaces, as composed in train.py:548-573,
//   utils/loss_utils.py:77-104  compute_scale_and_shift  (masked least squares  min sum m (s d + t - y)^2, closed form)
//   train.py:552,567            scale = |scale|;  aligned = scale * depth + shift
//   utils/loss_utils.py:26-30   l1_loss / l1_loss_masked on the aligned depth
//   utils/loss_utils.py:58-74   gradient_loss at four scales (aligned[:, ::2^k, ::2^k], k = 0..3), batch-based reduction
// i.e.   L = lambda_l1 * mean(|a - y| * w) + sum_k 0.5 * lambda_s * (sum |dx (g (a-y))| g g' + sum |dy ...|) / sum(g_k)
// with a = |s0| d + t,  (s0, t) the least-squares fit over the mask m,  w / g optional weight / gradient masks (1 if
// absent).  The gradient flows through the fit as well, as it does through the reference's autograd graph:
//   dL/dd = s G + m (A + B d + C y),   G = dL/da per pixel,  A, B, C scalars of (sum G, sum G d, the normal equations).
// The torch path is ~60 small kernels (slicing, masked products, four pyramids, two reductions each) + their autograd
// mirror; here: one pass of sums, one stencil pass producing G (the fit in its prologue), one one-block finisher, one pass for the gradient.
// All sums are per-workgroup partials in double finished in a fixed order: bit-reproducible.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gsr_common.h"

#define GDL_THREADS 256
#define GDL_SCALES 4

struct GdlParams {  // device-resident scalars shared by the passes
    double a00, a01, a11, b0, b1, det, s0, t;
    double Ms[GDL_SCALES];       // sum of the gradient mask on each lattice
    float s, sgn, shift, pad;    // s = |s0|, sgn = d|s0|/ds0
    float cs[GDL_SCALES];        // 0.5 lambda_s / Ms (0 when Ms == 0: reduction_batch_based returns 0)
    float cl1;                   // lambda_l1 / (H W)
    float A, B, C;               // fit-chain coefficients of the gradient
    float loss, l1_mean, smooth, pad2;
};

// Workgroup sums of NV values with ONE barrier: lanes by xor-shuffle, then the four waves in order; the totals end up in tot[NV]
// (LDS), valid for every thread after the call.  (The first version reduced one value at a time, two barriers each: the sums kernel
// spent its time in 18 barriers, the one-block finishers in 9 dependent rounds of strided loads.)
template <int NV>
__device__ __forceinline__ void gdl_block_sums(double (&v)[NV], double* red /*[NV * 4]*/, double* tot /*[NV]*/)
{
#pragma unroll
    for (int i = 0; i < NV; i++) {
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) v[i] += __shfl_xor(v[i], d, 64);
        if ((threadIdx.x & 63) == 0) red[i * 4 + (threadIdx.x >> 6)] = v[i];
    }
    __syncthreads();
    if (threadIdx.x < NV) tot[threadIdx.x] = ((red[threadIdx.x * 4] + red[threadIdx.x * 4 + 1]) + red[threadIdx.x * 4 + 2]) + red[threadIdx.x * 4 + 3];
    __syncthreads();
}

// Totals of the per-workgroup partials part[nparts][NV] (one block): every thread takes whole rows (NV independent loads each,
// rows t, t + 256, ... in order), then one workgroup sum.
template <int NV>
__device__ __forceinline__ void gdl_total_partials(int nparts, const double* __restrict__ part, double* red, double* tot)
{
    double v[NV];
#pragma unroll
    for (int i = 0; i < NV; i++) v[i] = 0.0;
    for (int b = threadIdx.x; b < nparts; b += GDL_THREADS) {
        double r[NV];
#pragma unroll
        for (int i = 0; i < NV; i++) r[i] = part[(size_t)b * NV + i];
#pragma unroll
        for (int i = 0; i < NV; i++) v[i] += r[i];
    }
    gdl_block_sums<NV>(v, red, tot);
}

// pass 1: normal-equation sums over the fit mask, and the gradient-mask sums of the four lattices
__global__ void __launch_bounds__(GDL_THREADS) gdl_sums_kernel(int H, int W, const float* __restrict__ depth,
                                                               const float* __restrict__ target,
                                                               const float* __restrict__ lsq_mask,
                                                               const float* __restrict__ grad_mask, double* __restrict__ part)
{
    __shared__ double red[(5 + GDL_SCALES) * 4], tot[5 + GDL_SCALES];
    double acc[5 + GDL_SCALES] = { 0, 0, 0, 0, 0, 0, 0, 0, 0 };
    const int N = H * W;
    for (int p = blockIdx.x * GDL_THREADS + threadIdx.x; p < N; p += gridDim.x * GDL_THREADS) {
        const float d = depth[p], y = target[p], m = lsq_mask ? lsq_mask[p] : 1.0f;
        acc[0] += (double)(m * d * d); acc[1] += (double)(m * d); acc[2] += (double)m;
        acc[3] += (double)(m * d * y); acc[4] += (double)(m * y);
        const int x = p % W, yy = p / W;
        const float g = grad_mask ? grad_mask[p] : 1.0f;
#pragma unroll
        for (int k = 0; k < GDL_SCALES; k++)
            if (((x | yy) & ((1 << k) - 1)) == 0) acc[5 + k] += (double)g;
    }
    gdl_block_sums<5 + GDL_SCALES>(acc, red, tot);
    if (threadIdx.x < 5 + GDL_SCALES) part[(size_t)blockIdx.x * (5 + GDL_SCALES) + threadIdx.x] = tot[threadIdx.x];
}

// the least-squares fit and the constants of the passes from the totals of pass 1 (one thread)
__device__ __forceinline__ GdlParams gdl_fit(const double* tot /*[5 + GDL_SCALES]*/, int H, int W, float lambda_l1, float lambda_smooth)
{
    GdlParams q;
    q.a00 = tot[0]; q.a01 = tot[1]; q.a11 = tot[2]; q.b0 = tot[3]; q.b1 = tot[4];
    q.det = q.a00 * q.a11 - q.a01 * q.a01;                       // loss_utils.py:97
    q.s0 = q.det != 0.0 ? (q.a11 * q.b0 - q.a01 * q.b1) / q.det : 0.0;   // :100
    q.t = q.det != 0.0 ? (-q.a01 * q.b0 + q.a00 * q.b1) / q.det : 0.0;   // :101
    q.s = (float)fabs(q.s0);                                      // train.py:552
    q.sgn = q.s0 > 0.0 ? 1.f : (q.s0 < 0.0 ? -1.f : 0.f);
    q.shift = (float)q.t;
    for (int k = 0; k < GDL_SCALES; k++) {
        q.Ms[k] = tot[5 + k];
        q.cs[k] = q.Ms[k] != 0.0 ? (float)(0.5 * (double)lambda_smooth / q.Ms[k]) : 0.f;  // :40-49
    }
    q.cl1 = (float)((double)lambda_l1 / ((double)H * (double)W));
    q.A = q.B = q.C = q.loss = q.l1_mean = q.smooth = q.pad = q.pad2 = 0.f;
    return q;
}

__device__ __forceinline__ float gdl_sign(float v) { return v > 0.f ? 1.f : (v < 0.f ? -1.f : 0.f); }

// pass 2: G = dL/d(aligned) per pixel (L1 term + the four gradient-loss stencils) and the sums the finisher needs
__global__ void __launch_bounds__(GDL_THREADS) gdl_stencil_kernel(int H, int W, const float* __restrict__ depth,
                                                                  const float* __restrict__ target,
                                                                  const float* __restrict__ l1_weight,
                                                                  const float* __restrict__ grad_mask,
                                                                  int nparts1, const double* __restrict__ part1, float lambda_l1,
                                                                  float lambda_smooth, GdlParams* __restrict__ P, float* __restrict__ G,
                                                                  double* __restrict__ part)
{
    __shared__ double red[(5 + GDL_SCALES) * 4], tot[5 + GDL_SCALES];
    __shared__ GdlParams sq;
    // The fit is folded in: EVERY workgroup totals the partials of pass 1 (nparts1 x 9 doubles, L2 hits; the same fixed order
    // everywhere, so all of them get the same bits) and solves the 2 x 2 system -- a launch of its own for that cost 4.6 us between two
    // kernels of 6 and 17.  Workgroup 0 leaves the parameters for the finisher and the backward.
    gdl_total_partials<5 + GDL_SCALES>(nparts1, part1, red, tot);
    if (threadIdx.x == 0) {
        sq = gdl_fit(tot, H, W, lambda_l1, lambda_smooth);
        if (blockIdx.x == 0) *P = sq;
    }
    __syncthreads();
    const float s = sq.s, t = sq.shift, cl1 = sq.cl1;
    float cs[GDL_SCALES];
#pragma unroll
    for (int k = 0; k < GDL_SCALES; k++) cs[k] = sq.cs[k];
    __syncthreads();  // (red / tot are used again for this pass's own sums)
    double acc[3 + GDL_SCALES] = { 0, 0, 0, 0, 0, 0, 0 };  // l1 sum, sum G, sum G d, edge sums per scale
    auto diff = [&](int x, int y) {  // g (a - y) at a pixel (gradient_loss :62-63)
        const int q = y * W + x;
        const float gm = grad_mask ? grad_mask[q] : 1.0f;
        return gm * ((s * depth[q] + t) - target[q]);
    };
    auto gm_at = [&](int x, int y) { return grad_mask ? grad_mask[y * W + x] : 1.0f; };
    // (a 2-D walk -- 256-pixel row segments per workgroup, no division per pixel -- was measured: 21.5 instead of 17.5 us)
    const int N = H * W;
    for (int p = blockIdx.x * GDL_THREADS + threadIdx.x; p < N; p += gridDim.x * GDL_THREADS) {
        const int x = p % W, y = p / W;
        const float d = depth[p], r = (s * d + t) - target[p];
        const float w = l1_weight ? l1_weight[p] : 1.0f;
        float g = cl1 * gdl_sign(r) * w;                       // d/da of lambda_l1 mean(|a - y| w)
        acc[0] += (double)(fabsf(r) * w);
        const float gm = gm_at(x, y), here = gm * r;
#pragma unroll
        for (int k = 0; k < GDL_SCALES; k++) {
            const int step = 1 << k;
            if (((x | y) & (step - 1)) != 0) continue;         // not on the ::step lattice
            float dsum = 0.f;                                  // d(sum of |edge| terms)/d(diff at p)
            if (x + step < W) {                                // edge to the right neighbour, counted here (:65-67)
                const float m2 = gm * gm_at(x + step, y), e = diff(x + step, y) - here;
                acc[3 + k] += (double)(fabsf(e) * m2);
                dsum -= gdl_sign(e) * m2;
            }
            if (x - step >= 0) {
                const float m2 = gm * gm_at(x - step, y), e = here - diff(x - step, y);
                dsum += gdl_sign(e) * m2;
            }
            if (y + step < H) {                                // :69-71
                const float m2 = gm * gm_at(x, y + step), e = diff(x, y + step) - here;
                acc[3 + k] += (double)(fabsf(e) * m2);
                dsum -= gdl_sign(e) * m2;
            }
            if (y - step >= 0) {
                const float m2 = gm * gm_at(x, y - step), e = here - diff(x, y - step);
                dsum += gdl_sign(e) * m2;
            }
            g += cs[k] * gm * dsum;
        }
        G[p] = g;
        acc[1] += (double)g;
        acc[2] += (double)g * (double)d;
    }
    gdl_block_sums<3 + GDL_SCALES>(acc, red, tot);
    if (threadIdx.x < 3 + GDL_SCALES) part[(size_t)blockIdx.x * (3 + GDL_SCALES) + threadIdx.x] = tot[threadIdx.x];
}

__global__ void __launch_bounds__(GDL_THREADS) gdl_finish_kernel(int nparts, int H, int W, const double* __restrict__ part,
                                                                 float lambda_l1, GdlParams* __restrict__ P, float* __restrict__ out)
{
    __shared__ double red[(3 + GDL_SCALES) * 4], tot[3 + GDL_SCALES];
    gdl_total_partials<3 + GDL_SCALES>(nparts, part, red, tot);
    if (threadIdx.x == 0) {
        GdlParams q = *P;
        const double l1_mean = tot[0] / ((double)H * (double)W);
        double smooth = 0.0;
        for (int k = 0; k < GDL_SCALES; k++) smooth += (double)q.cs[k] * tot[3 + k];  // = sum_k 0.5 lambda_s edge_k / M_k
        const double SG = tot[1], SGd = tot[2];
        if (q.det != 0.0) {  // chain through the least-squares fit (zero when the fit degenerated to s = t = 0)
            const double sg = (double)q.sgn;
            q.A = (float)((SGd * sg * (-q.b1 + 2.0 * q.s0 * q.a01) + SG * (-q.b0 + 2.0 * q.t * q.a01)) / q.det);
            q.B = (float)((SGd * sg * (-2.0 * q.s0 * q.a11) + SG * (2.0 * q.b1 - 2.0 * q.t * q.a11)) / q.det);
            q.C = (float)((SGd * sg * q.a11 + SG * (-q.a01)) / q.det);
        }
        q.l1_mean = (float)l1_mean;
        q.smooth = (float)smooth;
        q.loss = (float)((double)lambda_l1 * l1_mean + smooth);
        *P = q;
        out[0] = q.loss; out[1] = q.l1_mean; out[2] = q.smooth; out[3] = q.s; out[4] = q.shift;
    }
}

// pass 3: dL/d(depth)
__global__ void __launch_bounds__(GDL_THREADS) gdl_backward_kernel(int N, const float* __restrict__ depth,
                                                                   const float* __restrict__ target,
                                                                   const float* __restrict__ lsq_mask,
                                                                   const GdlParams* __restrict__ P, const float* __restrict__ G,
                                                                   const float* __restrict__ upstream, float* __restrict__ dL_dd)
{
    const int p = blockIdx.x * GDL_THREADS + threadIdx.x;
    if (p >= N) return;
    const float up = upstream ? upstream[0] : 1.0f;
    const float m = lsq_mask ? lsq_mask[p] : 1.0f;
    dL_dd[p] = up * (P->s * G[p] + m * (P->A + P->B * depth[p] + P->C * target[p]));
}

// ---- host side -------------------------------------------------------------------------------------------------
#ifndef GDL_BLOCKS
#define GDL_BLOCKS 512
#endif
struct GdlWorkspace {
    float* G;
    double *part1, *part2;
    GdlParams* params;
    size_t bytes;
};
static GdlWorkspace gdl_carve(void* base, int H, int W)
{
    GdlWorkspace w;
    char* b = (char*)base;
    size_t off = 0;
    const size_t n = (size_t)(H > 0 ? H : 1) * (W > 0 ? W : 1);
    w.G = (float*)(b + off); off += gsr_align(n * 4);
    w.part1 = (double*)(b + off); off += gsr_align((size_t)GDL_BLOCKS * (5 + GDL_SCALES) * 8);
    w.part2 = (double*)(b + off); off += gsr_align((size_t)GDL_BLOCKS * (3 + GDL_SCALES) * 8);
    w.params = (GdlParams*)(b + off); off += gsr_align(sizeof(GdlParams));
    w.bytes = off;
    return w;
}
size_t gdl_workspace_bytes(int H, int W) { return gdl_carve(nullptr, H, W).bytes; }

hipError_t gdl_launch_forward(int H, int W, const float* depth, const float* target, const float* lsq_mask,
                              const float* l1_weight, const float* grad_mask, float lambda_l1, float lambda_smooth,
                              void* workspace, float* out5, hipStream_t stream)
{
    const GdlWorkspace w = gdl_carve(workspace, H, W);
    const int N = H * W;
    int nb = (N + GDL_THREADS - 1) / GDL_THREADS;
    if (nb > GDL_BLOCKS) nb = GDL_BLOCKS;  // (1024 / 2304 workgroups measured: no change / slower)
    hipLaunchKernelGGL(gdl_sums_kernel, dim3(nb), dim3(GDL_THREADS), 0, stream, H, W, depth, target, lsq_mask, grad_mask, w.part1);
    hipLaunchKernelGGL(gdl_stencil_kernel, dim3(nb), dim3(GDL_THREADS), 0, stream, H, W, depth, target, l1_weight, grad_mask,
                       nb, w.part1, lambda_l1, lambda_smooth, w.params, w.G, w.part2);
    hipLaunchKernelGGL(gdl_finish_kernel, dim3(1), dim3(GDL_THREADS), 0, stream, nb, H, W, w.part2, lambda_l1, w.params, out5);
    return hipGetLastError();
}

hipError_t gdl_launch_backward(int H, int W, const float* depth, const float* target, const float* lsq_mask,
                               const void* workspace, const float* upstream, float* dL_ddepth, hipStream_t stream)
{
    const GdlWorkspace w = gdl_carve(const_cast<void*>(workspace), H, W);
    const int N = H * W;
    hipLaunchKernelGGL(gdl_backward_kernel, dim3((N + GDL_THREADS - 1) / GDL_THREADS), dim3(GDL_THREADS), 0, stream, N, depth,
                       target, lsq_mask, w.params, w.G, upstream, dL_ddepth);
    return hipGetLastError();
}
