// loss.hip -- fused image-space RGB loss: weighted L1 + weighted SSIM (11x11 Gaussian window), value and gradient.
//
// the step right after the rasterizer.  Replaces, in GScream,
//   utils/loss_utils.py:26-30   l1_loss, l1_loss_masked
//   utils/loss_utils.py:113-121 gaussian / create_window (sigma 1.5, outer product of the normalised 1-D window)
//   utils/loss_utils.py:131-190 ssim / _ssim, ssim_masked / _ssim_masked (five zero-padded depthwise conv2d each)
// as composed in train.py:538-545:  loss = w ((1 - lambda) L1 + lambda (1 - SSIM)).
// One generic objective covers every combination the trainer uses:
//   L = a_l1 * mean(|img - gt| * m)  +  a_ssim * mean(ssim_map(img, gt) * m)        m = weight[H,W] (1 if absent),
// means over C*H*W like torch's .mean() of the broadcast product.
//
// Mapping.  One 256-thread workgroup per 32x16 pixel tile and channel.  The (32+10)x(16+10) halo of both images is
// staged in LDS once; the 2-D window is separable, so a horizontal pass produces four row-filtered planes
// (x, y, x^2 + y^2, xy) in LDS and a vertical pass finishes them per pixel: 16 taps-equivalents per quantity instead
// of 121, no intermediate image ever reaches HBM (the torch path writes 5 convolved planes + ~10 elementwise
// temporaries per call).  The forward also emits the three partial-derivative planes the backward needs
// (d/d mu1, d/d E[x^2], d/d E[xy] of the weighted map); the backward filters those with the same two passes and
// contracts them with the images:  dL/dx = G*(d_mu1) + 2x G*(d_E11) + y G*(d_E12)  (+ the L1 sign term).
// Sums are reduced per workgroup and finished by one block in a fixed order: bit-reproducible.
// HBM traffic per pixel-channel: forward 8 B in (+4/C weight) + 12 B out, backward 20 B in + 4 B out.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gsr_common.h"

#define GSL_TW 32
#define GSL_TH 16
#define GSL_R 5                      // window radius (window_size 11)
#define GSL_HW (GSL_TW + 2 * GSL_R)  // 42
#define GSL_HH (GSL_TH + 2 * GSL_R)  // 26

// normalised 1-D Gaussian, sigma = 1.5, 11 taps (loss_utils.py:113-115), rounded to fp32 like torch.Tensor(...).  The
// kernels take their taps as an argument (GslTaps, in the kernel-argument segment: scalar loads like __constant__): a smaller
// odd window is the same 11-tap machinery with zeros at both ends -- identical to conv2d with padding window_size // 2.
struct GslTaps { float g[11]; };
static const float GSL_G11[11] = { 0x1.0d956cp-10f, 0x1.f1fe02p-8f, 0x1.26eb18p-5f, 0x1.bff0fep-4f, 0x1.b43c3ep-3f, 0x1.10656p-2f,
                                 0x1.b43c3ep-3f, 0x1.bff0fep-4f, 0x1.26eb18p-5f, 0x1.f1fe02p-8f, 0x1.0d956cp-10f };

struct GslPartial { double l1, ssim; };

__device__ __forceinline__ float gsl_load(const float* __restrict__ p, int x, int y, int W, int H)
{
    return (x >= 0 && x < W && y >= 0 && y < H) ? p[(size_t)y * W + x] : 0.f;  // conv2d zero padding
}

template <bool STATE>
__global__ void __launch_bounds__(256) gsl_forward_kernel(int H, int W, const float* __restrict__ img,
                                                          const float* __restrict__ gt,
                                                          const float* __restrict__ weight, float* __restrict__ d_mu1,
                                                          float* __restrict__ d_e11, float* __restrict__ d_e12,
                                                          GslPartial* __restrict__ partial, const GslTaps taps)
{
    const float* GSL_G = taps.g;
    __shared__ float sx[GSL_HH][GSL_HW + 1], sy[GSL_HH][GSL_HW + 1];
    __shared__ float hx[4][GSL_HH][GSL_TW];
    __shared__ double red[2][4];
    const int t = threadIdx.x, ch = blockIdx.z;
    const int x0 = blockIdx.x * GSL_TW, y0 = blockIdx.y * GSL_TH;
    const size_t plane = (size_t)H * W;
    const float* ip = img + ch * plane;
    const float* gp = gt + ch * plane;
    {   // halo tile: all of a thread's loads are issued before the first LDS store (one memory round trip, not five)
        constexpr int NL = (GSL_HH * GSL_HW + 255) / 256;
        float vx[NL], vy[NL];
#pragma unroll
        for (int k = 0; k < NL; k++) {
            const int i = t + k * 256, r = i / GSL_HW, c = i - r * GSL_HW;
            const bool in = i < GSL_HH * GSL_HW;
            vx[k] = in ? gsl_load(ip, x0 - GSL_R + c, y0 - GSL_R + r, W, H) : 0.f;
            vy[k] = in ? gsl_load(gp, x0 - GSL_R + c, y0 - GSL_R + r, W, H) : 0.f;
        }
#pragma unroll
        for (int k = 0; k < NL; k++) {
            const int i = t + k * 256, r = i / GSL_HW, c = i - r * GSL_HW;
            if (i < GSL_HH * GSL_HW) { sx[r][c] = vx[k]; sy[r][c] = vy[k]; }
        }
    }
    __syncthreads();
    // horizontal pass: a thread produces four adjacent outputs of one row from a 14-value register window (28 LDS
    // reads instead of 88)
    if (t < GSL_HH * (GSL_TW / 4)) {
        const int r = t / (GSL_TW / 4), c = (t % (GSL_TW / 4)) * 4;
        float xs[14], ys[14];
#pragma unroll
        for (int k = 0; k < 14; k++) { xs[k] = sx[r][c + k]; ys[k] = sy[r][c + k]; }
        // FOUR filtered quantities, not five: the map needs sigma1^2 + sigma2^2 = E[x^2 + y^2] - mu1^2 - mu2^2 and sigma12 = E[xy] - mu1 mu2,
        // never E[x^2] and E[y^2] apart (and d ssim / d E[x^2] = d ssim / d E[x^2 + y^2]: the backward's planes are unchanged)
        float s1[4] = { 0, 0, 0, 0 }, s2[4] = { 0, 0, 0, 0 }, sS[4] = { 0, 0, 0, 0 }, s12[4] = { 0, 0, 0, 0 };
#pragma unroll
        for (int k = 0; k < 14; k++) {
            const float x = xs[k], y = ys[k], ss = x * x + y * y, xy = x * y;
#pragma unroll
            for (int o = 0; o < 4; o++) {
                const int tap = k - o;
                if (tap >= 0 && tap < 11) {
                    const float g = GSL_G[tap];
                    s1[o] += g * x; s2[o] += g * y; sS[o] += g * ss; s12[o] += g * xy;
                }
            }
        }
#pragma unroll
        for (int o = 0; o < 4; o++) {
            hx[0][r][c + o] = s1[o]; hx[1][r][c + o] = s2[o]; hx[2][r][c + o] = sS[o]; hx[3][r][c + o] = s12[o];
        }
    }
    __syncthreads();
    double l1sum = 0.0, ssum = 0.0;
    // vertical pass: a thread finishes two vertically adjacent pixels from a 12-row register window
    const int vc = t % GSL_TW, vr = (t / GSL_TW) * 2;
    float col[4][12];
#pragma unroll
    for (int q = 0; q < 4; q++)
#pragma unroll
        for (int k = 0; k < 12; k++) col[q][k] = hx[q][vr + k][vc];
#pragma unroll
    for (int half = 0; half < 2; half++) {
        const int r = vr + half, c = vc;
        const int px = x0 + c, py = y0 + r;
        float mu1 = 0.f, mu2 = 0.f, eS = 0.f, e12 = 0.f;
#pragma unroll
        for (int k = 0; k < 11; k++) {
            const float g = GSL_G[k];
            mu1 += g * col[0][half + k]; mu2 += g * col[1][half + k];
            eS += g * col[2][half + k]; e12 += g * col[3][half + k];
        }
        if (px < W && py < H) {
            const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;  // loss_utils.py:153-154
            const float mu1_sq = mu1 * mu1, mu2_sq = mu2 * mu2, mu12 = mu1 * mu2;
            const float sig12 = e12 - mu12;
            const float A = 2.f * mu12 + C1, B = 2.f * sig12 + C2, Cc = mu1_sq + mu2_sq + C1, D = (eS - mu1_sq - mu2_sq) + C2;
            const float inv = 1.0f / (Cc * D);
            const float ssim = A * B * inv;
            const float m = weight ? weight[(size_t)py * W + px] : 1.0f;
            const float x = sx[r + GSL_R][c + GSL_R], y = sy[r + GSL_R][c + GSL_R];
            l1sum += (double)(fabsf(x - y) * m);
            ssum += (double)(ssim * m);
            if (STATE) {
                // partial derivatives of ssim w.r.t. mu1, E[x^2], E[xy] (sigma1^2 = E11 - mu1^2, sigma12 = E12 - mu1 mu2)
                const float dmu1 = (2.f * mu2 * (B - A) - 2.f * mu1 * ssim * (D - Cc)) * inv;
                const float de11 = -ssim / D;
                const float de12 = 2.f * A * inv;
                const size_t o = ch * plane + (size_t)py * W + px;
                d_mu1[o] = dmu1 * m; d_e11[o] = de11 * m; d_e12[o] = de12 * m;
            }
        }
    }
    // workgroup sums (fixed order: lanes by xor-shuffle, then the four waves)
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { l1sum += __shfl_xor(l1sum, d, 64); ssum += __shfl_xor(ssum, d, 64); }
    if ((t & 63) == 0) { red[0][t >> 6] = l1sum; red[1][t >> 6] = ssum; }
    __syncthreads();
    if (t == 0) {
        GslPartial p;
        p.l1 = ((red[0][0] + red[0][1]) + red[0][2]) + red[0][3];
        p.ssim = ((red[1][0] + red[1][1]) + red[1][2]) + red[1][3];
        partial[(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] = p;
    }
}

// one block: sums the workgroup partials in a fixed order; out = {L, mean(|d| m), mean(ssim m)}
__global__ void __launch_bounds__(256) gsl_finish_kernel(int nparts, const GslPartial* __restrict__ partial, double count,
                                                         float a_l1, float a_ssim, float* __restrict__ out)
{
    __shared__ double red[2][4];
    double l1 = 0.0, ss = 0.0;
    for (int i = threadIdx.x; i < nparts; i += 256) { l1 += partial[i].l1; ss += partial[i].ssim; }
    // (lanes by xor-shuffle, then the four waves: one barrier instead of the eight of an LDS tree)
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { l1 += __shfl_xor(l1, d, 64); ss += __shfl_xor(ss, d, 64); }
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = l1; red[1][threadIdx.x >> 6] = ss; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const double ml1 = (((red[0][0] + red[0][1]) + red[0][2]) + red[0][3]) / count, mss = (((red[1][0] + red[1][1]) + red[1][2]) + red[1][3]) / count;
        out[0] = (float)((double)a_l1 * ml1 + (double)a_ssim * mss);
        out[1] = (float)ml1;
        out[2] = (float)mss;
    }
}

__global__ void __launch_bounds__(256) gsl_backward_kernel(int H, int W, const float* __restrict__ img,
                                                           const float* __restrict__ gt,
                                                           const float* __restrict__ weight,
                                                           const float* __restrict__ d_mu1, const float* __restrict__ d_e11,
                                                           const float* __restrict__ d_e12, float c_l1, float c_ssim,
                                                           const float* __restrict__ upstream, float* __restrict__ dL_dimg, const GslTaps taps)
{
    const float* GSL_G = taps.g;
    __shared__ float sm[3][GSL_HH][GSL_HW + 1];
    __shared__ float hx[3][GSL_HH][GSL_TW];
    const int t = threadIdx.x, ch = blockIdx.z;
    const int x0 = blockIdx.x * GSL_TW, y0 = blockIdx.y * GSL_TH;
    const size_t plane = (size_t)H * W;
    const float* p0 = d_mu1 + ch * plane;
    const float* p1 = d_e11 + ch * plane;
    const float* p2 = d_e12 + ch * plane;
    {   // halo tiles of the three derivative planes; loads first, LDS stores after (see the forward)
        constexpr int NL = (GSL_HH * GSL_HW + 255) / 256;
        float v0[NL], v1[NL], v2[NL];
#pragma unroll
        for (int k = 0; k < NL; k++) {
            const int i = t + k * 256, r = i / GSL_HW, c = i - r * GSL_HW;
            const int x = x0 - GSL_R + c, y = y0 - GSL_R + r;
            const bool in = i < GSL_HH * GSL_HW;
            v0[k] = in ? gsl_load(p0, x, y, W, H) : 0.f;
            v1[k] = in ? gsl_load(p1, x, y, W, H) : 0.f;
            v2[k] = in ? gsl_load(p2, x, y, W, H) : 0.f;
        }
#pragma unroll
        for (int k = 0; k < NL; k++) {
            const int i = t + k * 256, r = i / GSL_HW, c = i - r * GSL_HW;
            if (i < GSL_HH * GSL_HW) { sm[0][r][c] = v0[k]; sm[1][r][c] = v1[k]; sm[2][r][c] = v2[k]; }
        }
    }
    __syncthreads();
    if (t < GSL_HH * (GSL_TW / 4)) {  // horizontal pass, four adjacent outputs per thread (see the forward)
        const int r = t / (GSL_TW / 4), c = (t % (GSL_TW / 4)) * 4;
        float a[3][4] = { { 0, 0, 0, 0 }, { 0, 0, 0, 0 }, { 0, 0, 0, 0 } };
#pragma unroll
        for (int k = 0; k < 14; k++) {
            const float v0 = sm[0][r][c + k], v1 = sm[1][r][c + k], v2 = sm[2][r][c + k];
#pragma unroll
            for (int o = 0; o < 4; o++) {
                const int tap = k - o;
                if (tap >= 0 && tap < 11) {
                    const float g = GSL_G[tap];
                    a[0][o] += g * v0; a[1][o] += g * v1; a[2][o] += g * v2;
                }
            }
        }
#pragma unroll
        for (int o = 0; o < 4; o++) { hx[0][r][c + o] = a[0][o]; hx[1][r][c + o] = a[1][o]; hx[2][r][c + o] = a[2][o]; }
    }
    __syncthreads();
    const float up = upstream ? upstream[0] : 1.0f;
    const int vc = t % GSL_TW, vr = (t / GSL_TW) * 2;
    float col[3][12];
#pragma unroll
    for (int q = 0; q < 3; q++)
#pragma unroll
        for (int k = 0; k < 12; k++) col[q][k] = hx[q][vr + k][vc];
#pragma unroll
    for (int half = 0; half < 2; half++) {
        const int r = vr + half, c = vc;
        const int px = x0 + c, py = y0 + r;
        float f0 = 0.f, f1 = 0.f, f2 = 0.f;
#pragma unroll
        for (int k = 0; k < 11; k++) {
            const float g = GSL_G[k];
            f0 += g * col[0][half + k]; f1 += g * col[1][half + k]; f2 += g * col[2][half + k];
        }
        if (px < W && py < H) {
            const size_t o = ch * plane + (size_t)py * W + px;
            const float x = img[o], y = gt[o];
            const float m = weight ? weight[(size_t)py * W + px] : 1.0f;
            const float d = x - y;
            const float sgn = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);  // torch.abs backward: sign(0) = 0
            dL_dimg[o] = up * (c_ssim * (f0 + 2.f * x * f1 + y * f2) + c_l1 * sgn * m);
        }
    }
}

// ---- host side -------------------------------------------------------------------------------------------------
struct GslWorkspace {
    float *d_mu1, *d_e11, *d_e12;
    GslPartial* partial;
    size_t bytes;
    int gx, gy;
};

static GslWorkspace gsl_carve(void* base, int C, int H, int W)
{
    GslWorkspace w;
    char* b = (char*)base;
    size_t off = 0;
    const size_t n = (size_t)(C > 0 ? C : 1) * (H > 0 ? H : 1) * (W > 0 ? W : 1);
    w.gx = (W + GSL_TW - 1) / GSL_TW; w.gy = (H + GSL_TH - 1) / GSL_TH;
    w.d_mu1 = (float*)(b + off); off += gsr_align(n * 4);
    w.d_e11 = (float*)(b + off); off += gsr_align(n * 4);
    w.d_e12 = (float*)(b + off); off += gsr_align(n * 4);
    w.partial = (GslPartial*)(b + off); off += gsr_align((size_t)(w.gx > 0 ? w.gx : 1) * (w.gy > 0 ? w.gy : 1) * (C > 0 ? C : 1) * sizeof(GslPartial));
    w.bytes = off;
    return w;
}

size_t gsl_workspace_bytes(int C, int H, int W) { return gsl_carve(nullptr, C, H, W).bytes; }

// taps of create_window(window_size) (loss_utils.py:113-121) centred in the 11-tap frame; window_size odd, 1..11
static bool gsl_make_taps(int window_size, GslTaps& t)
{
    if (window_size < 1 || window_size > 11 || (window_size & 1) == 0) return false;
    for (int k = 0; k < 11; k++) t.g[k] = 0.f;
    if (window_size == 11) {
        for (int k = 0; k < 11; k++) t.g[k] = GSL_G11[k];
        return true;
    }
    const int r = window_size / 2;
    float g[11], sum = 0.f;
    for (int x = 0; x < window_size; x++) {  // torch.Tensor([exp(...)]) rounds each double to fp32; the sum and the division are fp32
        g[x] = (float)exp(-(double)((x - r) * (x - r)) / (2.0 * 1.5 * 1.5));
        sum += g[x];
    }
    for (int x = 0; x < window_size; x++) t.g[GSL_R - r + x] = g[x] / sum;
    return true;
}

hipError_t gsl_launch_forward(int C, int H, int W, const float* img, const float* gt, const float* weight, float a_l1,
                              float a_ssim, void* workspace, float* out, int keep_state, int window_size, hipStream_t stream)
{
    GslTaps taps;
    if (!gsl_make_taps(window_size, taps)) return hipErrorInvalidValue;
    const GslWorkspace w = gsl_carve(workspace, C, H, W);
    const dim3 grid(w.gx, w.gy, C);
    if (keep_state)
        hipLaunchKernelGGL(gsl_forward_kernel<true>, grid, dim3(256), 0, stream, H, W, img, gt, weight, w.d_mu1, w.d_e11,
                           w.d_e12, w.partial, taps);
    else
        hipLaunchKernelGGL(gsl_forward_kernel<false>, grid, dim3(256), 0, stream, H, W, img, gt, weight, w.d_mu1, w.d_e11,
                           w.d_e12, w.partial, taps);
    hipLaunchKernelGGL(gsl_finish_kernel, dim3(1), dim3(256), 0, stream, w.gx * w.gy * C, w.partial,
                       (double)C * (double)H * (double)W, a_l1, a_ssim, out);
    return hipGetLastError();
}

hipError_t gsl_launch_backward(int C, int H, int W, const float* img, const float* gt, const float* weight, float a_l1,
                               float a_ssim, const void* workspace, const float* upstream, float* dL_dimg,
                               int window_size, hipStream_t stream)
{
    GslTaps taps;
    if (!gsl_make_taps(window_size, taps)) return hipErrorInvalidValue;
    const GslWorkspace w = gsl_carve(const_cast<void*>(workspace), C, H, W);
    const double count = (double)C * (double)H * (double)W;
    hipLaunchKernelGGL(gsl_backward_kernel, dim3(w.gx, w.gy, C), dim3(256), 0, stream, H, W, img, gt, weight, w.d_mu1,
                       w.d_e11, w.d_e12, (float)((double)a_l1 / count), (float)((double)a_ssim / count), upstream, dL_dimg, taps);
    return hipGetLastError();
}
