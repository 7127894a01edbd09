// decode.hip -- fused neural-Gaussian decode + compaction (the step right BEFORE the rasterizer).
//
// Replaces the body of GScream's gaussian_renderer/__init__.py:18-102 generate_neural_gaussians after the
// visible-anchor gather (:25-28): view vector / distance (:30-35), the four per-anchor MLPs 36 -> 32 -> {10, 10, 30, 70}
// (scene/gaussian_model.py:118-144: opacity+Tanh, uncertainty+Sigmoid, color+Sigmoid, cov linear), the opacity mask
// (:57-60), the [N*K, 23] concat + boolean-mask compaction (:78-87) and the post-processing (:90-96:
// scaling = grid_scaling[:,3:] * sigmoid(.), rot = normalize(.), xyz = anchor + offset * grid_scaling[:,:3]).
// The torch path materialises ~20 intermediates of up to [N*K, 23] floats; here one thread owns one anchor, nothing
// but the compacted per-Gaussian outputs reaches HBM in the forward.
//
// Mapping.  Thread = anchor.  The MLP weights are wave-uniform operands, staged once per workgroup in LDS and read as
// broadcasts, so a multiply-accumulate is one VALU op per 64 anchors.  Layer 1 (32 x 36) is fully unrolled against the
// register-resident input -- every index is a compile-time constant; layer 2 walks its output rows in a real loop
// against the 32 hidden registers.
//   pass A  gsd_count_kernel : opacity MLP only -> neural_opacity[N*K], mask[N*K], per-anchor survivor count
//   scan    gsd_scan_kernel  : exclusive scan of the counts (one block; N ~ 2e5) -> first output row per anchor, total
//   pass B  gsd_emit_kernel  : all four MLPs, writes the surviving offsets' rows in the reference's order
//                              (anchor-major, offset-minor = boolean-mask order)
//   bwd     gsd_backward_mlp_kernel<M> x4 : recompute MLP M's activations, turn the per-Gaussian upstream gradients into
//                              its layer deltas (stored feature-major with the activations for the caller's weight-
//                              gradient GEMMs, delta @ activations^T);
//           gsd_backward_input_kernel : W1^T deltas -> gradients of feat / anchor, and the offset / grid-scaling geometry.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gsr_common.h"
#include "gsr_math.h"

#define GSD_F 32        // feat_dim (arguments/__init__.py:50)
#define GSD_IN 36       // feat + view(3) + dist(1)
#define GSD_HID 32
#define GSD_MAXK 10     // n_offsets (arguments/__init__.py:51); the kernels handle K <= 10
#define GSD_THREADS 256
// Layout of the per-anchor arrays D2 / D1 / H / X (deltas and activations kept for the weight gradients): chunks of 64
// anchors, feature-major inside a chunk -- element (row, anchor n) of an array with ROWS rows lives at
// ((n / 64) * ROWS + row) * 64 + n % 64.  A wavefront (64 consecutive anchors) writes 256 contiguous bytes per row and all
// rows of its chunk lie within 64 KB; the weight-gradient kernel reads 16-byte groups of 4 anchors.  (Plain [row][N] kept
// the 64 anchors contiguous too, but put the rows 800 KB apart: every wave touched ~100 pages per step.)  Arrays are
// sized for N rounded up to a whole chunk.
__host__ __device__ static inline int gsd_ld(int N) { return (N + 63) & ~63; }
#define GSD_AT(ROWS, row, n) ((((size_t)(n) >> 6) * (size_t)(ROWS) + (size_t)(row)) * 64 + ((size_t)(n) & 63))

struct GsdMlps {  // device pointers; m = 0 opacity (K, tanh), 1 uncertainty (K, sigmoid), 2 color (3K, sigmoid), 3 cov (7K)
    const float* w1[4];  // [32][36] row-major (torch Linear.weight)
    const float* b1[4];  // [32]
    const float* w2[4];  // [out][32]
    const float* b2[4];  // [out]
};

__device__ __forceinline__ float gsd_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }

// input vector of one anchor (gaussian_renderer/__init__.py:30-47): [feat(32), ob_view(3), ob_dist]
__device__ __forceinline__ void gsd_input(const float* __restrict__ feat, const float* __restrict__ anchor,
                                          const float* __restrict__ campos, int a /* row in the model's tensors */,
                                          float x[GSD_IN], float& dist)
{
    const float4* f4 = reinterpret_cast<const float4*>(feat + (size_t)a * GSD_F);
#pragma unroll
    for (int i = 0; i < GSD_F / 4; i++) {
        const float4 v = f4[i];
        x[4 * i] = v.x; x[4 * i + 1] = v.y; x[4 * i + 2] = v.z; x[4 * i + 3] = v.w;
    }
    const float vx = anchor[3 * (size_t)a] - campos[0], vy = anchor[3 * (size_t)a + 1] - campos[1], vz = anchor[3 * (size_t)a + 2] - campos[2];
    dist = sqrtf(vx * vx + vy * vy + vz * vz);
    x[32] = vx / dist; x[33] = vy / dist; x[34] = vz / dist; x[35] = dist;
}

// The weights of the MLPs a kernel needs are staged in LDS once per workgroup and read back as broadcasts
// (same address in every lane: one ds_read_b128 feeds four FMAs of 64 anchors each).  Scalar loads were measured
// first: they keep the FMA at one VALU op per 64 anchors too, but every 16-weight s_load is a ~200-cycle round trip
// the wave has to wait out, and the kernels ran at a quarter of their VALU bound.
struct GsdLds {  // offsets (floats) into the staging buffer
    int w1[4], b1[4], w2[4], b2[4];
};
#define GSD_LDS_FLOATS(K) (4 * GSD_HID * GSD_IN + 4 * GSD_HID + 12 * (K) * GSD_HID + 12 * (K))
__device__ __forceinline__ GsdLds gsd_stage_weights(const GsdMlps& P, int K, float* sw, int m_lo, int m_hi, bool first_layer_only = false)
{
    GsdLds L;
    int off = 0;
    const int outs[4] = { K, K, 3 * K, 7 * K };
#pragma unroll
    for (int m = 0; m < 4; m++) { L.w1[m] = off; off += GSD_HID * GSD_IN; }
#pragma unroll
    for (int m = 0; m < 4; m++) { L.b1[m] = off; off += GSD_HID; }
#pragma unroll
    for (int m = 0; m < 4; m++) { L.w2[m] = off; off += outs[m] * GSD_HID; }
#pragma unroll
    for (int m = 0; m < 4; m++) { L.b2[m] = off; off += outs[m]; }
    for (int m = m_lo; m <= m_hi; m++) {
        for (int i = threadIdx.x; i < GSD_HID * GSD_IN; i += blockDim.x) sw[L.w1[m] + i] = P.w1[m][i];
        if (first_layer_only) continue;
        for (int i = threadIdx.x; i < GSD_HID; i += blockDim.x) sw[L.b1[m] + i] = P.b1[m][i];
        for (int i = threadIdx.x; i < outs[m] * GSD_HID; i += blockDim.x) sw[L.w2[m] + i] = P.w2[m][i];
        for (int i = threadIdx.x; i < outs[m]; i += blockDim.x) sw[L.b2[m] + i] = P.b2[m][i];
    }
    __syncthreads();
    return L;
}

// The multiply-accumulates run as v_pk_fma_f32 on PAIRS of adjacent weights (a row of W as float2s) against pairs of
// inputs: one packed FMA = two MACs in 4.2 issue cycles, where the scalar form acc += w * x reads three VGPRs and costs
// 3.8-4.0 per MAC (tools/microbench/valu_issue.hip) -- the packed form is the only way to the fp32 FMA rate when both
// factors live in VGPRs.  Dot products therefore come out as (sum over even terms) + (sum over odd terms).
typedef float gsd_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ gsd_f2 gsd_fma2(gsd_f2 a, gsd_f2 b, gsd_f2 c) { return __builtin_elementwise_fma(a, b, c); }

// layer 1 of MLP m, fully unrolled into registers (post-ReLU): every index is a compile-time constant.
__device__ __forceinline__ void gsd_layer1(const float* sw, const GsdLds& L, int m, const float x[GSD_IN], float h[GSD_HID])
{
    const float* w = sw + L.w1[m];
    const float* b = sw + L.b1[m];
#pragma unroll
    for (int j = 0; j < GSD_HID; j++) {
        gsd_f2 s = {b[j], 0.f};
#pragma unroll
        for (int i = 0; i < GSD_IN; i += 2) {
            const gsd_f2 wv = *reinterpret_cast<const gsd_f2*>(w + j * GSD_IN + i), xv = {x[i], x[i + 1]};
            s = gsd_fma2(wv, xv, s);
        }
        h[j] = fmaxf(s.x + s.y, 0.0f);
    }
}
__device__ __forceinline__ float gsd_out(const float* sw, const GsdLds& L, int m, int o, const float h[GSD_HID])
{
    const float* w = sw + L.w2[m] + o * GSD_HID;
    gsd_f2 s = {sw[L.b2[m] + o], 0.f};
#pragma unroll
    for (int j = 0; j < GSD_HID; j += 2) {
        const gsd_f2 wv = *reinterpret_cast<const gsd_f2*>(w + j), hv = {h[j], h[j + 1]};
        s = gsd_fma2(wv, hv, s);
    }
    return s.x + s.y;
}

// ---- pass A: opacity MLP, mask, count ---------------------------------------------------------------------------
__global__ void __launch_bounds__(GSD_THREADS) gsd_count_kernel(int N, int K, GsdMlps P, const int32_t* __restrict__ vis,
                                                                const float* __restrict__ feat,
                                                                const float* __restrict__ anchor,
                                                                const float* __restrict__ campos,
                                                                float* __restrict__ neural_opacity,
                                                                uint8_t* __restrict__ mask, uint8_t* __restrict__ count,
                                                                uint32_t* __restrict__ block_sum)
{
    __shared__ uint32_t bs;
    extern __shared__ __attribute__((aligned(16))) float sw[];
    const int n = blockIdx.x * GSD_THREADS + threadIdx.x;
    if (threadIdx.x == 0) bs = 0u;
    const GsdLds L = gsd_stage_weights(P, K, sw, 0, 0);  // (contains the barrier)
    int c = 0;
    if (n < N) {
        float x[GSD_IN], dist, h[GSD_HID];
        gsd_input(feat, anchor, campos, vis ? vis[n] : n, x, dist);
        gsd_layer1(sw, L, 0, x, h);
#pragma unroll 1
        for (int k = 0; k < K; k++) {
            const float op = tanhf(gsd_out(sw, L, 0, k, h));
            const bool keep = op > 0.0f;  // gaussian_renderer/__init__.py:59
            neural_opacity[(size_t)n * K + k] = op;
            mask[(size_t)n * K + k] = keep ? 1 : 0;
            c += keep ? 1 : 0;
        }
        count[n] = (uint8_t)c;
    }
    // survivors of this block of 256 anchors: the scan below only has to cover N/256 block totals
    const uint32_t ws = gsr_wave_scan_add((uint32_t)c);
    if ((threadIdx.x & 63) == 63 && ws != 0u) atomicAdd(&bs, ws);
    __syncthreads();
    if (threadIdx.x == 0) block_sum[blockIdx.x] = bs;
}

// ---- exclusive scan of the block totals (single block of 1024; thread i owns a contiguous run of them) -------------
__global__ void __launch_bounds__(1024) gsd_scan_kernel(int nblocks, uint32_t* __restrict__ block_sum /* in place -> exclusive */,
                                                        uint32_t* __restrict__ total)
{
    __shared__ uint32_t wsum[16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int per = (nblocks + 1023) / 1024, i0 = threadIdx.x * per;
    uint32_t s = 0;
    for (int i = 0; i < per; i++) s += i0 + i < nblocks ? block_sum[i0 + i] : 0u;
    const uint32_t incl = gsr_wave_scan_add(s);
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    uint32_t run = incl - s, tot = 0;
#pragma unroll
    for (int w = 0; w < 16; w++) { const uint32_t sw = wsum[w]; run += w < wave ? sw : 0u; tot += sw; }
    for (int i = 0; i < per && i0 + i < nblocks; i++) { const uint32_t v = block_sum[i0 + i]; block_sum[i0 + i] = run; run += v; }
    if (threadIdx.x == 0) total[0] = tot;
}

// first output row of every anchor = its block's base + the exclusive scan of the counts inside the block
__global__ void __launch_bounds__(GSD_THREADS) gsd_first_kernel(int N, const uint8_t* __restrict__ count,
                                                                const uint32_t* __restrict__ block_base, uint32_t* __restrict__ first)
{
    __shared__ uint32_t wsum[GSD_THREADS / 64];
    const int n = blockIdx.x * GSD_THREADS + threadIdx.x;
    const uint32_t c = n < N ? count[n] : 0u;
    const uint32_t incl = gsr_wave_scan_add(c);
    if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = incl;
    __syncthreads();
    uint32_t run = block_base[blockIdx.x] + incl - c;
    for (int w = 0; w < (int)(threadIdx.x >> 6); w++) run += wsum[w];
    if (n < N) first[n] = run;
}

// ---- pass B: full decode, compacted output ------------------------------------------------------------------------
__global__ void __launch_bounds__(GSD_THREADS) gsd_emit_kernel(
    int N, int K, GsdMlps P, const int32_t* __restrict__ vis, const float* __restrict__ feat, const float* __restrict__ anchor,
    const float* __restrict__ offsets /*[N,K,3]*/, const float* __restrict__ gscale /*[N,6]*/,
    const float* __restrict__ campos, const float* __restrict__ neural_opacity, const uint8_t* __restrict__ mask,
    const uint32_t* __restrict__ first, float* __restrict__ xyz, float* __restrict__ color, float* __restrict__ opacity, float* __restrict__ uncertainty,
    float* __restrict__ scaling, float* __restrict__ rot)
{
    extern __shared__ __attribute__((aligned(16))) float sw[];
    const GsdLds L = gsd_stage_weights(P, K, sw, 1, 3);
    const int n = blockIdx.x * GSD_THREADS + threadIdx.x;
    if (n >= N) return;
    const int a = vis ? vis[n] : n;  // visible-anchor gather (gaussian_renderer/__init__.py:25-28) folded in
    float x[GSD_IN], dist, h[GSD_HID];
    gsd_input(feat, anchor, campos, a, x, dist);
    uint32_t keep = 0;
    for (int k = 0; k < K; k++) keep |= mask[(size_t)n * K + k] ? (1u << k) : 0u;
    if (keep == 0u) return;
    const uint32_t row0 = first[n];
    const float ax = anchor[3 * (size_t)a], ay = anchor[3 * (size_t)a + 1], az = anchor[3 * (size_t)a + 2];
    float gs[6];
#pragma unroll
    for (int i = 0; i < 6; i++) gs[i] = gscale[6 * (size_t)a + i];

    // opacity = neural_opacity[mask] (:63): copied from pass A, bit for bit; geometry of the offsets
#pragma unroll 1
    for (int k = 0; k < K; k++) {
        if (!((keep >> k) & 1u)) continue;
        const uint32_t r = row0 + (uint32_t)__popc(keep & ((1u << k) - 1u));
        opacity[r] = neural_opacity[(size_t)n * K + k];
        const float* of = offsets + ((size_t)a * K + k) * 3;
        xyz[3 * (size_t)r] = ax + of[0] * gs[0];       // :94-95
        xyz[3 * (size_t)r + 1] = ay + of[1] * gs[1];
        xyz[3 * (size_t)r + 2] = az + of[2] * gs[2];
    }
    gsd_layer1(sw, L, 1, x, h);
#pragma unroll 1
    for (int k = 0; k < K; k++) {
        if (!((keep >> k) & 1u)) continue;
        const uint32_t r = row0 + (uint32_t)__popc(keep & ((1u << k) - 1u));
        uncertainty[r] = gsd_sigmoid(gsd_out(sw, L, 1, k, h));
    }
    gsd_layer1(sw, L, 2, x, h);
#pragma unroll 1
    for (int k = 0; k < K; k++) {
        if (!((keep >> k) & 1u)) continue;
        const uint32_t r = row0 + (uint32_t)__popc(keep & ((1u << k) - 1u));
#pragma unroll
        for (int c = 0; c < 3; c++) color[3 * (size_t)r + c] = gsd_sigmoid(gsd_out(sw, L, 2, 3 * k + c, h));
    }
    gsd_layer1(sw, L, 3, x, h);
#pragma unroll 1
    for (int k = 0; k < K; k++) {
        if (!((keep >> k) & 1u)) continue;
        const uint32_t r = row0 + (uint32_t)__popc(keep & ((1u << k) - 1u));
        float sr[7];
#pragma unroll
        for (int c = 0; c < 7; c++) sr[c] = gsd_out(sw, L, 3, 7 * k + c, h);
#pragma unroll
        for (int c = 0; c < 3; c++) scaling[3 * (size_t)r + c] = gs[3 + c] * gsd_sigmoid(sr[c]);  // :90
        const float nrm = fmaxf(sqrtf(sr[3] * sr[3] + sr[4] * sr[4] + sr[5] * sr[5] + sr[6] * sr[6]), 1e-12f);  // F.normalize
#pragma unroll
        for (int c = 0; c < 4; c++) rot[4 * (size_t)r + c] = sr[3 + c] / nrm;  // :91
    }
}

// ---- backward -------------------------------------------------------------------------------------------------------
// Per anchor: upstream gradients of its surviving rows -> d(out) of the four second layers -> d(hidden) -> d(input).
// Writes d_feat[N,32], d_anchor[N,3], d_offsets[N,K,3], d_gscale[N,6] and, for the weight-gradient GEMMs of the caller,
//   D2[12K, N] (opacity K | uncertainty K | color 3K | cov 7K)   = dL/d(second-layer pre-activations)
//   D1[128, N], H[128, N]  (four blocks of 32)                    = dL/d(first-layer pre-activations), hidden activations
//   X [36, N]                                                     = the MLP input
// all feature-major, so that the 64 anchors of a wave store 64 consecutive floats
// One MLP per launch (template M): keeps the live state at input + hidden + d(hidden) registers, so several
// waves fit per SIMD (the all-in-one version needed 256 VGPRs + 51 AGPRs and spilled SGPRs: one wave per SIMD).
template <int M>
__global__ void __launch_bounds__(GSD_THREADS) gsd_backward_mlp_kernel(
    int N, int K, GsdMlps P, const int32_t* __restrict__ vis, const float* __restrict__ feat, const float* __restrict__ anchor,
    const float* __restrict__ gscale, const float* __restrict__ campos, const uint8_t* __restrict__ mask,
    const uint32_t* __restrict__ first, const float* __restrict__ g_color, const float* __restrict__ g_opacity,
    const float* __restrict__ g_unc, const float* __restrict__ g_scaling, const float* __restrict__ g_rot,
    float* __restrict__ d_gscale, float* __restrict__ D2, float* __restrict__ D1, float* __restrict__ Hout,
    float* __restrict__ Xout)
{
    extern __shared__ __attribute__((aligned(16))) float sw[];
    const GsdLds L = gsd_stage_weights(P, K, sw, M, M);
    const int n = blockIdx.x * GSD_THREADS + threadIdx.x;
    if (n >= N) return;
    const int a = vis ? vis[n] : n;
    float x[GSD_IN], dist, h[GSD_HID], dh[GSD_HID];
    gsd_input(feat, anchor, campos, a, x, dist);
    if (M == 0) {
#pragma unroll
        for (int i = 0; i < GSD_IN; i++) Xout[GSD_AT(GSD_IN, i, n)] = x[i];
    }
    uint32_t keep = 0;
    for (int k = 0; k < K; k++) keep |= mask[(size_t)n * K + k] ? (1u << k) : 0u;
    const uint32_t row0 = first[n];
    float gs3[3] = { 0, 0, 0 }, dgs3[3] = { 0, 0, 0 };
    if (M == 3) {
#pragma unroll
        for (int c = 0; c < 3; c++) gs3[c] = gscale[6 * (size_t)a + 3 + c];
    }
    gsd_layer1(sw, L, M, x, h);
#pragma unroll
    for (int j = 0; j < GSD_HID; j++) { Hout[GSD_AT(128, M * 32 + j, n)] = h[j]; dh[j] = 0.f; }
    constexpr int per = M == 0 || M == 1 ? 1 : (M == 2 ? 3 : 7);
    const int out_base = M == 0 ? 0 : (M == 1 ? K : (M == 2 ? 2 * K : 5 * K));
#pragma unroll 1
    for (int k = 0; k < K; k++) {
        const bool on = (keep >> k) & 1u;
        const uint32_t r = row0 + (uint32_t)__popc(keep & ((1u << k) - 1u));
        float dz[per];  // dL/d(pre-activation) of this offset's outputs
#pragma unroll
        for (int c = 0; c < per; c++) dz[c] = 0.f;
        if (on) {
            if (M == 0) {  // opacity = tanh(z)
                const float t = tanhf(gsd_out(sw, L, 0, k, h));
                dz[0] = g_opacity[r] * (1.0f - t * t);
            } else if (M == 1) {  // sigmoid
                const float sg = gsd_sigmoid(gsd_out(sw, L, 1, k, h));
                dz[0] = g_unc[r] * sg * (1.0f - sg);
            } else if (M == 2) {
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    const float sg = gsd_sigmoid(gsd_out(sw, L, 2, 3 * k + c, h));
                    dz[c % per] = g_color[3 * (size_t)r + c] * sg * (1.0f - sg);
                }
            } else {
                float sr[7];
#pragma unroll
                for (int c = 0; c < 7; c++) {
                    sr[c] = gsd_out(sw, L, 3, 7 * k + c, h);
                    __builtin_amdgcn_sched_barrier(0);  // one weight row in flight at a time: 7 x 32 live scalars do not fit
                }
#pragma unroll
                for (int c = 0; c < 3; c++) {  // scaling = gs[3+c] * sigmoid(sr[c])
                    const float sg = gsd_sigmoid(sr[c]), g = g_scaling[3 * (size_t)r + c];
                    dz[c % per] = g * gs3[c] * sg * (1.0f - sg);
                    dgs3[c] += g * sg;
                }
                // rot = q / max(|q|, eps): d q = (g - rot (rot . g)) / |q|
                const float nrm = fmaxf(sqrtf(sr[3] * sr[3] + sr[4] * sr[4] + sr[5] * sr[5] + sr[6] * sr[6]), 1e-12f);
                float gq[4], dot = 0.f;
#pragma unroll
                for (int c = 0; c < 4; c++) { gq[c] = g_rot[4 * (size_t)r + c]; dot += gq[c] * (sr[3 + c] / nrm); }
#pragma unroll
                for (int c = 0; c < 4; c++) dz[(3 + c) % per] = (gq[c] - (sr[3 + c] / nrm) * dot) / nrm;
            }
        }
#pragma unroll
        for (int c = 0; c < per; c++) {
            const int o = per * k + c;
            D2[GSD_AT(12 * K, out_base + o, n)] = dz[c];
            const float* w = sw + L.w2[M] + o * GSD_HID;
#pragma unroll
            for (int j = 0; j < GSD_HID; j += 2) {  // two hidden units per packed FMA
                gsd_f2 d = {dh[j], dh[j + 1]};
                d = gsd_fma2(*reinterpret_cast<const gsd_f2*>(w + j), (gsd_f2){dz[c], dz[c]}, d);
                dh[j] = d.x; dh[j + 1] = d.y;
            }
            if (per > 1) __builtin_amdgcn_sched_barrier(0);
        }
    }
#pragma unroll
    for (int j = 0; j < GSD_HID; j++) D1[GSD_AT(128, M * 32 + j, n)] = h[j] > 0.0f ? dh[j] : 0.0f;  // through the ReLU
    if (M == 3) {
#pragma unroll
        for (int c = 0; c < 3; c++) d_gscale[6 * (size_t)a + 3 + c] = dgs3[c];
    }
}

// dL/d(input) = sum over the four MLPs of W1^T d(pre1) (read back feature-major, coalesced), then feature / anchor
// gradients; also the geometry part: xyz = anchor + offset * gs[0:3].
__global__ void __launch_bounds__(GSD_THREADS) gsd_backward_input_kernel(
    int N, int K, GsdMlps P, const int32_t* __restrict__ vis, const float* __restrict__ anchor,
    const float* __restrict__ offsets, const float* __restrict__ gscale, const float* __restrict__ campos,
    const uint8_t* __restrict__ mask, const uint32_t* __restrict__ first, const float* __restrict__ g_xyz,
    const float* __restrict__ D1, float* __restrict__ d_feat, float* __restrict__ d_anchor, float* __restrict__ d_offsets,
    float* __restrict__ d_gscale)
{
    extern __shared__ __attribute__((aligned(16))) float sw[];
    const GsdLds L = gsd_stage_weights(P, K, sw, 0, 3, true);
    const int n = blockIdx.x * GSD_THREADS + threadIdx.x;
    if (n >= N) return;
    const int a = vis ? vis[n] : n;
    float dx[GSD_IN];
#pragma unroll
    for (int i = 0; i < GSD_IN; i++) dx[i] = 0.f;
#pragma unroll 1
    for (int m = 0; m < 4; m++) {
        const float* w1 = sw + L.w1[m];
#pragma unroll 4
        for (int j = 0; j < GSD_HID; j++) {
            const float d1 = D1[GSD_AT(128, m * 32 + j, n)];
#pragma unroll
            for (int i = 0; i < GSD_IN; i += 2) {  // two inputs per packed FMA
                gsd_f2 d = {dx[i], dx[i + 1]};
                d = gsd_fma2(*reinterpret_cast<const gsd_f2*>(w1 + j * GSD_IN + i), (gsd_f2){d1, d1}, d);
                dx[i] = d.x; dx[i + 1] = d.y;
            }
        }
    }
#pragma unroll
    for (int i = 0; i < GSD_F; i++) d_feat[(size_t)a * GSD_F + i] = dx[i];
    // view vector / distance back to the anchor (v = a - c, dist = |v|, view = v / dist)
    const float vx = anchor[3 * (size_t)a] - campos[0], vy = anchor[3 * (size_t)a + 1] - campos[1], vz = anchor[3 * (size_t)a + 2] - campos[2];
    const float dist = sqrtf(vx * vx + vy * vy + vz * vz);
    const float ux = vx / dist, uy = vy / dist, uz = vz / dist;
    const float gdot = dx[32] * ux + dx[33] * uy + dx[34] * uz;
    float da[3] = { (dx[32] - ux * gdot) / dist + dx[35] * ux, (dx[33] - uy * gdot) / dist + dx[35] * uy,
                    (dx[34] - uz * gdot) / dist + dx[35] * uz };
    uint32_t keep = 0;
    for (int k = 0; k < K; k++) keep |= mask[(size_t)n * K + k] ? (1u << k) : 0u;
    const uint32_t row0 = first[n];
    float gs[3], dgs[3] = { 0, 0, 0 };
#pragma unroll
    for (int c = 0; c < 3; c++) gs[c] = gscale[6 * (size_t)a + c];
    for (int k = 0; k < K; k++) {
        float* dof = d_offsets + ((size_t)a * K + k) * 3;
        if (!((keep >> k) & 1u)) { dof[0] = dof[1] = dof[2] = 0.f; continue; }
        const uint32_t r = row0 + (uint32_t)__popc(keep & ((1u << k) - 1u));
        const float* of = offsets + ((size_t)a * K + k) * 3;
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const float g = g_xyz[3 * (size_t)r + c];
            da[c] += g; dof[c] = g * gs[c]; dgs[c] += g * of[c];
        }
    }
#pragma unroll
    for (int c = 0; c < 3; c++) { d_anchor[3 * (size_t)a + c] = da[c]; d_gscale[6 * (size_t)a + c] = dgs[c]; }
}

// ---- weight gradients: D @ A^T over the anchors, on the f32 matrix cores ------------------------------------------------
// gw2[m][o][j] = sum_n D2[base_m + o][n] H[32 m + j][n],  gb2[m][o] = sum_n D2[base_m + o][n]
// gw1[m][j][i] = sum_n D1[32 m + j][n]  X[i][n],           gb1[m][j] = sum_n D1[32 m + j][n]
// (only the diagonal blocks of the two big products exist: MLP m's deltas meet MLP m's activations).  All reduction, no
// reuse beyond the tile: the kernel is bound by reading the four arrays once (1.3 KB per anchor).  v_mfma_f32_16x16x4_f32
// (exact fp32, k-ordered fma chain) with the ANCHORS as the K dimension: lane l supplies A[row l & 15][k = l >> 4] and
// B[k = l >> 4][col l & 15]; one 16-byte load per lane and tile (row l & 15, anchors n0 + 4 (l >> 4) .. + 3) feeds four
// MFMA steps (step e takes component e: both operands use the same anchor -> k mapping, which is all that matters).
// ROLE 0 = second layers (9 delta tiles x 2 hidden tiles per 16 anchors = 18 MFMA per step), ROLE 1 = first layers (per
// MLP 2 delta tiles x 3 input tiles = 24 MFMA per step; input row 36 is a row of ones, so gb1 falls out of the product).
// gb2: each lane adds up the delta components it loads (VALU), reduced over the four k-groups at the end.
// A wave owns every (total waves)-th group of 16 anchors; the four waves of a workgroup are summed through LDS in wave
// order, workgroup partials go to the workspace and gsd_weight_grad_finish_kernel adds them in workgroup order (double):
// bit-reproducible.
typedef float gsd_f4 __attribute__((ext_vector_type(4)));
#define GSD_WG_BLOCKS 512
#define GSD_WG2_ROWS (12 * GSD_MAXK)   // delta rows of the second layers (K | K | 3K | 7K)
#define GSD_WG2_COLS 33                // 32 hidden + bias
#define GSD_WG1_ROWS 128
#define GSD_WG1_COLS 48                // 36 inputs + ones row (bias) + padding of the third tile

__device__ __forceinline__ float4 gsd_load_tile(const float* __restrict__ base, int rows, int row, int row_end, int n, int N)
{
    // 4 consecutive anchors of one row.  The load itself is unconditional (row clamped into the array; n < the padded
    // anchor count by construction), so that all the loads of a step are in flight together; rows beyond the block and
    // anchors beyond N are zeroed afterwards by selects (the padding may hold anything, NaNs included).
    const bool rok = row < row_end;
    float4 v = *reinterpret_cast<const float4*>(base + GSD_AT(rows, rok ? row : 0, n));
    v.x = (rok && n < N) ? v.x : 0.f;
    v.y = (rok && n + 1 < N) ? v.y : 0.f;
    v.z = (rok && n + 2 < N) ? v.z : 0.f;
    v.w = (rok && n + 3 < N) ? v.w : 0.f;
    return v;
}
__device__ __forceinline__ float gsd_comp(const float4& v, int e) { return e == 0 ? v.x : e == 1 ? v.y : e == 2 ? v.z : v.w; }

template <int ROLE>
__global__ void __launch_bounds__(256) gsd_weight_grad_kernel(int N, int K, const float* __restrict__ Dm /* D2 | D1 */,
                                                              const float* __restrict__ Am /* H | X */,
                                                              const float* __restrict__ Hm /* ROLE 1: unused */,
                                                              float* __restrict__ partial)
{
    constexpr int ROWS = ROLE == 0 ? GSD_WG2_ROWS : GSD_WG1_ROWS, COLS = ROLE == 0 ? GSD_WG2_COLS : GSD_WG1_COLS;
    __shared__ float red[ROWS * COLS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r16 = lane & 15, kg = lane >> 4;
    const int npairs = gridDim.x * 2, pair = blockIdx.x * 2 + (wave >> 1), par = wave & 1;
    const int nchunks = (N + 63) >> 6;
    for (int i = threadIdx.x; i < ROWS * COLS; i += 256) red[i] = 0.f;
    if (ROLE == 0) {
        // delta tiles: MLP m owns rows [base_m, base_m + out_m), ceil(out_m / 16) tiles; tile list for K = 10: 1, 1, 2, 5
        const int outs[4] = { K, K, 3 * K, 7 * K }, base[4] = { 0, K, 2 * K, 5 * K };
        constexpr int NT[4] = { 1, 1, 2, 5 }, T0[4] = { 0, 1, 2, 4 };  // tiles per MLP at GSD_MAXK, first tile index
        gsd_f4 acc[9][2];
        float bsum[9];
#pragma unroll
        for (int t = 0; t < 9; t++) { bsum[t] = 0.f; acc[t][0] = acc[t][1] = (gsd_f4){0.f, 0.f, 0.f, 0.f}; }
        // A chunk of 64 anchors is shared by a PAIR of waves of the workgroup: wave parity p takes the 16-anchor steps p and
        // p + 2, so the two 64-byte halves of every 128-byte line are requested by the two waves at the same time and the
        // line is fetched from HBM once (one wave walking steps 0..3 fetched every line twice: PMC 373 MB for 198 MB --
        // the in-flight working set of 2048 waves is larger than the L2s).
        for (int g = pair * 4 + par; g < nchunks * 4; g = (g & 2) ? g + 4 * npairs - 2 : g + 2) {
            const int n = g * 16 + kg * 4;
            float4 d[9], h[8];
#pragma unroll
            for (int m = 0; m < 4; m++) {
#pragma unroll
                for (int tt = 0; tt < NT[m]; tt++) d[T0[m] + tt] = gsd_load_tile(Dm, 12 * K, base[m] + tt * 16 + r16, base[m] + outs[m], n, N);
#pragma unroll
                for (int c = 0; c < 2; c++) h[2 * m + c] = gsd_load_tile(Am, 128, 32 * m + c * 16 + r16, 128, n, N);
            }
#pragma unroll
            for (int t = 0; t < 9; t++) bsum[t] += (d[t].x + d[t].y) + (d[t].z + d[t].w);
#pragma unroll
            for (int e = 0; e < 4; e++) {
#pragma unroll
                for (int m = 0; m < 4; m++) {
#pragma unroll
                    for (int tt = 0; tt < NT[m]; tt++) {
#pragma unroll
                        for (int c = 0; c < 2; c++)
                            acc[T0[m] + tt][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(gsd_comp(d[T0[m] + tt], e), gsd_comp(h[2 * m + c], e), acc[T0[m] + tt][c], 0, 0, 0);
                    }
                }
            }
        }
        // wave -> LDS, in wave order (C/D layout: col = lane & 15, row = 4 (lane >> 4) + reg)
        for (int w = 0; w < 4; w++) {
            __syncthreads();
            if (wave != w) continue;
#pragma unroll
            for (int m = 0; m < 4; m++) {
#pragma unroll
                for (int tt = 0; tt < NT[m]; tt++) {
                    const int t = T0[m] + tt;
#pragma unroll
                    for (int rg = 0; rg < 4; rg++) {
                        const int o = tt * 16 + kg * 4 + rg;  // output row inside the MLP's block
                        if (o < outs[m]) {
#pragma unroll
                            for (int c = 0; c < 2; c++) red[(base[m] + o) * COLS + c * 16 + r16] += acc[t][c][rg];
                        }
                    }
                    // bias: lanes r16, r16 + 16, r16 + 32, r16 + 48 hold the four k-group partial sums of row tt * 16 + r16
                    float b = bsum[t];
                    b += __shfl_xor(b, 16, 64);
                    b += __shfl_xor(b, 32, 64);
                    if (kg == 0 && tt * 16 + r16 < outs[m]) red[(base[m] + tt * 16 + r16) * COLS + 32] += b;
                }
            }
        }
    } else {
        gsd_f4 acc[4][2][3];
#pragma unroll
        for (int m = 0; m < 4; m++)
#pragma unroll
            for (int a = 0; a < 2; a++)
#pragma unroll
                for (int c = 0; c < 3; c++) acc[m][a][c] = (gsd_f4){0.f, 0.f, 0.f, 0.f};
        for (int g = pair * 4 + par; g < nchunks * 4; g = (g & 2) ? g + 4 * npairs - 2 : g + 2) {  // wave pairs share chunks, see ROLE 0
            const int n = g * 16 + kg * 4;
            float4 d[8], x[3];
#pragma unroll
            for (int t = 0; t < 8; t++) d[t] = gsd_load_tile(Dm, 128, t * 16 + r16, 128, n, N);
#pragma unroll
            for (int c = 0; c < 3; c++) x[c] = gsd_load_tile(Am, GSD_IN, c * 16 + r16, 36, n, N);
            if (r16 == 4) {  // row 36 of the input tile: ones where the anchor exists -> the bias gradient
                x[2] = make_float4(n < N ? 1.f : 0.f, n + 1 < N ? 1.f : 0.f, n + 2 < N ? 1.f : 0.f, n + 3 < N ? 1.f : 0.f);
            }
#pragma unroll
            for (int e = 0; e < 4; e++) {
#pragma unroll
                for (int m = 0; m < 4; m++)
#pragma unroll
                    for (int a = 0; a < 2; a++)
#pragma unroll
                        for (int c = 0; c < 3; c++)
                            acc[m][a][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(gsd_comp(d[2 * m + a], e), gsd_comp(x[c], e), acc[m][a][c], 0, 0, 0);
            }
        }
        for (int w = 0; w < 4; w++) {
            __syncthreads();
            if (wave != w) continue;
#pragma unroll
            for (int m = 0; m < 4; m++)
#pragma unroll
                for (int a = 0; a < 2; a++)
#pragma unroll
                    for (int c = 0; c < 3; c++)
#pragma unroll
                        for (int rg = 0; rg < 4; rg++) red[(32 * m + 16 * a + kg * 4 + rg) * COLS + c * 16 + r16] += acc[m][a][c][rg];
        }
    }
    __syncthreads();
    float* dst = partial + (size_t)blockIdx.x * ROWS * COLS;
    for (int i = threadIdx.x; i < ROWS * COLS; i += 256) dst[i] = red[i];
}

// sums the workgroup partials in workgroup order (double) and scatters them into the 16 gradient tensors
struct GsdGrads { float* g[16]; };  // { gw1[4], gb1[4], gw2[4], gb2[4] }
__global__ void __launch_bounds__(256) gsd_weight_grad_finish_kernel(int K, int nblocks, const float* __restrict__ partial2,
                                                                     const float* __restrict__ partial1, GsdGrads G)
{
    // 64 elements per workgroup, four threads per element: thread (q, l) adds the q-th quarter of the workgroup partials of
    // element l (loads unrolled), the four quarter sums meet in LDS and are added in order
    __shared__ double quarter[4][64];
    const int l = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + l;
    constexpr int n2 = GSD_WG2_ROWS * GSD_WG2_COLS, n1 = GSD_WG1_ROWS * GSD_WG1_COLS;
    const bool live = i < n2 + n1, second = i < n2;
    const int e = second ? i : i - n2;
    const float* p = second ? partial2 : partial1;
    const int stride = second ? n2 : n1;
    double s = 0.0;
    if (live) {
        const int per = (nblocks + 3) / 4, b0 = q * per, b1 = min(nblocks, b0 + per);
        int b = b0;
        for (; b + 8 <= b1; b += 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) v[u] = p[(size_t)(b + u) * stride + e];
#pragma unroll
            for (int u = 0; u < 8; u++) s += (double)v[u];
        }
        for (; b < b1; b++) s += (double)p[(size_t)b * stride + e];
    }
    quarter[q][l] = s;
    __syncthreads();
    if (q != 0 || !live) return;
    const float v = (float)(((quarter[0][l] + quarter[1][l]) + quarter[2][l]) + quarter[3][l]);
    if (second) {
        const int row = e / GSD_WG2_COLS, col = e - row * GSD_WG2_COLS;
        if (row >= 12 * K) return;
        const int m = row < K ? 0 : row < 2 * K ? 1 : row < 5 * K ? 2 : 3;
        const int o = row - (m == 0 ? 0 : m == 1 ? K : m == 2 ? 2 * K : 5 * K);
        if (col < 32) G.g[8 + m][o * 32 + col] = v;
        else G.g[12 + m][o] = v;
    } else {
        const int row = e / GSD_WG1_COLS, col = e - row * GSD_WG1_COLS;
        const int m = row >> 5, j = row & 31;
        if (col < 36) G.g[m][j * 36 + col] = v;
        else if (col == 36) G.g[4 + m][j] = v;
    }
}

// ---- host side ----------------------------------------------------------------------------------------------------
static GsdMlps gsd_pack(const float* const* w)  // w[16] = {w1[4], b1[4], w2[4], b2[4]}
{
    GsdMlps P;
    for (int m = 0; m < 4; m++) { P.w1[m] = w[m]; P.b1[m] = w[4 + m]; P.w2[m] = w[8 + m]; P.b2[m] = w[12 + m]; }
    return P;
}

hipError_t gsd_launch_count(int N, int K, const float* const* weights, const int32_t* vis, const float* feat, const float* anchor,
                            const float* campos, float* neural_opacity, uint8_t* mask, uint8_t* count, uint32_t* first,
                            uint32_t* total, uint32_t* block_scratch, hipStream_t stream)
{
    if (N <= 0) return hipSuccess;
    const int nb = (N + GSD_THREADS - 1) / GSD_THREADS;
    hipLaunchKernelGGL(gsd_count_kernel, dim3(nb), dim3(GSD_THREADS), GSD_LDS_FLOATS(K) * sizeof(float), stream, N, K, gsd_pack(weights), vis, feat, anchor, campos,
                       neural_opacity, mask, count, block_scratch);
    hipLaunchKernelGGL(gsd_scan_kernel, dim3(1), dim3(1024), 0, stream, nb, block_scratch, total);
    hipLaunchKernelGGL(gsd_first_kernel, dim3(nb), dim3(GSD_THREADS), 0, stream, N, count, block_scratch, first);
    return hipGetLastError();
}

hipError_t gsd_launch_emit(int N, int K, const float* const* weights, const int32_t* vis, const float* feat, const float* anchor,
                           const float* offsets, const float* gscale, const float* campos, const float* neural_opacity,
                           const uint8_t* mask, const uint32_t* first, float* xyz, float* color, float* opacity,
                           float* uncertainty, float* scaling, float* rot, hipStream_t stream)
{
    if (N <= 0) return hipSuccess;
    hipLaunchKernelGGL(gsd_emit_kernel, dim3((N + GSD_THREADS - 1) / GSD_THREADS), dim3(GSD_THREADS), GSD_LDS_FLOATS(K) * sizeof(float), stream, N, K,
                       gsd_pack(weights), vis, feat, anchor, offsets, gscale, campos, neural_opacity, mask, first, xyz, color,
                       opacity, uncertainty, scaling, rot);
    return hipGetLastError();
}

hipError_t gsd_launch_backward(int N, int K, const float* const* weights, const int32_t* vis, const float* feat, const float* anchor,
                               const float* offsets, const float* gscale, const float* campos, const uint8_t* mask,
                               const uint32_t* first, const float* g_xyz, const float* g_color, const float* g_opacity,
                               const float* g_unc, const float* g_scaling, const float* g_rot, float* d_feat,
                               float* d_anchor, float* d_offsets, float* d_gscale, float* D2, float* D1, float* H, float* X,
                               hipStream_t stream)
{
    if (N <= 0) return hipSuccess;
    const dim3 grid((N + GSD_THREADS - 1) / GSD_THREADS), block(GSD_THREADS);
    const GsdMlps P = gsd_pack(weights);
#define GSD_BWD(M)                                                                                                          \
    hipLaunchKernelGGL(gsd_backward_mlp_kernel<M>, grid, block, GSD_LDS_FLOATS(K) * sizeof(float), stream, N, K, P, vis, feat, anchor, gscale, campos, mask,  \
                       first, g_color, g_opacity, g_unc, g_scaling, g_rot, d_gscale, D2, D1, H, X)
    GSD_BWD(0); GSD_BWD(1); GSD_BWD(2); GSD_BWD(3);
#undef GSD_BWD
    hipLaunchKernelGGL(gsd_backward_input_kernel, grid, block, 4 * GSD_HID * GSD_IN * sizeof(float), stream, N, K, P, vis, anchor, offsets, gscale, campos, mask,
                       first, g_xyz, D1, d_feat, d_anchor, d_offsets, d_gscale);
    return hipGetLastError();
}

size_t gsd_weight_grad_workspace_bytes() { return (size_t)GSD_WG_BLOCKS * (GSD_WG2_ROWS * GSD_WG2_COLS + GSD_WG1_ROWS * GSD_WG1_COLS) * sizeof(float); }
int gsd_leading_dim(int N) { return gsd_ld(N); }

hipError_t gsd_launch_weight_grads(int N, int K, const float* D2, const float* D1, const float* H, const float* X, void* workspace,
                                   float* const* grads16, hipStream_t stream)
{
    GsdGrads G;
    for (int i = 0; i < 16; i++) G.g[i] = grads16[i];
    float* p2 = (float*)workspace;
    float* p1 = p2 + (size_t)GSD_WG_BLOCKS * GSD_WG2_ROWS * GSD_WG2_COLS;
    // N == 0: the partials are zeros (nothing to add up) -- the kernels still run so that every output is written
    hipLaunchKernelGGL(gsd_weight_grad_kernel<0>, dim3(GSD_WG_BLOCKS), dim3(256), 0, stream, N, K, D2, H, (const float*)nullptr, p2);
    hipLaunchKernelGGL(gsd_weight_grad_kernel<1>, dim3(GSD_WG_BLOCKS), dim3(256), 0, stream, N, K, D1, X, (const float*)nullptr, p1);
    const int total = GSD_WG2_ROWS * GSD_WG2_COLS + GSD_WG1_ROWS * GSD_WG1_COLS;
    hipLaunchKernelGGL(gsd_weight_grad_finish_kernel, dim3((total + 63) / 64), dim3(256), 0, stream, K, GSD_WG_BLOCKS, p2, p1, G);
    return hipGetLastError();
}
