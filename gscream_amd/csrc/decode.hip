// decode.hip -- fused neural-Gaussian decode + compaction (the step right BEFORE the rasterizer).
//
// Replaces the body of GScream's gaussian_renderer/__init__.py:18-102 generate_neural_gaussians after the
// visible-anchor gather (:25-28): view vector / distance (:30-35), the four per-anchor MLPs 36 -> 32 -> {10, 10, 30, 70}
// (scene/gaussian_model.py:118-144: opacity+Tanh, uncertainty+Sigmoid, color+Sigmoid, cov linear), the opacity mask
// (:57-60), the [N*K, 23] concat + boolean-mask compaction (:78-87) and the post-processing (:90-96:
// scaling = grid_scaling[:,3:] * sigmoid(.), rot = normalize(.), xyz = anchor + offset * grid_scaling[:,:3]).
// The torch path materialises ~20 intermediates of up to [N*K, 23] floats; here nothing but the compacted per-Gaussian
// outputs reaches HBM in the forward, and nothing but the gradients themselves in the backward.
//
// Mapping.  The MLPs run on the f32 matrix cores, evaluated transposed: weights = A operand (per-lane operand tables in
// LDS, staged once per workgroup), 16 anchors = the columns of a product, per-anchor data = B operand; the registers a
// product leaves are the next product's B operand as they stand (section "the MLPs on the f32 matrix cores").  fp32 MFMA
// runs at the rate of packed fp32 VALU math and the two do NOT overlap on a SIMD (measured: matrix-core busy + VALU busy
// + memory waits add up to the kernel time whatever the number of waves), so what it buys is operand delivery: the
// thread-per-anchor first version read ~1800 broadcast LDS words per 64 anchors and sat at 30 % of either unit.
//   pass A  gsd_count_kernel : opacity MLP only -> neural_opacity[N*K], mask[N*K], per-anchor survivor count
//   scan    gsd_scan_kernel  : exclusive scan of the counts (one block; N ~ 2e5) -> first output row per anchor, total
//   pass B  gsd_emit_kernel  : the other three MLPs, writes the surviving offsets' rows in the reference's order
//                              (anchor-major, offset-minor = boolean-mask order); persistent waves
//   bwd     gsd_backward_fused_kernel : the whole backward, one wave per SIMD: all four MLPs recomputed, input /
//                              geometry gradients, and the 16 weight / bias gradients accumulated in registers
//           gsd_weight_grad_finish_kernel : adds the workgroup partials of the weight gradients (fixed order, double)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <type_traits>

#include "gsr_common.h"
#include "gsr_math.h"

#define GSD_F 32        // feat_dim (arguments/__init__.py:50)
#define GSD_IN 36       // feat + view(3) + dist(1)
#define GSD_HID 32
#define GSD_MAXK 10     // n_offsets (arguments/__init__.py:51); the kernels handle K <= 10
#define GSD_THREADS 256
struct GsdMlps {  // device pointers; m = 0 opacity (K, tanh), 1 uncertainty (K, sigmoid), 2 color (3K, sigmoid), 3 cov (7K)
    const float* w1[4];  // [32][36] row-major (torch Linear.weight)
    const float* b1[4];  // [32]
    const float* w2[4];  // [out][32]
    const float* b2[4];  // [out]
};

// sigmoid through the hardware exp2 / reciprocal (each within 1 ulp): relative error ~3e-7, far inside the 2e-5 the outputs
// are held to; the IEEE expf + division of the first version cost ~25 VALU instructions per value, 21 values per lane
// and 16-anchor step -- as much SIMD time as the matrix-core products themselves (-DGSD_PRECISE_ACT restores it).
#ifdef GSD_PRECISE_ACT
__device__ __forceinline__ float gsd_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }
__device__ __forceinline__ float gsd_rcp(float x) { return 1.0f / x; }
__device__ __forceinline__ float gsd_tanh(float x) { return tanhf(x); }
#else
__device__ __forceinline__ float gsd_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x)); }
__device__ __forceinline__ float gsd_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float gsd_tanh(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(2.8853900817779268f * x)); }  // backward only: enters as 1 - t^2
#endif

// ---- the MLPs on the f32 matrix cores ----------------------------------------------------------------------------------
// v_mfma_f32_16x16x4_f32 computes D[16x16] += A[16x4] B[4x16]; lane l = (g = l >> 4, a = l & 15) supplies A[row a][k g] and
// B[k g][col a] and holds D[row 4g + r][col a] in register r.  The layers are evaluated TRANSPOSED, out[o][anchor] =
// sum_i W[o][i] in[i][anchor]: the weights are the A operand -- loaded once per wave into registers and kept there for
// every anchor the wave ever sees --, sixteen anchors are the columns, and the B operand is per-anchor data in lane
// (g, anchor a).  The sum over k is order-free, so every step may pair k-group g with ANY input index as long as the A
// operand uses the same map:
//   layer 1, step s < 4: input 4g + s, step 4 <= s < 8: input 12 + 4g + s (lane g holds feat[4g .. 4g+3] and feat[16+4g .. 16+4g+3]:
//   two 16-byte loads, each touching one 64-byte line per anchor), step 8: input 32 + g (view, dist);
//   layer 2, step (jt, r): hidden unit 16 jt + 4g + r -- exactly register r of layer-1 tile jt as the matrix core left it.
// No LDS, no cross-lane traffic between the layers.  The rows of a second-layer tile are ours to choose as well: row
// 4g + r carries component r of Gaussian k = 4q + g (tile q), so that after the product lane (g, a) holds, in the four
// registers of a tile, all components of ONE offset of ITS anchor (opacity / uncertainty: k = 4r + g in one tile).
// fp32 throughout (the matrix-core fp32 path is an exact fma chain); rounding differs from a sequential dot product only
// by the order of the sum.
typedef float gsd_v4 __attribute__((ext_vector_type(4)));
#define GSD_MFMA(A, B, C) __builtin_amdgcn_mfma_f32_16x16x4f32((A), (B), (C), 0, 0, 0)
#define GSD_GROUPS 4  // groups of 16 anchors per 64-anchor unit of a wave
// The A operands live in LDS as a table of per-lane values, entry e of lane l at sw[e * 64 + l] (a wave reads an entry
// with one conflict-free ds_read_b32), staged once per workgroup: registers stay free for several waves per SIMD, and a
// 16-anchor step reads ~200 entries where the thread-per-anchor form needed ~1800 broadcast reads per 64 anchors.
//   first layer of an MLP : 26 entries = w[jt][s] at 9 jt + s (18), bias (the C operand) at 18 + 4 jt + r
//   second-layer tile     : 12 entries = w[4 jt + r] (8), bias at 8 + r
#define GSD_L1_ENTRIES 26
#define GSD_L2_ENTRIES 12
__device__ __forceinline__ float gsd_l1_entry(const float* __restrict__ w1, const float* __restrict__ b1, int e, int g, int a)
{
    if (e < 18) {
        const int jt = e / 9, s = e - 9 * jt;
        return w1[(16 * jt + a) * GSD_IN + (s < 4 ? 4 * g + s : s < 8 ? 12 + 4 * g + s : 32 + g)];
    }
    return b1[16 * ((e - 18) >> 2) + 4 * g + ((e - 18) & 3)];
}
// head = true: the opacity / uncertainty heads, one tile, row 4g' + r' = offset k = 4r' + g' (r' < 3); otherwise tile q of
// a head with `comps` components per offset: row 4g' + r' = component r' of offset k = 4q + g', output k * stride + first + r'
__device__ __forceinline__ float gsd_l2_entry(const float* __restrict__ w2, const float* __restrict__ b2, int K, bool head, int q,
                                              int comps, int stride, int first, int e, int g, int a)
{
    if (e < 8) {
        const int gp = a >> 2, rp = a & 3;
        const int k = head ? 4 * rp + gp : 4 * q + gp;
        const bool used = head ? (rp < 3 && k < K) : (rp < comps && k < K);
        const int o = head ? k : k * stride + first + rp;
        return used ? w2[o * GSD_HID + 16 * (e >> 2) + 4 * g + (e & 3)] : 0.0f;
    }
    const int r = e - 8;
    const int k = head ? 4 * r + g : 4 * q + g;
    const bool used = head ? (r < 3 && k < K) : (r < comps && k < K);
    return used ? b2[head ? k : k * stride + first + r] : 0.0f;
}
// B operands of the first layer for lane (g, a): eight features of its anchor (map above) and component g of (view, dist).  Loading and
// finishing are separate so that a wave can have the next group's loads in flight without touching their results.
struct GsdIn {
    float f[8];
    float v;
};
struct GsdRaw {
    float4 u, w;
    float ax, ay, az;
};
__device__ __forceinline__ void gsd_load_raw(GsdRaw& R, const float* __restrict__ feat, const float* __restrict__ anchor, int ai, int g)
{
    const float4* f4 = reinterpret_cast<const float4*>(feat + (size_t)ai * GSD_F + 4 * g);
    R.u = f4[0]; R.w = f4[4];  // floats 4g .. 4g+3 and 16 + 4g .. 16 + 4g+3: one 64-byte line per anchor and load
    R.ax = anchor[3 * (size_t)ai]; R.ay = anchor[3 * (size_t)ai + 1]; R.az = anchor[3 * (size_t)ai + 2];
}
__device__ __forceinline__ void gsd_finish_in(GsdIn& X, const GsdRaw& R, float cx, float cy, float cz, int g)
{
    X.f[0] = R.u.x; X.f[1] = R.u.y; X.f[2] = R.u.z; X.f[3] = R.u.w; X.f[4] = R.w.x; X.f[5] = R.w.y; X.f[6] = R.w.z; X.f[7] = R.w.w;
    const float vx = R.ax - cx, vy = R.ay - cy, vz = R.az - cz;
    const float dist = sqrtf(vx * vx + vy * vy + vz * vz);
    X.v = g == 0 ? vx / dist : g == 1 ? vy / dist : g == 2 ? vz / dist : dist;
}
__device__ __forceinline__ void gsd_load_in(GsdIn& X, const float* __restrict__ feat, const float* __restrict__ anchor, float cx,
                                            float cy, float cz, int ai, int g)
{
    GsdRaw R;
    gsd_load_raw(R, feat, anchor, ai, g);
    gsd_finish_in(X, R, cx, cy, cz, g);
}
// hidden layer (post-ReLU) of 16 anchors: h[jt][r] = unit 16 jt + 4g + r of the lane's anchor; t = the layer's table (+ lane)
__device__ __forceinline__ void gsd_mfma_l1(const float* t, const GsdIn& X, gsd_v4 (&h)[2])
{
    // the two hidden tiles are independent accumulation chains: stepped together, so that a product never waits for the
    // one before it
    gsd_v4 acc0 = {t[18 * 64], t[19 * 64], t[20 * 64], t[21 * 64]};
    gsd_v4 acc1 = {t[22 * 64], t[23 * 64], t[24 * 64], t[25 * 64]};
#pragma unroll
    for (int s = 0; s < 8; s++) {
        acc0 = GSD_MFMA(t[s * 64], X.f[s], acc0);
        acc1 = GSD_MFMA(t[(9 + s) * 64], X.f[s], acc1);
    }
    acc0 = GSD_MFMA(t[8 * 64], X.v, acc0);
    acc1 = GSD_MFMA(t[17 * 64], X.v, acc1);
#pragma unroll
    for (int r = 0; r < 4; r++) { h[0][r] = fmaxf(acc0[r], 0.0f); h[1][r] = fmaxf(acc1[r], 0.0f); }
}
// NT second-layer tiles (consecutive in the table) of one hidden layer: NT independent chains, stepped together
template <int NT>
__device__ __forceinline__ void gsd_mfma_l2n(const float* t, const gsd_v4 (&h)[2], gsd_v4 (&z)[NT])
{
#pragma unroll
    for (int u = 0; u < NT; u++) z[u] = (gsd_v4){t[(u * GSD_L2_ENTRIES + 8) * 64], t[(u * GSD_L2_ENTRIES + 9) * 64], t[(u * GSD_L2_ENTRIES + 10) * 64], t[(u * GSD_L2_ENTRIES + 11) * 64]};
#pragma unroll
    for (int jt = 0; jt < 2; jt++)
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
            for (int u = 0; u < NT; u++) z[u] = GSD_MFMA(t[(u * GSD_L2_ENTRIES + 4 * jt + r) * 64], h[jt][r], z[u]);
}
__device__ __forceinline__ gsd_v4 gsd_mfma_l2(const float* t, const gsd_v4 (&h)[2])
{
    gsd_v4 acc = {t[8 * 64], t[9 * 64], t[10 * 64], t[11 * 64]};
#pragma unroll
    for (int jt = 0; jt < 2; jt++)
#pragma unroll
        for (int r = 0; r < 4; r++) acc = GSD_MFMA(t[(4 * jt + r) * 64], h[jt][r], acc);
    return acc;
}

// Staging of an operand table: wave W of NW takes entries W, W + NW, ...; W is a template parameter (switch on the wave
// index) and the entry index reaches the callee as a type (std::integral_constant), so that every weight pointer and
// offset is a constant expression and the wave's loads are all in flight before its first LDS write.  (With a run-time
// entry index the MLP pointers are fetched from the argument block by dependent loads: the staging alone took tens of us.)
template <int W, int NW, int TOTAL, typename Entry, int... I>
__device__ __forceinline__ void gsd_stage_part_impl(float* sw, int lane, Entry entry, std::integer_sequence<int, I...>)
{
    // entry(std::integral_constant<int, e>) -> the lane's value of entry e; e is a constant expression inside the callee
    const float v[sizeof...(I)] = { entry(std::integral_constant<int, (W + I * NW < TOTAL ? W + I * NW : W)>{})... };
    ((W + I * NW < TOTAL ? (void)(sw[(W + I * NW) * 64 + lane] = v[I]) : (void)0), ...);
}
template <int W, int NW, int TOTAL, typename Entry>
__device__ __forceinline__ void gsd_stage_part(float* sw, int lane, Entry entry)
{
    gsd_stage_part_impl<W, NW, TOTAL>(sw, lane, entry, std::make_integer_sequence<int, (TOTAL + NW - 1) / NW>{});
}
template <int NW, int TOTAL, typename Entry>
__device__ __forceinline__ void gsd_stage(float* sw, int wave, int lane, Entry entry)
{
    static_assert(NW == 4 || NW == 8, "waves per workgroup");
    switch (wave) {
    case 0: gsd_stage_part<0, NW, TOTAL>(sw, lane, entry); break;
    case 1: gsd_stage_part<1, NW, TOTAL>(sw, lane, entry); break;
    case 2: gsd_stage_part<2, NW, TOTAL>(sw, lane, entry); break;
    case 3: gsd_stage_part<3, NW, TOTAL>(sw, lane, entry); break;
    case 4: if (NW > 4) gsd_stage_part<4 % NW, NW, TOTAL>(sw, lane, entry); break;
    case 5: if (NW > 4) gsd_stage_part<5 % NW, NW, TOTAL>(sw, lane, entry); break;
    case 6: if (NW > 4) gsd_stage_part<6 % NW, NW, TOTAL>(sw, lane, entry); break;
    default: if (NW > 4) gsd_stage_part<7 % NW, NW, TOTAL>(sw, lane, entry); break;
    }
}

// ---- pass A: opacity MLP, mask, count ---------------------------------------------------------------------------
// Workgroups walk 256-anchor blocks (the unit of the scan); wave w of a block takes its anchors 64 w .. 64 w + 63 as four
// groups of 16.
// `vis_count` (optional): the number of valid entries of `vis`, read HERE instead of on the host (the row list was compacted on the
// device, gsd_rows_*): N is then only the upper bound the launch and the buffers were sized for; anchors / blocks past the count get
// count = 0 / block total = 0, so the scans downstream need not know.
__global__ void __launch_bounds__(GSD_THREADS) gsd_count_kernel(int N, int K, GsdMlps P, const int32_t* __restrict__ vis,
                                                                const uint32_t* __restrict__ vis_count,
                                                                const float* __restrict__ feat,
                                                                const float* __restrict__ anchor,
                                                                const float* __restrict__ campos,
                                                                float* __restrict__ neural_opacity,
                                                                uint8_t* __restrict__ mask, uint8_t* __restrict__ count,
                                                                uint32_t* __restrict__ block_sum)
{
    __shared__ uint32_t bs;
    __shared__ float sw[(GSD_L1_ENTRIES + GSD_L2_ENTRIES) * 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, a = lane & 15;
    gsd_stage<GSD_THREADS / 64, GSD_L1_ENTRIES + GSD_L2_ENTRIES>(sw, wave, lane, [&](auto ec) {
        constexpr int e = decltype(ec)::value;
        if constexpr (e < GSD_L1_ENTRIES) return gsd_l1_entry(P.w1[0], P.b1[0], e, g, a);
        else return gsd_l2_entry(P.w2[0], P.b2[0], K, true, 0, 0, 0, 0, e - GSD_L1_ENTRIES, g, a);
    });
    const float* t1 = sw + lane;
    const float* t2 = sw + GSD_L1_ENTRIES * 64 + lane;
    const float cx = campos[0], cy = campos[1], cz = campos[2];
    const int nb = (N + GSD_THREADS - 1) / GSD_THREADS;
    const int Nv = vis_count ? min(N, (int)vis_count[0]) : N;  // rows that exist
    for (int blk = blockIdx.x; blk < nb; blk += gridDim.x) {
        if (blk * GSD_THREADS >= Nv) {  // a block past the row count (block-uniform): nothing survives here
            const int n = blk * GSD_THREADS + (int)threadIdx.x;
            if (n < N) count[n] = 0;
            if (threadIdx.x == 0) block_sum[blk] = 0u;
            continue;
        }
        if (threadIdx.x == 0) bs = 0u;
        __syncthreads();  // (also orders the table writes above against the first reads)
        uint32_t mine = 0;  // survivors among this lane's offsets
#pragma unroll
        for (int t = 0; t < GSD_GROUPS; t++) {
            const int n = blk * GSD_THREADS + wave * 64 + t * 16 + a;
            const bool live = n < Nv;
            const int nn = live ? n : Nv - 1;
            GsdIn X;
            gsd_load_in(X, feat, anchor, cx, cy, cz, vis ? vis[nn] : nn, g);
            gsd_v4 h[2];
            gsd_mfma_l1(t1, X, h);
            const gsd_v4 z = gsd_mfma_l2(t2, h);
            uint32_t c = 0;
#pragma unroll
            for (int r = 0; r < 3; r++) {
                const int k = 4 * r + g;
                if (live && k < K) {
                    const float op = tanhf(z[r]);
                    const bool keep = op > 0.0f;  // gaussian_renderer/__init__.py:59
                    neural_opacity[(size_t)n * K + k] = op;
                    mask[(size_t)n * K + k] = keep ? 1 : 0;
                    c += keep ? 1u : 0u;
                }
            }
            mine += c;
            c += (uint32_t)__shfl_xor((int)c, 16, 64);
            c += (uint32_t)__shfl_xor((int)c, 32, 64);
            if (g == 0 && n < N) count[n] = live ? (uint8_t)c : (uint8_t)0;
        }
        // survivors of this block of 256 anchors: the scan below only has to cover N/256 block totals
        const uint32_t ws = gsr_wave_scan_add(mine);
        if (lane == 63 && ws != 0u) atomicAdd(&bs, ws);
        __syncthreads();
        if (threadIdx.x == 0) block_sum[blk] = bs;
    }
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
// Uncertainty, colour and covariance MLPs of 16 anchors per step on the matrix cores (layout above); lane (g, a) then owns
// the offsets k = g, 4 + g, 8 + g of its anchor: activations, geometry and the stores of their output rows.  A wave's
// unit of work is 64 consecutive anchors (four steps); 8 waves share one operand table.
#define GSD_EMIT_THREADS 512
#define GSD_EMIT_ENTRIES (3 * GSD_L1_ENTRIES + 10 * GSD_L2_ENTRIES)  // tiles: uncertainty | colour q = 0..2 | scale q | rotation q
__global__ void __launch_bounds__(GSD_EMIT_THREADS) gsd_emit_kernel(
    int N, int K, GsdMlps P, const int32_t* __restrict__ vis, const uint32_t* __restrict__ vis_count, const float* __restrict__ feat,
    const float* __restrict__ anchor, const float* __restrict__ offsets /*[N,K,3]*/, const float* __restrict__ gscale /*[N,6]*/,
    const float* __restrict__ campos, const float* __restrict__ neural_opacity, const uint8_t* __restrict__ mask,
    const uint32_t* __restrict__ first, float* __restrict__ xyz, float* __restrict__ color, float* __restrict__ opacity, float* __restrict__ uncertainty,
    float* __restrict__ scaling, float* __restrict__ rot)
{
    __shared__ float sw[GSD_EMIT_ENTRIES * 64];
    if (vis_count) N = min(N, (int)vis_count[0]);  // the row count, read on the device (see gsd_count_kernel)
    if (N <= 0) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, a = lane & 15;
    gsd_stage<GSD_EMIT_THREADS / 64, GSD_EMIT_ENTRIES>(sw, wave, lane, [&](auto ec) {
        constexpr int e = decltype(ec)::value;
        if constexpr (e < 3 * GSD_L1_ENTRIES) {
            constexpr int m = e / GSD_L1_ENTRIES;
            return gsd_l1_entry(P.w1[m + 1], P.b1[m + 1], e - m * GSD_L1_ENTRIES, g, a);
        } else {
            constexpr int t = (e - 3 * GSD_L1_ENTRIES) / GSD_L2_ENTRIES, j = (e - 3 * GSD_L1_ENTRIES) - t * GSD_L2_ENTRIES;
            if constexpr (t == 0) return gsd_l2_entry(P.w2[1], P.b2[1], K, true, 0, 0, 0, 0, j, g, a);
            else if constexpr (t < 4) return gsd_l2_entry(P.w2[2], P.b2[2], K, false, t - 1, 3, 3, 0, j, g, a);
            else if constexpr (t < 7) return gsd_l2_entry(P.w2[3], P.b2[3], K, false, t - 4, 3, 7, 0, j, g, a);
            else return gsd_l2_entry(P.w2[3], P.b2[3], K, false, t - 7, 4, 7, 3, j, g, a);
        }
    });
    __syncthreads();
    const float* tl1 = sw + lane;
    const float* tl2 = sw + 3 * GSD_L1_ENTRIES * 64 + lane;
    const float cx = campos[0], cy = campos[1], cz = campos[2];
    // A wave walks groups of 16 anchors, group index = first + i * stride, with the NEXT group's inputs in flight while the
    // current one is computed (the gather index two groups ahead): a wave's life is one chain of dependent round trips
    // otherwise, and there are too few waves per SIMD to hide them behind each other.
    struct In {  // raw loads only: nothing here is looked at before the group is computed
        GsdRaw R;
        float gs[6], nop[3], of[9];
        uint32_t row0;
    };
    const int groups = (N + 15) / 16, stride = gridDim.x * (GSD_EMIT_THREADS / 64);
    auto anchor_row = [&](int grp) {  // (group index clamped: the loads of a group past the end are issued and never used)
        const int n = min(grp, groups - 1) * 16 + a;
        const int nn = n < N ? n : N - 1;
        return vis ? vis[nn] : nn;  // visible-anchor gather (gaussian_renderer/__init__.py:25-28) folded in
    };
    auto load = [&](In& I, int grp, int ai) {
        const int n = min(grp, groups - 1) * 16 + a;
        const int nn = n < N ? n : N - 1;
        gsd_load_raw(I.R, feat, anchor, ai, g);
#pragma unroll
        for (int i = 0; i < 6; i++) I.gs[i] = gscale[6 * (size_t)ai + i];
        I.row0 = first[nn];
#pragma unroll
        for (int q = 0; q < 3; q++) {
            const int k = min(4 * q + g, K - 1);  // (clamped: the duplicate of a lane without a third offset is never used)
            I.nop[q] = neural_opacity[(size_t)nn * K + k];
            const float* of = offsets + ((size_t)ai * K + k) * 3;
            I.of[3 * q] = of[0]; I.of[3 * q + 1] = of[1]; I.of[3 * q + 2] = of[2];
        }
    };
    auto compute = [&](const In& I, int grp) {
        const bool live = grp * 16 + a < N;
        GsdIn X;
        gsd_finish_in(X, I.R, cx, cy, cz, g);
        uint32_t keep = 0;  // the lane's own offsets, then OR-ed over the anchor's four lanes
#pragma unroll
        for (int q = 0; q < 3; q++)
            if (live && 4 * q + g < K && I.nop[q] > 0.0f) keep |= 1u << (4 * q + g);  // = mask of pass A (:59)
        keep |= (uint32_t)__shfl_xor((int)keep, 16, 64);
        keep |= (uint32_t)__shfl_xor((int)keep, 32, 64);
        uint32_t row[3];
        bool on[3];
#pragma unroll
        for (int q = 0; q < 3; q++) {
            const int k = 4 * q + g;
            on[q] = k < K && ((keep >> k) & 1u);
            row[q] = I.row0 + (uint32_t)__popc(keep & ((1u << k) - 1u));
        }
        // The products and the activations of all three heads run branch-free (one basic block: the scheduler can put the
        // activations of one tile into the shadow of the next tile's matrix-core steps); the stores of the lane's up to
        // three surviving rows follow at the end.
        gsd_v4 h[2];
        float unc[3], col[3][3], scl[3][3];
        float4 rt[3];
        gsd_mfma_l1(tl1, X, h);
        {
            const gsd_v4 z = gsd_mfma_l2(tl2, h);  // register r = offset 4r + g: offset 4q + g sits in register q
#pragma unroll
            for (int q = 0; q < 3; q++) unc[q] = gsd_sigmoid(z[q]);
        }
        gsd_mfma_l1(tl1 + GSD_L1_ENTRIES * 64, X, h);
        {
            gsd_v4 zc[3];
            gsd_mfma_l2n<3>(tl2 + GSD_L2_ENTRIES * 64, h, zc);
#pragma unroll
            for (int q = 0; q < 3; q++)
#pragma unroll
                for (int c = 0; c < 3; c++) col[q][c] = gsd_sigmoid(zc[q][c]);
        }
        gsd_mfma_l1(tl1 + 2 * GSD_L1_ENTRIES * 64, X, h);
        gsd_v4 zv[6];  // scale q = 0..2, rotation q = 0..2
        gsd_mfma_l2n<6>(tl2 + 4 * GSD_L2_ENTRIES * 64, h, zv);
#pragma unroll
        for (int q = 0; q < 3; q++) {
            const gsd_v4 zs = zv[q], zr = zv[3 + q];
#pragma unroll
            for (int c = 0; c < 3; c++) scl[q][c] = I.gs[3 + c] * gsd_sigmoid(zs[c]);  // :90
            const float nrm = fmaxf(sqrtf(zr[0] * zr[0] + zr[1] * zr[1] + zr[2] * zr[2] + zr[3] * zr[3]), 1e-12f);  // F.normalize
            const float rn = gsd_rcp(nrm);
            rt[q] = make_float4(zr[0] * rn, zr[1] * rn, zr[2] * rn, zr[3] * rn);  // :91
        }
#pragma unroll
        for (int q = 0; q < 3; q++) {
            if (!on[q]) continue;
            const size_t r = row[q];
            opacity[r] = I.nop[q];  // = neural_opacity[mask] (:63), bit for bit
            xyz[3 * r] = I.R.ax + I.of[3 * q] * I.gs[0];       // :94-95
            xyz[3 * r + 1] = I.R.ay + I.of[3 * q + 1] * I.gs[1];
            xyz[3 * r + 2] = I.R.az + I.of[3 * q + 2] * I.gs[2];
            uncertainty[r] = unc[q];
#pragma unroll
            for (int c = 0; c < 3; c++) { color[3 * r + c] = col[q][c]; scaling[3 * r + c] = scl[q][c]; }
            reinterpret_cast<float4*>(rot)[r] = rt[q];
        }
    };
    // Two input buffers in alternation: while a group is computed, the loads of the wave's next group are in flight into
    // the other buffer (and the gather index of the one after that), untouched until their turn.
    int gi = blockIdx.x * (GSD_EMIT_THREADS / 64) + wave;
    if (gi >= groups) return;
    In A, B;
    int aiA = anchor_row(gi), aiB = anchor_row(gi + stride);
    load(A, gi, aiA);
    aiA = anchor_row(gi + 2 * stride);
    for (;;) {
        load(B, gi + stride, aiB);
        aiB = anchor_row(gi + 3 * stride);
        compute(A, gi);
        gi += stride;
        if (gi >= groups) break;
        load(A, gi + stride, aiA);
        aiA = anchor_row(gi + 3 * stride);
        compute(B, gi);
        gi += stride;
        if (gi >= groups) break;
    }
}

// ---- backward -------------------------------------------------------------------------------------------------------
// One persistent kernel (gsd_backward_fused_kernel below) + a small finishing kernel that adds the workgroup partials of
// the 16 weight / bias gradients in workgroup order (double): bit-reproducible.  Partials: gw2 | gb2 as [12 K rows (opacity K
// | uncertainty K | colour 3K | cov 7K)][32 hidden + bias], gw1 | gb1 as [4 x 32 rows][36 inputs + bias (the ones row) + pad].
typedef float gsd_f4 __attribute__((ext_vector_type(4)));
#define GSD_WG_BLOCKS 512
#define GSD_WG2_ROWS (12 * GSD_MAXK)   // delta rows of the second layers (K | K | 3K | 7K)
#define GSD_WG2_COLS 33                // 32 hidden + bias
#define GSD_WG1_ROWS 128
#define GSD_WG1_COLS 48                // 36 inputs + ones row (bias) + padding of the third tile

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

// ---- the whole backward in one persistent kernel ---------------------------------------------------------------------------
// One wave per SIMD walks groups of 16 anchors through all four MLPs and keeps EVERYTHING it accumulates in registers:
// the input gradients of the group (W1^T d(pre1) over the four MLPs) and, over all of its groups, the 16 weight / bias
// gradients.  Nothing per-anchor goes to HBM but the outputs themselves (no delta / activation arrays, no second pass).
// The weight gradients need the ANCHORS as the reduction dimension of the matrix-core product, gW[o][j] = sum_a dz[o][a]
// h[j][a], while the layer products leave anchors along the columns: a 16 x 16 tile is turned through a per-wave LDS
// buffer (row stride 20 floats; written as four rows per lane group, read back as 16 bytes = 4 anchors per lane, one
// register per k-step) -- for the deltas as the A operand, for the hidden layer / the input as the B operand.
//   table (entries of 64 lane values): first layers 4 x 26 | second-layer tiles 11 x 12 (opacity, uncertainty, colour q,
//   scale q, rotation q) | W2^T operands 36 (tile, component) pairs x 2 hidden tiles | W1^T operands 4 x 3 input tiles x 8
#define GSD_FB_THREADS 256
#define GSD_FB_L2 (4 * GSD_L1_ENTRIES)
#define GSD_FB_W2T (GSD_FB_L2 + 11 * GSD_L2_ENTRIES)
#define GSD_FB_W1T (GSD_FB_W2T + 2 * 36)
#define GSD_FB_ENTRIES (GSD_FB_W1T + 4 * 24)
#define GSD_TS 20                                   // row stride of the transposition buffers
#define GSD_FB_WAVEBUF ((48 + 32 + 16) * GSD_TS)    // per wave: input rows | hidden rows | one 16-row scratch tile
#define GSD_FB_LDS_FLOATS (GSD_FB_ENTRIES * 64 + (GSD_FB_THREADS / 64) * GSD_FB_WAVEBUF)
struct GsdTile { int mlp, head, q, comps, stride, first, pair0; };
__host__ __device__ constexpr GsdTile gsd_tile(int t)
{
    return t == 0 ? GsdTile{0, 1, 0, 3, 1, 0, 0}
         : t == 1 ? GsdTile{1, 1, 0, 3, 1, 0, 3}
         : t < 5  ? GsdTile{2, 0, t - 2, 3, 3, 0, 6 + 3 * (t - 2)}
         : t < 8  ? GsdTile{3, 0, t - 5, 3, 7, 0, 15 + 3 * (t - 5)}
                  : GsdTile{3, 0, t - 8, 4, 7, 3, 24 + 4 * (t - 8)};
}
__host__ __device__ constexpr int gsd_pair_tile(int pr)  // tile of (tile, component) pair pr
{
    int t = 0;
    for (int u = 1; u < 11; u++)
        if (pr >= gsd_tile(u).pair0) t = u;
    return t;
}
__host__ __device__ constexpr int gsd_tile0(int m) { return m == 0 ? 0 : m == 1 ? 1 : m == 2 ? 2 : 5; }   // first tile of MLP m
__host__ __device__ constexpr int gsd_ntiles(int m) { return m < 2 ? 1 : m == 2 ? 3 : 6; }
// row of the D2 numbering (opacity K | uncertainty K | colour 3K | cov 7K) that tile t carries in row 4g + r, -1 if none
__device__ __forceinline__ int gsd_tile_row(int t, int g, int r, int K)
{
    const GsdTile T = gsd_tile(t);
    const int base = T.mlp == 0 ? 0 : T.mlp == 1 ? K : T.mlp == 2 ? 2 * K : 5 * K;
    const int k = T.head ? 4 * r + g : 4 * T.q + g;
    if (r >= T.comps || k >= K) return -1;
    return base + (T.head ? k : k * T.stride + T.first + r);
}
__device__ __forceinline__ float gsd_comp4(const float4& v, int e) { return e == 0 ? v.x : e == 1 ? v.y : e == 2 ? v.z : v.w; }

__global__ void __launch_bounds__(GSD_FB_THREADS) __attribute__((amdgpu_waves_per_eu(1, 1))) gsd_backward_fused_kernel(
    int N, int K, GsdMlps P, const int32_t* __restrict__ vis, const float* __restrict__ feat, const float* __restrict__ anchor,
    const float* __restrict__ offsets, const float* __restrict__ gscale, const float* __restrict__ campos,
    const uint8_t* __restrict__ mask, const uint32_t* __restrict__ first, const float* __restrict__ g_xyz,
    const float* __restrict__ g_color, const float* __restrict__ g_opacity, const float* __restrict__ g_unc,
    const float* __restrict__ g_scaling, const float* __restrict__ g_rot, float* __restrict__ d_feat,
    float* __restrict__ d_anchor, float* __restrict__ d_offsets, float* __restrict__ d_gscale, float* __restrict__ partial2,
    float* __restrict__ partial1)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const sw = smem;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, a = lane & 15;
    float* const xT = smem + GSD_FB_ENTRIES * 64 + wave * GSD_FB_WAVEBUF;
    float* const hT = xT + 48 * GSD_TS;
    float* const sT = hT + 32 * GSD_TS;
    gsd_stage<GSD_FB_THREADS / 64, GSD_FB_ENTRIES>(sw, wave, lane, [&](auto ec) {
        constexpr int e = decltype(ec)::value;
        if constexpr (e < GSD_FB_L2) {
            constexpr int m = e / GSD_L1_ENTRIES;
            return gsd_l1_entry(P.w1[m], P.b1[m], e - m * GSD_L1_ENTRIES, g, a);
        } else if constexpr (e < GSD_FB_W2T) {
            constexpr int t = (e - GSD_FB_L2) / GSD_L2_ENTRIES, j = (e - GSD_FB_L2) - t * GSD_L2_ENTRIES;
            constexpr GsdTile T = gsd_tile(t);
            return gsd_l2_entry(P.w2[T.mlp], P.b2[T.mlp], K, T.head, T.q, T.comps, T.stride, T.first, j, g, a);
        } else if constexpr (e < GSD_FB_W1T) {
            // W2^T operand of k-step (tile t, component r), hidden tile jt: W2[rho(t, 4g + r)][16 jt + a]
            constexpr int pr = (e - GSD_FB_W2T) >> 1, jt = (e - GSD_FB_W2T) & 1;
            constexpr int t = gsd_pair_tile(pr);
            constexpr GsdTile T = gsd_tile(t);
            constexpr int r = pr - T.pair0;
            const int k = T.head ? 4 * r + g : 4 * T.q + g;
            const int o = T.head ? k : k * T.stride + T.first + r;
            return k < K ? P.w2[T.mlp][o * GSD_HID + 16 * jt + a] : 0.0f;
        } else {
            // W1^T operand of MLP m, input tile it, k-step (jt, r): W1_m[16 jt + 4g + r][16 it + a]
            constexpr int m = (e - GSD_FB_W1T) / 24, it = ((e - GSD_FB_W1T) - 24 * m) >> 3, ks = (e - GSD_FB_W1T) & 7;
            const int i = 16 * it + a;
            return i < GSD_IN ? P.w1[m][(16 * (ks >> 2) + 4 * g + (ks & 3)) * GSD_IN + i] : 0.0f;
        }
    });
    for (int i = lane; i < 48 * GSD_TS; i += 64) xT[i] = i / GSD_TS == 36 ? 1.f : 0.f;  // input rows 37..47 stay zero; row 36 = ones: the bias column of the first-layer gradients
    __syncthreads();
    const float* const tl = sw + lane;
    const float cx = campos[0], cy = campos[1], cz = campos[2];

    gsd_v4 gW2[11][2], gW1[4][2][3];
    float bs2[11][4];
#pragma unroll
    for (int t = 0; t < 11; t++) {
        gW2[t][0] = gW2[t][1] = (gsd_v4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 4; r++) bs2[t][r] = 0.f;
    }
#pragma unroll
    for (int m = 0; m < 4; m++)
#pragma unroll
        for (int jt = 0; jt < 2; jt++) {
#pragma unroll
            for (int it = 0; it < 3; it++) gW1[m][jt][it] = (gsd_v4){0.f, 0.f, 0.f, 0.f};
        }

    // A wave has its SIMD to itself (the 16 gradients fill the register file), so nothing hides a load behind another wave:
    // the per-anchor inputs of the NEXT group are requested while the current one is computed (gather index two groups
    // ahead), the upstream gradients of the current group -- their rows depend on its mask -- right at its start, and
    // all four first layers run before the first delta is needed.
    struct In {  // raw loads only: nothing here is looked at before the group's turn
        GsdRaw R;
        float gs[6], of[9];
        uint32_t row0;
        uint8_t mk[3];
    };
    const int groups = (N + 15) / 16, stride = gridDim.x * (GSD_FB_THREADS / 64);
    auto anchor_row = [&](int grp) {  // (group index clamped: the loads of a group past the end are issued and never used)
        const int n = min(grp, groups - 1) * 16 + a;
        const int nn = n < N ? n : N - 1;
        return vis ? vis[nn] : nn;
    };
    auto load = [&](In& I, int grp, int ai) {
        const int n = min(grp, groups - 1) * 16 + a;
        const int nn = n < N ? n : N - 1;
        gsd_load_raw(I.R, feat, anchor, ai, g);
#pragma unroll
        for (int c = 0; c < 6; c++) I.gs[c] = gscale[6 * (size_t)ai + c];
        I.row0 = first[nn];
#pragma unroll
        for (int q = 0; q < 3; q++) {
            const int k = min(4 * q + g, K - 1);  // (clamped: the duplicate of a lane without a third offset is never used)
            I.mk[q] = mask[(size_t)nn * K + k];
            const float* of = offsets + ((size_t)ai * K + k) * 3;
            I.of[3 * q] = of[0]; I.of[3 * q + 1] = of[1]; I.of[3 * q + 2] = of[2];
        }
    };
    int grp = blockIdx.x * (GSD_FB_THREADS / 64) + wave;
    In cur;
    int ai_cur = 0, ai_nxt = 0;
    if (grp < groups) {
        ai_cur = anchor_row(grp);
        load(cur, grp, ai_cur);
        ai_nxt = anchor_row(grp + stride);
    }
#pragma unroll 1
    for (; grp < groups; grp += stride) {
        const int n = grp * 16 + a;
        const bool live = n < N;
        const int ai = ai_cur;
        const In I = cur;
        uint32_t keep = 0;  // the lane's own offsets, then OR-ed over the anchor's four lanes
#pragma unroll
        for (int q = 0; q < 3; q++) {
            const int k = 4 * q + g;
            if (live && k < K && I.mk[q]) keep |= 1u << k;
        }
        keep |= (uint32_t)__shfl_xor((int)keep, 16, 64);
        keep |= (uint32_t)__shfl_xor((int)keep, 32, 64);
        size_t row[3];
        bool on[3];
#pragma unroll
        for (int q = 0; q < 3; q++) {
            const int k = 4 * q + g;
            on[q] = k < K && ((keep >> k) & 1u);
            row[q] = I.row0 + (uint32_t)__popc(keep & ((1u << k) - 1u));
        }
        // upstream gradients of the lane's rows (zeros where the offset did not survive): the covariance head's now -- it
        // runs first --, the others when it is done
        float up_o[3], up_u[3], up_c[3][3], up_s[3][3], up_x[3][3];
        float4 up_r[3];
#pragma unroll
        for (int q = 0; q < 3; q++) {
            up_r[q] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int c = 0; c < 3; c++) up_s[q][c] = 0.f;
            if (on[q]) {  // (a NULL upstream tensor = no gradient flows into that output: zeros, no fill kernel on the host side)
                if (g_rot) up_r[q] = reinterpret_cast<const float4*>(g_rot)[row[q]];
                if (g_scaling) {
#pragma unroll
                    for (int c = 0; c < 3; c++) up_s[q][c] = g_scaling[3 * row[q] + c];
                }
            }
        }
        auto load_upstream_rest = [&]() {
#pragma unroll
            for (int q = 0; q < 3; q++) {
                up_o[q] = 0.f; up_u[q] = 0.f;
#pragma unroll
                for (int c = 0; c < 3; c++) { up_c[q][c] = 0.f; up_x[q][c] = 0.f; }
                if (on[q]) {
                    const size_t r = row[q];
                    if (g_opacity) up_o[q] = g_opacity[r];
                    if (g_unc) up_u[q] = g_unc[r];
#pragma unroll
                    for (int c = 0; c < 3; c++) {
                        if (g_color) up_c[q][c] = g_color[3 * r + c];
                        if (g_xyz) up_x[q][c] = g_xyz[3 * r + c];
                    }
                }
            }
        };
        const float gs[6] = { I.gs[0], I.gs[1], I.gs[2], I.gs[3], I.gs[4], I.gs[5] };
        const GsdRaw R = I.R;

        GsdIn X;
        gsd_finish_in(X, R, cx, cy, cz, g);
        // the input tile, turned: row = input index, 16 anchors along the row
#pragma unroll
        for (int s = 0; s < 8; s++) xT[(s < 4 ? 4 * g + s : 12 + 4 * g + s) * GSD_TS + a] = X.f[s];
        xT[(32 + g) * GSD_TS + a] = X.v;
        float4 XB[3];
#pragma unroll
        for (int it = 0; it < 3; it++) XB[it] = *reinterpret_cast<const float4*>(xT + (16 * it + a) * GSD_TS + 4 * g);

        gsd_v4 dx[3] = { {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f} };
        float dgs3[3] = { 0.f, 0.f, 0.f };
        auto head = [&](auto mc) {
            constexpr int m = decltype(mc)::value;
            float up[3][7];  // this head's upstream gradients
#pragma unroll
            for (int q = 0; q < 3; q++) {
#pragma unroll
                for (int c = 0; c < 7; c++) up[q][c] = 0.f;
                if (m == 0) up[q][0] = up_o[q];
                if (m == 1) up[q][0] = up_u[q];
                if (m == 2) { up[q][0] = up_c[q][0]; up[q][1] = up_c[q][1]; up[q][2] = up_c[q][2]; }
                if (m == 3) {
                    up[q][0] = up_s[q][0]; up[q][1] = up_s[q][1]; up[q][2] = up_s[q][2];
                    up[q][3] = up_r[q].x; up[q][4] = up_r[q].y; up[q][5] = up_r[q].z; up[q][6] = up_r[q].w;
                }
            }
            gsd_v4 h[2];
            gsd_mfma_l1(tl + m * GSD_L1_ENTRIES * 64, X, h);
            float4 HB[2];
#pragma unroll
            for (int jt = 0; jt < 2; jt++)
#pragma unroll
                for (int r = 0; r < 4; r++) hT[(16 * jt + 4 * g + r) * GSD_TS + a] = h[jt][r];
#pragma unroll
            for (int jt = 0; jt < 2; jt++) HB[jt] = *reinterpret_cast<const float4*>(hT + (16 * jt + a) * GSD_TS + 4 * g);

            gsd_v4 dh[2] = { {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f} };
            // one second-layer tile: deltas dz -> weight gradients (turned through sT), d(hidden), bias sums
            auto tile_back = [&](const int t, const gsd_v4 dz) {
                const GsdTile T = gsd_tile(t);
#pragma unroll
                for (int r = 0; r < 4; r++) sT[(4 * g + r) * GSD_TS + a] = dz[r];
                const float4 A4 = *reinterpret_cast<const float4*>(sT + a * GSD_TS + 4 * g);
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    gW2[t][0] = GSD_MFMA(gsd_comp4(A4, e), gsd_comp4(HB[0], e), gW2[t][0]);
                    gW2[t][1] = GSD_MFMA(gsd_comp4(A4, e), gsd_comp4(HB[1], e), gW2[t][1]);
                }
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    if (r < T.comps) {
                        dh[0] = GSD_MFMA(tl[(GSD_FB_W2T + 2 * (T.pair0 + r)) * 64], dz[r], dh[0]);
                        dh[1] = GSD_MFMA(tl[(GSD_FB_W2T + 2 * (T.pair0 + r) + 1) * 64], dz[r], dh[1]);
                        bs2[t][r] += dz[r];
                    }
                }
            };
            const float* tl2 = tl + GSD_FB_L2 * 64;
            if (m < 2) {
                const gsd_v4 z = gsd_mfma_l2(tl2 + m * GSD_L2_ENTRIES * 64, h);  // register r = offset 4r + g
                gsd_v4 dz;
#pragma unroll
                for (int q = 0; q < 3; q++) {
                    float d = 0.f;
                    if (m == 0) { const float t = gsd_tanh(z[q]); d = up[q][0] * (1.0f - t * t); }       // opacity = tanh(z)
                    else { const float sg = gsd_sigmoid(z[q]); d = up[q][0] * sg * (1.0f - sg); }   // sigmoid
                    dz[q] = on[q] ? d : 0.f;
                }
                dz[3] = 0.f;
                tile_back(m, dz);
            } else if (m == 2) {
                gsd_v4 zc[3];
                gsd_mfma_l2n<3>(tl2 + 2 * GSD_L2_ENTRIES * 64, h, zc);
#pragma unroll
                for (int q = 0; q < 3; q++) {
                    const gsd_v4 z = zc[q];
                    gsd_v4 dz;
#pragma unroll
                    for (int c = 0; c < 3; c++) {
                        const float sg = gsd_sigmoid(z[c]);
                        dz[c] = on[q] ? up[q][c] * sg * (1.0f - sg) : 0.f;
                    }
                    dz[3] = 0.f;
                    tile_back(2 + q, dz);
                }
            } else {
                gsd_v4 zv[6];  // scale q = 0..2, rotation q = 0..2
                gsd_mfma_l2n<6>(tl2 + 5 * GSD_L2_ENTRIES * 64, h, zv);
#pragma unroll
                for (int q = 0; q < 3; q++) {
                    const gsd_v4 zs = zv[q], zr = zv[3 + q];
                    gsd_v4 dzs, dzr;
#pragma unroll
                    for (int c = 0; c < 3; c++) {  // scaling = gs[3+c] * sigmoid(z)
                        const float sg = gsd_sigmoid(zs[c]);
                        dzs[c] = on[q] ? up[q][c] * gs[3 + c] * sg * (1.0f - sg) : 0.f;
                        dgs3[c] += on[q] ? up[q][c] * sg : 0.f;
                    }
                    dzs[3] = 0.f;
                    // rot = v / max(|v|, eps): d v = (g - rot (rot . g)) / |v|
                    const float nrm = fmaxf(sqrtf(zr[0] * zr[0] + zr[1] * zr[1] + zr[2] * zr[2] + zr[3] * zr[3]), 1e-12f);
                    float rt[4], dot = 0.f;
                    const float rn = gsd_rcp(nrm);
#pragma unroll
                    for (int c = 0; c < 4; c++) { rt[c] = zr[c] * rn; dot += up[q][3 + c] * rt[c]; }
#pragma unroll
                    for (int c = 0; c < 4; c++) dzr[c] = on[q] ? (up[q][3 + c] - rt[c] * dot) * rn : 0.f;
                    tile_back(5 + q, dzs);
                    tile_back(8 + q, dzr);
                }
            }
            // through the ReLU; first-layer weight gradients, input gradients, bias sums
#pragma unroll
            for (int jt = 0; jt < 2; jt++) {
                gsd_v4 d1;
#pragma unroll
                for (int r = 0; r < 4; r++) d1[r] = h[jt][r] > 0.0f ? dh[jt][r] : 0.0f;
#pragma unroll
                for (int r = 0; r < 4; r++) sT[(4 * g + r) * GSD_TS + a] = d1[r];
                const float4 A4 = *reinterpret_cast<const float4*>(sT + a * GSD_TS + 4 * g);
#pragma unroll
                for (int e = 0; e < 4; e++)
#pragma unroll
                    for (int it = 0; it < 3; it++) gW1[m][jt][it] = GSD_MFMA(gsd_comp4(A4, e), gsd_comp4(XB[it], e), gW1[m][jt][it]);
#pragma unroll
                for (int r = 0; r < 4; r++) {
#pragma unroll
                    for (int it = 0; it < 3; it++)
                        dx[it] = GSD_MFMA(tl[(GSD_FB_W1T + 24 * m + 8 * it + 4 * jt + r) * 64], d1[r], dx[it]);
                }
            }
        };
        head(std::integral_constant<int, 3>{});
        // the next group's inputs (the current ones are half done with)
        load(cur, grp + stride, ai_nxt);
        ai_cur = ai_nxt;
        ai_nxt = anchor_row(grp + 2 * stride);
        load_upstream_rest();
        head(std::integral_constant<int, 2>{});
        head(std::integral_constant<int, 1>{});
        head(std::integral_constant<int, 0>{});
        // geometry of the lane's offsets: xyz = anchor + offset * gs[0:3]
        float da[3] = { 0.f, 0.f, 0.f }, dgs[3] = { 0.f, 0.f, 0.f };
#pragma unroll
        for (int q = 0; q < 3; q++) {
            const int k = 4 * q + g;
            if (!live || k >= K) continue;
            float* dof = d_offsets + ((size_t)ai * K + k) * 3;
#pragma unroll
            for (int c = 0; c < 3; c++) {  // (up_x is zero where the offset did not survive)
                da[c] += up_x[q][c]; dof[c] = up_x[q][c] * gs[c]; dgs[c] += up_x[q][c] * I.of[3 * q + c];
            }
        }
        // per-anchor outputs
#pragma unroll
        for (int c = 0; c < 3; c++) {
            da[c] += __shfl_xor(da[c], 16, 64); da[c] += __shfl_xor(da[c], 32, 64);
            dgs[c] += __shfl_xor(dgs[c], 16, 64); dgs[c] += __shfl_xor(dgs[c], 32, 64);
            dgs3[c] += __shfl_xor(dgs3[c], 16, 64); dgs3[c] += __shfl_xor(dgs3[c], 32, 64);
        }
        if (live) {
            float4* df = reinterpret_cast<float4*>(d_feat + (size_t)ai * GSD_F + 4 * g);
            df[0] = make_float4(dx[0][0], dx[0][1], dx[0][2], dx[0][3]);
            df[4] = make_float4(dx[1][0], dx[1][1], dx[1][2], dx[1][3]);
            if (g == 0) {
                // view vector / distance back to the anchor (v = a - c, dist = |v|, view = v / dist); dx[2] = d(view, dist)
                const float vx = R.ax - cx, vy = R.ay - cy, vz = R.az - cz;
                const float dist = sqrtf(vx * vx + vy * vy + vz * vz);
                const float ux = vx / dist, uy = vy / dist, uz = vz / dist;
                const float gdot = dx[2][0] * ux + dx[2][1] * uy + dx[2][2] * uz;
                d_anchor[3 * (size_t)ai] = da[0] + (dx[2][0] - ux * gdot) / dist + dx[2][3] * ux;
                d_anchor[3 * (size_t)ai + 1] = da[1] + (dx[2][1] - uy * gdot) / dist + dx[2][3] * uy;
                d_anchor[3 * (size_t)ai + 2] = da[2] + (dx[2][2] - uz * gdot) / dist + dx[2][3] * uz;
#pragma unroll
                for (int c = 0; c < 3; c++) { d_gscale[6 * (size_t)ai + c] = dgs[c]; d_gscale[6 * (size_t)ai + 3 + c] = dgs3[c]; }
            }
        }
    }

    // ---- the wave's weight gradients -> workgroup partial (LDS, in wave order: bit-reproducible) -> workspace ----
    // bias sums over the 16 anchors of a lane group (the lanes of one DPP row)
#pragma unroll
    for (int t = 0; t < 11; t++)
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
            for (int d = 1; d < 16; d <<= 1) bs2[t][r] += __shfl_xor(bs2[t][r], d, 64);
    __syncthreads();  // every wave is done with the operand table: the partials take its place
    float* const red2 = smem;
    float* const red1 = smem + GSD_WG2_ROWS * GSD_WG2_COLS;
    for (int i = threadIdx.x; i < GSD_WG2_ROWS * GSD_WG2_COLS + GSD_WG1_ROWS * GSD_WG1_COLS; i += GSD_FB_THREADS) smem[i] = 0.f;
    for (int w = 0; w < GSD_FB_THREADS / 64; w++) {
        __syncthreads();
        if (wave != w) continue;
#pragma unroll
        for (int t = 0; t < 11; t++) {
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int o = gsd_tile_row(t, g, r, K);
                if (o < 0) continue;
                red2[o * GSD_WG2_COLS + a] += gW2[t][0][r];
                red2[o * GSD_WG2_COLS + 16 + a] += gW2[t][1][r];
                if (a == 0) red2[o * GSD_WG2_COLS + 32] += bs2[t][r];
            }
        }
#pragma unroll
        for (int m = 0; m < 4; m++)
#pragma unroll
            for (int jt = 0; jt < 2; jt++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int j = 32 * m + 16 * jt + 4 * g + r;
#pragma unroll
                    for (int it = 0; it < 3; it++) red1[j * GSD_WG1_COLS + 16 * it + a] += gW1[m][jt][it][r];
                }
    }
    __syncthreads();
    float* dst2 = partial2 + (size_t)blockIdx.x * GSD_WG2_ROWS * GSD_WG2_COLS;
    float* dst1 = partial1 + (size_t)blockIdx.x * GSD_WG1_ROWS * GSD_WG1_COLS;
    for (int i = threadIdx.x; i < GSD_WG2_ROWS * GSD_WG2_COLS; i += GSD_FB_THREADS) dst2[i] = red2[i];
    for (int i = threadIdx.x; i < GSD_WG1_ROWS * GSD_WG1_COLS; i += GSD_FB_THREADS) dst1[i] = red1[i];
}

// ---- host side ----------------------------------------------------------------------------------------------------
#define GSD_MLP_GRID 256  // CUs: the matrix-core MLP kernels launch a few workgroups per CU, each with its operand table in LDS
static GsdMlps gsd_pack(const float* const* w)  // w[16] = {w1[4], b1[4], w2[4], b2[4]}
{
    GsdMlps P;
    for (int m = 0; m < 4; m++) { P.w1[m] = w[m]; P.b1[m] = w[4 + m]; P.w2[m] = w[8 + m]; P.b2[m] = w[12 + m]; }
    return P;
}

hipError_t gsd_launch_count(int N, int K, const float* const* weights, const int32_t* vis, const uint32_t* vis_count, const float* feat, const float* anchor,
                            const float* campos, float* neural_opacity, uint8_t* mask, uint8_t* count, uint32_t* first,
                            uint32_t* total, uint32_t* block_scratch, hipStream_t stream)
{
    if (N <= 0) return hipSuccess;
    const int nb = (N + GSD_THREADS - 1) / GSD_THREADS;
    hipLaunchKernelGGL(gsd_count_kernel, dim3(nb < 8 * GSD_MLP_GRID ? nb : 8 * GSD_MLP_GRID), dim3(GSD_THREADS), 0, stream, N, K, gsd_pack(weights), vis, vis_count, feat, anchor, campos,
                       neural_opacity, mask, count, block_scratch);
    hipLaunchKernelGGL(gsd_scan_kernel, dim3(1), dim3(1024), 0, stream, nb, block_scratch, total);
    hipLaunchKernelGGL(gsd_first_kernel, dim3(nb), dim3(GSD_THREADS), 0, stream, N, count, block_scratch, first);
    return hipGetLastError();
}

hipError_t gsd_launch_emit(int N, int K, const float* const* weights, const int32_t* vis, const uint32_t* vis_count, const float* feat, const float* anchor,
                           const float* offsets, const float* gscale, const float* campos, const float* neural_opacity,
                           const uint8_t* mask, const uint32_t* first, float* xyz, float* color, float* opacity,
                           float* uncertainty, float* scaling, float* rot, hipStream_t stream)
{
    if (N <= 0) return hipSuccess;
    // persistent waves: 2 per SIMD (one workgroup of 8 per CU), each walks ~6 groups of 16 anchors at 200k anchors
    const int emit_grid = GSD_MLP_GRID;
    const int nb = ((N + 15) / 16 + GSD_EMIT_THREADS / 64 - 1) / (GSD_EMIT_THREADS / 64);
    hipLaunchKernelGGL(gsd_emit_kernel, dim3(nb < emit_grid ? nb : emit_grid), dim3(GSD_EMIT_THREADS), 0, stream, N, K,
                       gsd_pack(weights), vis, vis_count, feat, anchor, offsets, gscale, campos, neural_opacity, mask, first, xyz, color,
                       opacity, uncertainty, scaling, rot);
    return hipGetLastError();
}

hipError_t gsd_launch_backward(int N, int K, const float* const* weights, const int32_t* vis, const float* feat, const float* anchor,
                                     const float* offsets, const float* gscale, const float* campos, const uint8_t* mask,
                                     const uint32_t* first, const float* g_xyz, const float* g_color, const float* g_opacity,
                                     const float* g_unc, const float* g_scaling, const float* g_rot, float* d_feat,
                                     float* d_anchor, float* d_offsets, float* d_gscale, void* workspace, float* const* grads16,
                                     hipStream_t stream)
{
    GsdGrads G;
    for (int i = 0; i < 16; i++) G.g[i] = grads16[i];
    float* p2 = (float*)workspace;
    float* p1 = p2 + (size_t)GSD_WG_BLOCKS * GSD_WG2_ROWS * GSD_WG2_COLS;
    // dynamic LDS beyond 64 KiB: one-time opt-in per device
    static thread_local uint64_t done_mask = 0;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const size_t lds = (size_t)GSD_FB_LDS_FLOATS * sizeof(float);
    if (!(dev >= 0 && dev < 64 && ((done_mask >> dev) & 1))) {
        e = hipFuncSetAttribute((const void*)gsd_backward_fused_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        if (dev >= 0 && dev < 64) done_mask |= 1ull << dev;
    }
    const int fused_grid = GSD_MLP_GRID;  // one workgroup of four waves per CU: one wave per SIMD
    const int want = ((N > 0 ? (N + 15) / 16 : 1) + GSD_FB_THREADS / 64 - 1) / (GSD_FB_THREADS / 64);
    int grid = want < fused_grid ? want : fused_grid;
    if (grid > GSD_WG_BLOCKS) grid = GSD_WG_BLOCKS;
    // N == 0: the partials are zeros (nothing to add up) -- the kernels still run so that every output is written
    hipLaunchKernelGGL(gsd_backward_fused_kernel, dim3(grid), dim3(GSD_FB_THREADS), lds, stream, N, K, gsd_pack(weights), vis, feat, anchor,
                       offsets, gscale, campos, mask, first, g_xyz, g_color, g_opacity, g_unc, g_scaling, g_rot, d_feat, d_anchor,
                       d_offsets, d_gscale, p2, p1);
    const int total = GSD_WG2_ROWS * GSD_WG2_COLS + GSD_WG1_ROWS * GSD_WG1_COLS;
    hipLaunchKernelGGL(gsd_weight_grad_finish_kernel, dim3((total + 63) / 64), dim3(256), 0, stream, K, grid, p2, p1, G);
    return hipGetLastError();
}

size_t gsd_weight_grad_workspace_bytes() { return (size_t)GSD_WG_BLOCKS * (GSD_WG2_ROWS * GSD_WG2_COLS + GSD_WG1_ROWS * GSD_WG1_COLS) * sizeof(float); }

// Gradient rows of the anchors a decode did NOT touch (hidden by the visibility mask): exact zeros, as the reference's
// x[visible_mask] indexing leaves them.  One thread per model row; the visible rows are written by the backward kernel, so
// the caller needs no model-sized zero-fill (57 MB per iteration at 200k anchors; typically a tenth of the rows are hidden).
__global__ void __launch_bounds__(256) gsd_zero_hidden_kernel(int N, int K, const uint8_t* __restrict__ visible_mask, float* __restrict__ d_feat,
                                                              float* __restrict__ d_anchor, float* __restrict__ d_off, float* __restrict__ d_gs)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= N || visible_mask[r]) return;
    float4* f = reinterpret_cast<float4*>(d_feat + (size_t)r * 32);
#pragma unroll
    for (int i = 0; i < 8; i++) f[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int i = 0; i < 3; i++) d_anchor[(size_t)r * 3 + i] = 0.f;
    for (int i = 0; i < 3 * K; i++) d_off[(size_t)r * 3 * K + i] = 0.f;
    for (int i = 0; i < 6; i++) d_gs[(size_t)r * 6 + i] = 0.f;
}

hipError_t gsd_launch_zero_hidden(int N, int K, const uint8_t* visible_mask, float* d_feat, float* d_anchor, float* d_off, float* d_gs,
                                  hipStream_t stream)
{
    if (N <= 0) return hipSuccess;
    hipLaunchKernelGGL(gsd_zero_hidden_kernel, dim3((N + 255) / 256), dim3(256), 0, stream, N, K, visible_mask, d_feat, d_anchor, d_off, d_gs);
    return hipGetLastError();
}

// ---- visible-row list on the device (replaces torch.nonzero(visible_mask) and its host round trip) ------------------
// rows[] = the indices r with visible_mask[r] != 0 in ascending order, count[0] = how many.  Three small launches: per-block
// counts, the one-block scan above, per-block placement.
__global__ void __launch_bounds__(256) gsd_rows_count_kernel(int N, const uint8_t* __restrict__ visible_mask, uint32_t* __restrict__ block_sum)
{
    __shared__ uint32_t wsum[4];
    const int n = blockIdx.x * 256 + threadIdx.x;
    const bool v = n < N && visible_mask[n] != 0;
    const unsigned long long b = __ballot(v);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = (uint32_t)__popcll(b);
    __syncthreads();
    if (threadIdx.x == 0) block_sum[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

__global__ void __launch_bounds__(256) gsd_rows_place_kernel(int N, const uint8_t* __restrict__ visible_mask, const uint32_t* __restrict__ block_base,
                                                             int32_t* __restrict__ rows)
{
    __shared__ uint32_t wsum[4];
    const int n = blockIdx.x * 256 + threadIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool v = n < N && visible_mask[n] != 0;
    const unsigned long long b = __ballot(v);
    if (lane == 0) wsum[wave] = (uint32_t)__popcll(b);
    __syncthreads();
    uint32_t base = block_base[blockIdx.x];
    for (int w = 0; w < wave; w++) base += wsum[w];
    if (v) rows[base + __builtin_amdgcn_mbcnt_hi((uint32_t)(b >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b, 0u))] = n;
}

hipError_t gsd_launch_visible_rows(int N, const uint8_t* visible_mask, int32_t* rows, uint32_t* count, uint32_t* block_scratch, hipStream_t stream)
{
    if (N <= 0) return hipMemsetAsync(count, 0, 4, stream);
    const int nb = (N + 255) / 256;
    hipLaunchKernelGGL(gsd_rows_count_kernel, dim3(nb), dim3(256), 0, stream, N, visible_mask, block_scratch);
    hipLaunchKernelGGL(gsd_scan_kernel, dim3(1), dim3(1024), 0, stream, nb, block_scratch, count);
    hipLaunchKernelGGL(gsd_rows_place_kernel, dim3(nb), dim3(256), 0, stream, N, visible_mask, block_scratch, rows);
    return hipGetLastError();
}
