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
#include <stdlib.h>

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

#ifdef GSD_EXP_FASTMATH
__device__ __forceinline__ float gsd_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x)); }
#elif defined(GSD_EXP_NOPOST)
__device__ __forceinline__ float gsd_sigmoid(float x) { return x; }
#else
__device__ __forceinline__ float gsd_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }
#endif

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
#ifdef GSD_EXP_NOMFMA
#define GSD_MFMA(A, B, C) ((C) + (A) * (B))
#else
#define GSD_MFMA(A, B, C) __builtin_amdgcn_mfma_f32_16x16x4f32((A), (B), (C), 0, 0, 0)
#endif
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
#pragma unroll
    for (int jt = 0; jt < 2; jt++) {
        gsd_v4 acc = {t[(18 + 4 * jt) * 64], t[(19 + 4 * jt) * 64], t[(20 + 4 * jt) * 64], t[(21 + 4 * jt) * 64]};
#pragma unroll
        for (int s = 0; s < 8; s++) acc = GSD_MFMA(t[(9 * jt + s) * 64], X.f[s], acc);
        acc = GSD_MFMA(t[(9 * jt + 8) * 64], X.v, acc);
#pragma unroll
        for (int r = 0; r < 4; r++) h[jt][r] = fmaxf(acc[r], 0.0f);
    }
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
// index) so that every entry index -- and with it every weight pointer and offset -- is a compile-time constant and the
// wave's loads are all in flight before its first LDS write.  (With a run-time entry index the MLP pointers are fetched
// from the argument block by dependent loads and the staging alone took ~50 us.)
template <int W, int NW, int TOTAL, typename Entry>
__device__ __forceinline__ void gsd_stage_part(float* sw, int lane, Entry entry)
{
    constexpr int PER = (TOTAL + NW - 1) / NW;
    float v[PER];
#pragma unroll
    for (int i = 0; i < PER; i++) v[i] = (W + i * NW < TOTAL) ? entry(W + i * NW) : 0.0f;
#pragma unroll
    for (int i = 0; i < PER; i++)
        if (W + i * NW < TOTAL) sw[(W + i * NW) * 64 + lane] = v[i];
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
__global__ void __launch_bounds__(GSD_THREADS) gsd_count_kernel(int N, int K, GsdMlps P, const int32_t* __restrict__ vis,
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
    gsd_stage<GSD_THREADS / 64, GSD_L1_ENTRIES + GSD_L2_ENTRIES>(sw, wave, lane, [&](int e) {
        return e < GSD_L1_ENTRIES ? gsd_l1_entry(P.w1[0], P.b1[0], e, g, a)
                                  : gsd_l2_entry(P.w2[0], P.b2[0], K, true, 0, 0, 0, 0, e - GSD_L1_ENTRIES, g, a);
    });
    const float* t1 = sw + lane;
    const float* t2 = sw + GSD_L1_ENTRIES * 64 + lane;
    const float cx = campos[0], cy = campos[1], cz = campos[2];
    const int nb = (N + GSD_THREADS - 1) / GSD_THREADS;
    for (int blk = blockIdx.x; blk < nb; blk += gridDim.x) {
        if (threadIdx.x == 0) bs = 0u;
        __syncthreads();  // (also orders the table writes above against the first reads)
        uint32_t mine = 0;  // survivors among this lane's offsets
#pragma unroll
        for (int t = 0; t < GSD_GROUPS; t++) {
            const int n = blk * GSD_THREADS + wave * 64 + t * 16 + a;
            const bool live = n < N;
            const int nn = live ? n : N - 1;
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
            if (g == 0 && live) count[n] = (uint8_t)c;
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
    int N, int K, GsdMlps P, const int32_t* __restrict__ vis, const float* __restrict__ feat, const float* __restrict__ anchor,
    const float* __restrict__ offsets /*[N,K,3]*/, const float* __restrict__ gscale /*[N,6]*/,
    const float* __restrict__ campos, const float* __restrict__ neural_opacity, const uint8_t* __restrict__ mask,
    const uint32_t* __restrict__ first, float* __restrict__ xyz, float* __restrict__ color, float* __restrict__ opacity, float* __restrict__ uncertainty,
    float* __restrict__ scaling, float* __restrict__ rot)
{
    __shared__ float sw[GSD_EMIT_ENTRIES * 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, a = lane & 15;
    gsd_stage<GSD_EMIT_THREADS / 64, GSD_EMIT_ENTRIES>(sw, wave, lane, [&](int e) {
        if (e < 3 * GSD_L1_ENTRIES) {
            const int m = e / GSD_L1_ENTRIES;
            return gsd_l1_entry(P.w1[m + 1], P.b1[m + 1], e - m * GSD_L1_ENTRIES, g, a);
        }
        const int t = (e - 3 * GSD_L1_ENTRIES) / GSD_L2_ENTRIES, j = (e - 3 * GSD_L1_ENTRIES) - t * GSD_L2_ENTRIES;
        if (t == 0) return gsd_l2_entry(P.w2[1], P.b2[1], K, true, 0, 0, 0, 0, j, g, a);
        if (t < 4) return gsd_l2_entry(P.w2[2], P.b2[2], K, false, t - 1, 3, 3, 0, j, g, a);
        if (t < 7) return gsd_l2_entry(P.w2[3], P.b2[3], K, false, t - 4, 3, 7, 0, j, g, a);
        return gsd_l2_entry(P.w2[3], P.b2[3], K, false, t - 7, 4, 7, 3, j, g, a);
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
#ifdef GSD_EXP_NOSTORE
            on[q] = k < K - 1000 && ((keep >> k) & 1u);
#else
            on[q] = k < K && ((keep >> k) & 1u);
#endif
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
#pragma unroll
        for (int q = 0; q < 3; q++) {
            const gsd_v4 z = gsd_mfma_l2(tl2 + (1 + q) * GSD_L2_ENTRIES * 64, h);
#pragma unroll
            for (int c = 0; c < 3; c++) col[q][c] = gsd_sigmoid(z[c]);
        }
        gsd_mfma_l1(tl1 + 2 * GSD_L1_ENTRIES * 64, X, h);
#pragma unroll
        for (int q = 0; q < 3; q++) {
            const gsd_v4 zs = gsd_mfma_l2(tl2 + (4 + q) * GSD_L2_ENTRIES * 64, h);
            const gsd_v4 zr = gsd_mfma_l2(tl2 + (7 + q) * GSD_L2_ENTRIES * 64, h);
#pragma unroll
            for (int c = 0; c < 3; c++) scl[q][c] = I.gs[3 + c] * gsd_sigmoid(zs[c]);  // :90
            const float nrm = fmaxf(sqrtf(zr[0] * zr[0] + zr[1] * zr[1] + zr[2] * zr[2] + zr[3] * zr[3]), 1e-12f);  // F.normalize
            rt[q] = make_float4(zr[0] / nrm, zr[1] / nrm, zr[2] / nrm, zr[3] / nrm);  // :91
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
#ifdef GSD_EXP_DESYNC
    if (wave >= 4) for (int i = 0; i < GSD_EXP_DESYNC; i++) __builtin_amdgcn_s_sleep(127);
#endif
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
// Per anchor: upstream gradients of its surviving rows -> d(out) of the four second layers -> d(hidden) -> d(input).
// Writes d_feat[N,32], d_anchor[N,3], d_offsets[N,K,3], d_gscale[N,6] and, for the weight-gradient GEMMs of the caller,
//   D2[12K, N] (opacity K | uncertainty K | color 3K | cov 7K)   = dL/d(second-layer pre-activations)
//   D1[128, N], H[128, N]  (four blocks of 32)                    = dL/d(first-layer pre-activations), hidden activations
//   X [36, N]                                                     = the MLP input
// all feature-major, so that the 64 anchors of a wave store 64 consecutive floats
// One MLP per launch (template M): keeps the live state at input + hidden + d(hidden) registers, so several
// waves fit per SIMD (the all-in-one version needed 256 VGPRs + 51 AGPRs and spilled SGPRs: one wave per SIMD).
// On the matrix cores like the forward (layout above): recompute the hidden layer and the head's pre-activations of 16
// anchors, turn the upstream gradients of the lane's own offsets into deltas dz (register r of a tile, exactly where the
// forward product left z), and feed those registers straight back as the B operand of W2^T dz -> d(hidden): step (tile t,
// component r) pairs k-group g with output rho(t, 4g + r), the A operand holds W2[rho(t, 4g + r)][16 jt + a].  d(hidden)
// comes out in the layout of the hidden layer itself, so the ReLU mask is elementwise.  All four arrays are written
// feature-major in 64-anchor chunks: a register of a tile is 4 rows x 16 consecutive anchors = four 64-byte runs.
template <int M> struct GsdBwd {
    static constexpr int NT = M < 2 ? 1 : (M == 2 ? 3 : 6);      // second-layer tiles
    static constexpr int NC = M < 2 ? 3 : (M == 2 ? 9 : 21);     // (tile, component) pairs = k-steps of W2^T dz
    static constexpr int W2T = GSD_L1_ENTRIES + NT * GSD_L2_ENTRIES;
    static constexpr int ENTRIES = W2T + 2 * NC;
    // tile t: head (M < 2) | colour q = t | scale q = t (t < 3), rotation q = t - 3
    __device__ static constexpr int comps(int t) { return M < 2 ? 3 : (M == 2 ? 3 : (t < 3 ? 3 : 4)); }
    __device__ static constexpr int q(int t) { return M == 3 && t >= 3 ? t - 3 : t; }
    __device__ static constexpr int stride() { return M < 2 ? 1 : (M == 2 ? 3 : 7); }
    __device__ static constexpr int first(int t) { return M == 3 && t >= 3 ? 3 : 0; }
    __device__ static constexpr int pair0(int t) { return M == 3 ? (t < 3 ? 3 * t : 9 + 4 * (t - 3)) : 3 * t; }  // first (tile, component) pair of tile t
};
#define GSD_BWD_THREADS 512
template <int M>
__global__ void __launch_bounds__(GSD_BWD_THREADS) gsd_backward_mlp_kernel(
    int N, int K, GsdMlps P, const int32_t* __restrict__ vis, const float* __restrict__ feat, const float* __restrict__ anchor,
    const float* __restrict__ gscale, const float* __restrict__ campos, const uint8_t* __restrict__ mask,
    const uint32_t* __restrict__ first, const float* __restrict__ g_color, const float* __restrict__ g_opacity,
    const float* __restrict__ g_unc, const float* __restrict__ g_scaling, const float* __restrict__ g_rot,
    float* __restrict__ d_gscale, float* __restrict__ D2, float* __restrict__ D1, float* __restrict__ Hout,
    float* __restrict__ Xout)
{
    typedef GsdBwd<M> C;
    __shared__ float sw[C::ENTRIES * 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, a = lane & 15;
    gsd_stage<GSD_BWD_THREADS / 64, C::ENTRIES>(sw, wave, lane, [&](int e) {
        if (e < GSD_L1_ENTRIES) return gsd_l1_entry(P.w1[M], P.b1[M], e, g, a);
        if (e < C::W2T) {
            const int t = (e - GSD_L1_ENTRIES) / GSD_L2_ENTRIES, j = (e - GSD_L1_ENTRIES) - t * GSD_L2_ENTRIES;
            return gsd_l2_entry(P.w2[M], P.b2[M], K, M < 2, C::q(t), C::comps(t), C::stride(), C::first(t), j, g, a);
        }
        // W2^T operand of k-step (tile t, component r), hidden tile jt: W2[rho(t, 4g + r)][16 jt + a]
        const int pr = (e - C::W2T) >> 1, jt = (e - C::W2T) & 1;
        int t = 0;
        for (int u = 1; u < C::NT; u++)
            if (pr >= C::pair0(u)) t = u;
        const int r = pr - C::pair0(t);
        const int k = M < 2 ? 4 * r + g : 4 * C::q(t) + g;
        const int o = M < 2 ? k : k * C::stride() + C::first(t) + r;
        return k < K ? P.w2[M][o * GSD_HID + 16 * jt + a] : 0.0f;
    });
    __syncthreads();
    const float* tl1 = sw + lane;
    const float* tl2 = sw + GSD_L1_ENTRIES * 64 + lane;
    const float* tw2t = sw + C::W2T * 64 + lane;
    const float cx = campos[0], cy = campos[1], cz = campos[2];
    const int out_base = M == 0 ? 0 : (M == 1 ? K : (M == 2 ? 2 * K : 5 * K));
    const int groups = (N + 15) / 16, stride = gridDim.x * (GSD_BWD_THREADS / 64);
#pragma unroll 1
    for (int grp = blockIdx.x * (GSD_BWD_THREADS / 64) + wave; grp < groups; grp += stride) {
        const int n = grp * 16 + a;
        const bool live = n < N;
        const int nn = live ? n : N - 1;
        const int ai = vis ? vis[nn] : nn;
        GsdRaw R;
        gsd_load_raw(R, feat, anchor, ai, g);
        uint32_t keep = 0;  // the lane's own offsets, then OR-ed over the anchor's four lanes
#pragma unroll
        for (int q = 0; q < 3; q++) {
            const int k = 4 * q + g;
            if (live && k < K && mask[(size_t)nn * K + k]) keep |= 1u << k;
        }
        const uint32_t row0 = first[nn];
        float gs3[3] = { 0.f, 0.f, 0.f };
        if (M == 3) {
#pragma unroll
            for (int c = 0; c < 3; c++) gs3[c] = gscale[6 * (size_t)ai + 3 + c];
        }
        keep |= (uint32_t)__shfl_xor((int)keep, 16, 64);
        keep |= (uint32_t)__shfl_xor((int)keep, 32, 64);
        uint32_t row[3];
        bool on[3];
#pragma unroll
        for (int q = 0; q < 3; q++) {
            const int k = 4 * q + g;
            on[q] = k < K && ((keep >> k) & 1u);
            row[q] = row0 + (uint32_t)__popc(keep & ((1u << k) - 1u));
        }
        // upstream gradients of the lane's rows (requested before the products)
        float up[3][7];
#pragma unroll
        for (int q = 0; q < 3; q++) {
#pragma unroll
            for (int c = 0; c < 7; c++) up[q][c] = 0.f;
            if (on[q]) {
                const size_t r = row[q];
                if (M == 0) up[q][0] = g_opacity[r];
                if (M == 1) up[q][0] = g_unc[r];
                if (M == 2) { up[q][0] = g_color[3 * r]; up[q][1] = g_color[3 * r + 1]; up[q][2] = g_color[3 * r + 2]; }
                if (M == 3) {
                    up[q][0] = g_scaling[3 * r]; up[q][1] = g_scaling[3 * r + 1]; up[q][2] = g_scaling[3 * r + 2];
                    const float4 gr = reinterpret_cast<const float4*>(g_rot)[r];
                    up[q][3] = gr.x; up[q][4] = gr.y; up[q][5] = gr.z; up[q][6] = gr.w;
                }
            }
        }
        GsdIn X;
        gsd_finish_in(X, R, cx, cy, cz, g);
        if (M == 0 && live) {  // the MLP input, feature-major (feature map of the first layer)
#pragma unroll
            for (int s = 0; s < 8; s++) Xout[GSD_AT(GSD_IN, s < 4 ? 4 * g + s : 12 + 4 * g + s, n)] = X.f[s];
            Xout[GSD_AT(GSD_IN, 32 + g, n)] = X.v;
        }
        gsd_v4 h[2];
        gsd_mfma_l1(tl1, X, h);
        if (live) {
#pragma unroll
            for (int jt = 0; jt < 2; jt++)
#pragma unroll
                for (int r = 0; r < 4; r++) Hout[GSD_AT(128, M * 32 + 16 * jt + 4 * g + r, n)] = h[jt][r];
        }
        // deltas of the second layer, tile by tile
        gsd_v4 dz[C::NT];
        float dgs3[3] = { 0.f, 0.f, 0.f };
        if (M < 2) {
            const gsd_v4 z = gsd_mfma_l2(tl2, h);  // register r = offset 4r + g
#pragma unroll
            for (int q = 0; q < 3; q++) {
                float d = 0.f;
                if (on[q]) {
                    if (M == 0) { const float t = tanhf(z[q]); d = up[q][0] * (1.0f - t * t); }       // opacity = tanh(z)
                    else { const float sg = gsd_sigmoid(z[q]); d = up[q][0] * sg * (1.0f - sg); }   // sigmoid
                }
                dz[0][q] = d;
            }
            dz[0][3] = 0.f;
        } else if (M == 2) {
#pragma unroll
            for (int q = 0; q < 3; q++) {
                const gsd_v4 z = gsd_mfma_l2(tl2 + q * GSD_L2_ENTRIES * 64, h);
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    const float sg = gsd_sigmoid(z[c]);
                    dz[q][c] = on[q] ? up[q][c] * sg * (1.0f - sg) : 0.f;
                }
                dz[q][3] = 0.f;
            }
        } else {
#pragma unroll
            for (int q = 0; q < 3; q++) {
                const gsd_v4 zs = gsd_mfma_l2(tl2 + q * GSD_L2_ENTRIES * 64, h);
                const gsd_v4 zr = gsd_mfma_l2(tl2 + (3 + q) * GSD_L2_ENTRIES * 64, h);
#pragma unroll
                for (int c = 0; c < 3; c++) {  // scaling = gs[3+c] * sigmoid(z)
                    const float sg = gsd_sigmoid(zs[c]);
                    dz[q][c] = on[q] ? up[q][c] * gs3[c] * sg * (1.0f - sg) : 0.f;
                    dgs3[c] += on[q] ? up[q][c] * sg : 0.f;
                }
                dz[q][3] = 0.f;
                // rot = v / max(|v|, eps): d v = (g - rot (rot . g)) / |v|
                const float nrm = fmaxf(sqrtf(zr[0] * zr[0] + zr[1] * zr[1] + zr[2] * zr[2] + zr[3] * zr[3]), 1e-12f);
                float rt[4], dot = 0.f;
#pragma unroll
                for (int c = 0; c < 4; c++) { rt[c] = zr[c] / nrm; dot += up[q][3 + c] * rt[c]; }
#pragma unroll
                for (int c = 0; c < 4; c++) dz[3 + q][c] = on[q] ? (up[q][3 + c] - rt[c] * dot) / nrm : 0.f;
            }
        }
        // D2 rows and d(hidden) = W2^T dz
        gsd_v4 dh[2] = { {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f} };
#pragma unroll
        for (int t = 0; t < C::NT; t++) {
#pragma unroll
            for (int r = 0; r < C::comps(t); r++) {
                const int k = M < 2 ? 4 * r + g : 4 * C::q(t) + g;
                const int o = M < 2 ? k : k * C::stride() + C::first(t) + r;
                if (live && k < K) D2[GSD_AT(12 * K, out_base + o, n)] = dz[t][r];
                const int pr = C::pair0(t) + r;
                dh[0] = GSD_MFMA(tw2t[(2 * pr) * 64], dz[t][r], dh[0]);
                dh[1] = GSD_MFMA(tw2t[(2 * pr + 1) * 64], dz[t][r], dh[1]);
            }
        }
        if (live) {
#pragma unroll
            for (int jt = 0; jt < 2; jt++)
#pragma unroll
                for (int r = 0; r < 4; r++)
                    D1[GSD_AT(128, M * 32 + 16 * jt + 4 * g + r, n)] = h[jt][r] > 0.0f ? dh[jt][r] : 0.0f;  // through the ReLU
        }
        if (M == 3) {  // d grid_scaling[3:6]: over the anchor's offsets = over its four lanes
#pragma unroll
            for (int c = 0; c < 3; c++) {
                dgs3[c] += __shfl_xor(dgs3[c], 16, 64);
                dgs3[c] += __shfl_xor(dgs3[c], 32, 64);
            }
            if (live && g == 0) {
#pragma unroll
                for (int c = 0; c < 3; c++) d_gscale[6 * (size_t)ai + 3 + c] = dgs3[c];
            }
        }
    }
}

// dL/d(input) = sum over the four MLPs of W1^T d(pre1), then feature / anchor gradients; also the geometry part:
// xyz = anchor + offset * gs[0:3].  On the matrix cores: rows = the 36 inputs (three tiles), columns = 16 anchors, K = the
// 4 x 32 first-layer deltas, read back from D1 (step s of MLP m: lane (g, a) takes row 32 m + 4 s + g of its anchor:
// four 64-byte runs per load).  Lane (g, a) ends up with d(input) 4g .. 4g+3, 16 + 4g .. 16 + 4g+3 of its anchor (two
// 16-byte stores into d_feat) and lane (0, a) with the view / distance gradients; the offsets k = g, 4 + g, 8 + g of
// the anchor are the lane's share of the geometry part.
#define GSD_INK_ENTRIES (4 * 3 * 8)
__global__ void __launch_bounds__(GSD_BWD_THREADS) gsd_backward_input_kernel(
    int N, int K, GsdMlps P, const int32_t* __restrict__ vis, const float* __restrict__ anchor,
    const float* __restrict__ offsets, const float* __restrict__ gscale, const float* __restrict__ campos,
    const uint8_t* __restrict__ mask, const uint32_t* __restrict__ first, const float* __restrict__ g_xyz,
    const float* __restrict__ D1, float* __restrict__ d_feat, float* __restrict__ d_anchor, float* __restrict__ d_offsets,
    float* __restrict__ d_gscale)
{
    __shared__ float sw[GSD_INK_ENTRIES * 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, a = lane & 15;
    gsd_stage<GSD_BWD_THREADS / 64, GSD_INK_ENTRIES>(sw, wave, lane, [&](int e) {
        // entry (m, it, s): W1_m[4 s + g][16 it + a] (input rows beyond 35: zero)
        const int m = e / 24, it = (e - 24 * m) >> 3, st = e & 7, i = 16 * it + a;
        return i < GSD_IN ? P.w1[m][(4 * st + g) * GSD_IN + i] : 0.0f;
    });
    __syncthreads();
    const float* tw = sw + lane;
    const float cx = campos[0], cy = campos[1], cz = campos[2];
    const int groups = (N + 15) / 16, stride = gridDim.x * (GSD_BWD_THREADS / 64);
#pragma unroll 1
    for (int grp = blockIdx.x * (GSD_BWD_THREADS / 64) + wave; grp < groups; grp += stride) {
        const int n = grp * 16 + a;
        const bool live = n < N;
        const int nn = live ? n : N - 1;
        const int ai = vis ? vis[nn] : nn;
        float d1[32];
#pragma unroll
        for (int m = 0; m < 4; m++)
#pragma unroll
            for (int st = 0; st < 8; st++) d1[8 * m + st] = D1[GSD_AT(128, 32 * m + 4 * st + g, nn)];
        uint32_t keep = 0;  // the lane's own offsets, then OR-ed over the anchor's four lanes
#pragma unroll
        for (int q = 0; q < 3; q++) {
            const int k = 4 * q + g;
            if (live && k < K && mask[(size_t)nn * K + k]) keep |= 1u << k;
        }
        const uint32_t row0 = first[nn];
        const float ax = anchor[3 * (size_t)ai], ay = anchor[3 * (size_t)ai + 1], az = anchor[3 * (size_t)ai + 2];
        float gs[3];
#pragma unroll
        for (int c = 0; c < 3; c++) gs[c] = gscale[6 * (size_t)ai + c];
        keep |= (uint32_t)__shfl_xor((int)keep, 16, 64);
        keep |= (uint32_t)__shfl_xor((int)keep, 32, 64);
        // geometry of the lane's offsets
        float da[3] = { 0.f, 0.f, 0.f }, dgs[3] = { 0.f, 0.f, 0.f };
#pragma unroll
        for (int q = 0; q < 3; q++) {
            const int k = 4 * q + g;
            if (!live || k >= K) continue;
            float* dof = d_offsets + ((size_t)ai * K + k) * 3;
            if (!((keep >> k) & 1u)) { dof[0] = 0.f; dof[1] = 0.f; dof[2] = 0.f; continue; }
            const size_t r = row0 + (uint32_t)__popc(keep & ((1u << k) - 1u));
            const float* of = offsets + ((size_t)ai * K + k) * 3;
#pragma unroll
            for (int c = 0; c < 3; c++) {
                const float gu = g_xyz[3 * r + c];
                da[c] += gu; dof[c] = gu * gs[c]; dgs[c] += gu * of[c];
            }
        }
        gsd_v4 dx[3] = { {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f} };
#pragma unroll
        for (int m = 0; m < 4; m++)
#pragma unroll
            for (int st = 0; st < 8; st++)
#pragma unroll
                for (int it = 0; it < 3; it++) dx[it] = GSD_MFMA(tw[(24 * m + 8 * it + st) * 64], d1[8 * m + st], dx[it]);
#pragma unroll
        for (int c = 0; c < 3; c++) {
            da[c] += __shfl_xor(da[c], 16, 64); da[c] += __shfl_xor(da[c], 32, 64);
            dgs[c] += __shfl_xor(dgs[c], 16, 64); dgs[c] += __shfl_xor(dgs[c], 32, 64);
        }
        if (live) {
            float4* df = reinterpret_cast<float4*>(d_feat + (size_t)ai * GSD_F + 4 * g);
            df[0] = make_float4(dx[0][0], dx[0][1], dx[0][2], dx[0][3]);
            df[4] = make_float4(dx[1][0], dx[1][1], dx[1][2], dx[1][3]);
            if (g == 0) {
                // view vector / distance back to the anchor (v = a - c, dist = |v|, view = v / dist); dx[2] = d(view, dist)
                const float vx = ax - cx, vy = ay - cy, vz = az - cz;
                const float dist = sqrtf(vx * vx + vy * vy + vz * vz);
                const float ux = vx / dist, uy = vy / dist, uz = vz / dist;
                const float gdot = dx[2][0] * ux + dx[2][1] * uy + dx[2][2] * uz;
                d_anchor[3 * (size_t)ai] = da[0] + (dx[2][0] - ux * gdot) / dist + dx[2][3] * ux;
                d_anchor[3 * (size_t)ai + 1] = da[1] + (dx[2][1] - uy * gdot) / dist + dx[2][3] * uy;
                d_anchor[3 * (size_t)ai + 2] = da[2] + (dx[2][2] - uz * gdot) / dist + dx[2][3] * uz;
#pragma unroll
                for (int c = 0; c < 3; c++) d_gscale[6 * (size_t)ai + c] = dgs[c];
            }
        }
    }
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
    gsd_stage<GSD_FB_THREADS / 64, GSD_FB_ENTRIES>(sw, wave, lane, [&](int e) {
        if (e < GSD_FB_L2) {
            const int m = e / GSD_L1_ENTRIES;
            return gsd_l1_entry(P.w1[m], P.b1[m], e - m * GSD_L1_ENTRIES, g, a);
        }
        if (e < GSD_FB_W2T) {
            const int t = (e - GSD_FB_L2) / GSD_L2_ENTRIES, j = (e - GSD_FB_L2) - t * GSD_L2_ENTRIES;
            const GsdTile T = gsd_tile(t);
            return gsd_l2_entry(P.w2[T.mlp], P.b2[T.mlp], K, T.head, T.q, T.comps, T.stride, T.first, j, g, a);
        }
        if (e < GSD_FB_W1T) {
            // W2^T operand of k-step (tile t, component r), hidden tile jt: W2[rho(t, 4g + r)][16 jt + a]
            const int pr = (e - GSD_FB_W2T) >> 1, jt = (e - GSD_FB_W2T) & 1;
            int t = 0;
            for (int u = 1; u < 11; u++)
                if (pr >= gsd_tile(u).pair0) t = u;
            const GsdTile T = gsd_tile(t);
            const int r = pr - T.pair0;
            const int k = T.head ? 4 * r + g : 4 * T.q + g;
            const int o = T.head ? k : k * T.stride + T.first + r;
            return k < K ? P.w2[T.mlp][o * GSD_HID + 16 * jt + a] : 0.0f;
        }
        // W1^T operand of MLP m, input tile it, k-step (jt, r): W1_m[16 jt + 4g + r][16 it + a]
        const int m = (e - GSD_FB_W1T) / 24, it = ((e - GSD_FB_W1T) - 24 * m) >> 3, ks = (e - GSD_FB_W1T) & 7, i = 16 * it + a;
        return i < GSD_IN ? P.w1[m][(16 * (ks >> 2) + 4 * g + (ks & 3)) * GSD_IN + i] : 0.0f;
    });
    for (int i = lane; i < 48 * GSD_TS; i += 64) xT[i] = 0.f;  // rows 36..47 of the input tile stay zero
    __syncthreads();
    const float* const tl = sw + lane;
    const float cx = campos[0], cy = campos[1], cz = campos[2];

    gsd_v4 gW2[11][2], gW1[4][2][3];
    float bs2[11][4], bs1[4][2][4];
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
#pragma unroll
            for (int r = 0; r < 4; r++) bs1[m][jt][r] = 0.f;
        }

    const int groups = (N + 15) / 16, stride = gridDim.x * (GSD_FB_THREADS / 64);
#pragma unroll 1
    for (int grp = blockIdx.x * (GSD_FB_THREADS / 64) + wave; grp < groups; grp += stride) {
        const int n = grp * 16 + a;
        const bool live = n < N;
        const int nn = live ? n : N - 1;
        const int ai = vis ? vis[nn] : nn;
        GsdRaw R;
        gsd_load_raw(R, feat, anchor, ai, g);
        uint32_t keep = 0;  // the lane's own offsets, then OR-ed over the anchor's four lanes
#pragma unroll
        for (int q = 0; q < 3; q++) {
            const int k = 4 * q + g;
            if (live && k < K && mask[(size_t)nn * K + k]) keep |= 1u << k;
        }
        const uint32_t row0 = first[nn];
        float gs[6];
#pragma unroll
        for (int c = 0; c < 6; c++) gs[c] = gscale[6 * (size_t)ai + c];
        keep |= (uint32_t)__shfl_xor((int)keep, 16, 64);
        keep |= (uint32_t)__shfl_xor((int)keep, 32, 64);
        size_t row[3];
        bool on[3];
#pragma unroll
        for (int q = 0; q < 3; q++) {
            const int k = 4 * q + g;
            on[q] = k < K && ((keep >> k) & 1u);
            row[q] = row0 + (uint32_t)__popc(keep & ((1u << k) - 1u));
        }
        // geometry of the lane's offsets: xyz = anchor + offset * gs[0:3]
        float da[3] = { 0.f, 0.f, 0.f }, dgs[3] = { 0.f, 0.f, 0.f };
#pragma unroll
        for (int q = 0; q < 3; q++) {
            const int k = 4 * q + g;
            if (!live || k >= K) continue;
            float* dof = d_offsets + ((size_t)ai * K + k) * 3;
            if (!on[q]) { dof[0] = 0.f; dof[1] = 0.f; dof[2] = 0.f; continue; }
            const float* of = offsets + ((size_t)ai * K + k) * 3;
#pragma unroll
            for (int c = 0; c < 3; c++) {
                const float gu = g_xyz[3 * row[q] + c];
                da[c] += gu; dof[c] = gu * gs[c]; dgs[c] += gu * of[c];
            }
        }
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
#pragma unroll
        for (int m = 0; m < 4; m++) {
            // upstream gradients of the lane's rows for this head
            float up[3][7];
#pragma unroll
            for (int q = 0; q < 3; q++) {
#pragma unroll
                for (int c = 0; c < 7; c++) up[q][c] = 0.f;
                if (on[q]) {
                    const size_t r = row[q];
                    if (m == 0) up[q][0] = g_opacity[r];
                    if (m == 1) up[q][0] = g_unc[r];
                    if (m == 2) { up[q][0] = g_color[3 * r]; up[q][1] = g_color[3 * r + 1]; up[q][2] = g_color[3 * r + 2]; }
                    if (m == 3) {
                        up[q][0] = g_scaling[3 * r]; up[q][1] = g_scaling[3 * r + 1]; up[q][2] = g_scaling[3 * r + 2];
                        const float4 gr = reinterpret_cast<const float4*>(g_rot)[r];
                        up[q][3] = gr.x; up[q][4] = gr.y; up[q][5] = gr.z; up[q][6] = gr.w;
                    }
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
                    if (m == 0) { const float t = tanhf(z[q]); d = up[q][0] * (1.0f - t * t); }       // opacity = tanh(z)
                    else { const float sg = gsd_sigmoid(z[q]); d = up[q][0] * sg * (1.0f - sg); }   // sigmoid
                    dz[q] = on[q] ? d : 0.f;
                }
                dz[3] = 0.f;
                tile_back(m, dz);
            } else if (m == 2) {
#pragma unroll
                for (int q = 0; q < 3; q++) {
                    const gsd_v4 z = gsd_mfma_l2(tl2 + (2 + q) * GSD_L2_ENTRIES * 64, h);
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
#pragma unroll
                for (int q = 0; q < 3; q++) {
                    const gsd_v4 zs = gsd_mfma_l2(tl2 + (5 + q) * GSD_L2_ENTRIES * 64, h);
                    const gsd_v4 zr = gsd_mfma_l2(tl2 + (8 + q) * GSD_L2_ENTRIES * 64, h);
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
#pragma unroll
                    for (int c = 0; c < 4; c++) { rt[c] = zr[c] / nrm; dot += up[q][3 + c] * rt[c]; }
#pragma unroll
                    for (int c = 0; c < 4; c++) dzr[c] = on[q] ? (up[q][3 + c] - rt[c] * dot) / nrm : 0.f;
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
                    bs1[m][jt][r] += d1[r];
                }
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
#pragma unroll
    for (int m = 0; m < 4; m++)
#pragma unroll
        for (int jt = 0; jt < 2; jt++)
#pragma unroll
            for (int r = 0; r < 4; r++)
#pragma unroll
                for (int d = 1; d < 16; d <<= 1) bs1[m][jt][r] += __shfl_xor(bs1[m][jt][r], d, 64);
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
        __builtin_amdgcn_s_waitcnt(0);  // (the tile sums above land before the bias column, which shares words with input row 36)
#pragma unroll
        for (int m = 0; m < 4; m++)
#pragma unroll
            for (int jt = 0; jt < 2; jt++)
#pragma unroll
                for (int r = 0; r < 4; r++)
                    if (a == 0) red1[(32 * m + 16 * jt + 4 * g + r) * GSD_WG1_COLS + 36] += bs1[m][jt][r];
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

hipError_t gsd_launch_count(int N, int K, const float* const* weights, const int32_t* vis, const float* feat, const float* anchor,
                            const float* campos, float* neural_opacity, uint8_t* mask, uint8_t* count, uint32_t* first,
                            uint32_t* total, uint32_t* block_scratch, hipStream_t stream)
{
    if (N <= 0) return hipSuccess;
    const int nb = (N + GSD_THREADS - 1) / GSD_THREADS;
    hipLaunchKernelGGL(gsd_count_kernel, dim3(nb < 8 * GSD_MLP_GRID ? nb : 8 * GSD_MLP_GRID), dim3(GSD_THREADS), 0, stream, N, K, gsd_pack(weights), vis, feat, anchor, campos,
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
    // persistent waves: 2 per SIMD (one workgroup of 8 per CU), each walks ~6 groups of 16 anchors at 200k anchors
    static const int emit_grid = getenv("GSD_EMIT_GRID") ? atoi(getenv("GSD_EMIT_GRID")) : GSD_MLP_GRID;
    const int nb = ((N + 15) / 16 + GSD_EMIT_THREADS / 64 - 1) / (GSD_EMIT_THREADS / 64);
    hipLaunchKernelGGL(gsd_emit_kernel, dim3(nb < emit_grid ? nb : emit_grid), dim3(GSD_EMIT_THREADS), 0, stream, N, K,
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
    const GsdMlps P = gsd_pack(weights);
    static const int bwd_grid = getenv("GSD_BWD_GRID") ? atoi(getenv("GSD_BWD_GRID")) : 2 * GSD_MLP_GRID;
    const int ngrp = ((N + 15) / 16 + GSD_BWD_THREADS / 64 - 1) / (GSD_BWD_THREADS / 64);
#define GSD_BWD(M)                                                                                                          \
    hipLaunchKernelGGL(gsd_backward_mlp_kernel<M>, dim3(ngrp < bwd_grid ? ngrp : bwd_grid), dim3(GSD_BWD_THREADS), 0, stream, N, K, P, vis, feat, anchor, gscale, campos, mask,  \
                       first, g_color, g_opacity, g_unc, g_scaling, g_rot, d_gscale, D2, D1, H, X)
    GSD_BWD(0); GSD_BWD(1); GSD_BWD(2); GSD_BWD(3);
#undef GSD_BWD
    hipLaunchKernelGGL(gsd_backward_input_kernel, dim3(ngrp < bwd_grid ? ngrp : bwd_grid), dim3(GSD_BWD_THREADS), 0, stream, N, K, P, vis, anchor, offsets, gscale, campos, mask,
                       first, g_xyz, D1, d_feat, d_anchor, d_offsets, d_gscale);
    return hipGetLastError();
}

hipError_t gsd_launch_backward_fused(int N, int K, const float* const* weights, const int32_t* vis, const float* feat, const float* anchor,
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
    static const int fused_grid = getenv("GSD_FUSED_GRID") ? atoi(getenv("GSD_FUSED_GRID")) : GSD_MLP_GRID;
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
