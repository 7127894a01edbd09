// blend.hip -- tile-based forward / backward alpha blend of RGB + depth + one feature channel.
//
// Replaces DGR forward.cu:441-568 renderCUDA (fwd) and backward.cu:409-604 renderCUDA (bwd).
//
// Mapping onto CDNA4.  One 256-thread workgroup (4 wavefronts of 64) per 16x16 screen tile; wavefront w owns
// the 8x8 pixel QUADRANT (w & 1, w >> 1) of the tile, lane l the pixel (l & 7, l >> 3) inside it.
//   * Tile lists are consumed in batches of 256 instances.  Thread i gathers the 64-byte record of instance i
//     (three 16-byte loads fwd, four bwd) into LDS as float4 SoA arrays, and -- once per instance, not once per
//     pixel -- tests the Gaussian's alpha >= 1/255 ellipse against the four quadrant boxes (gsr_box_min_q).
//     Each wave then compacts, with ballot + mbcnt, the indices of the instances that can reach ITS quadrant
//     into a private LDS list and walks only those: a typical splat touches 1-2 of the 4 quadrants, so the
//     per-pixel test loop shrinks accordingly and the hit rate of what remains goes up.  The reference tests
//     every instance of the tile against all 256 pixels (forward.cu:513-553).
//   * Per-Gaussian operands reach the per-pixel math as SCALAR operands: each lane pulls ONE instance of its
//     wave's list out of LDS (3 ds_read_b128 serve 64 instances) and the inner loop broadcasts lane k's values
//     with v_readlane_b32 into SGPRs -- no LDS round trip per instance (measured: the broadcast ds_reads +
//     4-lane LDS atomics kept the LDS 66 % busy with the VALU at 34 %).  The reference re-reads colour and
//     depth from GLOBAL memory per contribution (forward.cu:545-546).
//   * Early-out: a wave leaves the batch when all its pixels are done (__all); in the backward a wave skips
//     gradient math + reduction when none of its pixels blends the instance (__any); the block leaves when
//     every wave is done (__syncthreads_and).
//   * Backward gradient scatter: the reference issues 11 global atomicAdd per (pixel, Gaussian) contribution
//     (backward.cu:554-601).  Here the 11 partials are summed over the wave's 64 lanes with 6 DPP adds
//     (quad_perm x2, row_half_mirror, row_mirror, row_bcast:15, row_bcast:31) and lane 63 adds the wave total
//     into an LDS accumulator [256][12] with one single-lane ds_add_f32 per value (built with
//     -amdgpu-atomic-optimizer-strategy=None so it stays one instruction).  After the batch,
//     thread i stores the 12 floats of instance i with three plain 16-byte stores into that instance's private
//     gradient slot (slot = Gaussian's scan offset + tile position inside its rectangle); gauss_bwd.hip sums
//     each Gaussian's slots.  No atomics on global memory at all.  (The LDS adds of a tile's 4 wavefronts
//     are unordered, so two runs agree to rounding, not bit for bit.)
//   * AUX = false specialises the backward for "no gradient flows into the depth and feature maps" (GScream's
//     RGB-only iterations): 9 instead of 11 reductions and no depth/feature recurrences.
//   * XCD awareness: workgroup b runs on XCD b % 8 (observed dispatch rule); the block->tile map hands each
//     XCD a contiguous band of tile rows so neighbouring tiles, which share most of their Gaussians, hit the
//     same 4 MiB L2.  Pure speed: any placement gives the same result.
#include "gsr_math.h"

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

// Sum over all 64 lanes; the total lands in the last DPP row (lanes 48..63).
__device__ __forceinline__ float gsr_wave_sum_to_row3(float v)
{
    v = gsr_row_sum16(v);
    // row_bcast:15 into rows 1,3 then row_bcast:31 into rows 2,3 (lanes outside the row mask add 0)
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x142, 0xa, 0xf, false));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x143, 0xc, 0xf, false));
    return v;
}

// Broadcast lane k's value to the whole wave through an SGPR (v_readlane_b32): per-Gaussian operands then
// enter the per-pixel math as scalar operands -- no LDS round trip and no VGPRs per operand.
__device__ __forceinline__ float gsr_bcast(float v, int k)
{
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), k));
}

// ---- A/B switches (compile-time; defaults chosen from measurements, see profiles/) ----
//   GSR_{FWD,BWD}_READLANE = 1 : per-instance operands via v_readlane from a per-lane copy (else LDS broadcast reads)
//   GSR_RED_MODE 0: 4-step row reduce per value, 4 row leaders issue one ds_add_f32 per value (9-11 LDS atomics)
//                1: 6-step wave reduce per value, lane 63 issues one ds_add_f32 per value
//                2: 4-step row reduce per value, pack value i into lane i of each row, 2 lane-aligned cross-row adds
//                   (v_permlane32_swap / v_permlane16_swap) on the packed register, lanes 0..10 issue ONE ds_add_f32
//                3: like 2 without the cross-row step: all four rows add (one instruction, 4 lanes per address)
#ifndef GSR_FWD_READLANE
#define GSR_FWD_READLANE 0
#endif
#ifndef GSR_BWD_READLANE
#define GSR_BWD_READLANE 0
#endif
#ifndef GSR_RED_MODE
#define GSR_RED_MODE 2
#endif

// 4-bit mask of the 8x8 quadrants of tile (tx, ty) in which the Gaussian can reach alpha >= 1/255.
__device__ __forceinline__ uint32_t gsr_quadrant_mask(const float4 A, const float4 B, const float tau, int tx, int ty,
                                                      int W, int H)
{
    const float rA = GSR_RCP(A.z), rC = GSR_RCP(B.x);
    uint32_t m = 0;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int x0 = tx * 16 + (q & 1) * 8, y0 = ty * 16 + (q >> 1) * 8;
        const float bx1 = (float)min(x0 + 7, W - 1), by1 = (float)min(y0 + 7, H - 1);
        const bool hit = x0 < W && y0 < H &&
                         !(gsr_box_min_q(A.x, A.y, A.z, A.w, B.x, rA, rC, (float)x0, bx1, (float)y0, by1) > tau);
        m |= hit ? (1u << q) : 0u;
    }
    return m;
}

// Wave-private compaction: indices i < cnt with bit `wave` set in sQ[i] and pred(i), in ascending order.
template <typename Pred>
__device__ __forceinline__ int gsr_compact(const uint32_t* sQ, uint16_t* list, int cnt, int wave, int lane, Pred pred)
{
    int n = 0;
#pragma unroll
    for (int c = 0; c < GSR_BATCH / 64; c++) {
        const int i = c * 64 + lane;
        const bool hit = i < cnt && ((sQ[i] >> wave) & 1u) && pred(i);
        const unsigned long long bal = __ballot(hit);
        if (hit) list[n + __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u))] = (uint16_t)i;
        n += __popcll(bal);
    }
    return n;
}

// ---------------------------------------------------------------------------------------------
// Forward
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) gsr_blend_fwd_kernel(
    const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list, const GsrRec* __restrict__ rec, int W,
    int H, int gx, int T, const float* __restrict__ bg, float* __restrict__ out_color, float* __restrict__ out_depth,
    float* __restrict__ out_feature, float* __restrict__ final_T, uint32_t* __restrict__ n_contrib)
{
    __shared__ float4 sA[GSR_BATCH], sB[GSR_BATCH], sC[GSR_BATCH];
    __shared__ uint32_t sQ[GSR_BATCH];
    __shared__ uint16_t sList[4][GSR_BATCH];

    const int tile = gsr_tile_of_block(blockIdx.x, T);
    const int tx = tile % gx, ty = tile / gx;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int px = tx * 16 + (wave & 1) * 8 + (lane & 7);
    const int py = ty * 16 + (wave >> 1) * 8 + (lane >> 3);
    const float pxf = (float)px, pyf = (float)py;
    const uint2 rg = ranges[tile];
    const int n = (int)(rg.y - rg.x);
    const bool inside = px < W && py < H;

    bool done = !inside;
    float Tr = 1.0f, C0 = 0.f, C1 = 0.f, C2 = 0.f, Dp = 0.f, Uf = 0.f;
    uint32_t last = 0;
    uint16_t* mylist = sList[wave];

    for (int base = 0; base < n; base += GSR_BATCH) {
        if (__syncthreads_and(done)) break;  // also fences the previous batch's LDS reads
        const int cnt = min(GSR_BATCH, n - base);
        if (t < cnt) {
            const float4* r = reinterpret_cast<const float4*>(rec + point_list[rg.x + base + t]);
            const float4 a = r[0], b = r[1], c = r[2];
            sA[t] = a; sB[t] = b; sC[t] = c;
            sQ[t] = gsr_quadrant_mask(a, b, gsr_cull_tau_fast(b.y), tx, ty, W, H);
        }
        __syncthreads();
        const int nw = gsr_compact(sQ, mylist, cnt, wave, lane, [](int) { return true; });
        __builtin_amdgcn_wave_barrier();

#if GSR_FWD_READLANE
        // Each lane fetches ONE instance of the wave's list from LDS (3 x ds_read_b128 serve 64 instances);
        // the k-loop then broadcasts lane k's operands with v_readlane instead of re-reading LDS per instance.
        for (int c0 = 0; c0 < nw; c0 += 64) {
            if (__all(done)) break;  // wave-uniform
            const int mine = min(c0 + lane, nw - 1);
            const int jj = mylist[mine];
            const float4 a = sA[jj], b = sB[jj], c = sC[jj];
            const int m = min(64, nw - c0);
            for (int k = 0; k < m; k++) {
                if (__all(done)) break;  // wave-uniform
                const int j = __builtin_amdgcn_readlane(jj, k);
                const float mx = gsr_bcast(a.x, k), my = gsr_bcast(a.y, k);
                const float cA = gsr_bcast(a.z, k), cB = gsr_bcast(a.w, k), cC = gsr_bcast(b.x, k), op = gsr_bcast(b.y, k);
#define GSR_FWD_C(i_) gsr_bcast(i_ == 0 ? c.x : i_ == 1 ? c.y : i_ == 2 ? c.z : i_ == 3 ? b.z : b.w, k)
#else
        {
            for (int k = 0; k < nw; k++) {
                if (__all(done)) break;  // wave-uniform
                const int j = mylist[k];
                const float4 A = sA[j], B = sB[j];
                const float mx = A.x, my = A.y, cA = A.z, cB = A.w, cC = B.x, op = B.y;
#define GSR_FWD_C(i_) (i_ == 0 ? sC[j].x : i_ == 1 ? sC[j].y : i_ == 2 ? sC[j].z : i_ == 3 ? B.z : B.w)
#endif
                const float dx = mx - pxf, dy = my - pyf;
                const float power = -0.5f * (cA * dx * dx + cC * dy * dy) - cB * dx * dy;
                const float alpha = fminf(0.99f, op * GSR_EXP(power));
                bool ok = !done && power <= 0.0f && alpha >= (1.0f / 255.0f);
                if (!__any(ok)) continue;  // wave-uniform
                const float test_T = Tr * (1.0f - alpha);
                const bool stop = ok && test_T < 0.0001f;
                done = done || stop;
                ok = ok && !stop;
                const float w = ok ? alpha * Tr : 0.0f;
                C0 += GSR_FWD_C(0) * w; C1 += GSR_FWD_C(1) * w; C2 += GSR_FWD_C(2) * w;
                Dp += GSR_FWD_C(3) * w; Uf += GSR_FWD_C(4) * w;
                Tr = ok ? test_T : Tr;
                last = ok ? (uint32_t)(base + j + 1) : last;
            }
        }
#undef GSR_FWD_C
    }

    if (inside) {
        const size_t HW = (size_t)H * W, pid = (size_t)py * W + px;
        final_T[pid] = Tr;
        n_contrib[pid] = last;
        out_color[pid] = C0 + Tr * bg[0];
        out_color[HW + pid] = C1 + Tr * bg[1];
        out_color[2 * HW + pid] = C2 + Tr * bg[2];
        out_depth[pid] = Dp;
        out_feature[pid] = Uf;
    }
}

// ---------------------------------------------------------------------------------------------
// Backward
// ---------------------------------------------------------------------------------------------
template <bool AUX>
__global__ void __launch_bounds__(256) gsr_blend_bwd_kernel(
    const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list, const GsrRec* __restrict__ rec, int W,
    int H, int gx, int T, const float* __restrict__ bg, const float* __restrict__ final_T,
    const uint32_t* __restrict__ n_contrib, const float* __restrict__ dL_dcolor, const float* __restrict__ dL_ddepth,
    const float* __restrict__ dL_dfeature, float4* __restrict__ slots)
{
    __shared__ float4 sA[GSR_BATCH], sB[GSR_BATCH], sC[GSR_BATCH];
    __shared__ __attribute__((aligned(16))) float acc[GSR_BATCH * GSR_SLOT_FLOATS];
    __shared__ uint32_t sSlot[GSR_BATCH], sQ[GSR_BATCH];
    __shared__ uint16_t sList[4][GSR_BATCH];
    __shared__ int sMax;

    const int tile = gsr_tile_of_block(blockIdx.x, T);
    const int tx = tile % gx, ty = tile / gx;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const uint2 rg = ranges[tile];
    const int n = (int)(rg.y - rg.x);
    if (n == 0) return;

    const int px = tx * 16 + (wave & 1) * 8 + (lane & 7);
    const int py = ty * 16 + (wave >> 1) * 8 + (lane >> 3);
    const float pxf = (float)px, pyf = (float)py;
    const float ddelx_dx = 0.5f * W, ddely_dy = 0.5f * H;
    const size_t HW = (size_t)H * W;
    const bool inside = px < W && py < H;
    const size_t pid = inside ? (size_t)py * W + px : 0;

    const float Tf = inside ? final_T[pid] : 0.f;
    float Tr = Tf;
    const int lastc = inside ? (int)n_contrib[pid] : 0;
    const float g0 = inside ? dL_dcolor[pid] : 0.f, g1 = inside ? dL_dcolor[HW + pid] : 0.f;
    const float g2 = inside ? dL_dcolor[2 * HW + pid] : 0.f;
    float gd = 0.f, gu = 0.f;
    if (AUX && inside) { gd = dL_ddepth[pid]; gu = dL_dfeature[pid]; }
    const float bgdot = bg[0] * g0 + bg[1] * g1 + bg[2] * g2;
    float ar0 = 0.f, ar1 = 0.f, ar2 = 0.f, ard = 0.f, aru = 0.f;
    float la = 0.f, lc0 = 0.f, lc1 = 0.f, lc2 = 0.f, lcd = 0.f, lcu = 0.f;

    // wave-level and block-level maxima of the last contributor: nothing at a position >= them is blended
    int wmax = lastc;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) wmax = max(wmax, __shfl_xor(wmax, d, 64));
    if (t == 0) sMax = 0;
    for (int i = t; i < GSR_BATCH * GSR_SLOT_FLOATS; i += 256) acc[i] = 0.f;
    __syncthreads();
    if (lane == 0) atomicMax(&sMax, wmax);
    __syncthreads();
    const int nproc = min(n, sMax);
    uint16_t* mylist = sList[wave];

    // back to front, in batches of 256 instances; local j = 0 is the backmost instance of the batch
    for (int hi = n; hi > 0; hi -= GSR_BATCH) {
        const int lo = max(0, hi - GSR_BATCH), cnt = hi - lo;
        const bool active = lo < nproc;
        if (t < cnt) {
            const GsrRec* r = rec + point_list[rg.x + (hi - 1 - t)];
            const uint4 d = r->d;
            const float4 c = r->c;
            // gradient slot = Gaussian's scan offset + rank of this tile among the surviving tiles of its rectangle
            const int x0 = d.y & 0xffff, y0 = d.y >> 16, wd = (int)__float_as_uint(c.w);
            const int pos = (ty - y0) * wd + (tx - x0);
            const unsigned long long mask = ((unsigned long long)d.w << 32) | d.z;
            sSlot[t] = d.x + (uint32_t)(pos < 64 ? __popcll(mask & ((1ull << pos) - 1ull)) : __popcll(mask) + (pos - 64));
            if (active) {
                const float4 a = r->a, b = r->b;
                sA[t] = a; sB[t] = b; sC[t] = c;
                sQ[t] = gsr_quadrant_mask(a, b, gsr_cull_tau_fast(b.y), tx, ty, W, H);
            }
        }
        __syncthreads();

        if (active) {
            // instance j sits at list position p = hi-1-j; this wave needs it only if p < wmax
            const int nw = gsr_compact(sQ, mylist, cnt, wave, lane, [=](int i) { return hi - 1 - i < wmax; });
            __builtin_amdgcn_wave_barrier();
#if GSR_BWD_READLANE
            for (int c0 = 0; c0 < nw; c0 += 64) {
                const int mine = min(c0 + lane, nw - 1);
                const int jj = mylist[mine];
                const float4 a = sA[jj], b = sB[jj], c = sC[jj];
                const int m = min(64, nw - c0);
                for (int k = 0; k < m; k++) {
                    const int j = __builtin_amdgcn_readlane(jj, k);
                    const float mx = gsr_bcast(a.x, k), my = gsr_bcast(a.y, k);
                    const float cA = gsr_bcast(a.z, k), cB = gsr_bcast(a.w, k), cC = gsr_bcast(b.x, k), op = gsr_bcast(b.y, k);
#define GSR_BWD_C(i_) gsr_bcast(i_ == 0 ? c.x : i_ == 1 ? c.y : i_ == 2 ? c.z : i_ == 3 ? b.z : b.w, k)
#else
            {
                for (int k = 0; k < nw; k++) {
                    const int j = mylist[k];
                    const float4 A = sA[j], B = sB[j];
                    const float mx = A.x, my = A.y, cA = A.z, cB = A.w, cC = B.x, op = B.y;
#define GSR_BWD_C(i_) (i_ == 0 ? sC[j].x : i_ == 1 ? sC[j].y : i_ == 2 ? sC[j].z : i_ == 3 ? B.z : B.w)
#endif
                    const int p = hi - 1 - j;
                    const float dx = mx - pxf, dy = my - pyf;
                    const float power = -0.5f * (cA * dx * dx + cC * dy * dy) - cB * dx * dy;
                    const float G = GSR_EXP(power);
                    const float alpha = fminf(0.99f, op * G);
                    const bool ok = p < lastc && power <= 0.0f && alpha >= (1.0f / 255.0f);
                    if (!__any(ok)) continue;  // wave-uniform: no pixel of this quadrant blends the instance

                    float s[11];
#pragma unroll
                    for (int v = 0; v < 11; v++) s[v] = 0.f;
                    const float c0r = GSR_BWD_C(0), c1r = GSR_BWD_C(1), c2r = GSR_BWD_C(2);
                    if (ok) {  // divergent: executed under the EXEC mask of the lanes that blend
                        const float rinv = GSR_RCP(1.0f - alpha);
                        const float Tn = Tr * rinv;  // T / (1 - alpha)
                        const float w = alpha * Tn;
                        const float oml = 1.0f - la;
                        ar0 = la * lc0 + oml * ar0; ar1 = la * lc1 + oml * ar1; ar2 = la * lc2 + oml * ar2;
                        float dL_dalpha = (c0r - ar0) * g0 + (c1r - ar1) * g1 + (c2r - ar2) * g2;
                        if (AUX) {
                            const float cdr = GSR_BWD_C(3), cur = GSR_BWD_C(4);
                            ard = la * lcd + oml * ard; aru = la * lcu + oml * aru;
                            dL_dalpha += (cdr - ard) * gd + (cur - aru) * gu;
                            s[3] = w * gd; s[4] = w * gu;
                            lcd = cdr; lcu = cur;
                        }
                        dL_dalpha *= Tn;
                        dL_dalpha += (-Tf * rinv) * bgdot;
                        const float dL_dG = op * dL_dalpha;
                        const float gdx = G * dx, gdy = G * dy;
                        const float dG_ddelx = -gdx * cA - gdy * cB;
                        const float dG_ddely = -gdy * cC - gdx * cB;
                        s[0] = w * g0; s[1] = w * g1; s[2] = w * g2;
                        s[5] = dL_dG * dG_ddelx * ddelx_dx; s[6] = dL_dG * dG_ddely * ddely_dy;
                        s[7] = -0.5f * gdx * dx * dL_dG; s[8] = -0.5f * gdx * dy * dL_dG; s[9] = -0.5f * gdy * dy * dL_dG;
                        s[10] = G * dL_dalpha;
                        Tr = Tn; la = alpha;
                        lc0 = c0r; lc1 = c1r; lc2 = c2r;
                    }
#if GSR_RED_MODE == 0 || GSR_RED_MODE == 1
#pragma unroll
                    for (int v = 0; v < 11; v++)
                        if (AUX || (v != 3 && v != 4)) s[v] = GSR_RED_MODE == 1 ? gsr_wave_sum_to_row3(s[v]) : gsr_row_sum16(s[v]);
                    if (GSR_RED_MODE == 1 ? lane == 63 : (lane & 15) == 0) {
                        float* ac = acc + j * GSR_SLOT_FLOATS;
#pragma unroll
                        for (int v = 0; v < 11; v++)
                            if (AUX || (v != 3 && v != 4)) atomicAdd(ac + v, s[v]);
                    }
#else
                    // Row totals of every value (4 DPP adds each), then lane i of each row keeps value i, so that
                    // ONE ds_add_f32 with 11 distinct addresses replaces 9-11 LDS atomics (an LDS atomic
                    // instruction costs ~13 LDS cycles whatever its lane count -- measured, profiles/).
                    float x = 0.f;
#pragma unroll
                    for (int v = 0; v < 11; v++)
                        if (AUX || (v != 3 && v != 4)) {
                            const float rs = gsr_row_sum16(s[v]);
                            x = (lane & 15) == v ? rs : x;
                        }
#if GSR_RED_MODE == 2
                    // lane-aligned cross-row adds (DPP cannot move a lane across rows): v_permlane32_swap brings
                    // lanes 32..63 under lanes 0..31, v_permlane16_swap brings row 1 under row 0 (gfx950)
                    x += __uint_as_float(__builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false)[1]);
                    x += __uint_as_float(__builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false)[1]);
                    if (lane < 11) atomicAdd(acc + j * GSR_SLOT_FLOATS + lane, x);
#else
                    if ((lane & 15) < 11) atomicAdd(acc + j * GSR_SLOT_FLOATS + (lane & 15), x);
#endif
#endif
                }
            }
#undef GSR_BWD_C
        }
        __syncthreads();
        if (t < cnt) {
            float4* a4 = reinterpret_cast<float4*>(acc + t * GSR_SLOT_FLOATS);
            float4* dst = slots + (size_t)sSlot[t] * 3;
            dst[0] = a4[0]; dst[1] = a4[1]; dst[2] = a4[2];
            if (active) { a4[0] = a4[1] = a4[2] = make_float4(0.f, 0.f, 0.f, 0.f); }
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
hipError_t gsr_launch_blend_forward(int W, int H, int gx, int T, const float* bg, const GsrGeom& geom,
                                    const GsrImage& image, const GsrBinning& bin, float* out_color, float* out_depth,
                                    float* out_feature, hipStream_t stream)
{
    if (T <= 0) return hipSuccess;
    hipLaunchKernelGGL(gsr_blend_fwd_kernel, dim3(T), dim3(256), 0, stream, image.ranges, bin.point_list, geom.rec, W, H,
                       gx, T, bg, out_color, out_depth, out_feature, image.final_T, image.n_contrib);
    return hipGetLastError();
}

hipError_t gsr_launch_blend_backward(int W, int H, int gx, int T, const float* bg, const GsrGeom& geom,
                                     const GsrImage& image, const GsrBinning& bin, const float* dL_dcolor,
                                     const float* dL_ddepth, const float* dL_dfeature, float* slots, hipStream_t stream)
{
    if (T <= 0) return hipSuccess;
    float4* s4 = reinterpret_cast<float4*>(slots);
    if (dL_ddepth && dL_dfeature)
        hipLaunchKernelGGL(gsr_blend_bwd_kernel<true>, dim3(T), dim3(256), 0, stream, image.ranges, bin.point_list,
                           geom.rec, W, H, gx, T, bg, image.final_T, image.n_contrib, dL_dcolor, dL_ddepth, dL_dfeature, s4);
    else
        hipLaunchKernelGGL(gsr_blend_bwd_kernel<false>, dim3(T), dim3(256), 0, stream, image.ranges, bin.point_list,
                           geom.rec, W, H, gx, T, bg, image.final_T, image.n_contrib, dL_dcolor, nullptr, nullptr, s4);
    return hipGetLastError();
}
