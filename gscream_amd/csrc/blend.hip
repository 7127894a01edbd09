// blend.hip -- tile-based forward / backward alpha blend of RGB + depth + one feature channel.
//
// Replaces DGR forward.cu:441-568 renderCUDA (fwd) and backward.cu:409-604 renderCUDA (bwd).
//
// Mapping onto CDNA4.  16x16 screen tiles as in the reference; wavefront = one 8x8 pixel QUADRANT of a tile, lane l
// the pixel (l & 7, l >> 3) inside it.
//   * Forward: 4T independent single-wave workgroups (see gsr_blend_fwd_kernel).  Each walks its tile's list on its own
//     in batches of 64 instances, tests every instance against ITS quadrant box (gsr_box_min_q: can alpha reach 1/255
//     anywhere in the box?), writes the survivors in list order into LDS as PAIR records and evaluates two instances
//     per loop iteration with packed fp32 instructions.  No workgroup barriers, no waiting for sibling quadrants.
//     The reference tests every instance of the tile against all 256 pixels (forward.cu:513-553) and re-reads colour
//     and depth from GLOBAL memory per contribution (forward.cu:545-546).
//   * Backward: one 128-thread workgroup per DEPTH SEGMENT of a tile (see gsr_blend_bwd_kernel): the forward stores,
//     per pixel and every 64 / 128 list positions, the transmittance and the sums of the segment that ends there, so a
//     segment can be walked back to front without the ones behind it.  Wavefront w owns the 16x8 strip w of the tile,
//     lane l two pixels of it; the per-pixel arithmetic runs on float2 (v_pk_*).  Thread i gathers the record of
//     instance i into LDS and computes the 4-bit quadrant mask once per instance; each wave compacts, with ballot +
//     mbcnt, the indices of the instances that can reach ITS strip into a private LDS list and walks only those.
//     Operands come back with same-address (broadcast) ds_read_b128, software-pipelined one instance ahead; the list
//     index comes from a per-lane copy via v_readlane (no dependent LDS read).
//   * Early-out: a forward wave stops when all its pixels are saturated; a backward wave skips gradient math +
//     reduction when none of its pixels blends the instance, and instances behind the tile's deepest contributor are
//     not even loaded.
//   * Backward gradient scatter: the reference issues 11 global atomicAdd per (pixel, Gaussian) contribution
//     (backward.cu:554-601).  Here each lane forms 9 (11) partials (colour, depth, feature and six moments of
//     G dL/dalpha), the wave reduces them with a select-free transposing butterfly (gsr_bank_reduce + permlane swaps,
//     23-26 VALU operations) and 9 (11) lanes add into an LDS accumulator [128][12] with ONE ds_add_f32 (built with
//     -amdgpu-atomic-optimizer-strategy=None so it stays one instruction).  After the batch, thread i stores the
//     12 floats of instance i with three plain 16-byte stores into that instance's private gradient slot (slot =
//     Gaussian's scan offset + rank of the tile among the surviving tiles of its rectangle); gauss_bwd.hip sums each
//     Gaussian's slots.  No atomics on global memory at all; each of a task's 2 wavefronts has its own LDS accumulator and
//     the flush adds them in a fixed order, so the backward is bit-reproducible.
//   * AUX = false specialises the backward for "no gradient flows into the depth and feature maps" (GScream's
//     RGB-only iterations): 9 instead of 11 reductions and no depth/feature recurrences.
//   * XCD awareness: workgroup b runs on XCD b % 8 (observed dispatch rule); the block->tile maps hand each
//     XCD a contiguous band of tile rows so neighbouring tiles, which share most of their Gaussians, hit the
//     same 4 MiB L2.  Pure speed: any placement gives the same result.
//   * Scheduling (measured with the `make trace` build, tools/wave_trace.py): nearly all wavefronts of a launch are
//     resident from the start, so a launch lasts as long as its most loaded SIMD.  Hence independent quadrant waves in
//     the forward and uniform depth-segment tasks, several per wavefront slot, in the backward.
#include <stdlib.h>
#include <algorithm>
#include "gsr_math.h"

// Fast-math knobs of the blend inner loops.  Default: exp via v_exp_f32 (the records hold the quadratic form
// pre-multiplied by log2 e, so the exponent goes straight into the instruction; ~1 ulp) and
// T/(1-alpha) via v_rcp_f32 (1 ulp).  -DGSR_PRECISE_MATH selects libm-accurate expf and IEEE division
// (diagnostic build, used to attribute parity differences; not shipped).
// PARITY build (-DGSR_PRECISE_MATH, libgsraster_precise.so, compiled with -ffp-contract=off): the records keep the raw
// conic (preprocess.hip) and the falloff is the reference's own expression, power = -0.5 (A dx^2 + C dy^2) - B dx dy,
// alpha = min(0.99, o * expf(power)) (DGR forward.cu:528-533, backward.cu:516-524), with libm-accurate expf and IEEE
// division -- no pre-scaling, no v_exp_f32, no v_rcp_f32, no FMA contraction.  Used by tests/test_gpu_precise.py to show
// that the outliers of the shipped build against the oracle are alpha = 1/255 / T = 1e-4 threshold flips.
#define GSR_QSCALE(v) ((v) * GSR_LOG2E)     // raw conic entry -> entry of the base-2 quadratic form used by the box tests
#define GSR_ALPHA_C (1.0f / 255.0f)        // the blend threshold of forward.cu:534
#ifdef GSR_PRECISE_MATH
#define GSR_RCP(x) (1.0f / (x))
__device__ __forceinline__ float gsr_power1(float cA, float cB, float cC, float dx, float dy) { return -0.5f * (cA * dx * dx + cC * dy * dy) - cB * dx * dy; }
__device__ __forceinline__ float gsr_gauss1(float power) { return expf(power); }
#else
#define GSR_RCP(x) __builtin_amdgcn_rcpf(x)
__device__ __forceinline__ float gsr_gauss1(float power) { return __builtin_amdgcn_exp2f(power); }
// Raw conic of the records -> the pre-scaled quadratic form of the blend loops (once per staged instance)
#define GSR_HA(conA) ((conA) * (-0.5f * GSR_LOG2E))
#define GSR_HB(conB) ((conB) * (-GSR_LOG2E))
#define GSR_HC(conC) ((conC) * (-0.5f * GSR_LOG2E))
// GUARD BAND around the alpha >= 1/255 decision (round 4).  The fast form's alpha differs from the reference expression's by
// a few 1e-7 relative (pre-scaled quadratic form, FMA contraction, v_exp_f32), so a (pixel, Gaussian) pair whose alpha lies
// that close to 1/255 may be blended by one implementation and skipped by the other -- the "threshold flips" that were the
// shipped build's only outliers against the oracle.  Both blend loops now accept candidates from
// (1 - GSR_BAND)/255 on and, when a candidate lies below (1 + GSR_BAND)/255 -- a wave-uniform, rare branch: ~1e-5 of the
// evaluated pairs -- decide it with the reference's own expression on the raw conic (gsr_blends_exact: forward.cu:528-534 /
// backward.cu:516-524, no contraction, accurate expf).  Forward and backward evaluate alpha with the same instructions, so
// they agree on band membership and on the decision.
#ifndef GSR_BAND
#define GSR_BAND 3.0e-5f
#endif
#define GSR_ALPHA_LO (GSR_ALPHA_C * (1.0f - GSR_BAND))
#define GSR_ALPHA_HI (GSR_ALPHA_C * (1.0f + GSR_BAND))
// (A band that widens PER INSTANCE with the magnitude of the quadratic form's terms -- the reference's own fp32 `power` carries ~2e-7 x
// that magnitude of absolute error, which for large thin tilted splats exceeds the fixed band -- was built and measured in round 5:
// tools/experiments/r05_per_instance_guard_band.patch.  It cost config 2 +2 us in either blend and config 4 +11 / +8 us, and removed no
// outlier on any workload: what is left beyond 1e-3 lies inside the reference's own summation-order noise or is an expf tie)
__device__ __forceinline__ bool gsr_blends_exact(const float cA, const float cB, const float cC, const float op, const float dx, const float dy)
{
#pragma clang fp contract(off)
    const float power = -0.5f * (cA * dx * dx + cC * dy * dy) - cB * dx * dy;
    const float alpha = fminf(0.99f, op * expf(power));
    return !(power > 0.0f) && !(alpha < 1.0f / 255.0f);
}
// EXACT REPLAY of the pixels that came near the T = 1e-4 stop (round 5).  The transmittance is accumulated state: the fast path's
// T differs from the reference chain's by the accumulated alpha errors (measured: tools/t_chain_error.py), so a pixel whose test_T
// lands that close to 1e-4 (forward.cu:537) may stop one instance earlier or later than the reference -- the "T flips" that were
// the shipped build's last decision-type outliers.  A guard band cannot settle them in place (the whole chain in front of the
// decision has to be the reference's), so the forward REMEMBERS which of its pixels came within GSR_TBAND (relative) of the stop --
// at the stop itself (test_T just below 1e-4: one rarely taken branch per blend) and at the end of the walk (final T just above) --
// and the quadrant wave then walks those pixels' lists again, all 64 lanes on one pixel: lane i evaluates instance i of a batch with
// the reference's own expressions (raw conic, accurate expf, no contraction: forward.cu:528-534), the T chain runs over the
// blending lanes in list order exactly as forward.cu:536-549 rounds it, and the pixel's outputs, final T, contributor count and
// depth checkpoints are replaced.  ~GSR_TBAND * 1e7 pixels per 1008x567 frame.
#ifndef GSR_TBAND
#define GSR_TBAND 1.0e-4f
#endif
struct GsrExactAlpha { float alpha, one_minus; bool blends, in_band; };
__device__ __forceinline__ GsrExactAlpha gsr_alpha_exact(const float cA, const float cB, const float cC, const float op, const float dx, const float dy)
{
#pragma clang fp contract(off)
    GsrExactAlpha r;
    const float power = -0.5f * (cA * dx * dx + cC * dy * dy) - cB * dx * dy;
    const float raw = op * expf(power);
    r.alpha = fminf(0.99f, raw);
    r.blends = !(power > 0.0f) && !(r.alpha < 1.0f / 255.0f);
    r.one_minus = 1.0f - r.alpha;
    // (twice the guard band's width: the backward decides band membership on the FAST alpha, a few 1e-7 away from this one)
    // (twice the guard band's width: the backward decides band membership on the FAST alpha, a few 1e-7 away from this one)
    r.in_band = !(power > 0.0f) && fabsf(raw - GSR_ALPHA_C) < GSR_ALPHA_C * (2.0f * GSR_BAND);
    return r;
}
__device__ __forceinline__ float gsr_mul_exact(const float a, const float b)
{
#pragma clang fp contract(off)
    return a * b;
}
// Sum over the wavefront, result wave-uniform: quad and row steps as DPP adds, rows combined with row_bcast:15 / :31, lane 63
// read back (six short-latency VALU steps; the __shfl_xor butterfly is six dependent LDS-crossbar round trips).
__device__ __forceinline__ float gsr_wave_sum(float v)
{
    auto dpp = [](const float x, const int ctrl_tag) -> float {
        switch (ctrl_tag) {
            case 0: return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0xB1, 0xf, 0xf, true));    // quad_perm [1,0,3,2]
            case 1: return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x4E, 0xf, 0xf, true));    // quad_perm [2,3,0,1]
            case 2: return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x124, 0xf, 0xf, true));   // row_ror:4
            case 3: return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x128, 0xf, 0xf, true));   // row_ror:8
            case 4: return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x142, 0xa, 0xf, false));  // row_bcast:15 -> rows 1, 3
            default: return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x143, 0xc, 0xf, false)); // row_bcast:31 -> rows 2, 3
        }
    };
    v += dpp(v, 0); v += dpp(v, 1); v += dpp(v, 2); v += dpp(v, 3);   // every lane: its row's sum
    v += dpp(v, 4); v += dpp(v, 5);                                   // lane 63: the wave's sum
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
#endif

// -DGSR_TRACE (diagnostic build `make trace`, not shipped): every wavefront of the backward blend records its start and
// end time (100 MHz wall clock) and the hardware slot it ran on (HW_ID, XCC_ID); tools/wave_trace.py reads them back.
#ifdef GSR_TRACE
__device__ unsigned long long gsr_trace_buf[2 * 4 * 4 * 36864];  // [backward | forward]
#define GSR_TRACE_BEGIN const unsigned long long gsr_tr0 = wall_clock64();
#define GSR_TRACE_END(NW) GSR_TRACE_END_AT(NW, 0)
#define GSR_TRACE_END_AT(NW, BASE)                                                             \
    if ((threadIdx.x & 63) == 0) {                                                             \
        const size_t o = (size_t)(BASE) + ((size_t)blockIdx.x * (NW) + (threadIdx.x >> 6)) * 4; \
        gsr_trace_buf[o] = gsr_tr0; gsr_trace_buf[o + 1] = wall_clock64();                     \
        gsr_trace_buf[o + 2] = __builtin_amdgcn_s_getreg(63492); /* HW_ID, 32 bits */          \
        gsr_trace_buf[o + 3] = __builtin_amdgcn_s_getreg(63508); /* XCC_ID */                  \
    }
extern "C" int gsr_debug_trace(void* host_dst, size_t bytes)
{
    return (int)hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(gsr_trace_buf), bytes);
}
#else
#define GSR_TRACE_BEGIN
#define GSR_TRACE_END(NW)
#define GSR_TRACE_END_AT(NW, BASE)
#endif

// -DGSR_COUNT (diagnostic build, `make variant SRC=blend NAME=count FLAGS=-DGSR_COUNT`; tools/blend_counts.py): how many
// (instance, strip) iterations the backward runs, how many of them blend at least one pixel, and how many pixels blend.
#ifdef GSR_COUNT
__device__ unsigned long long gsr_cnt[8];
#define GSR_COUNT_ADD(i, v) do { if ((threadIdx.x & 63) == 0) atomicAdd(&gsr_cnt[i], (unsigned long long)(v)); } while (0)
extern "C" int gsr_debug_counters(void* host_dst, int reset)
{
    hipError_t e = hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(gsr_cnt), sizeof(gsr_cnt));
    if (e == hipSuccess && reset) { unsigned long long z[8] = {}; e = hipMemcpyToSymbol(HIP_SYMBOL(gsr_cnt), z, sizeof(z)); }
    return (int)e;
}
#else
#define GSR_COUNT_ADD(i, v)
#endif

// Lane selects on wave-uniform 64-bit masks (compares write SGPR pairs, the logic between them is scalar).
__device__ __forceinline__ float gsr_sel(unsigned long long m, float if_set, float if_clear) { return __builtin_amdgcn_inverse_ballot_w64(m) ? if_set : if_clear; }
__device__ __forceinline__ float gsr_sel0(unsigned long long m, float if_set) { return __builtin_amdgcn_inverse_ballot_w64(m) ? if_set : 0.0f; }

template <int CTRL>
__device__ __forceinline__ float gsr_dpp(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
// Row-level part of the transposing wave reduction of the backward blend.  A pair step merges two registers (a, b)
// into one whose even banks (4-lane groups of a DPP row) hold a + rot(a) and whose odd banks hold b + rot(b): two
// v_add_f32_dpp with complementary bank masks, no select.  The second step (row_ror:8) does the same with the bank
// pairs {0,1} / {2,3}.  After both, bank k of the result holds, in each of its four lanes, the sum over the row's lanes
// of equal (lane & 3) of one value (which value sits in which bank: gsr_bank_reduce_dyf below).
// Inline assembly because the masked write of a DPP destination has no IR form; the s_nop cover the VALU-write ->
// DPP-read hazard against the surrounding compiler-scheduled code (inside the block producers and consumers are
// at least two instructions apart).
#define GSR_PAIR4(dst, a, b)                                                     \
    "v_add_f32_dpp " dst ", " a ", " a " row_ror:4 row_mask:0xf bank_mask:0x5\n\t" \
    "v_add_f32_dpp " dst ", " b ", " b " row_ror:4 row_mask:0xf bank_mask:0xa\n\t"
#define GSR_PAIR8(dst, a, b)                                                     \
    "v_add_f32_dpp " dst ", " a ", " a " row_ror:8 row_mask:0xf bank_mask:0x3\n\t" \
    "v_add_f32_dpp " dst ", " b ", " b " row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
// The backward's lane map is  pixel row = 2 (lane >> 4) + (lane & 1), column = (lane >> 1) & 7:  the four lanes a DPP row's
// bank steps add up (l, l+4, l+8, l+12) then share their pixel row, i.e. dy, so the three dy-weighted moments need not
// enter the butterfly at all -- they are dy and dy^2 times the first step's sums of (sum g, sum g dx), formed between the
// two steps: 6 (8) values are reduced instead of 9 (11)  (round 3: 161 -> 155 us).
//   non-AUX: q0 = {c0, c1, c2, Mxx}  q1 = {M0, Mx, My, Mxy}  q2 = {Myy, -, -, -}
//   AUX:     q0 = {c0, c1, c2, d}    q1 = {u, Mxx, M0, Mx}   q2 = {My, Mxy, Myy, -}
template <bool AUX>
__device__ __forceinline__ void gsr_bank_reduce_dyf(float c0, float c1, float c2, float d, float u, float sx, float mxx, float gs,
                                                    float dy, float& q0, float& q1, float& q2)
{
    float r0, r1, r2, r3, r4, r5;
    if (AUX) {
        asm volatile("s_nop 1\n\t" GSR_PAIR4("%0", "%4", "%5") GSR_PAIR4("%1", "%6", "%7") GSR_PAIR4("%2", "%8", "%9")
                     GSR_PAIR4("%3", "%10", "%11") "s_nop 1"
                     : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3)
                     : "v"(c0), "v"(c1), "v"(c2), "v"(d), "v"(u), "v"(mxx), "v"(gs), "v"(sx));
        r4 = r3 * dy;   // even banks: dy sum g (My), odd banks: dy sum g dx (Mxy)
        r5 = r4 * dy;   // even banks: dy^2 sum g (Myy)
        asm volatile("s_nop 1\n\t" GSR_PAIR8("%0", "%3", "%4") GSR_PAIR8("%1", "%5", "%6") GSR_PAIR8("%2", "%7", "%8") "s_nop 1"
                     : "=&v"(q0), "=&v"(q1), "=&v"(q2)
                     : "v"(r0), "v"(r1), "v"(r2), "v"(r3), "v"(r4), "v"(r5));
    } else {
        asm volatile("s_nop 1\n\t" GSR_PAIR4("%0", "%3", "%4") GSR_PAIR4("%1", "%5", "%6") GSR_PAIR4("%2", "%7", "%8") "s_nop 1"
                     : "=&v"(r0), "=&v"(r1), "=&v"(r2)
                     : "v"(c0), "v"(c1), "v"(c2), "v"(mxx), "v"(gs), "v"(sx));
        r3 = r2 * dy;
        r4 = r3 * dy;
        asm volatile("s_nop 1\n\t" GSR_PAIR8("%0", "%3", "%4") GSR_PAIR8("%1", "%5", "%6")
                     "v_add_f32_dpp %2, %7, %7 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
                     "s_nop 1"
                     : "=&v"(q0), "=&v"(q1), "=&v"(q2)
                     : "v"(r0), "v"(r1), "v"(r2), "v"(r3), "v"(r4));
    }
}

// The instance loops are software-pipelined: the wave's list indices sit one per lane (v_readlane instead of a
// dependent LDS read) and the operands of instance k+1 are in flight while instance k is computed (measured winner
// against the plain loop, profiles/).

// 4-bit mask of the 8x8 quadrants of tile (tx, ty) in which the Gaussian can reach alpha >= 1/255.
__device__ __forceinline__ uint32_t gsr_quadrant_mask(const float4 A, const float4 B, const float tau, int tx, int ty,
                                                      int W, int H)
{
    // q2 = log2(e) q = 0.5 (a dx^2 + c dy^2) + b dx dy with (a, b, c) = log2(e) conic; the threshold scales the same way
    const float ca = GSR_QSCALE(A.z), cb = GSR_QSCALE(A.w), cc = GSR_QSCALE(B.x);  // A, B: the record's raw conic
    const float rA = GSR_RCP(ca), rC = GSR_RCP(cc);
    uint32_t m = 0;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int x0 = tx * 16 + (q & 1) * 8, y0 = ty * 16 + (q >> 1) * 8;
        const float bx1 = (float)min(x0 + 7, W - 1), by1 = (float)min(y0 + 7, H - 1);
        const bool hit = x0 < W && y0 < H &&
                         !(gsr_box_min_q(A.x, A.y, ca, cb, cc, rA, rC, (float)x0, bx1, (float)y0, by1) > tau);
        m |= hit ? (1u << q) : 0u;
    }
    return m;
}

// ---------------------------------------------------------------------------------------------
// Forward, one independent wavefront per 8x8 quadrant, two instances per loop iteration.
//
// A 64-thread workgroup owns ONE quadrant of a tile and walks the tile's list on its own: lane i fetches the record of
// instance i of the current 64-instance batch (the next batch's loads are in flight while this one is blended), tests
// it against the quadrant box, and the survivors are written, in list order, to LDS as PAIR records
//     {x_a, x_b, y_a, y_b} {hA_a, hA_b, hB_a, hB_b} {hC_a, hC_b, op_a, op_b} {i_a, i_b, feature_a, feature_b}
// so that four same-address ds_read_b128 hand the alpha evaluation of two instances to v_pk_* instructions
// (16 VALU operations for two instances instead of 14 for one) and give each wave two independent dependency chains.
// Nothing is shared between the quadrants of a tile: no workgroup barriers, a quadrant whose pixels are saturated stops
// while its siblings go on, and the launch consists of 4T independently scheduled wavefronts instead of T groups of
// four that wait for each other.  Each record is fetched by the four quadrant waves of its tile (L2 hits: the four
// run on the same XCD).
// ---------------------------------------------------------------------------------------------
#define GSR_FWB 64
// Stop threshold of the forward's fast walk (forward.cu:537 tests test_T < 0.0001f).  With the exact replay (GSR_TBAND) the fast walk
// stops only below 1e-4 (1 - band): given that its T is within the band of the reference chain's, it then never stops BEFORE the
// reference does; where it blends an instance the reference would have stopped at, its T ends below 1e-4 (1 + band) -- so ONE
// compare at the end of the walk (final T < 1e-4 (1 + band)) finds every pixel whose stop could differ, at no cost inside the loop
// (a per-blend test of the stopping test_T cost the launch 5 %).
#if !defined(GSR_PRECISE_MATH) && !defined(GSR_NO_TREPLAY)
#define GSR_T_STOP (0.0001f * (1.0f - GSR_TBAND))
#else
#define GSR_T_STOP 0.0001f
#endif
// Dynamic LDS requested (and never touched) per forward workgroup: it caps how many quadrant waves are resident per CU, so
// that the rest of the 4T workgroups are handed out as earlier ones finish (the dispatcher then balances the SIMDs; with
// every wave resident from the start a launch lasts as long as its most loaded SIMD).  Compile-time experiment knob
// (make variant FLAGS=-DGSR_FWD_LDS_PAD=...; measured: every non-zero value is slower); the shipped library
// reads no environment variable.
#ifndef GSR_FWD_LDS_PAD
#define GSR_FWD_LDS_PAD 0
#endif
static_assert(GSR_FWD_LDS_PAD >= 0 && GSR_FWD_LDS_PAD <= 60 * 1024, "GSR_FWD_LDS_PAD out of range");
typedef float gsr_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ gsr_f2 gsr_splat(float v) { gsr_f2 r = {v, v}; return r; }
__device__ __forceinline__ gsr_f2 gsr_fma2(gsr_f2 a, gsr_f2 b, gsr_f2 c) { return __builtin_elementwise_fma(a, b, c); }

// Checkpoint planes (gsr_common.h): slot k of a pixel = float4 {T_k (last slot: checkpoints passed), r, g, b} + float2
// {depth, feature}; slot k < GSR_SEG_MAX-1 belongs to list position gsr_ckpt_pos(k) (three tiers, gsr_common.h), the last one to the end.
__device__ __forceinline__ size_t gsr_ckpt_stride(size_t HW) { return (HW + 3) & ~(size_t)3; }  // keeps the float4 slots aligned
__device__ __forceinline__ float4* gsr_ckpt_a(float* ckpt, int k, size_t HW) { return reinterpret_cast<float4*>(ckpt + (size_t)k * 6 * gsr_ckpt_stride(HW)); }
__device__ __forceinline__ float2* gsr_ckpt_b(float* ckpt, int k, size_t HW) { return reinterpret_cast<float2*>(ckpt + ((size_t)k * 6 + 4) * gsr_ckpt_stride(HW)); }
__device__ __forceinline__ const float4* gsr_ckpt_a(const float* ckpt, int k, size_t HW) { return reinterpret_cast<const float4*>(ckpt + (size_t)k * 6 * gsr_ckpt_stride(HW)); }
__device__ __forceinline__ const float2* gsr_ckpt_b(const float* ckpt, int k, size_t HW) { return reinterpret_cast<const float2*>(ckpt + ((size_t)k * 6 + 4) * gsr_ckpt_stride(HW)); }

// Waves per SIMD the forward's register allocation is held to.  Round 6: 7 (72 VGPRs + one 8-byte spill that is reloaded once per
// 64-instance batch) instead of the 74 registers = 6 waves the allocator settles on by itself: 7 168 instead of 6 144 resident quadrant
// waves of the launch's ~9 000 -- fewer late starters.  Same-box A/B (profiles/r06_waves_per_simd_ab.txt), forward blend us: config 2 without
// the per-view order 104.5 -> 100.2, with it 93.5 -> 93.2; config 4 207 -> 200; init-state 277 -> 268-276; 8 waves (64 VGPRs, 40 B of
// scratch) loses everywhere.  0 = no constraint.
#ifndef GSR_DEEP_FLAG_AT
#define GSR_DEEP_FLAG_AT (GSR_SEG1 + 4)  // checkpoints passed when a walk counts as deep for the backward's launch order: position gsr_ckpt_pos(10) = 19 L
#endif
#ifndef GSR_FWD_WAVES
#define GSR_FWD_WAVES 7
#endif
#if GSR_FWD_WAVES > 0
#define GSR_FWD_ATTR __attribute__((amdgpu_waves_per_eu(GSR_FWD_WAVES, GSR_FWD_WAVES)))
#else
#define GSR_FWD_ATTR
#endif

// Deepest walks first (gsr_tuning.walk_depths).  A forward launch has 4 T quadrant waves for ~6 k wave slots: the last third starts when
// the first waves end, and the launch lasts until the deepest of those late starters is through (tools/wave_trace.py: config 2, started at
// 42-47 us of 105, 55-63 us of life).  How deep a quadrant will walk is not in this frame's data (list length: correlation 0.01 on the
// bench scene) but it barely changes between two visits of the same VIEW, and training revisits its views every epoch: a caller that keeps
// one array of 4 T words per view gets its tasks dispatched in the order of what each quadrant walked last time, per XCD (the tile -> XCD
// map stays).  Measured with the previous step's depths (the same view again = what an epoch later looks like): forward blend config 2
// 104 -> 97 us, config 3 109 -> 98, config 4 229 -> 225, init-state 2 x 200 -> 2 x 168, `surfaces` 51.5 -> 52.7 (incl. this launch).
// The ordering itself: gsr_common.h gsr_fwd_order_block -- eight workgroups beside the binning's column scan in the one-call forward,
// this launch of its own (4.6 us in front of the blend) in the two-stage form.
__global__ void __launch_bounds__(1024) gsr_fwd_order_kernel(int T, int xt, const uint32_t* __restrict__ walk_depths, uint32_t* __restrict__ order)
{
    __shared__ uint32_t hist[256], start[256];
    gsr_fwd_order_block<1024>((int)blockIdx.x, T, xt, walk_depths, order, hist, start);
}

// TRAIN = false: the inference forward (gsr_tuning.inference; render under no_grad): no depth checkpoints are stored (the sums
// still restart at the segment boundaries, so that they associate as in the training forward), and final T, the last
// contributor and the running sums are stored only by a quadrant that ran off a partially sorted prefix (its resume state);
// the tile's traversal depth is not recorded.  Same arithmetic in the same order: images bit-identical.
template <bool TRAIN>
__global__ void __launch_bounds__(64) GSR_FWD_ATTR gsr_blend_fwd_kernel(
    const uint2* __restrict__ ranges, uint32_t* point_list /* bit 31 of an entry: guard-band flag, set here */, const GsrRec* __restrict__ rec, int W,
    int H, int gx, int T, const float* __restrict__ bg, float* __restrict__ out_color, float* __restrict__ out_depth,
    float* __restrict__ out_feature, float* __restrict__ final_T, uint32_t* __restrict__ n_contrib,
    uint32_t* __restrict__ tile_work, float* __restrict__ ckpt, int seg_len, uint32_t capacity, uint32_t longest_sorted,
    const uint32_t* __restrict__ sorted_len, uint32_t* __restrict__ need_full, const uint32_t* __restrict__ only_flagged,
    uint32_t* __restrict__ qresume, uint32_t* __restrict__ deep_walks /* info[3]: quadrant waves that entered the second tier */,
    const uint32_t* __restrict__ qorder /* dispatch order of the XCDs' task slots, or NULL */, uint32_t* __restrict__ walk_out /* gsr_tuning.walk_depths or NULL */,
    uint32_t* __restrict__ ranoff_report /* host-mapped word (fix-up pass only, may be NULL): a resumed quadrant stores `serial` */, uint32_t serial)
{
    __shared__ float4 sPair[GSR_FWB / 2][4];
    __shared__ float4 sC[GSR_FWB];

    GSR_TRACE_BEGIN
    // quadrant tasks: workgroup b = quadrant (b >> 3) & 3 of the (b >> 5)-th tile of XCD b & 7 (gsr_xcd_tile)
    // (the order array: gsr_fwd_order_kernel; without one -- no per-view depths, or the fix-up pass -- slot b >> 3 is the task itself)
    const int qslot = qorder ? (int)qorder[(size_t)(blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3)] : (int)(blockIdx.x >> 3);
    const int tile = gsr_xcd_tile((int)(blockIdx.x & 7u), qslot >> 2, T), quad = qslot & 3;
    if (tile < 0) return;
    const int u = 4 * tile + quad;
    const int tx = tile % gx, ty = tile / gx;
    const int lane = threadIdx.x;
    const int qx0 = tx * 16 + (quad & 1) * 8, qy0 = ty * 16 + (quad >> 1) * 8;
    if (qx0 >= W || qy0 >= H) return;  // quadrant entirely off the image
    const int px = qx0 + (lane & 7), py = qy0 + (lane >> 3);
    const float pxf = (float)px, pyf = (float)py;
    const float bx0 = (float)qx0, by0 = (float)qy0, bx1 = (float)min(qx0 + 7, W - 1), by1 = (float)min(qy0 + 7, H - 1);
    if (only_flagged && !only_flagged[tile]) return;  // fix-up pass after a full sort: only the tiles that asked for it
    const uint2 rg = ranges[tile];
    // Lists that were not (completely) scattered by a speculative launch, or that were longer than the LDS the sort was
    // provisioned with from a stale hint, are treated as empty: the host redoes stage 2.  Lists beyond GSR_NEAR_CAP
    // are in depth order only up to sorted_len[tile] (binning.hip): the walk ends there, and if a pixel is still
    // blending the tile is flagged and redone after a full sort.
    const int nlist = (int)(rg.y - rg.x), nsort = (int)sorted_len[tile];
    const int seg2_len = gsr_seg2_len(nlist, seg_len);  // unit of the second tier's (fixed) boundaries (gsr_common.h)
    const int n = (rg.y > capacity || (nsort == nlist && (uint32_t)nlist > longest_sorted)) ? 0 : min(nlist, nsort);
    const bool inside = px < W && py < H;
    const uint32_t HW = (uint32_t)H * (uint32_t)W, pid = inside ? (uint32_t)py * (uint32_t)W + (uint32_t)px : 0u;  // <= 2^24 tiles (api.hip) = at most 2^32 pixels

    const unsigned long long full = __builtin_amdgcn_ballot_w64(true);
    unsigned long long donem = __builtin_amdgcn_ballot_w64(!inside);
    unsigned long long riskm = 0ull;  // pixels whose stop came within GSR_TBAND of T = 1e-4 (replayed exactly after the walk)
    float Tr = 1.0f, C0 = 0.f, C1 = 0.f, C2 = 0.f, Dp = 0.f, Uf = 0.f;
    uint32_t last = 0;
    int npass = 0;  // checkpoints passed (wave-uniform)
    // sums of the segments closed so far: touched once per 64 / 128 list positions, so they live in LDS (one word per lane and
    // sum), not in five registers -- the walk loop stays at 6 waves per SIMD with the guard band's re-check in it (round 4)
    __shared__ float sAcc[5][64];
    float A0 = 0.f, A1 = 0.f, A2 = 0.f, A3 = 0.f, A4 = 0.f;  // (prologue only: the resumed walk's sums before they go to sAcc)
    uint32_t* ids = point_list + rg.x;  // entries: Gaussian id | (guard-band flag << 31)
    // Fix-up pass (the tile's list has been sorted completely since): a quadrant that ran off the sorted prefix at list
    // position m RESUMES there instead of starting over -- the first m entries of the complete order are the prefix it
    // walked, and what it left behind when it finished is its whole state: T (final_T), the last contributor and whether the
    // pixel had stopped (n_contrib, top bit), the sums since the last checkpoint (the last checkpoint slot) and the closed
    // segments (the other slots, re-added in order).  Same arithmetic in the same order as one uninterrupted walk: same bits.
    // Quadrants of the tile that had finished inside the prefix are left alone.
    int start = 0;
    if (only_flagged) {
        start = (int)qresume[u];
        if (start == 0) return;  // wave-uniform
        // the partial sort lost its bet on this tile: tell the host which forward it was (api.hip, gsr_partial_bet: the next calls sort
        // lists of this length completely at once and spare themselves this whole second launch); every resumed wave stores the same word
        if (lane == 0 && ranoff_report) *ranoff_report = serial;
        bool stopped = !inside;
        if (inside) {
            const uint32_t nc = n_contrib[pid];
            last = nc & 0x3fffffffu;
            stopped = (nc >> 31) != 0u;
            riskm = (nc >> 30) & 1u;  // (per lane here; made a wave mask below)
            Tr = final_T[pid];
            const float4 fa = gsr_ckpt_a(ckpt, GSR_SEG_MAX - 1, HW)[pid];
            const float2 fb = gsr_ckpt_b(ckpt, GSR_SEG_MAX - 1, HW)[pid];
            npass = __float_as_int(fa.x);
            C0 = fa.y; C1 = fa.z; C2 = fa.w; Dp = fb.x; Uf = fb.y;
        }
        npass = __builtin_amdgcn_readfirstlane(npass);  // wave-uniform by construction; lane 0 (the quadrant's first pixel) is inside
        if (!TRAIN) {
            // the inference forward stores no checkpoints: a quadrant that ran off left the sum of its closed segments in slot 0
            if (inside) {
                const float4 sa = gsr_ckpt_a(ckpt, 0, HW)[pid];
                const float2 sb = gsr_ckpt_b(ckpt, 0, HW)[pid];
                A0 = sa.y; A1 = sa.z; A2 = sa.w; A3 = sb.x; A4 = sb.y;
            }
        } else if (inside)
            for (int k = 0; k < npass; k++) {
                const float4 sa = gsr_ckpt_a(ckpt, k, HW)[pid];
                const float2 sb = gsr_ckpt_b(ckpt, k, HW)[pid];
                A0 += sa.y; A1 += sa.z; A2 += sa.w; A3 += sb.x; A4 += sb.y;
            }
        donem = __builtin_amdgcn_ballot_w64(stopped);
        riskm = __builtin_amdgcn_ballot_w64(inside && riskm != 0ull);
        // the walk state above came from memory: settle it here, or the blend loop waits for `Tr` with a counter that also
        // covers the NEXT batch's record loads (the prefetch would be waited for at the first blend of every batch)
        __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
    }

    sAcc[0][lane] = A0; sAcc[1][lane] = A1; sAcc[2][lane] = A2; sAcc[3][lane] = A3; sAcc[4][lane] = A4;  // (a lane reads only its own words)
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a, c = a;
    // List entries are requested TWO batches ahead, records one (round 6): entry -> record is a chain of two dependent round trips, and a
    // wave that has its SIMD to itself -- every deep walk of a frame that does not saturate early -- waited for both at the top of every
    // batch with few survivors (the blend loop of such a batch is shorter than one round trip): ~56 ns per list position, which is what
    // bounds the forward of the init-state and `fitted` frames.
    uint32_t idn = 0u;  // the NEXT batch's list entry of this lane
    {
        const int cnt0 = min(GSR_FWB - (start & (GSR_FWB - 1)), n - start);  // batches end at multiples of 64 list positions
        if (start + cnt0 + lane < n) idn = ids[start + cnt0 + lane];
        if (lane < cnt0) {
            const float4* r = reinterpret_cast<const float4*>(rec + (ids[start + lane] & 0x7fffffffu));
            a = r[0]; b = r[1]; c = r[2];
        }
    }
    int base = start, cnt = 0;
    for (; base < n; base += cnt) {
        if (donem == full) break;  // wave-uniform
        // Issue priority grows with the depth reached: the launch lasts as long as its deepest quadrants, which share
        // their SIMD fairly with up to seven shallower ones for most of their life; letting the waves that are still
        // going at depth 128 / 256 / 384 issue first shortens exactly those.
        if (base == 128) __builtin_amdgcn_s_setprio(1);
        else if (base == 256) __builtin_amdgcn_s_setprio(2);
        else if (base == 384) __builtin_amdgcn_s_setprio(3);
        // Checkpoint at list position gsr_ckpt_pos(k) (k = 0 .. GSR_SEG_MAX-2), from which the backward starts its depth
        // segment k - 1: the transmittance T_k in front of the position and the sums S_k over the segment that ends
        // there; C0.. restart from zero and the image sums are the sums of the segment sums.  (The
        // backward needs the sum BEHIND a position to a relative accuracy that final - prefix cannot give once T is
        // small: sums of small terms have to stay small.)
        // (positions are multiples of 64 and batches end at multiples of 64: a checkpoint is always the start of a batch)
        if (npass < GSR_SEG_MAX - 1 && base == gsr_ckpt_pos(npass, seg_len, seg2_len)) {
            if (TRAIN && inside) {  // (inference: the same restarts, so that the sums associate identically, but nothing is stored)
                gsr_ckpt_a(ckpt, npass, HW)[pid] = make_float4(Tr, C0, C1, C2);
                gsr_ckpt_b(ckpt, npass, HW)[pid] = make_float2(Dp, Uf);
            }
            npass++;
            // entering the second tier of depth segments: counted (one fire-and-forget atomic per quadrant wave that gets there), so
            // that the backward knows whether its big tasks are the rule or the exception on this frame
            if (TRAIN && npass == GSR_SEG1 && lane == 0) atomicAdd(deep_walks, 1u);
            // ... and a walk that passes checkpoint GSR_DEEP_FLAG_AT (list position 19 L = 1 216 with L = 64) sets the word's top bit: behind
            // it lie the 8 L backward tasks, which must not start last whatever the count says (round 6: a frame late in an optimisation
            // run has a few dozen deep tiles among 2 268 and kept them at the grid's tail: backward blend 133 -> 106 us)
            if (TRAIN && npass == GSR_DEEP_FLAG_AT && lane == 0) atomicOr(deep_walks, 0x80000000u);
            sAcc[0][lane] += C0; sAcc[1][lane] += C1; sAcc[2][lane] += C2; sAcc[3][lane] += Dp; sAcc[4][lane] += Uf;
            C0 = 0.f; C1 = 0.f; C2 = 0.f; Dp = 0.f; Uf = 0.f;
        }
        cnt = min(GSR_FWB - (base & (GSR_FWB - 1)), n - base);
        bool hit = false;
        if (lane < cnt) {
            const float ca = GSR_QSCALE(a.z), cb = GSR_QSCALE(a.w), cc = GSR_QSCALE(b.x);
            // (c.w: ln(255 opacity) + the fixed margin + the rounding of the reference's own fp32 `power` over this Gaussian's whole
            // rectangle, gsr_math.h GSR_CULL_ERR -- computed once per Gaussian by preprocess)
            hit = !(gsr_box_min_q(a.x, a.y, ca, cb, cc, GSR_RCP(ca), GSR_RCP(cc), bx0, bx1, by0, by1) > c.w * GSR_LOG2E);
        }
        const unsigned long long bal = __builtin_amdgcn_ballot_w64(hit);
        const int nq = __popcll(bal);
        if (hit) {
            const int pos = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
            float* dst = reinterpret_cast<float*>(&sPair[pos >> 1][0]) + (pos & 1);
#ifdef GSR_PRECISE_MATH
            dst[0] = a.x; dst[2] = a.y; dst[4] = a.z; dst[6] = a.w; dst[8] = b.x; dst[10] = b.y;
#else
            dst[0] = a.x; dst[2] = a.y; dst[4] = GSR_HA(a.z); dst[6] = GSR_HB(a.w); dst[8] = GSR_HC(b.x); dst[10] = b.y;
#endif
            dst[12] = __int_as_float(lane); dst[14] = b.w;
            sC[lane] = make_float4(c.x, c.y, c.z, b.z);
        }
        if ((nq & 1) && lane == 0) {  // odd count: the second half of the last pair is an instance with opacity 0
            float* dst = reinterpret_cast<float*>(&sPair[nq >> 1][0]) + 1;
            dst[0] = 0.f; dst[2] = 0.f; dst[4] = 0.f; dst[6] = 0.f; dst[8] = 0.f; dst[10] = 0.f; dst[12] = 0.f; dst[14] = 0.f;
        }
        {  // next batch's records (their list entries arrived during the previous batch) + the list entries of the batch after it:
           // in flight during the blend loop
            const int i = base + cnt + lane;  // (the next batch is a full one, or the list's tail; it starts at a multiple of 64)
            if (i < n) {
                const float4* r = reinterpret_cast<const float4*>(rec + (idn & 0x7fffffffu));
                a = r[0]; b = r[1]; c = r[2];
            }
            if (i + GSR_FWB < n) idn = ids[i + GSR_FWB];
        }
        __syncthreads();  // single-wave workgroup: orders the LDS writes above against the reads below

        const int np = (nq + 1) >> 1;
        float4 P0 = sPair[0][0], P1 = sPair[0][1], P2 = sPair[0][2], P3 = sPair[0][3];
        for (int k = 0; k < np; k++) {
            if (donem == full) break;  // wave-uniform
            const int kn = min(k + 1, np - 1);
            const float4 N0 = sPair[kn][0], N1 = sPair[kn][1], N2 = sPair[kn][2], N3 = sPair[kn][3];
            const gsr_f2 X = {P0.x, P0.y}, Y = {P0.z, P0.w}, HA = {P1.x, P1.y}, HB = {P1.z, P1.w};
            const gsr_f2 HC = {P2.x, P2.y}, OP = {P2.z, P2.w};
            const gsr_f2 dx = X - pxf, dy = Y - pyf;
            // log2 of the falloff, evaluated as fma(dy, hC dy, dx * fma(hA, dx, hB dy)) like every blend kernel
#ifdef GSR_PRECISE_MATH
            const gsr_f2 power = {gsr_power1(HA.x, HB.x, HC.x, dx.x, dy.x), gsr_power1(HA.y, HB.y, HC.y, dx.y, dy.y)};
#else
            const gsr_f2 inner = gsr_fma2(HA, dx, HB * dy);
            const gsr_f2 power = gsr_fma2(dy, HC * dy, dx * inner);
#endif
            const gsr_f2 G = {gsr_gauss1(power.x), gsr_gauss1(power.y)};
            const gsr_f2 al = OP * G;
            // alpha >= 1/255 is tested on the unclamped product (0.99 > 1/255: same truth value); the clamp of
            // forward.cu:531 is applied only to instances that blend
#ifdef GSR_PRECISE_MATH
            const unsigned long long cma = __builtin_amdgcn_ballot_w64(power.x <= 0.0f) & __builtin_amdgcn_ballot_w64(al.x >= (1.0f / 255.0f));
            const unsigned long long cmb = __builtin_amdgcn_ballot_w64(power.y <= 0.0f) & __builtin_amdgcn_ballot_w64(al.y >= (1.0f / 255.0f));
#else       // candidates from the lower edge of the guard band on; the ones inside the band are settled exactly, for BOTH instances of
            // the pair in front of the first blend: one rarely taken branch per pair iteration whose compares issue with the others,
            // instead of a compare -> scalar test -> branch in the dependency chain of every blend (a wave with its SIMD to itself,
            // i.e. the second half of every launch, runs at the speed of that chain: per-blend checks cost the launch 9 %, this 2 %)
            unsigned long long cma = __builtin_amdgcn_ballot_w64(power.x <= 0.0f) & __builtin_amdgcn_ballot_w64(al.x >= GSR_ALPHA_LO);
            unsigned long long cmb = __builtin_amdgcn_ballot_w64(power.y <= 0.0f) & __builtin_amdgcn_ballot_w64(al.y >= GSR_ALPHA_LO);
            // (|alpha - 1/255| of both instances -> one minimum -> ONE compare whose result is branched on directly: the test adds
            // three VALU instructions and no scalar logic to the chain in front of the blends)
            const gsr_f2 dband = al - gsr_splat(GSR_ALPHA_C);
#ifdef GSR_NO_BAND  // diagnostic: the band test compiled out (what the check costs)
            if (false) {
#else
            if (__builtin_amdgcn_ballot_w64(fminf(fabsf(dband.x), fabsf(dband.y)) < GSR_ALPHA_C * GSR_BAND) != 0ull) {  // rare: inside the guard band
#endif
                const unsigned long long banda = __builtin_amdgcn_ballot_w64(al.x < GSR_ALPHA_HI) & cma;
                const unsigned long long bandb = __builtin_amdgcn_ballot_w64(al.y < GSR_ALPHA_HI) & cmb;
                // -> the reference's own expression decides
                auto settle = [&](const unsigned long long bandm, unsigned long long& cm, const int j) {
                    if (bandm == 0ull) return;
                    // (the list index as an opaque per-lane value, made scalar again INSIDE the branch: with a plain scalar use here the
                    // compiler moves `j` -- and the whole {i_a, i_b, f_a, f_b} record -- to SGPRs with four v_readfirstlane in the hot
                    // path and waits early for the next iteration's LDS reads (forward 86 -> 101 us); with per-lane loads the kernel
                    // needs 88 VGPRs instead of 80: one wave per SIMD less)
                    int jv = j;
                    asm volatile("" : "+v"(jv));
                    // The list entry is FLAGGED (bit 31): the backward, whose pixels are a subset of the ones that met this instance
                    // here, runs its own band check only on flagged instances -- everywhere else alpha >= lower edge is the
                    // decision (no pixel of the tile is inside the band) and the two compares per iteration are saved.
                    const int pos = base + __builtin_amdgcn_readfirstlane(jv);
                    const uint32_t gid = ids[pos] & 0x7fffffffu;
                    if (lane == 0) ids[pos] = gid | 0x80000000u;  // (the four quadrant waves of a tile may all store this same word)
                    const GsrRec* r = rec + gid;
                    const float4 ra = r->a;
                    const float2 rb = *reinterpret_cast<const float2*>(&r->b);
                    const bool keep = gsr_blends_exact(ra.z, ra.w, rb.x, rb.y, ra.x - pxf, ra.y - pyf);
                    cm &= ~bandm | __builtin_amdgcn_ballot_w64(keep);
                    GSR_COUNT_ADD(7, 1);
                };
                settle(banda, cma, __float_as_int(P3.x));
                settle(bandb, cmb, __float_as_int(P3.y));
            }
#endif
            // colour / depth of both instances requested up front (addressed per lane, no scalar round trip): their LDS latency
            // passes behind the falloff arithmetic instead of sitting in the blend chain.  That chain is what a wave that has its
            // SIMD (nearly) to itself is bound by, and the second half of every launch is such waves (tools/wave_trace.py,
            // tools/lone_wave_probe.py: 123 -> 109 ns per instance for a lone wave, the launch 92.5 -> 87 us).
            const int ja = __float_as_int(P3.x), jb = __float_as_int(P3.y);
            float4 Ca = sC[ja];
            const float4 Cb = sC[jb];
            auto blend = [&](const unsigned long long okm, const float alu, const int j, const float4 C, const float feat) {
                const float al1 = __builtin_amdgcn_fmed3f(alu, 0.99f, -3.0e38f);  // = min(0.99, alu), one instruction (no NaN canonicalisation in front)
                const float test_T = Tr * (1.0f - al1);
                // (GSR_T_STOP: 1e-4, lowered by the replay band in the shipped build -- the fast walk never stops BEFORE the reference would,
                // so every pixel whose stop could differ ends with its T inside the band, where the end-of-walk test finds it)
                const unsigned long long stopm = __builtin_amdgcn_ballot_w64(test_T < GSR_T_STOP) & okm;
                donem |= stopm;
                const unsigned long long okf = okm & ~stopm;
#ifdef GSR_FWD_EXEC_MASK
                if (__builtin_amdgcn_inverse_ballot_w64(okf)) {  // EXEC = the pixels that blend: no selects
                    const float w = al1 * Tr;
                    C0 += C.x * w; C1 += C.y * w; C2 += C.z * w;
                    Dp += C.w * w; Uf += feat * w;
                    Tr = test_T;
                    last = (uint32_t)(base + j + 1);
                }
#else
                const float w = gsr_sel0(okf, al1 * Tr);
                C0 += C.x * w; C1 += C.y * w; C2 += C.z * w;
                Dp += C.w * w; Uf += feat * w;
                Tr = gsr_sel(okf, test_T, Tr);
                last = __float_as_uint(gsr_sel(okf, __uint_as_float((uint32_t)(base + j + 1)), __uint_as_float(last)));
#endif
            };
            const unsigned long long okma = cma & ~donem;
            GSR_COUNT_ADD(4, 1);                                 // forward: pair iterations
            GSR_COUNT_ADD(5, (okma != 0ull) + ((cmb & ~donem) != 0ull));  // ... instances of them that blend a pixel
            GSR_COUNT_ADD(6, __popcll(okma) + __popcll(cmb & ~donem));
            // (an opaque use in front of the branch: the compiler would otherwise sink the first read into the branch)
            asm volatile("" : "+v"(Ca.x), "+v"(Ca.y), "+v"(Ca.z), "+v"(Ca.w));
            if (okma != 0ull) blend(okma, al.x, ja, Ca, P3.z);
            const unsigned long long okmb = cmb & ~donem;  // after a: pixels it finished no longer blend b
            if (okmb != 0ull) blend(okmb, al.y, jb, Cb, P3.w);
            P0 = N0; P1 = N1; P2 = N2; P3 = N3;
        }
        __syncthreads();
    }

    // ran off the sorted prefix with pixels still blending: the tile is sorted completely and this quadrant resumes at n
    if (walk_out && lane == 0) walk_out[u] = (uint32_t)min(base + GSR_FWB, n);  // list positions this walk covered (a resumed walk: up to its end)
    const bool ran_off = nsort < nlist && rg.y <= capacity && donem != full;

    int npl = npass;  // checkpoints passed, per lane: a replayed pixel's own walk may end in another segment than the wave's
#if !defined(GSR_PRECISE_MATH) && !defined(GSR_NO_TREPLAY)
    // ---- exact replay of the pixels near the T = 1e-4 stop (see GSR_TBAND, GSR_T_STOP) ----
    // final T inside the band around 1e-4 (it can lie BELOW 1e-4: the fast walk stops late rather than early): the last blend passed
    // the reference's stop test by less than the band, or failed it by less
    riskm |= __builtin_amdgcn_ballot_w64(inside && Tr < 0.0001f * (1.0f + GSR_TBAND));
    // (a quadrant that ran off a partially sorted prefix replays at the end of its resumed walk, over the complete list: the
    // flags travel in bit 30 of n_contrib)
    if (riskm != 0ull && !ran_off) {
        // (a replay is one dependent chain at the end of this wave's life while its SIMD still serves up to five others: let it issue first)
        __builtin_amdgcn_s_setprio(3);
        for (unsigned long long todo = riskm; todo != 0ull; todo &= todo - 1ull) {
            const int f = (int)__builtin_ctzll(todo);  // wave-uniform: the pixel all 64 lanes work on
            const float fx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(pxf), f));
            const float fy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(pyf), f));
            const uint32_t fpid = (uint32_t)__builtin_amdgcn_readlane((int)pid, f);
            float Tq = 1.0f;  // the reference's T (wave-uniform)
            float S0 = 0.f, S1 = 0.f, S2 = 0.f, S3 = 0.f, S4 = 0.f;  // sums of the open segment (uniform)
            // (sums of the closed segments: in LDS, sRep[0..4], touched once per segment -- five registers the walk loop's
            // allocation does not have to leave room for: 74 VGPRs = 6 waves per SIMD with them, 72 = 7 without)
            __shared__ float sRep[8];
            __shared__ float sRepCk[(GSR_SEG_MAX - 1) * 6];  // the replay's depth checkpoints {T, r, g, b, depth, feature}, stored once it is accepted
            if (lane < 5) sRep[lane] = 0.0f;
            int np2 = 0;
            uint32_t lastq = 0u;
            bool stoppedq = false;
            // Software pipeline over the batches: list entries two batches ahead and records one batch ahead are in flight while a
            // batch is evaluated (a replay is a chain of dependent L2 round trips at the END of a wave's life: its latency, not its
            // instruction count, is what the launch pays -- first version, loads inside the loop + a scalar T loop: forward 94 -> 114 us)
            auto fetch_id = [&](const int b) -> uint32_t { return b + lane < n ? (ids[b + lane] & 0x7fffffffu) : 0xffffffffu; };
            struct RecRegs { float4 a; float4 b; };  // (the colour is requested at the top of its own batch: the ripple covers its latency)
            auto fetch_rec = [&](const uint32_t gid) -> RecRegs {
                RecRegs r = {make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f)};
                if (gid != 0xffffffffu) { const GsrRec* q = rec + gid; r.a = q->a; r.b = q->b; }
                return r;
            };
            uint32_t gid_cur = fetch_id(0), gid_nxt = fetch_id(GSR_FWB);
            RecRegs rc_cur = fetch_rec(gid_cur);
            for (int b0 = 0; b0 < n && !stoppedq; b0 += GSR_FWB) {
                const uint32_t gid_nn = fetch_id(b0 + 2 * GSR_FWB);
                const RecRegs rc_nxt = fetch_rec(gid_nxt);
                float4 col = make_float4(0.f, 0.f, 0.f, 0.f);
                if (gid_cur != 0xffffffffu) col = rec[gid_cur].c;
                if (np2 < GSR_SEG_MAX - 1 && b0 == gsr_ckpt_pos(np2, seg_len, seg2_len)) {  // same restarts as the walk above
                    // (staged, not stored: a replay that is dropped below must leave the pixel's checkpoints as the fast walk wrote
                    // them -- they belong to the final T / sums the pixel then keeps)
                    if (TRAIN && lane == 0) {
                        float* c = sRepCk + np2 * 6;
                        c[0] = Tq; c[1] = S0; c[2] = S1; c[3] = S2; c[4] = S3; c[5] = S4;
                    }
                    np2++;
                    if (lane < 5) sRep[lane] += lane == 0 ? S0 : lane == 1 ? S1 : lane == 2 ? S2 : lane == 3 ? S3 : S4;
                    S0 = 0.f; S1 = 0.f; S2 = 0.f; S3 = 0.f; S4 = 0.f;
                }
                GsrExactAlpha ea = {0.f, 1.f, false, false};
                if (gid_cur != 0xffffffffu) {
                    ea = gsr_alpha_exact(rc_cur.a.z, rc_cur.a.w, rc_cur.b.x, rc_cur.b.y, rc_cur.a.x - fx, rc_cur.a.y - fy);
                    // the backward runs its own exact alpha check only on flagged list entries: this walk may visit entries
                    // the pixel's fast walk never reached
                    if (ea.in_band) ids[b0 + lane] = gid_cur | 0x80000000u;
                }
                const unsigned long long m = __builtin_amdgcn_ballot_w64(ea.blends);
                float w = 0.0f;
                if (m != 0ull) {  // (wave-uniform)
                    // The T chain in list order, rounded like forward.cu:536-549 -- T <- fl(T * (1 - alpha)) over the blending instances --
                    // as a RIPPLE over the lanes: X[i] <- X[i-1] * f[i] (f = 1 - alpha on blending lanes, exactly 1 elsewhere; lane 0
                    // takes the batch's incoming T) repeated until the highest blending lane has its value: after step s the lanes <= s
                    // hold the sequentially rounded products.  One DPP wave shift + one multiply per step, no scalar round trips.
                    const float f = ea.blends ? ea.one_minus : 1.0f;
                    const int hi = 63 - (int)__builtin_clzll(m);
                    // X starts as T_in * f (final for lane 0); a step multiplies lane i's factor onto lane i-1's value IN PLACE: the DPP
                    // source of lane 0 is invalid, so with bound_ctrl off lane 0 is simply not written.  One v_mul_f32_dpp + the two
                    // wait states a DPP read of a just-written VGPR needs; no fused multiply-add can form inside the asm.
                    float X = gsr_mul_exact(Tq, f);
                    // (four steps per trip: the loop's own scalar bookkeeping cost as much as a step; extra steps past `hi` only touch
                    // lanes whose values nobody reads)
#define GSR_RIPPLE_STEP "s_nop 1\n\tv_mul_f32_dpp %0, %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
                    for (int st = 0; st < hi; st += 4)
                        asm volatile(GSR_RIPPLE_STEP GSR_RIPPLE_STEP GSR_RIPPLE_STEP GSR_RIPPLE_STEP : "+v"(X) : "v"(f));
#undef GSR_RIPPLE_STEP
                    asm volatile("s_nop 1" ::: "memory");
                    const float Y = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(Tq), __float_as_int(X), 0x138 /* wave_shr:1 */, 0xf, 0xf, false));
                    // X = T behind each lane's instance, Y = T in front of it (both final for lanes <= hi); T is non-increasing, so
                    // the first blending lane whose product is below 1e-4 is where forward.cu:537 stops
                    const unsigned long long stopm = __builtin_amdgcn_ballot_w64(ea.blends && X < 0.0001f);
                    unsigned long long blended = m;
                    if (stopm != 0ull) {
                        stoppedq = true;
                        blended = m & ((1ull << __builtin_ctzll(stopm)) - 1ull);
                    }
                    if (blended != 0ull) {
                        const int lb = 63 - (int)__builtin_clzll(blended);
                        Tq = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(X), lb));
                        lastq = (uint32_t)(b0 + lb + 1);
                    }
                    if ((blended >> lane) & 1ull) w = ea.alpha * Y;
                }
                S0 += gsr_wave_sum(col.x * w); S1 += gsr_wave_sum(col.y * w); S2 += gsr_wave_sum(col.z * w);
                S3 += gsr_wave_sum(rc_cur.b.z * w); S4 += gsr_wave_sum(rc_cur.b.w * w);
                gid_cur = gid_nxt; gid_nxt = gid_nn; rc_cur = rc_nxt;
            }
            // (a replay that reaches the end of a partially sorted prefix without stopping would need the entries behind it: it is
            // dropped and the pixel keeps its fast walk -- a flip exactly at the cut of a list beyond 2048 entries)
            if (!(nsort < nlist && !stoppedq)) {
                if (TRAIN && lane < np2) {  // (single-wave workgroup: lane 0's LDS stores above are ordered before these reads)
                    const float* c = sRepCk + lane * 6;
                    gsr_ckpt_a(ckpt, lane, HW)[fpid] = make_float4(c[0], c[1], c[2], c[3]);
                    gsr_ckpt_b(ckpt, lane, HW)[fpid] = make_float2(c[4], c[5]);
                }
                if (lane == f) {
                    Tr = Tq; C0 = S0; C1 = S1; C2 = S2; Dp = S3; Uf = S4; last = lastq; npl = np2;
                    sAcc[0][lane] = sRep[0]; sAcc[1][lane] = sRep[1]; sAcc[2][lane] = sRep[2]; sAcc[3][lane] = sRep[3]; sAcc[4][lane] = sRep[4];
                }
                GSR_COUNT_ADD(3, 1);
            }
        }
    }
#endif
    if (lane == 0) {
        if (ran_off) need_full[tile] = 1u;
        qresume[u] = ran_off ? (uint32_t)n : 0u;
    }

    // deepest contributor of the quadrant -> of the tile: what the backward has to traverse (drives its launch order)
    uint32_t wl = last;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) wl = max(wl, (uint32_t)__shfl_xor((int)wl, d, 64));
    if (TRAIN && lane == 0 && wl) atomicMax(&tile_work[tile], wl);  // zeroed by gsr_tile_scan_kernel

    if (inside) {
        const float E0 = sAcc[0][lane], E1 = sAcc[1][lane], E2 = sAcc[2][lane], E3 = sAcc[3][lane], E4 = sAcc[4][lane];
        if (TRAIN || ran_off) {  // (ran_off is wave-uniform)
            // sums behind the last checkpoint + how many checkpoints were passed
            gsr_ckpt_a(ckpt, GSR_SEG_MAX - 1, HW)[pid] = make_float4(__int_as_float(npl), C0, C1, C2);
            gsr_ckpt_b(ckpt, GSR_SEG_MAX - 1, HW)[pid] = make_float2(Dp, Uf);
            final_T[pid] = Tr;
            // top bits: for the resume only (31: the pixel had stopped, 30: its stop was inside the T band)
            n_contrib[pid] = last | ((ran_off && __builtin_amdgcn_inverse_ballot_w64(donem)) ? 0x80000000u : 0u) |
                             ((ran_off && __builtin_amdgcn_inverse_ballot_w64(riskm)) ? 0x40000000u : 0u);
            if (!TRAIN) {  // + the closed segments' sums, which the training forward keeps in its checkpoint slots
                gsr_ckpt_a(ckpt, 0, HW)[pid] = make_float4(0.f, E0, E1, E2);
                gsr_ckpt_b(ckpt, 0, HW)[pid] = make_float2(E3, E4);
            }
        }
        C0 += E0; C1 += E1; C2 += E2; Dp += E3; Uf += E4;  // image sums = sum of the segment sums
        out_color[pid] = C0 + Tr * bg[0];
        out_color[HW + pid] = C1 + Tr * bg[1];
        out_color[2 * (size_t)HW + pid] = C2 + Tr * bg[2];
        out_depth[pid] = Dp;
        out_feature[pid] = Uf;
    }
    GSR_TRACE_END_AT(1, 4 * 4 * 36864)
#ifdef GSR_TRACE  // + list length, deepest contributor and the position the walk ended at
    if (lane == 0) {
        gsr_trace_buf[4 * 4 * 36864 + (size_t)blockIdx.x * 4 + 3] |= ((unsigned long long)(uint32_t)n << 8) | ((unsigned long long)wl << 32);
        gsr_trace_buf[4 * 4 * 36864 + (size_t)blockIdx.x * 4 + 2] |= (unsigned long long)(uint32_t)base << 32;
    }
#endif
}

// ---------------------------------------------------------------------------------------------
// Backward
//
// Work unit = one DEPTH SEGMENT of one tile: GSR_SEG_LEN consecutive list positions (the last segment of a tile takes
// everything behind (GSR_SEG_MAX-1) * GSR_SEG_LEN).  The reference walks a tile's list back to front carrying
// T and the colour accumulated behind the current instance (backward.cu:500-566); both are functions of the forward's
// prefix sums, T_e and (C_final - C_front(e)) / T_e, which the forward stores per pixel at the segment boundaries, so
// every segment can start on its own.  On the bench scene that turns 2268 tile-sized work units (1.8 per wavefront
// slot: the launch lasted as long as its most loaded SIMD) into ~8000 uniform ones.
//
// One 128-thread workgroup (2 wavefronts) per task; wavefront w owns the 16x8 pixel STRIP w of the tile and lane l the
// two pixels (l & 7, l >> 3) and (8 + (l & 7), l >> 3) of it.  The per-pixel arithmetic runs on float2 values, i.e.
// v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 (two fp32 operations per lane and instruction), the blend recurrences
// are made branch-free by zeroing alpha and G of the pixels that do not blend (two selects per pixel; everything
// downstream is then exact: T / (1 - 0) = T, 0 * C + 1 * acc = acc, weights 0), and the wave reduction -- the
// largest fixed cost per (wave, instance) -- is paid once per 128 pixels instead of once per 64.
// ---------------------------------------------------------------------------------------------
// Wave-private compaction: indices i < cnt whose quadrant mask sQ[i] meets `qmask` and pred(i), in ascending order.
template <int SL, typename Pred>
__device__ __forceinline__ int gsr_compact2(const uint32_t* sQ, uint16_t* list, int cnt, uint32_t qmask, int lane, Pred pred)
{
    int n = 0;
#pragma unroll
    for (int c = 0; c < SL / 64; c++) {
        const int i = c * 64 + lane;
        const bool hit = i < cnt && (sQ[i] & qmask) && pred(i);
        const unsigned long long bal = __ballot(hit);
        if (hit) list[n + __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u))] = (uint16_t)i;
        n += __popcll(bal);
    }
    return n;
}

// Waves per SIMD of the backward (64-entry form).  Round 6: 6 (80 VGPRs; the RGB-only variant without a spill, the AUX variant with four
// dwords spilled in its prologue) instead of 5 (90 / 98 VGPRs): the kernel is VALU-issue-bound at ~77 % of the issue slots, and a sixth
// wave per SIMD fills some of the rest -- backward blend us, same box: config 2 152 -> 147-150, config 3 172 -> 168.5, config 4 386 ->
// 372-378, init-state 343 -> 332, `surfaces` 158 -> 153 (profiles/r06_waves_per_simd_ab.txt).  7 (72 VGPRs, 24 / 52 B of scratch): config 2
// the same, config 3 +9 us.  (Rounds 2-4 had measured 6 as a loss on the kernels of their time.)
#ifndef GSR_BWD_WAVES
#define GSR_BWD_WAVES 6
#endif
// SL = the launch's segment length (64 up to 4096 tiles, 128 beyond): sizes the LDS arrays.
template <bool AUX, int SL>
// (the 128-entry form is held to 4 waves per SIMD by its 19.5 KB of LDS per workgroup: its register budget says so too)
__global__ void __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(SL == 64 ? GSR_BWD_WAVES : 4, SL == 64 ? GSR_BWD_WAVES : 4))) gsr_blend_bwd_kernel(
    const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list, const GsrRec* __restrict__ rec, int W,
    int H, int gx, const float* __restrict__ bg, const float* __restrict__ final_T,
    const uint32_t* __restrict__ n_contrib, const float* __restrict__ ckpt, const float* __restrict__ dL_dcolor,
    const float* __restrict__ dL_ddepth, const float* __restrict__ dL_dfeature, const uint32_t* __restrict__ tile_work,
    int T, int seg_len, const uint32_t* __restrict__ offsets, uint8_t* __restrict__ slot_written, float4* __restrict__ slots,
    uint32_t* __restrict__ heavy_groups, const uint32_t* __restrict__ need_full, int nseg /* segments the grid covers per tile */,
    const uint32_t* __restrict__ deep_walks /* info[3] of the forward */, int xcd_tiles /* gsr_xcd_tiles(T) */)
{
    // the per-Gaussian backward that follows appends its heavy groups to a list: this launch, which always precedes it, resets
    // the counter (gauss_bwd.hip)
    if (blockIdx.x == 0 && threadIdx.x == 0) { heavy_groups[0] = 0u; heavy_groups[1] = 0u; }
    static_assert(SL == 64 || SL == GSR_SEG_LEN, "segment lengths of gsr_seg_len()");
    __shared__ float4 sA[SL], sB[SL], sC[SL];
    // one accumulator array PER WAVEFRONT: a wave's LDS adds are program-ordered and nobody else touches its copy, the flush
    // adds the two copies in a fixed order -> the backward is bit-reproducible (it was not while both waves added into one
    // array in whatever order they arrived)
    __shared__ __attribute__((aligned(16))) float acc[2][SL * GSR_SLOT_FLOATS];
    __shared__ uint32_t sSlot[SL], sQ[SL];
    __shared__ uint16_t sList[2][SL];

    GSR_TRACE_BEGIN
#ifdef GSR_TRACE
    int gsr_tr_it = 0, gsr_tr_bl = 0;
#endif
    // Workgroup b runs on XCD b % 8 and takes the (b >> 3)-th slot of that XCD: slot i = (depth segment rank i / xcd_tiles, the
    // XCD's (i % xcd_tiles)-th tile, gsr_xcd_tile: the forward's tile -> XCD map): all tasks of one rank come first, then the next
    // rank's, ...; a tile has min(GSR_SEG_MAX, ...) segments and the workgroups of the ones it does not have leave at once.
    const int band = (int)(blockIdx.x & 7u), slot = (int)(blockIdx.x >> 3);
    // Launch order = heaviest tasks first.  On a frame whose pixels saturate, the FIRST segments are the heaviest (every pixel still
    // blends) and the grid runs first segments first, as in rounds 2-4.  On a frame whose long lists are walked to their ends the
    // second-tier segments are up to 12 x longer than a first-tier one and lead the grid instead (with them at the END the launch
    // drained through them: init-state frame 570 -> 480 us).  Which kind of frame this is, the forward says: info[3] = the number of
    // quadrant walks that entered the second tier; "the rule" = more than one quadrant in sixteen.  (Leading unconditionally, or
    // whenever ONE walk got there -- config 2's deepest ends at 463 of 448 --, cost config 2 +5 us, config 4 +8 us: empty workgroups in
    // front of the real ones.)
    const int rank = slot / xcd_tiles, tile = gsr_xcd_tile(band, slot - rank * xcd_tiles, T);
    if (rank >= nseg || tile < 0) return;
    const int nbig = nseg - GSR_SEG1;  // segments behind the first tier
#ifdef GSR_BWD_BIG_FIRST  // (A/B knob: force one order)
    const bool big_first = GSR_BWD_BIG_FIRST;
#else
    const uint32_t dw = *deep_walks;  // (wave-uniform scalar load) low bits: quadrant walks that entered the second tier; top bit: one went past 19 L
    const bool big_first = (dw >> 31) != 0u || (dw & 0x7fffffffu) * 4u >= (uint32_t)T;
#endif
#ifdef GSR_BWD_TIER2_ASCENDING  // (A/B knob: second-tier segments in list order instead of longest first)
    const int seg = big_first ? (rank < nbig ? GSR_SEG1 + rank : rank - nbig) : rank;
#else
    // (the second tier's segments grow with depth: the deepest -- longest -- one first)
    const int seg = big_first ? (rank < nbig ? nseg - 1 - rank : rank - nbig) : rank;
#endif
    const int tx = tile % gx, ty = tile / gx;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const uint2 rg = ranges[tile];
    // Instances behind the tile's deepest contributor were blended by no pixel: they are not traversed and their
    // gradient slots are NOT written; slot_written[] (zeroed per call) tells the per-Gaussian kernel which slots exist.
    const int nproc = min((int)(rg.y - rg.x), (int)tile_work[tile]);
    const int seg2_len = gsr_seg2_len((int)(rg.y - rg.x), seg_len);  // unit of the second tier's (fixed) boundaries, as in the forward
    const int seg_lo = seg == 0 ? 0 : gsr_ckpt_pos(seg - 1, seg_len, seg2_len);
    // (the grid's last segment takes everything that is left: with the forward's own longest list as the grid's measure that is what its end
    // is anyway; with a SMALLER figure from the caller the lists beyond it are still traversed completely, just by one longer task)
    const int seg_hi = seg == nseg - 1 ? nproc : min(nproc, gsr_ckpt_pos(seg, seg_len, seg2_len));
    if (seg_hi <= seg_lo) return;
#if defined(GSR_BWD_DIAG) && GSR_BWD_DIAG == 1  // diagnostic: dispatch + task lookup only
    return;
#endif

    // lanes l, l+4, l+8, l+12 of a DPP row share their pixel row (see gsr_bank_reduce_dyf)
    const int pxa = tx * 16 + ((lane >> 1) & 7), pxb = pxa + 8;
    const int py = ty * 16 + wave * 8 + 2 * (lane >> 4) + (lane & 1);
    const gsr_f2 pxf = {(float)pxa, (float)pxb};
    const float pyf = (float)py;
    const float ddelx_dx = 0.5f * W, ddely_dy = 0.5f * H;
    const size_t HW = (size_t)H * W;
    const bool ina = pxa < W && py < H, inb = pxb < W && py < H;
    const size_t pa = ina ? (size_t)py * W + pxa : 0, pb = inb ? (size_t)py * W + pxb : 0;

    const gsr_f2 Tf = {ina ? final_T[pa] : 0.f, inb ? final_T[pb] : 0.f};
    // (bits 31 / 30 are the forward's resume flags: cleared by its fix-up pass, masked here in case the caller skipped that pass)
    const int lastca = ina ? (int)(n_contrib[pa] & 0x3fffffffu) : 0, lastcb = inb ? (int)(n_contrib[pb] & 0x3fffffffu) : 0;
    const gsr_f2 g0 = {ina ? dL_dcolor[pa] : 0.f, inb ? dL_dcolor[pb] : 0.f};
    const gsr_f2 g1 = {ina ? dL_dcolor[HW + pa] : 0.f, inb ? dL_dcolor[HW + pb] : 0.f};
    const gsr_f2 g2 = {ina ? dL_dcolor[2 * HW + pa] : 0.f, inb ? dL_dcolor[2 * HW + pb] : 0.f};
    gsr_f2 gd = {0.f, 0.f}, gu = {0.f, 0.f};
    if (AUX) {  // either auxiliary map may be absent on its own (NULL = no gradient flows into it)
        if (dL_ddepth) {
            if (ina) gd.x = dL_ddepth[pa];
            if (inb) gd.y = dL_ddepth[pb];
        }
        if (dL_dfeature) {
            if (ina) gu.x = dL_dfeature[pa];
            if (inb) gu.y = dL_dfeature[pb];
        }
    }
    const gsr_f2 nTfb = -Tf * (bg[0] * g0 + bg[1] * g1 + bg[2] * g2);  // -T_final * (bg . dL/dC)

    // State behind the segment.  A pixel whose last contributor lies inside or in front of the segment starts from its
    // final state (T_final, nothing accumulated behind) exactly like the reference; a pixel that blends instances behind
    // the segment end e starts from the forward's checkpoints: T_e, and the sums of the segments behind e (added back
    // to front, smallest first) divided by T_e.
    // The reference carries the colour / depth / feature accumulated behind the current instance per channel
    // (accum_rec[ch], backward.cu:546,559,566) and forms dL/dalpha = sum_ch (c_ch - accum_rec_ch) dL/dC_ch.  Both are
    // linear in the channels, so ONE scalar per pixel carries the same information: Ar = sum_ch accum_rec_ch dL/dC_ch obeys
    // the same recurrence with q = sum_ch c_ch dL/dC_ch in the place of the colour, and dL/dalpha = q - Ar.  One
    // recurrence instead of three (five with the depth and feature heads): 6 packed operations fewer per iteration.
    gsr_f2 Tr = Tf, Ar = {0.f, 0.f};
    {
        auto behind = [&](const size_t p, const float gc0, const float gc1, const float gc2, const float gdd, const float guu,
                          float& T_, float& A_) {
            // two memory round trips instead of up to eight: {last slot, slot `seg`} first (the count of checkpoints the pixel
            // passed is in the last slot), then every slot behind this segment at once; the sums are formed back to front in
            // the same order as a slot-by-slot loop.
            const float4 fa = gsr_ckpt_a(ckpt, GSR_SEG_MAX - 1, HW)[p];
            const float Te = gsr_ckpt_a(ckpt, seg, HW)[p].x;
            float2 fb = make_float2(0.f, 0.f);
            if (AUX) fb = gsr_ckpt_b(ckpt, GSR_SEG_MAX - 1, HW)[p];
            const int np = __float_as_int(fa.x);  // checkpoints this pixel passed (>= seg + 1 here)
            // the segments behind this one, back to front (smallest sums first), GSR_BEHIND_CHUNK slots per memory round trip: one
            // trip for walks inside tier 1 (as before the second tier existed), up to three for the deepest
            constexpr int GSR_BEHIND_CHUNK = GSR_SEG1 - 1;
            float t0 = fa.y, t1 = fa.z, t2 = fa.w, td = fb.x, tu = fb.y;
            for (int khi = np - 1; khi >= seg + 1; khi -= GSR_BEHIND_CHUNK) {
                float4 sa[GSR_BEHIND_CHUNK];
                float2 sb[GSR_BEHIND_CHUNK];
#pragma unroll
                for (int c = 0; c < GSR_BEHIND_CHUNK; c++) {
                    const int k = khi - c;
                    sa[c] = make_float4(0.f, 0.f, 0.f, 0.f); sb[c] = make_float2(0.f, 0.f);
                    if (k >= seg + 1) {
                        sa[c] = gsr_ckpt_a(ckpt, k, HW)[p];
                        if (AUX) sb[c] = gsr_ckpt_b(ckpt, k, HW)[p];
                    }
                }
#pragma unroll
                for (int c = 0; c < GSR_BEHIND_CHUNK; c++) {
                    if (khi - c >= seg + 1) {
                        t0 += sa[c].y; t1 += sa[c].z; t2 += sa[c].w;
                        if (AUX) { td += sb[c].x; tu += sb[c].y; }
                    }
                }
            }
            const float r = 1.0f / Te;
            T_ = Te;
            A_ = (t0 * r) * gc0 + (t1 * r) * gc1 + (t2 * r) * gc2;
            if (AUX) A_ += (td * r) * gdd + (tu * r) * guu;
        };
        if (lastca > seg_hi) {  // only possible for seg < GSR_SEG_MAX - 1
            float T_, A_;
            behind(pa, g0.x, g1.x, g2.x, gd.x, gu.x, T_, A_);
            Tr.x = T_; Ar.x = A_;
        }
        if (lastcb > seg_hi) {
            float T_, A_;
            behind(pb, g0.y, g1.y, g2.y, gd.y, gu.y, T_, A_);
            Tr.y = T_; Ar.y = A_;
        }
    }

#if defined(GSR_BWD_DIAG) && GSR_BWD_DIAG == 2  // diagnostic: ... + the pixel prologue (final T, gradients, checkpoint sums)
    if (Tr.x + Ar.x + Tr.y + Ar.y + g0.x + g1.x + g2.x == 12345.678f) slots[0] = make_float4(Tr.x, Ar.x, Tr.y, Ar.y);
    return;
#endif
    // wave-level maximum of the last contributor: nothing at a position >= it is blended by this strip
    int wmax = max(lastca, lastcb);
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) wmax = max(wmax, __shfl_xor(wmax, d, 64));
    for (int i = t; i < 2 * SL * GSR_SLOT_FLOATS; i += 128) (&acc[0][0])[i] = 0.f;
    uint16_t* mylist = sList[wave];
    const uint32_t qmask = 3u << (2 * wave);  // the two 8x8 quadrants of this strip
    int accfield = -1;  // accumulator field this lane reports after the wave reduction (see gsr_bank_reduce)
    {
        const int row = lane >> 4, bank = (lane >> 2) & 3;
        if ((lane & 3) == 0) {
            // slot fields: 0-2 colour, 3 depth, 4 feature, 5 Mx, 6 My, 7 Mxx, 8 Mxy, 9 Myy, 10 M0
            const int f0a[4] = {0, 1, 2, 3}, f1a[4] = {4, 7, 10, 5}, f2a[4] = {6, 8, 9, -1};
            const int f0n[4] = {0, 1, 2, 7}, f1n[4] = {10, 5, 6, 8}, f2n[4] = {9, -1, -1, -1};
            if (AUX) accfield = row == 0 ? f0a[bank] : row == 2 ? f1a[bank] : row == 1 ? f2a[bank] : -1;
            else accfield = row == 0 ? f0n[bank] : row == 2 ? f1n[bank] : row == 1 ? f2n[bank] : -1;
        }
    }

    // back to front, in batches of GSR_SEG_LEN instances (one batch, except in a tile's last segment); local j = 0 is
    // the backmost instance of the batch
    // (SL = the batch length the LDS arrays are sized for; the segment, i.e. the checkpoint spacing seg_len, may be longer)
    unsigned long long bandm64 = ~0ull;  // (the 128-entry form checks every instance)
    // a tile whose list was sorted a second time (it ran off its partially sorted prefix) lost the flags of its first pass: there
    // every instance is checked
    const bool band_flags_valid = need_full[tile] == 0u;
    (void)bandm64; (void)band_flags_valid;  // (unused in the parity build)
    for (int hi = seg_hi; hi > seg_lo; hi -= SL) {
        const int lo = max(seg_lo, hi - SL), cnt = hi - lo;
        if (SL == 64) {
            // both wavefronts stage: wave 0 fetches {a, b} of instance `lane`, wave 1 {c, d} + the slot offset; the strip
            // test below then runs on both (one box test per wave and instance instead of four quadrant tests on wave 0)
            uint32_t idf = 0u;
            if (lane < cnt) idf = point_list[rg.x + (hi - 1 - lane)];
#ifndef GSR_PRECISE_MATH
            bandm64 = band_flags_valid ? __ballot((idf >> 31) != 0u) : ~0ull;  // batch slots the forward met inside the guard band (see gsr_blend_fwd_kernel)
#endif
            if (lane < cnt) {
                const uint32_t id = idf & 0x7fffffffu;
                const GsrRec* r = rec + id;
                if (wave == 0) {
#ifdef GSR_PRECISE_MATH
                    sA[lane] = r->a; sB[lane] = r->b;
#else
                    const float4 a = r->a, b = r->b;  // raw conic -> the pre-scaled form the loop evaluates
                    sA[lane] = make_float4(a.x, a.y, GSR_HA(a.z), GSR_HB(a.w)); sB[lane] = make_float4(GSR_HC(b.x), b.y, b.z, b.w);
#endif
                } else {
                    const uint32_t slot0 = offsets[id];
                    const uint4 d = r->d;
                    const float4 c = r->c;
                    const int x0 = d.y & 0xffff, y0 = d.y >> 16, wd = (int)d.x;
                    const int pos = (ty - y0) * wd + (tx - x0);
                    const unsigned long long mask = ((unsigned long long)d.w << 32) | d.z;
                    sSlot[lane] = slot0 + (uint32_t)(pos < 64 ? __popcll(mask & ((1ull << pos) - 1ull)) : __popcll(mask) + (pos - 64));
                    sC[lane] = c;
                }
            }
        } else
        if (t < cnt) {
            const uint32_t id = point_list[rg.x + (hi - 1 - t)] & 0x7fffffffu;
            const GsrRec* r = rec + id;
            const uint32_t slot0 = offsets[id];  // first gradient slot of the Gaussian (exclusive scan of tiles[])
            const uint4 d = r->d;
            const float4 c = r->c;
            // gradient slot = Gaussian's scan offset + rank of this tile among the surviving tiles of its rectangle
            const int x0 = d.y & 0xffff, y0 = d.y >> 16, wd = (int)d.x;
            const int pos = (ty - y0) * wd + (tx - x0);
            const unsigned long long mask = ((unsigned long long)d.w << 32) | d.z;
            sSlot[t] = slot0 + (uint32_t)(pos < 64 ? __popcll(mask & ((1ull << pos) - 1ull)) : __popcll(mask) + (pos - 64));
            const float4 a = r->a, b = r->b;
            sQ[t] = gsr_quadrant_mask(a, b, c.w * GSR_LOG2E, tx, ty, W, H);
#ifdef GSR_PRECISE_MATH
            sA[t] = a; sB[t] = b; sC[t] = c;
#else
            sA[t] = make_float4(a.x, a.y, GSR_HA(a.z), GSR_HB(a.w)); sB[t] = make_float4(GSR_HC(b.x), b.y, b.z, b.w); sC[t] = c;
#endif
        }
        __syncthreads();
        {
            // instance j sits at list position p = hi-1-j; this wave needs it only if p < wmax
            int nw;
            if (SL == 64) {
                bool hit = false;
                if (lane < cnt && hi - 1 - lane < wmax) {
                    const float4 a = sA[lane], b = sB[lane];
#ifdef GSR_PRECISE_MATH
                    const float ca = GSR_QSCALE(a.z), cb = GSR_QSCALE(a.w), cc = GSR_QSCALE(b.x);
#else
                    const float ca = -2.0f * a.z, cb = -a.w, cc = -2.0f * b.x;
#endif
                    const int sx0 = tx * 16, sy0 = ty * 16 + wave * 8;  // this wavefront's 16x8 strip
                    hit = sy0 < H && !(gsr_box_min_q(a.x, a.y, ca, cb, cc, GSR_RCP(ca), GSR_RCP(cc), (float)sx0, (float)min(sx0 + 15, W - 1),
                                                     (float)sy0, (float)min(sy0 + 7, H - 1)) > sC[lane].w * GSR_LOG2E);  // (.w: the Gaussian's culling threshold)
                }
                const unsigned long long bal = __ballot(hit);
                if (hit) mylist[__builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u))] = (uint16_t)lane;
                nw = __popcll(bal);
            } else {
                nw = gsr_compact2<SL>(sQ, mylist, cnt, qmask, lane, [=](int i) { return hi - 1 - i < wmax; });
            }
            __builtin_amdgcn_wave_barrier();
#ifdef GSR_BWD_NOLOOP  // diagnostic: everything but the instance loop (what a task costs around it)
            for (int c0 = 0; c0 < 0; c0 += 64) {
#else
            for (int c0 = 0; c0 < nw; c0 += 64) {
#endif
                const int m = min(64, nw - c0);
                const int jj = mylist[min(c0 + lane, nw - 1)];
                int j = __builtin_amdgcn_readlane(jj, 0);
                float4 A = sA[j], B = sB[j];
                for (int k = 0; k < m; k++) {
                    const int jn = __builtin_amdgcn_readlane(jj, min(k + 1, m - 1));
                    const float4 An = sA[jn], Bn = sB[jn];
                    const int p = hi - 1 - j;
                    const gsr_f2 dx = gsr_splat(A.x) - pxf;
                    const float dy = A.y - pyf;
                    // log2 of the falloff, same evaluation order as the forward: fma(dy, hC dy, dx * fma(hA, dx, hB dy))
#ifdef GSR_PRECISE_MATH
                    const gsr_f2 power = {gsr_power1(A.z, A.w, B.x, dx.x, dy), gsr_power1(A.z, A.w, B.x, dx.y, dy)};
#else
                    const gsr_f2 inner = gsr_fma2(gsr_splat(A.z), dx, gsr_splat(A.w * dy));
                    const gsr_f2 power = gsr_fma2(gsr_splat(dy), gsr_splat(B.x * dy), dx * inner);
#endif
                    const gsr_f2 G = {gsr_gauss1(power.x), gsr_gauss1(power.y)};
                    const gsr_f2 al = B.y * G;
                    // alpha >= 1/255 tested on the unclamped product (0.99 > 1/255: same truth value, as in the forward); the
                    // clamp of backward.cu:524 is applied only in iterations that blend something
#ifdef GSR_PRECISE_MATH
                    const float alpha_min = 1.0f / 255.0f;
#else
                    const float alpha_min = GSR_ALPHA_LO;  // candidates from the lower edge of the guard band on (settled below)
#endif
                    unsigned long long okma = __builtin_amdgcn_ballot_w64(p < lastca) & __builtin_amdgcn_ballot_w64(power.x <= 0.0f) &
                                              __builtin_amdgcn_ballot_w64(al.x >= alpha_min);
                    unsigned long long okmb = __builtin_amdgcn_ballot_w64(p < lastcb) & __builtin_amdgcn_ballot_w64(power.y <= 0.0f) &
                                              __builtin_amdgcn_ballot_w64(al.y >= alpha_min);
                    GSR_COUNT_ADD(0, 1);
#ifdef GSR_TRACE
                    gsr_tr_it++;
#endif
                    if ((okma | okmb) != 0ull) {  // wave-uniform: some pixel of this strip blends the instance
#ifdef GSR_TRACE
                        gsr_tr_bl++;
#endif
#ifndef GSR_PRECISE_MATH
                        unsigned long long banda = 0ull, bandb = 0ull;
                        // (a second copy of the loop body without this test for batches with no flagged instance: no gain, 161 vs 158.5 us)
                        if (SL != 64 || ((bandm64 >> (j & 63)) & 1ull)) {  // wave-uniform (j is scalar): the forward flagged this instance
                            banda = __builtin_amdgcn_ballot_w64(al.x < GSR_ALPHA_HI) & okma;
                            bandb = __builtin_amdgcn_ballot_w64(al.y < GSR_ALPHA_HI) & okmb;
                        }
                        if ((banda | bandb) != 0ull) {  // rare: inside the guard band -> the reference's own expression decides (as in the forward)
                            const GsrRec* r = rec + (point_list[rg.x + p] & 0x7fffffffu);
                            const float4 ra = r->a;
                            const float2 rb = *reinterpret_cast<const float2*>(&r->b);
                            const float ey = ra.y - pyf;
                            const bool ka = gsr_blends_exact(ra.z, ra.w, rb.x, rb.y, ra.x - pxf.x, ey);
                            const bool kb = gsr_blends_exact(ra.z, ra.w, rb.x, rb.y, ra.x - pxf.y, ey);
                            okma &= ~banda | __builtin_amdgcn_ballot_w64(ka);
                            okmb &= ~bandb | __builtin_amdgcn_ballot_w64(kb);
                            GSR_COUNT_ADD(7, 1);
                        }
#endif
                        GSR_COUNT_ADD(1, 1);
                        GSR_COUNT_ADD(2, __popcll(okma) + __popcll(okmb));
                        GSR_COUNT_ADD(3, (okma != 0ull) != (okmb != 0ull));  // only one 8x8 half of the strip blends
                        const gsr_f2 alpha = {__builtin_amdgcn_fmed3f(al.x, 0.99f, -3.0e38f), __builtin_amdgcn_fmed3f(al.y, 0.99f, -3.0e38f)};
                        const gsr_f2 ae = {gsr_sel0(okma, alpha.x), gsr_sel0(okmb, alpha.y)};
                        const gsr_f2 Ge = {gsr_sel0(okma, G.x), gsr_sel0(okmb, G.y)};
                        const float4 C = sC[j];
                        const gsr_f2 oma = 1.0f - ae;
                        const gsr_f2 rinv = {GSR_RCP(oma.x), GSR_RCP(oma.y)};
                        const gsr_f2 Tn = Tr * rinv;  // T / (1 - alpha); unchanged where alpha was zeroed
                        const gsr_f2 w = ae * Tn;
                        // q = this instance's colour (depth, feature) projected on the pixels' upstream gradients; Ar = the
                        // same projection of what is accumulated BEHIND the instance (updated right after its use; the
                        // reference updates its per-channel accum_rec lazily at the top of the next contribution)
                        gsr_f2 q = C.x * g0 + C.y * g1 + C.z * g2;
                        if (AUX) q += B.z * gd + B.w * gu;
                        gsr_f2 dLda = q - Ar;
                        Ar = gsr_fma2(ae, dLda, Ar);  // = ae q + (1 - ae) Ar
                        dLda = dLda * Tn + nTfb * rinv;  // ... - T_final / (1 - alpha) * (bg . dL/dC)
                        const gsr_f2 g = Ge * dLda;
                        Tr = Tn;
                        // per-lane partials (both pixels summed): s0-2 colour, s3 depth, s4 feature, s5.. moments of
                        // g = G * dL/dalpha: sum g dx, sum g dy, sum g dx^2, sum g dx dy, sum g dy^2, sum g.  The
                        // per-Gaussian factors (conic, opacity, -1/2, viewport scale) are applied once per instance at
                        // flush time.
                        float q0, q1, q2;
                        {
                            const gsr_f2 w0 = w * g0, w1 = w * g1, w2 = w * g2;
                            float sd = 0.f, su = 0.f;
                            if (AUX) {
                                const gsr_f2 w3 = w * gd, w4 = w * gu;
                                sd = w3.x + w3.y; su = w4.x + w4.y;
                            }
                            const gsr_f2 gdx = g * dx, mxx = gdx * dx;
                            gsr_bank_reduce_dyf<AUX>(w0.x + w0.y, w1.x + w1.y, w2.x + w2.y, sd, su, gdx.x + gdx.y, mxx.x + mxx.y, g.x + g.y,
                                                     dy, q0, q1, q2);
                        }
                        float v0, v1;
                        {
                            const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(q0), __float_as_uint(q1), false, false);
                            v0 = __uint_as_float(r[0]) + __uint_as_float(r[1]);  // rows 0-1: q0 over halves, rows 2-3: q1
                        }
                        {
                            const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(q2), __float_as_uint(q2), false, false);
                            v1 = __uint_as_float(r[0]) + __uint_as_float(r[1]);
                        }
                        float x;
                        {
                            const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v0), __float_as_uint(v1), false, false);
                            x = __uint_as_float(r[0]) + __uint_as_float(r[1]);  // row 0: q0, row 2: q1, rows 1 and 3: q2
                        }
                        x += gsr_dpp<0xB1>(x);
                        x += gsr_dpp<0x4E>(x);
                        if (accfield >= 0) atomicAdd(&acc[wave][(uint32_t)j * GSR_SLOT_FLOATS + (uint32_t)accfield], x);
                    }
                    j = jn; A = An; B = Bn;
                }
            }
        }
        __syncthreads();
        if (t < cnt) {
            float4* a4 = reinterpret_cast<float4*>(&acc[0][t * GSR_SLOT_FLOATS]);
            float4* b4 = reinterpret_cast<float4*>(&acc[1][t * GSR_SLOT_FLOATS]);
            float4* dst = slots + (size_t)sSlot[t] * 3;
            const float4 p0 = b4[0], p1 = b4[1], p2 = b4[2];
            float4 o0 = a4[0], o1 = a4[1], o2 = a4[2];
            o0.x += p0.x; o0.y += p0.y; o0.z += p0.z; o0.w += p0.w;  // strip 0 + strip 1, always in this order
            o1.x += p1.x; o1.y += p1.y; o1.z += p1.z; o1.w += p1.w;
            o2.x += p2.x; o2.y += p2.y; o2.z += p2.z; o2.w += p2.w;
            {
                // moments -> the reference's per-instance sums (DGR backward.cu:586-601):
                //   dL/dmean2D = -o (cA Mx + cB My) W/2, -o (cC My + cB Mx) H/2;  dL/dconic = -o/2 (Mxx, Mxy, Myy);
                //   dL/dopacity = M0            with M* = sum over pixels of G dL/dalpha {dx, dy, dx^2, dx dy, dy^2, 1}
                const float4 A = sA[t], B = sB[t];
                const float op = B.y, Mx = o1.y, My = o1.z, Mxx = o1.w, Mxy = o2.x, Myy = o2.y;
#ifdef GSR_PRECISE_MATH
                const float cA = A.z, cB = A.w, cC = B.x;
#else
                const float k = -1.0f / GSR_LOG2E;  // back from the pre-scaled form to the conic (A, B, C)
                const float cA = 2.0f * k * A.z, cB = k * A.w, cC = 2.0f * k * B.x;
#endif
                o1.y = -op * (cA * Mx + cB * My) * ddelx_dx;
                o1.z = -op * (cC * My + cB * Mx) * ddely_dy;
                o1.w = -0.5f * op * Mxx; o2.x = -0.5f * op * Mxy; o2.y = -0.5f * op * Myy;
                a4[0] = a4[1] = a4[2] = b4[0] = b4[1] = b4[2] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
            dst[0] = o0; dst[1] = o1; dst[2] = o2;
            slot_written[sSlot[t]] = 1;
        }
        __syncthreads();
    }
    GSR_TRACE_END(2)
#ifdef GSR_TRACE  // + evaluated / blending iterations of the wave, its tile and segment (tools/wave_trace.py)
    if ((threadIdx.x & 63) == 0)
        gsr_trace_buf[((size_t)blockIdx.x * 2 + (threadIdx.x >> 6)) * 4 + 3] |= ((unsigned long long)(gsr_tr_it & 0xffff) << 4) | ((unsigned long long)(gsr_tr_bl & 0xffff) << 20) |
                                                                              ((unsigned long long)(tile & 0xffff) << 36) | ((unsigned long long)(seg & 0x3f) << 52);
#endif
}

// ---------------------------------------------------------------------------------------------
hipError_t gsr_launch_blend_forward(int W, int H, int gx, int T, const float* bg, const GsrGeom& geom,
                                    const GsrImage& image, const GsrBinning& bin, float* out_color, float* out_depth,
                                    float* out_feature, int capacity, int max_tile_count, bool only_flagged, bool inference,
                                    uint32_t* walk_depths, bool walk_depths_valid, bool already_ordered, uint32_t* ranoff_report, uint32_t serial,
                                    hipStream_t stream)
{
    if (T <= 0) return hipSuccess;
    // per-view walk depths: order the tasks by the previous visit's (not in the fix-up pass: a few flagged tiles), record this visit's
    const int xt = gsr_xcd_tiles(T);
    const uint32_t* qorder = nullptr;
    if (walk_depths && walk_depths_valid && !only_flagged && 4 * xt <= GSR_ORDER_MAX_SLOTS) {
        if (!already_ordered)  // (the one-call forward had the column scan's launch do it)
            hipLaunchKernelGGL(gsr_fwd_order_kernel, dim3(8), dim3(1024), 0, stream, T, xt, walk_depths, image.qorder);
        qorder = image.qorder;
    }
#define GSR_FWD_LAUNCH(TR)                                                                                                             \
    hipLaunchKernelGGL(gsr_blend_fwd_kernel<TR>, dim3(32 * xt), dim3(64), GSR_FWD_LDS_PAD, stream, image.ranges, bin.point_list, geom.rec, W, H, \
                       gx, T, bg, out_color, out_depth, out_feature, image.final_T, image.n_contrib, image.tile_work, image.ckpt,     \
                       gsr_seg_len(T), (uint32_t)capacity, max_tile_count < 0 ? 0x7fffffffu : (uint32_t)max_tile_count,               \
                       image.sorted_len, image.need_full, only_flagged ? image.need_full : (const uint32_t*)nullptr, image.qresume, image.info + 3,        \
                       qorder, walk_depths, only_flagged ? ranoff_report : (uint32_t*)nullptr, serial)
    if (inference) GSR_FWD_LAUNCH(false);
    else GSR_FWD_LAUNCH(true);
#undef GSR_FWD_LAUNCH
    return hipGetLastError();
}

hipError_t gsr_launch_blend_backward(int W, int H, int gx, int T, const float* bg, const GsrGeom& geom,
                                     const GsrImage& image, const GsrBinning& bin, const float* dL_dcolor,
                                     const float* dL_ddepth, const float* dL_dfeature, float* slots, uint8_t* slot_written,
                                     uint32_t* heavy_groups, int max_tile_count, hipStream_t stream)
{
    if (T <= 0) return hipSuccess;
    float4* s4 = reinterpret_cast<float4*>(slots);
    // the grid covers `nseg` segments of every tile -- as many as the frame's longest list has (round 6; rounds 4-5: all of them as soon
    // as one list reached the second tier) --; workgroups of segments a tile does not have leave at once
    const int sl = gsr_seg_len(T);
    const int nseg = std::max(GSR_SEG1 + 1, gsr_segments_for(max_tile_count, sl));
    const dim3 grid(8u * (uint32_t)gsr_xcd_tiles(T) * (uint32_t)nseg);
#define GSR_BWD_LAUNCH(A, SLEN, GD, GF)                                                                                          \
    hipLaunchKernelGGL((gsr_blend_bwd_kernel<A, SLEN>), grid, dim3(128), 0, stream, image.ranges, bin.point_list, geom.rec, W, H, \
                       gx, bg, image.final_T, image.n_contrib, image.ckpt, dL_dcolor, GD, GF, image.tile_work, T, sl,            \
                       geom.offsets, slot_written, s4, heavy_groups, image.need_full, nseg, image.info + 3, gsr_xcd_tiles(T))
#ifdef GSR_BWD_BATCH128  // long segments staged 128 instances at a time (19.5 KB of LDS: 4 waves per SIMD)
    const bool b64 = sl == 64;
#else                    // long segments in two batches of 64 (9.7 KB: 5 waves per SIMD)
    const bool b64 = true;
#endif
    if (dL_ddepth || dL_dfeature) {
        if (b64) GSR_BWD_LAUNCH(true, 64, dL_ddepth, dL_dfeature);
        else GSR_BWD_LAUNCH(true, GSR_SEG_LEN, dL_ddepth, dL_dfeature);
    } else {
        if (b64) GSR_BWD_LAUNCH(false, 64, nullptr, nullptr);
        else GSR_BWD_LAUNCH(false, GSR_SEG_LEN, nullptr, nullptr);
    }
#undef GSR_BWD_LAUNCH
    return hipGetLastError();
}
