// gsr_math.h -- per-Gaussian geometry shared by the forward preprocess and the per-Gaussian
// backward.  Both translation units are compiled with -ffp-contract=off so that this arithmetic
// is plain IEEE fp32 in the evaluation order of the reference source (GLM mat3 products expanded
// left to right, see oracle/gs_oracle.c): radii, tile rectangles, depth keys and therefore the
// whole binning are bit-exact against the oracle.
#pragma once
#include <hip/hip_runtime.h>
#include "gsr_common.h"

// CUDA's float->int conversion saturates and maps NaN to 0; make that explicit.
__device__ __forceinline__ int gsr_f2i(float v)
{
    if (!(v == v)) return 0;
    if (v >= 2147483648.0f) return 2147483647;
    if (v <= -2147483648.0f) return (-2147483647 - 1);
    return (int)v;
}

// DGR auxiliary.h:41-44 -- evaluated in double because of the 1.0 / 0.5 literals.
__device__ __forceinline__ float gsr_ndc2pix(float v, int S)
{
    return (float)((((double)v + 1.0) * (double)S - 1.0) * 0.5);
}

// DGR auxiliary.h:46-56 getRect
__device__ __forceinline__ void gsr_get_rect(float px, float py, int max_radius, int gx, int gy, int& x0, int& y0,
                                             int& x1, int& y1)
{
    const float r = (float)max_radius;
    x0 = min(gx, max(0, gsr_f2i((px - r) / 16.0f)));
    y0 = min(gy, max(0, gsr_f2i((py - r) / 16.0f)));
    x1 = min(gx, max(0, gsr_f2i((px + r + 15.0f) / 16.0f)));
    y1 = min(gy, max(0, gsr_f2i((py + r + 15.0f) / 16.0f)));
}

// Rotation from the quaternion as given (not normalised), stored R[c][r] like the glm::mat3 the
// reference builds at DGR forward.cu:136-140.
__device__ __forceinline__ void gsr_quat_to_R(const float4 q, float R[3][3])
{
    const float r = q.x, x = q.y, y = q.z, z = q.w;
    R[0][0] = 1.f - 2.f * (y * y + z * z); R[0][1] = 2.f * (x * y - r * z); R[0][2] = 2.f * (x * z + r * y);
    R[1][0] = 2.f * (x * y + r * z); R[1][1] = 1.f - 2.f * (x * x + z * z); R[1][2] = 2.f * (y * z - r * x);
    R[2][0] = 2.f * (x * z - r * y); R[2][1] = 2.f * (y * z + r * x); R[2][2] = 1.f - 2.f * (x * x + y * y);
}

// DGR forward.cu:120-154 computeCov3D:  Sigma = (S R)^T (S R), six unique entries.
__device__ __forceinline__ void gsr_cov3d(const float3 scale, float mod, const float4 rot, float cov[6])
{
    const float s[3] = { mod * scale.x, mod * scale.y, mod * scale.z };
    float R[3][3];
    gsr_quat_to_R(rot, R);
    float M[3][3];
#pragma unroll
    for (int c = 0; c < 3; c++)
#pragma unroll
        for (int r = 0; r < 3; r++) M[c][r] = s[r] * R[c][r];
#define GSR_SIG(c, r) (M[r][0] * M[c][0] + M[r][1] * M[c][1] + M[r][2] * M[c][2])
    cov[0] = GSR_SIG(0, 0); cov[1] = GSR_SIG(0, 1); cov[2] = GSR_SIG(0, 2);
    cov[3] = GSR_SIG(1, 1); cov[4] = GSR_SIG(1, 2); cov[5] = GSR_SIG(2, 2);
#undef GSR_SIG
}

struct GsrCov2D {
    float tx, ty, tz;      // view-space mean with x/z, y/z clamped to +-1.3 tan(fov/2)
    float txtz, tytz;      // unclamped ratios (the backward's grad multipliers test these)
    float limx, limy;
    float A0[3], A1[3];    // rows of the 2x3 matrix J * W_rot  (the reference's T[0][*], T[1][*])
    float a, b, c;         // 2D covariance WITH the 0.3 low-pass
};

// DGR forward.cu:76-115 computeCov2D and the recomputation at backward.cu:160-199.
__device__ __forceinline__ void gsr_cov2d(const float3 mean, const GsrCam& cam, const float cov3D[6], GsrCov2D& o)
{
    const float* vm = cam.view;
    float tx = vm[0] * mean.x + vm[4] * mean.y + vm[8] * mean.z + vm[12];
    float ty = vm[1] * mean.x + vm[5] * mean.y + vm[9] * mean.z + vm[13];
    const float tz = vm[2] * mean.x + vm[6] * mean.y + vm[10] * mean.z + vm[14];
    o.limx = 1.3f * cam.tan_fovx;
    o.limy = 1.3f * cam.tan_fovy;
    o.txtz = tx / tz;
    o.tytz = ty / tz;
    tx = fminf(o.limx, fmaxf(-o.limx, o.txtz)) * tz;
    ty = fminf(o.limy, fmaxf(-o.limy, o.tytz)) * tz;
    o.tx = tx; o.ty = ty; o.tz = tz;
    const float J00 = cam.focal_x / tz, J02 = -(cam.focal_x * tx) / (tz * tz);
    const float J11 = cam.focal_y / tz, J12 = -(cam.focal_y * ty) / (tz * tz);
    o.A0[0] = vm[0] * J00 + vm[2] * J02; o.A0[1] = vm[4] * J00 + vm[6] * J02; o.A0[2] = vm[8] * J00 + vm[10] * J02;
    o.A1[0] = vm[1] * J11 + vm[2] * J12; o.A1[1] = vm[5] * J11 + vm[6] * J12; o.A1[2] = vm[9] * J11 + vm[10] * J12;
    const float V[3][3] = { { cov3D[0], cov3D[1], cov3D[2] }, { cov3D[1], cov3D[3], cov3D[4] }, { cov3D[2], cov3D[4], cov3D[5] } };
    float X0[3], X1[3];
#pragma unroll
    for (int c = 0; c < 3; c++) {
        X0[c] = o.A0[0] * V[c][0] + o.A0[1] * V[c][1] + o.A0[2] * V[c][2];
        X1[c] = o.A1[0] * V[c][0] + o.A1[1] * V[c][1] + o.A1[2] * V[c][2];
    }
    const float c00 = X0[0] * o.A0[0] + X0[1] * o.A0[1] + X0[2] * o.A0[2];
    const float c01 = X1[0] * o.A0[0] + X1[1] * o.A0[1] + X1[2] * o.A0[2];
    const float c11 = X1[0] * o.A1[0] + X1[1] * o.A1[1] + X1[2] * o.A1[2];
    o.a = c00 + 0.3f; o.b = c01; o.c = c11 + 0.3f;
}

// Minimum over the pixel box [bx0,bx1] x [by0,by1] of q(d) = 0.5 (A dx^2 + C dy^2) + B dx dy with
// d = mean - pixel, i.e. of -power (DGR forward.cu:523).  q is convex with its minimum (0) at the mean:
//   * mean inside the box -> 0;
//   * otherwise, moving from any box point towards the mean lowers q and leaves the box through an edge that
//     FACES the mean, so the box minimum is the smaller of the clamped 1-D minima on the (at most two) facing
//     edges: the vertical edge nearest to the mean if the mean is outside in x, the horizontal one if in y.
// rA = 1/A, rC = 1/C are passed in (once per Gaussian); an approximate reciprocal only moves the evaluation
// point off the edge optimum by an ulp, which changes q by O(ulp^2).
__device__ __forceinline__ float gsr_box_min_q(float mx, float my, float A, float B, float C, float rA, float rC,
                                               float bx0, float bx1, float by0, float by1)
{
    const float dx0 = mx - bx1, dx1 = mx - bx0, dy0 = my - by1, dy1 = my - by0;  // ranges of d = mean - pixel
    const bool out_x = dx0 > 0.f || dx1 < 0.f, out_y = dy0 > 0.f || dy1 < 0.f;
    const float ex = dx0 > 0.f ? dx0 : dx1;  // d.x on the vertical edge nearest to the mean
    const float ey = dy0 > 0.f ? dy0 : dy1;
    const float dy = fminf(dy1, fmaxf(dy0, -B * ex * rC));
    const float dx = fminf(dx1, fmaxf(dx0, -B * ey * rA));
    const float qx = 0.5f * (A * ex * ex + C * dy * dy) + B * ex * dy;
    const float qy = 0.5f * (A * dx * dx + C * ey * ey) + B * dx * ey;
    float best = (out_x || out_y) ? 3.0e38f : 0.f;
    best = out_x ? qx : best;
    best = out_y ? fminf(best, qy) : best;
    return best;
}

// Culling threshold.  A (Gaussian, pixel) pair is blended only if alpha = opacity * exp(power) >= 1/255
// (DGR forward.cu:534 skips everything below), i.e. iff q = -power <= ln(255 opacity).  tau adds a 0.01
// safety margin on q -- orders of magnitude above fp32 rounding of `power` -- so a box (a 16x16 tile at
// binning time, an 8x8 quadrant inside the blend kernels) is skipped only when no pixel in it can pass.
// Skipped work contributes exactly nothing: images and gradients are unchanged.  NaNs compare false, so
// a degenerate conic or a non-positive opacity keeps the instance.
#define GSR_CULL_MARGIN 0.01f
// ... plus a margin that grows with the MAGNITUDE of the quadratic form's terms (round 5).  The reference evaluates
// power = -0.5 (A dx^2 + C dy^2) - B dx dy in fp32 (forward.cu:528); for a large, thin, tilted splat the three terms are of order
// 1e4 - 1e6 and cancel to a power of order 1-10, so the value the REFERENCE tests against 1/255 carries an absolute error of
// ~2e-7 x the sum of the terms' magnitudes -- 0.03 - 0.3 for such a splat, more than the fixed margin: a tile (quadrant, strip) that
// exact arithmetic rules out may still blend in the reference.  (Found on the initialised, untrained scene of round 5 -- anchors'
// scales up to 0.4 m next to a camera: the parity BUILD, which blends with the reference's own expression, disagreed with the oracle
// on the same 50 gradient elements as the shipped one.)  S bounds the terms over a box of pixel offsets; GSR_CULL_ERR * S is added
// to tau (same units as the conic passed in: natural or log2-scaled, the bound is homogeneous).
#define GSR_CULL_ERR 1.0e-6f
// (no contraction: the forward and the backward blend derive an instance's guard band from this value and must get the same bits
// whatever code surrounds the call)
__device__ __forceinline__ float gsr_terms_bound(float A, float B, float C, float DX, float DY)
{
#pragma clang fp contract(off)
    const float tA = fabsf(A) * DX * DX, tC = fabsf(C) * DY * DY, tB = fabsf(B) * DX * DY;
    return 0.5f * (tA + tC) + tB;
}
__device__ __forceinline__ float gsr_box_terms_bound(float mx, float my, float A, float B, float C, float bx0, float bx1, float by0, float by1)
{
    return gsr_terms_bound(A, B, C, fmaxf(fabsf(mx - bx0), fabsf(mx - bx1)), fmaxf(fabsf(my - by0), fabsf(my - by1)));
}
__device__ __forceinline__ float gsr_cull_tau(float opacity) { return logf(255.0f * opacity) + GSR_CULL_MARGIN; }
// v_log_f32 version for the blend kernels' quadrant test (1 ulp: irrelevant next to the margin)
__device__ __forceinline__ float gsr_cull_tau_fast(float opacity) { return __logf(255.0f * opacity) + GSR_CULL_MARGIN; }

__device__ __forceinline__ bool gsr_tile_survives(float mx, float my, float A, float B, float C, float rA, float rC,
                                                  float tau, int tx, int ty, int W, int H)
{
    const float bx0 = (float)(tx * 16), by0 = (float)(ty * 16);
    const float bx1 = (float)min(tx * 16 + 15, W - 1), by1 = (float)min(ty * 16 + 15, H - 1);
    return !(gsr_box_min_q(mx, my, A, B, C, rA, rC, bx0, bx1, by0, by1) > tau);
}
// Occlusion mass of a (Gaussian, tile) instance (gsr_tuning.occlusion_cut): -log2(1 - alpha_min) in fixed point, alpha_min the
// SMALLEST alpha the Gaussian has at any pixel of the tile, 0 unless that is safely above 1/255 (every pixel of the tile then
// blends the instance -- power <= 0 holds wherever alpha <= opacity -- with at least this alpha: DGR forward.cu:528-537).  q is
// convex, so its maximum over the tile's pixel box is at a corner.  tau_exact = ln(255 opacity).  Floors: never over-estimates.
__device__ __forceinline__ uint32_t gsr_tile_occlusion_mass(float mx, float my, float A, float B, float C, float tau_exact, int tx, int ty,
                                                            int W, int H)
{
    const float dx0 = mx - (float)(tx * 16), dx1 = mx - (float)min(tx * 16 + 15, W - 1);
    const float dy0 = my - (float)(ty * 16), dy1 = my - (float)min(ty * 16 + 15, H - 1);
    auto q = [&](float dx, float dy) { return 0.5f * (A * dx * dx + C * dy * dy) + B * dx * dy; };
    const float qmax = fmaxf(fmaxf(q(dx0, dy0), q(dx1, dy0)), fmaxf(q(dx0, dy1), q(dx1, dy1)));
    if (!(qmax <= tau_exact - 2.0f * GSR_CULL_MARGIN)) return 0u;  // (NaN-safe) not a whole-tile instance
    // alpha_min = exp(tau_exact - qmax) / 255, a hair low (1e-3 relative) against the rounding of q; clamp as the blend does
    const float amin = fminf(0.99f, __expf(tau_exact - qmax - 0.001f) * (1.0f / 255.0f));
    return (uint32_t)(-__log2f(1.0f - amin) * GSR_OCC_FIXED);
}

// survivor bit of rectangle position i (row-major); rectangles larger than 64 tiles keep their tail
__device__ __forceinline__ bool gsr_mask_bit(unsigned long long mask, int i) { return i >= 64 || ((mask >> i) & 1ull); }
// Calls f(x, y) for every surviving tile of the rectangle, in row-major order.  Walks the SET BITS of the
// mask (a typical Gaussian keeps 1-4 of its 4-16 rectangle tiles), one row at a time so that no integer
// division is needed; positions >= 64 (rectangles larger than 64 tiles) always survive.
template <typename F>
__device__ __forceinline__ void gsr_for_each_tile(const uint2 rc, unsigned long long mask, F f)
{
    const int x0 = rc.x & 0xffff, x1 = rc.x >> 16, y0 = rc.y & 0xffff, y1 = rc.y >> 16;
    const int w = x1 - x0;
    if (w <= 0) return;
    int base = 0;
    for (int y = y0; y < y1 && base < 64; y++, base += w) {
        unsigned long long rm = mask >> base;
        if (w < 64) rm &= (1ull << w) - 1ull;
        while (rm) {
            const int b = __builtin_ctzll(rm);
            rm &= rm - 1ull;
            f(x0 + b, y);
        }
    }
    const int area = w * (y1 - y0);
    for (int i = 64; i < area; i++) f(x0 + i % w, y0 + i / w);
}

// Index of the r-th (0-based) set bit of m; r < popcount(m).  Branch-free binary search on popcounts.
__device__ __forceinline__ uint32_t gsr_select_bit(unsigned long long m, uint32_t r)
{
    uint32_t w = (uint32_t)m, pos = 0;
    uint32_t c = (uint32_t)__popc(w);
    if (r >= c) { r -= c; w = (uint32_t)(m >> 32); pos = 32; }
#pragma unroll
    for (int half = 16; half >= 1; half >>= 1) {
        const uint32_t lowbits = w & ((1u << half) - 1u);
        c = (uint32_t)__popc(lowbits);
        const bool up = r >= c;
        r -= up ? c : 0u;
        pos += up ? half : 0;
        w = up ? (w >> half) : lowbits;
    }
    return pos;
}

// number of (Gaussian, surviving tile) instances of a rectangle + survivor mask (0 for the zero rectangle)
__device__ __forceinline__ uint32_t gsr_rect_count(const uint2 rc, const unsigned long long mask)
{
    const int w = (int)(rc.x >> 16) - (int)(rc.x & 0xffff), h = (int)(rc.y >> 16) - (int)(rc.y & 0xffff);
    const int area = w > 0 ? w * h : 0;
    const unsigned long long m = area >= 64 ? mask : mask & ((1ull << area) - 1ull);
    return (uint32_t)__popcll(m) + (uint32_t)max(area - 64, 0);
}

// Wave-wide inclusive scans on DPP (row_shr 1/2/4/8 inside the 16-lane rows, then row_bcast15 / row_bcast31 across
// rows): 6 VALU ops, no LDS traffic.  Lanes outside the source pattern contribute the identity 0.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t gsr_dpp_u32(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xf, false);
}
__device__ __forceinline__ uint32_t gsr_wave_scan_add(uint32_t v)
{
    v += gsr_dpp_u32<0x111, 0xf>(v);
    v += gsr_dpp_u32<0x112, 0xf>(v);
    v += gsr_dpp_u32<0x114, 0xf>(v);
    v += gsr_dpp_u32<0x118, 0xf>(v);
    v += gsr_dpp_u32<0x142, 0xa>(v);
    v += gsr_dpp_u32<0x143, 0xc>(v);
    return v;
}
__device__ __forceinline__ uint32_t gsr_wave_scan_max(uint32_t v)
{
    v = max(v, gsr_dpp_u32<0x111, 0xf>(v));
    v = max(v, gsr_dpp_u32<0x112, 0xf>(v));
    v = max(v, gsr_dpp_u32<0x114, 0xf>(v));
    v = max(v, gsr_dpp_u32<0x118, 0xf>(v));
    v = max(v, gsr_dpp_u32<0x142, 0xa>(v));
    v = max(v, gsr_dpp_u32<0x143, 0xc>(v));
    return v;
}

// Generic wave-dense enumeration: lane L contributes cnt_L items; the wave walks all sum(cnt) items 64 at a time and
// calls f(owner_lane, r, active) in EVERY lane each round (active = this lane holds an item: the r-th of `owner`), so
// that f may itself use cross-lane reads of the owner's registers.  `heads`: 64 words of LDS private to the wave.
template <typename F>
__device__ __forceinline__ void gsr_wave_dense(const uint32_t cnt, volatile uint32_t* heads, F f)
{
    const int lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    const uint32_t incl = gsr_wave_scan_add(cnt);
    const uint32_t excl = incl - cnt;
    const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    for (uint32_t base = 0; base < total; base += 64) {
        heads[lane] = 0u;
        if (cnt != 0u && excl < base + 64u && incl > base) heads[max(excl, base) - base] = (uint32_t)lane;
        const int owner = (int)gsr_wave_scan_max(heads[lane]);
        const uint32_t oexcl = __shfl(excl, owner, 64);
        f(owner, base + lane - oexcl, base + lane < total);
    }
}

// Wave-dense version of gsr_for_each_tile.  Every lane passes the rectangle and survivor mask of ITS Gaussian (a zero
// rectangle for none); the wave then enumerates all (Gaussian, surviving tile) instances of its 64 Gaussians 64 at a
// time, one instance per lane, and calls f(owner_lane, x, y, owner's payload) with all lanes (but the last round's tail) active.
// The per-Gaussian walk has as many rounds as the LARGEST footprint in the wave (~18 for a mean of 2.7), and every
// round costs one LDS atomic instruction whose latency does not depend on the number of active lanes; the dense form
// needs total/64 rounds.  Must be called by all 64 lanes (no divergence around the call).
template <bool WITH_POS, typename F>
__device__ __forceinline__ void gsr_wave_for_each_instance_t(const uint2 rc, const unsigned long long mask,
                                                             const uint32_t payload /* handed to f as the owner's value */,
                                                             volatile uint32_t* heads /* LDS, 64 words private to the wave */, F f)
{
    const int lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    const int x0 = rc.x & 0xffff, w = (int)(rc.x >> 16) - x0, y0 = rc.y & 0xffff, h = (int)(rc.y >> 16) - y0;
    const int area = w > 0 ? w * h : 0;
    const unsigned long long m = area >= 64 ? mask : mask & ((1ull << area) - 1ull);
    const uint32_t cnt = (uint32_t)__popcll(m) + (uint32_t)max(area - 64, 0);   // == gsr_survivors(mask, area)
    const uint32_t incl = gsr_wave_scan_add(cnt);
    const uint32_t excl = incl - cnt;
    const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    for (uint32_t base = 0; base < total; base += 64) {
        const uint32_t i = base + lane;
        // owner of instance i = the last lane whose first instance is <= i.  Every lane whose range meets this round
        // drops its id at the round-relative position of its first instance (ranges are disjoint, so the positions
        // are too); an inclusive max-scan over the lanes fills the gaps.  DPP only: no dependent LDS round trips.
        heads[lane] = 0u;
        if (cnt != 0u && excl < base + 64u && incl > base) heads[max(excl, base) - base] = (uint32_t)lane;
        const int lo = (int)gsr_wave_scan_max(heads[lane]);
        const uint32_t oincl = __shfl(incl, lo, 64), ocnt = __shfl(cnt, lo, 64);
        const uint32_t orx = __shfl(rc.x, lo, 64), ory = __shfl(rc.y, lo, 64);
        const uint32_t omlo = __shfl((uint32_t)m, lo, 64), omhi = __shfl((uint32_t)(m >> 32), lo, 64);
        const uint32_t opay = __shfl(payload, lo, 64);  // cross-lane reads stay outside the divergent part below
        if (i < total) {
            const uint32_t r = i - (oincl - ocnt);
            const unsigned long long om = ((unsigned long long)omhi << 32) | omlo;
            const uint32_t opc = (uint32_t)__popcll(om);
            const uint32_t pos = r < opc ? gsr_select_bit(om, r) : 64u + (r - opc);
            const int ox0 = orx & 0xffff, ow = (int)(orx >> 16) - ox0, oy0 = ory & 0xffff;
            int row = (int)((float)pos * __frcp_rn((float)ow));  // pos < 2^24: off by at most one, fixed below
            int col = (int)pos - row * ow;
            if (col < 0) { row--; col += ow; }
            if (col >= ow) { row++; col -= ow; }
            if constexpr (WITH_POS) f(lo, ox0 + col, oy0 + row, opay, pos);  // + the instance's position in the owner's rectangle
            else f(lo, ox0 + col, oy0 + row, opay);
        }
    }
}
template <typename F>
__device__ __forceinline__ void gsr_wave_for_each_instance(const uint2 rc, const unsigned long long mask, const uint32_t payload,
                                                           volatile uint32_t* heads, F f)
{
    gsr_wave_for_each_instance_t<false>(rc, mask, payload, heads, f);
}

// number of surviving tiles of a rectangle with `area` tiles
__device__ __forceinline__ int gsr_survivors(unsigned long long mask, int area)
{
    return area >= 64 ? __popcll(mask) + (area - 64) : __popcll(mask & ((1ull << area) - 1ull));
}

// Spherical-harmonics constants (DGR auxiliary.h:22-39)
#define GSR_SH_C0 0.28209479177387814f
#define GSR_SH_C1 0.4886025119029199f
static __device__ const float GSR_SH_C2[5] = { 1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                                                     -1.0925484305920792f, 0.5462742152960396f };
static __device__ const float GSR_SH_C3[7] = { -0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                                                     0.3731763325901154f, -0.4570457994644658f, 1.445305721320277f,
                                                     -0.5900435899266435f };
