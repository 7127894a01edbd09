// preprocess.hip -- per-Gaussian forward stage (one thread per Gaussian, pure streaming).
//
// Replaces DGR forward.cu:157-267 preprocessCUDA, :271-346 filter_preprocessCUDA, :352-433
// position2D_preprocessCUDA and rasterizer_impl.cu:54-66 checkFrustum.  Built with
// -ffp-contract=off (see gsr_math.h): outputs are bit-exact against oracle/gs_oracle.c.
//
// HBM traffic per Gaussian (mode 0): read 60 B (mean 12, scale 12, quat 16, opacity 4, feature 4,
// colour 12), write the 64-B record (whole waves write all their records, zeros for culled Gaussians: the
// stores go out coalesced through LDS) + rect 8 + depthkey 4 + tiles 4 + tmask 8 + radii 4 = 92 B.
// cov3D is NOT stored: the backward recomputes it from scale/quat (bit-identical, cheaper than
// 24 B out + 24 B in).
#include "gsr_math.h"

// DGR forward.cu:22-73 computeColorFromSH
__device__ __forceinline__ float3 gsr_sh_to_rgb(int idx, int deg, int M, float3 pos, const GsrCam& cam, const float* shs,
                                                uint8_t* clamped)
{
    float dx = pos.x - cam.campos[0], dy = pos.y - cam.campos[1], dz = pos.z - cam.campos[2];
    const float len = sqrtf(dx * dx + dy * dy + dz * dz);
    const float x = dx / len, y = dy / len, z = dz / len;
    const float* sh = shs + (size_t)idx * M * 3;
    float res[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
#define S(i) sh[(i) * 3 + k]
        float v = GSR_SH_C0 * S(0);
        if (deg > 0) {
            v = v - GSR_SH_C1 * y * S(1) + GSR_SH_C1 * z * S(2) - GSR_SH_C1 * x * S(3);
            if (deg > 1) {
                const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
                v = v + GSR_SH_C2[0] * xy * S(4) + GSR_SH_C2[1] * yz * S(5) + GSR_SH_C2[2] * (2.0f * zz - xx - yy) * S(6)
                      + GSR_SH_C2[3] * xz * S(7) + GSR_SH_C2[4] * (xx - yy) * S(8);
                if (deg > 2) {
                    v = v + GSR_SH_C3[0] * y * (3.0f * xx - yy) * S(9) + GSR_SH_C3[1] * xy * z * S(10)
                          + GSR_SH_C3[2] * y * (4.0f * zz - xx - yy) * S(11)
                          + GSR_SH_C3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy) * S(12)
                          + GSR_SH_C3[4] * x * (4.0f * zz - xx - yy) * S(13) + GSR_SH_C3[5] * z * (xx - yy) * S(14)
                          + GSR_SH_C3[6] * x * (xx - 3.0f * yy) * S(15);
                }
            }
        }
#undef S
        res[k] = v + 0.5f;
        clamped[3 * idx + k] = (res[k] < 0);
    }
    return make_float3(fmaxf(res[0], 0.0f), fmaxf(res[1], 0.0f), fmaxf(res[2], 0.0f));
}

// MODE 0: full preprocess (writes the geometry state)   MODE 1: radii only   MODE 2: radii + px/py
#ifndef GSR_PRE_WAVES
#define GSR_PRE_WAVES 0  // waves per SIMD the register allocation is held to (0 = the allocator's own choice: 76 VGPRs = 6 waves)
#endif
#if GSR_PRE_WAVES > 0
#define GSR_PRE_ATTR __attribute__((amdgpu_waves_per_eu(GSR_PRE_WAVES, GSR_PRE_WAVES)))
#else
#define GSR_PRE_ATTR
#endif
template <int MODE>
__global__ void __launch_bounds__(256) GSR_PRE_ATTR gsr_preprocess_kernel(
    int P, int D, int M, int tile_cull, const GsrCam cam, const float* __restrict__ means3D, const float* __restrict__ scales,
    const float* __restrict__ rotations, const float* __restrict__ opacities, const float* __restrict__ features,
    const float* __restrict__ shs, const float* __restrict__ cov3D_precomp, const float* __restrict__ colors_precomp,
    GsrRec* __restrict__ rec, uint2* __restrict__ rect, uint32_t* __restrict__ depthkey, uint32_t* __restrict__ tiles,
    unsigned long long* __restrict__ tmask, uint8_t* __restrict__ clamped, int32_t* __restrict__ radii, float* __restrict__ out_px, float* __restrict__ out_py,
    uint32_t* __restrict__ occ_mass)
{
    __shared__ uint32_t s_heads[MODE == 0 ? 256 : 1];
    __shared__ unsigned long long s_mask[MODE == 0 ? 256 : 1];
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= P) return;

    int radius = 0;
    uint2 rc = make_uint2(0u, 0u);
    uint32_t ntiles = 0;
    float pix = 0.f, piy = 0.f;

    const float3 p = make_float3(means3D[3 * idx], means3D[3 * idx + 1], means3D[3 * idx + 2]);
    // every input of the covariance path is requested together with the mean, not behind the near-plane test (one dependent
    // memory round trip less; the culled ~20 % of a scene cost 32 B each)
    float3 s_in = make_float3(0.f, 0.f, 0.f);
    float4 q_in = make_float4(0.f, 0.f, 0.f, 0.f);
    float op_in = 0.f;
    if (!cov3D_precomp) {
        s_in = make_float3(scales[3 * idx], scales[3 * idx + 1], scales[3 * idx + 2]);
        q_in = reinterpret_cast<const float4*>(rotations)[idx];
    }
    if (MODE == 0) op_in = opacities[idx];
    const float* pm = cam.proj;
    const float* vm = cam.view;
    // DGR auxiliary.h:139-164 in_frustum + forward.cu:200-204
    const float hx = pm[0] * p.x + pm[4] * p.y + pm[8] * p.z + pm[12];
    const float hy = pm[1] * p.x + pm[5] * p.y + pm[9] * p.z + pm[13];
    const float hw = pm[3] * p.x + pm[7] * p.y + pm[11] * p.z + pm[15];
    const float p_w = 1.0f / (hw + 0.0000001f);
    const float projx = hx * p_w, projy = hy * p_w;
    const float viewz = vm[2] * p.x + vm[6] * p.y + vm[10] * p.z + vm[14];

    float conx = 0.f, cony = 0.f, conz = 0.f, cova = 0.f, covc = 0.f;
    if (viewz > 0.2f) {
        float cov3D[6];
        if (cov3D_precomp) {
#pragma unroll
            for (int i = 0; i < 6; i++) cov3D[i] = cov3D_precomp[6 * idx + i];
        } else {
            gsr_cov3d(s_in, cam.scale_modifier, q_in, cov3D);
        }
        GsrCov2D c2;
        gsr_cov2d(p, cam, cov3D, c2);
        const float det = c2.a * c2.c - c2.b * c2.b;
        if (det != 0.0f) {
            const float det_inv = 1.f / det;
            conx = c2.c * det_inv; cony = -c2.b * det_inv; conz = c2.a * det_inv;
            cova = c2.a; covc = c2.c;
            const float mid = 0.5f * (c2.a + c2.c);
            const float lambda1 = mid + sqrtf(fmaxf(0.1f, mid * mid - det));
            const float lambda2 = mid - sqrtf(fmaxf(0.1f, mid * mid - det));
            const float my_radius = ceilf(3.f * sqrtf(fmaxf(lambda1, lambda2)));
            pix = gsr_ndc2pix(projx, cam.W);
            piy = gsr_ndc2pix(projy, cam.H);
            int x0, y0, x1, y1;
            gsr_get_rect(pix, piy, gsr_f2i(my_radius), cam.gx, cam.gy, x0, y0, x1, y1);
            // radius <= 0 can only come from a NaN covariance (ceil(3 sqrt(lambda)) >= 1 otherwise): the reference then
            // has radii = 0 but tiles_touched > 0 and reads uninitialised sort keys; here such a Gaussian is culled
            if ((x1 - x0) * (y1 - y0) != 0 && gsr_f2i(my_radius) > 0) {
                radius = gsr_f2i(my_radius);
                ntiles = (uint32_t)((y1 - y0) * (x1 - x0));
                rc = make_uint2((uint32_t)x0 | ((uint32_t)x1 << 16), (uint32_t)y0 | ((uint32_t)y1 << 16));
            }
        }
    }

    radii[idx] = radius;
    if (MODE == 1) return;
    if (MODE == 2) {
        out_px[idx] = radius > 0 ? pix : 0.f;
        out_py[idx] = radius > 0 ? piy : 0.f;
        return;
    }
    rect[idx] = rc;
    depthkey[idx] = __float_as_uint(viewz);
    // the record's remaining inputs: in flight during the tile test
    float3 col_in = make_float3(0.f, 0.f, 0.f);
    float feat_in = 0.f;
    if (radius > 0) {
        if (colors_precomp) col_in = make_float3(colors_precomp[3 * idx], colors_precomp[3 * idx + 1], colors_precomp[3 * idx + 2]);
        feat_in = features[idx];
    }
    unsigned long long mask = ~0ull;
    float tau = 0.f, tau_plain = 0.f;
    int4 cull_box = make_int4(0, 0, 0, 0);  // tiles (x, y, w, h) of the alpha >= 1/255 ellipse box inside the rectangle
    if (radius > 0) {
        tau_plain = gsr_cull_tau(op_in);
        // + the rounding of the reference's own fp32 `power` over this Gaussian's rectangle (gsr_math.h GSR_CULL_ERR): |d| <= radius + a tile
        const float reach = (float)radius + 16.0f;
        tau = tau_plain + GSR_CULL_ERR * gsr_terms_bound(conx, cony, conz, reach, reach);
        if (tile_cull) {
            const int x0 = rc.x & 0xffff, x1 = rc.x >> 16, y0 = rc.y & 0xffff, y1 = rc.y >> 16;
            // Tiles outside the axis-aligned bounding box of the alpha >= 1/255 ellipse {q <= tau} cannot
            // survive: |dx| <= sqrt(2 tau cov_xx), |dy| <= sqrt(2 tau cov_yy) (cov = inverse conic).  Only the
            // tiles inside box & rectangle run the exact test; for a typical splat that is 1-4 of the 4-16
            // rectangle tiles, and it bounds the trip count of the slowest lane of the wave.
            const float t2 = 2.0f * fmaxf(tau, 0.0f);
            const float ex = sqrtf(t2 * cova) + 1.0f, ey = sqrtf(t2 * covc) + 1.0f;  // +1 px slack
            const int bx0 = max(x0, gsr_f2i((pix - ex) * 0.0625f)), bx1 = min(x1, gsr_f2i((pix + ex) * 0.0625f) + 1);
            const int by0 = max(y0, gsr_f2i((piy - ey) * 0.0625f)), by1 = min(y1, gsr_f2i((piy + ey) * 0.0625f) + 1);
            cull_box = make_int4(bx0, by0, max(bx1 - bx0, 0), max(by1 - by0, 0));
            mask = 0ull;
            if (!(tau == tau)) {  // NaN tau (non-positive opacity): keep everything, like the box test would
                mask = ~0ull;
                cull_box.z = cull_box.w = 0;
            }
        }
    }
    if (tile_cull) {
        // The exact test runs once per (Gaussian, tile of its ellipse box).  A per-lane loop has as many trips as the
        // LARGEST box in the wave; full waves instead enumerate all their boxes' tiles densely, one per lane per
        // round (the owner's operands come over ds_bpermute, the survivor bits meet in an LDS word per Gaussian).
        const bool full_wave = blockIdx.x * blockDim.x + (threadIdx.x | 63u) < (unsigned)P;  // wave-uniform
        const int x0 = rc.x & 0xffff, y0 = rc.y & 0xffff, wd = (int)(rc.x >> 16) - x0;
        const float rA = 1.0f / conx, rC = 1.0f / conz;
        if (full_wave) {
            const int lane = threadIdx.x & 63;
            volatile uint32_t* heads = s_heads + (threadIdx.x & ~63u);
            unsigned long long* wmask = s_mask + (threadIdx.x & ~63u);
            wmask[lane] = 0ull;
            gsr_wave_dense((uint32_t)(cull_box.z * cull_box.w), heads, [&](int owner, uint32_t r, bool act) {
                const float o_pix = __shfl(pix, owner, 64), o_piy = __shfl(piy, owner, 64);
                const float o_cx = __shfl(conx, owner, 64), o_cy = __shfl(cony, owner, 64), o_cz = __shfl(conz, owner, 64);
                const float o_rA = __shfl(rA, owner, 64), o_rC = __shfl(rC, owner, 64), o_tau = __shfl(tau, owner, 64);
                const int o_bx0 = __shfl(cull_box.x, owner, 64), o_by0 = __shfl(cull_box.y, owner, 64), o_bw = __shfl(cull_box.z, owner, 64);
                const int o_x0 = __shfl(x0, owner, 64), o_y0 = __shfl(y0, owner, 64), o_wd = __shfl(wd, owner, 64);
                const uint32_t o_bkt = occ_mass ? (uint32_t)__shfl((int)gsr_occ_bucket(__float_as_uint(viewz)), owner, 64) : 0u;
                const float o_tau_plain = occ_mass ? __shfl(tau_plain, owner, 64) : 0.f;
                uint32_t* const occ_xcd = occ_mass ? occ_mass + (size_t)(__builtin_amdgcn_s_getreg(63508) & 7u) * (size_t)(cam.gx * cam.gy) * GSR_OCC_BUCKETS
                                                   : nullptr;  // (XCC_ID of the hardware slot this wave runs on)
                if (act) {
                    int row = (int)((float)r * __frcp_rn((float)o_bw));  // off by at most one, fixed below
                    int col = (int)r - row * o_bw;
                    if (col < 0) { row--; col += o_bw; }
                    if (col >= o_bw) { row++; col -= o_bw; }
                    const int x = o_bx0 + col, y = o_by0 + row;
                    const int i = (y - o_y0) * o_wd + (x - o_x0);
                    // (rectangle positions beyond 64 have no mask bit: always binned -- and, being binned, they count as occluders too)
                    if (i >= 64 ? occ_mass != nullptr : gsr_tile_survives(o_pix, o_piy, o_cx, o_cy, o_cz, o_rA, o_rC, o_tau, x, y, cam.W, cam.H)) {
                        if (i < 64) atomicOr(&wmask[owner], 1ull << i);
                        if (occ_mass) {  // (wave-uniform) occlusion cut-off: the instance's whole-tile mass into its (tile, depth bucket) sum
                            // (the EXACT threshold here, without either margin: a mass must never be over-estimated)
                            const uint32_t m = gsr_tile_occlusion_mass(o_pix, o_piy, o_cx, o_cy, o_cz, o_tau_plain - GSR_CULL_MARGIN, x, y, cam.W, cam.H);
                            // integer adds: order-free.  Into THIS XCD's copy of the table with an L2-local (workgroup-scope)
                            // atomic: every workgroup that touches the copy runs on this XCD, i.e. behind the same L2; device-scope
                            // atomics go to memory and cost this kernel 63 us on a large-splat frame
                            // ASSUMPTION (gfx942 / gfx950 only, stated in DESIGN 3.1): an XCD's workgroup-scope atomics execute in that XCD's
                            // L2 and reach memory at the kernel boundary, so all adds to one copy are ordered by that L2.  The HIP memory
                            // model does not promise this; any other target takes agent scope.  A lost add could only LOWER a mass, i.e.
                            // drop fewer instances: the cut-off stays conservative either way.
#if defined(__gfx942__) || defined(__gfx950__)
                            if (m) __hip_atomic_fetch_add(&occ_xcd[(size_t)(y * cam.gx + x) * GSR_OCC_BUCKETS + o_bkt], m, __ATOMIC_RELAXED,
                                                          __HIP_MEMORY_SCOPE_WORKGROUP);
#else
                            if (m) __hip_atomic_fetch_add(&occ_xcd[(size_t)(y * cam.gx + x) * GSR_OCC_BUCKETS + o_bkt], m, __ATOMIC_RELAXED,
                                                          __HIP_MEMORY_SCOPE_AGENT);
#endif
                        }
                    }
                }
            });
            if (cull_box.z * cull_box.w > 0) mask = wmask[lane];
        } else if (cull_box.z * cull_box.w > 0) {
            for (int y = cull_box.y; y < cull_box.y + cull_box.w; y++)
                for (int x = cull_box.x; x < cull_box.x + cull_box.z; x++) {
                    const int i = (y - y0) * wd + (x - x0);
                    if (i < 64 && gsr_tile_survives(pix, piy, conx, cony, conz, rA, rC, tau, x, y, cam.W, cam.H)) mask |= 1ull << i;
                }
        }
    }
    tmask[idx] = mask;
    tiles[idx] = radius > 0 ? (uint32_t)gsr_survivors(mask, (int)ntiles) : 0u;  // gradient slots of this Gaussian
    // The 64-byte record goes out through LDS: written lane-major, read back so that each of the wave's four store instructions
    // covers 1 KB of consecutive addresses -- four requests per record line otherwise (a lane's float4 stores are 64 bytes apart;
    // round 3: 39 -> 36 us).
    // Full waves only (the tail wave and the filter modes keep the direct stores); records of culled Gaussians are written too
    // (zeros: nothing reads them).
    __shared__ float4 s_rec[MODE == 0 ? 256 * 4 : 1];
    const bool full_wave_rec = blockIdx.x * blockDim.x + (threadIdx.x | 63u) < (unsigned)P;  // wave-uniform
    if (full_wave_rec) {
        float4 ra = make_float4(0.f, 0.f, 0.f, 0.f), rb = ra, rcc = ra;
        uint4 rd = make_uint4(0u, 0u, 0u, 0u);
        if (radius > 0) {
            float3 col = col_in;
            if (!colors_precomp) col = gsr_sh_to_rgb(idx, D, M, p, cam, shs, clamped);
            // the RAW conic (round 4; both builds): the blend kernels scale it once per staged instance and need the raw
            // values for the in-band re-check of the alpha = 1/255 decision (blend.hip, gsr_blends_exact)
            ra = make_float4(pix, piy, conx, cony);
            rb = make_float4(conz, op_in, viewz, feat_in);
            // c.w: the culling threshold of this Gaussian, ln(255 opacity) + both margins (gsr_math.h) -- the blend kernels' quadrant / strip
            // tests read it instead of recomputing a logarithm per staged instance; d.x: rectangle width (gradient-slot numbering)
            rcc = make_float4(col.x, col.y, col.z, tau);
            rd = make_uint4((rc.x >> 16) - (rc.x & 0xffff), (rc.x & 0xffff) | ((rc.y & 0xffff) << 16), (uint32_t)mask, (uint32_t)(mask >> 32));
        }
        const int lane = threadIdx.x & 63;
        float4* sw = s_rec + (threadIdx.x & ~63u) * 4;  // this wave's 256 float4
        sw[lane * 4 + 0] = ra; sw[lane * 4 + 1] = rb; sw[lane * 4 + 2] = rcc;
        sw[lane * 4 + 3] = make_float4(__uint_as_float(rd.x), __uint_as_float(rd.y), __uint_as_float(rd.z), __uint_as_float(rd.w));
        __builtin_amdgcn_wave_barrier();
        float4* dst = reinterpret_cast<float4*>(rec + (idx - lane));
#pragma unroll
        for (int j = 0; j < 4; j++) dst[j * 64 + lane] = sw[j * 64 + lane];
    } else if (radius > 0) {
        float3 col = col_in;
        if (!colors_precomp) col = gsr_sh_to_rgb(idx, D, M, p, cam, shs, clamped);
        GsrRec* r = rec + idx;
        r->a = make_float4(pix, piy, conx, cony);
        r->b = make_float4(conz, op_in, viewz, feat_in);
        r->c = make_float4(col.x, col.y, col.z, tau);  // .w = culling threshold (see above)
        r->d = make_uint4((rc.x >> 16) - (rc.x & 0xffff), (rc.x & 0xffff) | ((rc.y & 0xffff) << 16), (uint32_t)mask, (uint32_t)(mask >> 32));  // .x = rectangle width
    }
}

// DGR rasterizer_impl.cu:54-66 checkFrustum
__global__ void __launch_bounds__(256) gsr_mark_visible_kernel(int P, const float* __restrict__ means3D,
                                                               const float* __restrict__ vm,
                                                               uint8_t* __restrict__ present)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= P) return;
    const float z = vm[2] * means3D[3 * idx] + vm[6] * means3D[3 * idx + 1] + vm[10] * means3D[3 * idx + 2] + vm[14];
    present[idx] = z > 0.2f ? 1 : 0;
}

// `prefiltered` contract check (DGR auxiliary.h:154-162): the reference traps the kernel when a point the caller declared
// pre-filtered fails the near-plane test.  Counts such points; api.hip turns a non-zero count into an error (debug mode).
__global__ void __launch_bounds__(256) gsr_prefiltered_check_kernel(int P, const float* __restrict__ means3D,
                                                                    const float* __restrict__ vm, uint32_t* __restrict__ culled)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const bool bad = idx < P && !(vm[2] * means3D[3 * idx] + vm[6] * means3D[3 * idx + 1] + vm[10] * means3D[3 * idx + 2] + vm[14] > 0.2f);
    const unsigned long long m = __ballot(bad);
    if (m != 0ull && (threadIdx.x & 63) == 0) atomicAdd(culled, (uint32_t)__popcll(m));
}

hipError_t gsr_launch_prefiltered_check(int P, const float* means3D, const float* vm, uint32_t* culled, hipStream_t stream)
{
    if (P <= 0) return hipSuccess;
    hipLaunchKernelGGL(gsr_prefiltered_check_kernel, dim3((P + 255) / 256), dim3(256), 0, stream, P, means3D, vm, culled);
    return hipGetLastError();
}

hipError_t gsr_launch_preprocess(int mode, int P, int D, int M, const GsrCam& cam, const float* means3D,
                                 const float* scales, const float* rotations, const float* opacities,
                                 const float* features, const float* shs, const float* cov3D_precomp,
                                 const float* colors_precomp, const GsrGeom* g, int32_t* radii, float* px, float* py,
                                 int tile_cull, uint32_t* occ_mass, hipStream_t stream)
{
    if (P <= 0) return hipSuccess;
    const dim3 grid((P + 255) / 256), block(256);
    if (mode == 0)
        hipLaunchKernelGGL(gsr_preprocess_kernel<0>, grid, block, 0, stream, P, D, M, tile_cull, cam, means3D, scales, rotations,
                           opacities, features, shs, cov3D_precomp, colors_precomp, g->rec, g->rect, g->depthkey,
                           g->tiles, g->tmask, g->clamped, radii, nullptr, nullptr, tile_cull ? occ_mass : nullptr);
    else if (mode == 1)
        hipLaunchKernelGGL(gsr_preprocess_kernel<1>, grid, block, 0, stream, P, D, M, 0, cam, means3D, scales, rotations,
                           nullptr, nullptr, nullptr, cov3D_precomp, nullptr, nullptr, nullptr, nullptr, nullptr,
                           nullptr, nullptr, radii, nullptr, nullptr, nullptr);
    else
        hipLaunchKernelGGL(gsr_preprocess_kernel<2>, grid, block, 0, stream, P, D, M, 0, cam, means3D, scales, rotations,
                           nullptr, nullptr, nullptr, cov3D_precomp, nullptr, nullptr, nullptr, nullptr, nullptr,
                           nullptr, nullptr, radii, px, py, nullptr);
    return hipGetLastError();
}

hipError_t gsr_launch_mark_visible(int P, const float* means3D, const float* vm, uint8_t* present, hipStream_t stream)
{
    if (P <= 0) return hipSuccess;
    hipLaunchKernelGGL(gsr_mark_visible_kernel, dim3((P + 255) / 256), dim3(256), 0, stream, P, means3D, vm, present);
    return hipGetLastError();
}
