/*
 * gs_oracle.c -- CPU restatement of the GScream / Scaffold-GS differentiable
 * Gaussian rasterizer (reference: submodules/diff-gaussian-rasterization, "DGR").
 *
 * THIS FILE IS TEST INFRASTRUCTURE, NOT PRODUCT.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it.  The product path (gscream_amd/) never
 * links, imports or calls anything in oracle/.
 *
 * PARITY STATUS: *parity unpinned*.  The reference ships no tests, golden vectors or
 * fixtures for this path and its implementation is CUDA-only
 * (needs nvcc + the CUDA runtime headers + CUB, none of which exist in this image), so it
 * cannot be built or run here without writing stand-ins for those headers -- which the
 * task forbids.  The only reference-derived numbers available are the five filter-API
 * known answers recorded in SURVEY.md Appendix B-6; tests/test_oracle.py checks them.
 * Two pieces of the path ARE pinned by vectors the reference's own importable Python
 * computed in the build container (tests/golden/make_reference_vectors.py ->
 * tests/golden/ref_camera.npz, ref_sh.npz; checked by tests/test_reference_vectors.py):
 * the camera-matrix convention (utils/graphics_utils.py:38-76 as composed by
 * scene/cameras.py) and the SH colour expansion incl. the clamp flags
 * (utils/sh_utils.py:57-115, the Python twin of forward.cu:19-71).
 * Everything else is pinned by (a) closed-form known answers derived from the reference
 * source and (b) an independent float64 autograd restatement (oracle/naive_torch.py).
 *
 * The arithmetic follows the reference source line by line in evaluation order, in
 * IEEE fp32 with NO fused multiply-add (build with -ffp-contract=off), so that the
 * per-Gaussian stage can be compared bit-exactly with the HIP kernels (which are built
 * with -ffp-contract=off for that stage).  Every function cites the reference file:line
 * it restates.  GLM (third_party/glm, vendored by the reference) is column-major:
 * m[c][r]; its mat3*mat3 evaluates Result[c][r] = A[0][r]*B[c][0] + A[1][r]*B[c][1] +
 * A[2][r]*B[c][2] left to right (glm/detail/type_mat3x3.inl:486-520).  The expansions
 * below keep that order and drop only products with a literal 0 operand (exact).
 *
 * Scatter-sums of the backward blend (11 atomicAdd targets, backward.cu:554-601) have no
 * defined order in the reference; the oracle accumulates them in double and rounds once.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define TILE 16 /* config.h:16-17 BLOCK_X, BLOCK_Y */

/* CUDA float->int conversion saturates and maps NaN to 0; C leaves it undefined. */
static int f2i_sat(float v)
{
    if (!(v == v)) return 0;
    if (v >= 2147483648.0f) return 2147483647;
    if (v <= -2147483648.0f) return (-2147483647 - 1);
    return (int)v;
}
static int imin(int a, int b) { return a < b ? a : b; }
static int imax(int a, int b) { return a > b ? a : b; }

/* auxiliary.h:41-44 ndc2Pix: evaluated in double because of the 1.0 / 0.5 literals. */
static float ndc2pix(float v, int S)
{
    return (float)((((double)v + 1.0) * (double)S - 1.0) * 0.5);
}

/* auxiliary.h:46-56 getRect.  max_radius is an int parameter in the reference. */
static void get_rect(float px, float py, int max_radius, int gx, int gy, int *x0, int *y0, int *x1, int *y1)
{
    float r = (float)max_radius;
    *x0 = imin(gx, imax(0, f2i_sat((px - r) / (float)TILE)));
    *y0 = imin(gy, imax(0, f2i_sat((py - r) / (float)TILE)));
    *x1 = imin(gx, imax(0, f2i_sat((px + r + (float)(TILE - 1)) / (float)TILE)));
    *y1 = imin(gy, imax(0, f2i_sat((py + r + (float)(TILE - 1)) / (float)TILE)));
}

/* rasterizer_impl.cu:35-50 getHigherMsb (host) */
static uint32_t higher_msb(uint32_t n)
{
    uint32_t msb = sizeof(n) * 4, step = msb;
    while (step > 1) {
        step /= 2;
        if (n >> msb) msb += step; else msb -= step;
    }
    if (n >> msb) msb++;
    return msb;
}
int gso_sort_bits(int W, int H)
{
    int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE;
    return 32 + (int)higher_msb((uint32_t)(gx * gy));
}

/* forward.cu:120-154 computeCov3D.  Quaternion (r,x,y,z) used as given (no normalisation). */
static void cov3d_from_scale_rot(const float *scale, float mod, const float *rot, float *cov)
{
    float s[3] = { mod * scale[0], mod * scale[1], mod * scale[2] };
    float r = rot[0], x = rot[1], y = rot[2], z = rot[3];
    float R[3][3]; /* R[c][r], glm column-major as constructed at forward.cu:136-140 */
    R[0][0] = 1.f - 2.f * (y * y + z * z); R[0][1] = 2.f * (x * y - r * z); R[0][2] = 2.f * (x * z + r * y);
    R[1][0] = 2.f * (x * y + r * z); R[1][1] = 1.f - 2.f * (x * x + z * z); R[1][2] = 2.f * (y * z - r * x);
    R[2][0] = 2.f * (x * z - r * y); R[2][1] = 2.f * (y * z + r * x); R[2][2] = 1.f - 2.f * (x * x + y * y);
    float M[3][3]; /* M = S*R  => M[c][r] = s_r * R[c][r] */
    for (int c = 0; c < 3; c++) for (int q = 0; q < 3; q++) M[c][q] = s[q] * R[c][q];
    /* Sigma = transpose(M)*M => Sigma[c][r] = M[r][0]*M[c][0] + M[r][1]*M[c][1] + M[r][2]*M[c][2] */
#define SIG(c, r) (M[r][0] * M[c][0] + M[r][1] * M[c][1] + M[r][2] * M[c][2])
    cov[0] = SIG(0, 0); cov[1] = SIG(0, 1); cov[2] = SIG(0, 2);
    cov[3] = SIG(1, 1); cov[4] = SIG(1, 2); cov[5] = SIG(2, 2);
#undef SIG
}

/* Shared by forward.cu:76-115 computeCov2D and backward.cu:160-199: clamped view-space
 * mean t, the two non-zero rows of T (= A = J*W_rot, SURVEY A-13b) and the 2D covariance
 * (a,b,c) WITH the 0.3 low-pass. */
typedef struct { float tx, ty, tz, txtz, tytz, limx, limy; float A0[3], A1[3]; float a, b, c; } cov2d_t;

static void cov2d_eval(const float *mean, float fx, float fy, float tanfovx, float tanfovy,
                       const float *cov3D, const float *vm, cov2d_t *o)
{
    /* auxiliary.h:58-66 transformPoint4x3 */
    float tx = vm[0] * mean[0] + vm[4] * mean[1] + vm[8] * mean[2] + vm[12];
    float ty = vm[1] * mean[0] + vm[5] * mean[1] + vm[9] * mean[2] + vm[13];
    float tz = vm[2] * mean[0] + vm[6] * mean[1] + vm[10] * mean[2] + vm[14];
    o->limx = 1.3f * tanfovx; o->limy = 1.3f * tanfovy;
    o->txtz = tx / tz; o->tytz = ty / tz;
    tx = fminf(o->limx, fmaxf(-o->limx, o->txtz)) * tz;
    ty = fminf(o->limy, fmaxf(-o->limy, o->tytz)) * tz;
    o->tx = tx; o->ty = ty; o->tz = tz;
    float J00 = fx / tz, J02 = -(fx * tx) / (tz * tz);
    float J11 = fy / tz, J12 = -(fy * ty) / (tz * tz);
    /* T = W*J (glm) : T[0][r] = W[0][r]*J00 + W[2][r]*J02 ; T[1][r] = W[1][r]*J11 + W[2][r]*J12 */
    o->A0[0] = vm[0] * J00 + vm[2] * J02; o->A0[1] = vm[4] * J00 + vm[6] * J02; o->A0[2] = vm[8] * J00 + vm[10] * J02;
    o->A1[0] = vm[1] * J11 + vm[2] * J12; o->A1[1] = vm[5] * J11 + vm[6] * J12; o->A1[2] = vm[9] * J11 + vm[10] * J12;
    const float V[3][3] = { { cov3D[0], cov3D[1], cov3D[2] }, { cov3D[1], cov3D[3], cov3D[4] }, { cov3D[2], cov3D[4], cov3D[5] } };
    float X0[3], X1[3]; /* X = transpose(T)*transpose(Vrk): X[c][0], X[c][1] */
    for (int c = 0; c < 3; c++) {
        X0[c] = o->A0[0] * V[c][0] + o->A0[1] * V[c][1] + o->A0[2] * V[c][2];
        X1[c] = o->A1[0] * V[c][0] + o->A1[1] * V[c][1] + o->A1[2] * V[c][2];
    }
    float c00 = X0[0] * o->A0[0] + X0[1] * o->A0[1] + X0[2] * o->A0[2];
    float c01 = X1[0] * o->A0[0] + X1[1] * o->A0[1] + X1[2] * o->A0[2];
    float c11 = X1[0] * o->A1[0] + X1[1] * o->A1[1] + X1[2] * o->A1[2];
    o->a = c00 + 0.3f; o->b = c01; o->c = c11 + 0.3f; /* forward.cu:112-114 */
}

static const float SH_C0 = 0.28209479177387814f, SH_C1 = 0.4886025119029199f;
static const float SH_C2[5] = { 1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f, -1.0925484305920792f, 0.5462742152960396f };
static const float SH_C3[7] = { -0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f, 0.3731763325901154f,
                                -0.4570457994644658f, 1.445305721320277f, -0.5900435899266435f };

/* forward.cu:22-73 computeColorFromSH */
static void sh_to_rgb(int idx, int deg, int M, const float *means, const float *campos, const float *shs,
                      unsigned char *clamped, float *rgb)
{
    float dx = means[3 * idx] - campos[0], dy = means[3 * idx + 1] - campos[1], dz = means[3 * idx + 2] - campos[2];
    float len = sqrtf(dx * dx + dy * dy + dz * dz);
    float x = dx / len, y = dy / len, z = dz / len;
    const float *sh = shs + (size_t)idx * M * 3;
    float res[3];
    for (int k = 0; k < 3; k++) {
#define S(i) sh[(i) * 3 + k]
        float v = SH_C0 * S(0);
        if (deg > 0) {
            v = v - SH_C1 * y * S(1) + SH_C1 * z * S(2) - SH_C1 * x * S(3);
            if (deg > 1) {
                float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
                v = v + SH_C2[0] * xy * S(4) + SH_C2[1] * yz * S(5) + SH_C2[2] * (2.0f * zz - xx - yy) * S(6)
                      + SH_C2[3] * xz * S(7) + SH_C2[4] * (xx - yy) * S(8);
                if (deg > 2) {
                    v = v + SH_C3[0] * y * (3.0f * xx - yy) * S(9) + SH_C3[1] * xy * z * S(10)
                          + SH_C3[2] * y * (4.0f * zz - xx - yy) * S(11)
                          + SH_C3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy) * S(12)
                          + SH_C3[4] * x * (4.0f * zz - xx - yy) * S(13) + SH_C3[5] * z * (xx - yy) * S(14)
                          + SH_C3[6] * x * (xx - 3.0f * yy) * S(15);
                }
            }
        }
#undef S
        res[k] = v + 0.5f;
    }
    for (int k = 0; k < 3; k++) {
        clamped[3 * idx + k] = (res[k] < 0);
        rgb[3 * idx + k] = fmaxf(res[k], 0.0f);
    }
}

/* forward.cu:157-267 preprocessCUDA (fwd) + auxiliary.h:139-164 in_frustum.
 * mode 0 = full preprocess; 1 = filter_preprocessCUDA (forward.cu:271-346, radii only);
 * 2 = position2D_preprocessCUDA (forward.cu:352-433, radii + px/py written to means2D as
 *     two planar arrays: means2D[0..P) = x, means2D[P..2P) = y). */
void gso_preprocess(int mode, int P, int D, int M, const float *means3D, const float *scales, float scale_modifier,
                    const float *rotations, const float *opacities, const float *uncertainties, const float *shs,
                    const float *cov3D_precomp, const float *colors_precomp, const float *viewmatrix,
                    const float *projmatrix, const float *campos, int W, int H, float tanfovx, float tanfovy,
                    int *radii, float *means2D, float *depths, float *cov3Ds, float *conic_opacity, float *unc_out,
                    float *rgb, unsigned char *clamped, uint32_t *tiles_touched)
{
    const float fy = H / (2.0f * tanfovy), fx = W / (2.0f * tanfovx); /* rasterizer_impl.cu:226-227 */
    const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE;
    for (int idx = 0; idx < P; idx++) {
        radii[idx] = 0;
        if (mode == 0) tiles_touched[idx] = 0;
        if (mode == 2) { means2D[idx] = 0; means2D[P + idx] = 0; }
        const float *p = means3D + 3 * idx;
        const float *pm = projmatrix, *vm = viewmatrix;
        /* auxiliary.h:68-77 transformPoint4x4, :149-151 */
        float hx = pm[0] * p[0] + pm[4] * p[1] + pm[8] * p[2] + pm[12];
        float hy = pm[1] * p[0] + pm[5] * p[1] + pm[9] * p[2] + pm[13];
        float hw = pm[3] * p[0] + pm[7] * p[1] + pm[11] * p[2] + pm[15];
        float p_w = 1.0f / (hw + 0.0000001f);
        float projx = hx * p_w, projy = hy * p_w;
        float viewz = vm[2] * p[0] + vm[6] * p[1] + vm[10] * p[2] + vm[14];
        if (viewz <= 0.2f) continue; /* auxiliary.h:154 */

        float covbuf[6];
        const float *cov3D;
        if (cov3D_precomp) cov3D = cov3D_precomp + 6 * idx;
        else {
            cov3d_from_scale_rot(scales + 3 * idx, scale_modifier, rotations + 4 * idx, covbuf);
            if (cov3Ds) memcpy(cov3Ds + 6 * idx, covbuf, sizeof covbuf);
            cov3D = covbuf;
        }
        cov2d_t q;
        cov2d_eval(p, fx, fy, tanfovx, tanfovy, cov3D, vm, &q);
        float det = q.a * q.c - q.b * q.b;
        if (det == 0.0f) continue;
        float det_inv = 1.f / det;
        float conx = q.c * det_inv, cony = -q.b * det_inv, conz = q.a * det_inv;
        float mid = 0.5f * (q.a + q.c);
        float lambda1 = mid + sqrtf(fmaxf(0.1f, mid * mid - det));
        float lambda2 = mid - sqrtf(fmaxf(0.1f, mid * mid - det));
        float my_radius = ceilf(3.f * sqrtf(fmaxf(lambda1, lambda2)));
        float pix = ndc2pix(projx, W), piy = ndc2pix(projy, H);
        int x0, y0, x1, y1;
        get_rect(pix, piy, f2i_sat(my_radius), gx, gy, &x0, &y0, &x1, &y1);
        if ((x1 - x0) * (y1 - y0) == 0) continue;
        /* NaN covariance -> radius 0: the reference leaves radii = 0 with tiles_touched > 0 and then sorts
         * uninitialised keys (undefined behaviour); oracle and HIP path both cull such a Gaussian. */
        if (f2i_sat(my_radius) <= 0) continue;
        radii[idx] = f2i_sat(my_radius);
        if (mode == 1) continue;
        if (mode == 2) { means2D[idx] = pix; means2D[P + idx] = piy; continue; }
        if (!colors_precomp) sh_to_rgb(idx, D, M, means3D, campos, shs, clamped, rgb);
        depths[idx] = viewz;
        means2D[2 * idx] = pix; means2D[2 * idx + 1] = piy;
        conic_opacity[4 * idx] = conx; conic_opacity[4 * idx + 1] = cony; conic_opacity[4 * idx + 2] = conz;
        conic_opacity[4 * idx + 3] = opacities[idx];
        tiles_touched[idx] = (uint32_t)((y1 - y0) * (x1 - x0));
        unc_out[idx] = uncertainties[idx];
    }
}

/* rasterizer_impl.cu:54-66 checkFrustum / markVisible */
void gso_mark_visible(int P, const float *means3D, const float *viewmatrix, unsigned char *present)
{
    for (int i = 0; i < P; i++) {
        const float *p = means3D + 3 * i;
        float z = viewmatrix[2] * p[0] + viewmatrix[6] * p[1] + viewmatrix[10] * p[2] + viewmatrix[14];
        present[i] = !(z <= 0.2f);
    }
}

typedef struct { uint64_t key; uint32_t val; uint32_t seq; } kv_t;
static int kv_cmp(const void *a, const void *b)
{
    const kv_t *x = (const kv_t *)a, *y = (const kv_t *)b;
    if (x->key != y->key) return x->key < y->key ? -1 : 1;
    return x->seq < y->seq ? -1 : (x->seq > y->seq);
}

/* rasterizer_impl.cu:283-320: InclusiveSum, duplicateWithKeys (:70-111), stable SortPairs on
 * bits [0, 32+bit) (:306-314), identifyTileRanges (:116-138).  point_list/keys must hold
 * sum(tiles_touched) entries; ranges holds 2*tiles uint32 (zeroed here like :316). Returns R. */
int gso_bin(int P, const int *radii, const float *means2D, const float *depths, const uint32_t *tiles_touched,
            int W, int H, uint32_t *point_list, uint64_t *point_keys, uint32_t *ranges)
{
    const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE;
    size_t R = 0;
    for (int i = 0; i < P; i++) R += tiles_touched[i];
    memset(ranges, 0, sizeof(uint32_t) * 2 * (size_t)gx * gy);
    if (R == 0) return 0;
    kv_t *kv = (kv_t *)malloc(sizeof(kv_t) * R);
    size_t off = 0;
    for (int idx = 0; idx < P; idx++) {
        if (radii[idx] <= 0) continue;
        int x0, y0, x1, y1;
        get_rect(means2D[2 * idx], means2D[2 * idx + 1], radii[idx], gx, gy, &x0, &y0, &x1, &y1);
        uint32_t dbits; memcpy(&dbits, &depths[idx], 4);
        for (int y = y0; y < y1; y++)
            for (int x = x0; x < x1; x++) {
                kv[off].key = ((uint64_t)(uint32_t)(y * gx + x) << 32) | dbits;
                kv[off].val = (uint32_t)idx; kv[off].seq = (uint32_t)off; off++;
            }
    }
    /* All key bits above 32+bit are zero by construction, so masking is the identity. */
    qsort(kv, R, sizeof(kv_t), kv_cmp);
    for (size_t i = 0; i < R; i++) {
        point_list[i] = kv[i].val;
        if (point_keys) point_keys[i] = kv[i].key;
        uint32_t cur = (uint32_t)(kv[i].key >> 32);
        if (i == 0) ranges[2 * cur] = 0;
        else {
            uint32_t prev = (uint32_t)(kv[i - 1].key >> 32);
            if (cur != prev) { ranges[2 * prev + 1] = (uint32_t)i; ranges[2 * cur] = (uint32_t)i; }
        }
        if (i == R - 1) ranges[2 * cur + 1] = (uint32_t)R;
    }
    free(kv);
    return (int)R;
}

/* forward.cu:441-568 renderCUDA (fwd), restated per pixel (the block-wide early exit at
 * :496-498 does not change any pixel's result). */
void gso_render_forward(int W, int H, const uint32_t *ranges, const uint32_t *point_list, const float *means2D,
                        const float *colors, const float *depths, const float *unc, const float *conic_opacity,
                        const float *bg, float *out_color, float *out_depth, float *out_unc, float *final_T,
                        uint32_t *n_contrib, int nthreads)
{
    const int gx = (W + TILE - 1) / TILE;
    (void)nthreads;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 4) num_threads(nthreads > 0 ? nthreads : 1)
#endif
    for (int py = 0; py < H; py++)
        for (int px = 0; px < W; px++) {
            const int tile = (py / TILE) * gx + (px / TILE);
            const uint32_t r0 = ranges[2 * tile], r1 = ranges[2 * tile + 1];
            const float pixfx = (float)px, pixfy = (float)py;
            float T = 1.0f, C[3] = { 0, 0, 0 }, Dp = 0, U = 0;
            uint32_t contributor = 0, last = 0;
            for (uint32_t k = r0; k < r1; k++) {
                contributor++;
                const uint32_t g = point_list[k];
                float dx = means2D[2 * g] - pixfx, dy = means2D[2 * g + 1] - pixfy;
                const float *co = conic_opacity + 4 * g;
                float power = -0.5f * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
                if (power > 0.0f) continue;
                float alpha = fminf(0.99f, co[3] * expf(power));
                if (alpha < 1.0f / 255.0f) continue;
                float test_T = T * (1 - alpha);
                if (test_T < 0.0001f) break;
                for (int ch = 0; ch < 3; ch++) C[ch] += colors[3 * g + ch] * alpha * T;
                Dp += depths[g] * alpha * T;
                U += unc[g] * alpha * T;
                T = test_T;
                last = contributor;
            }
            const size_t pid = (size_t)py * W + px;
            final_T[pid] = T; n_contrib[pid] = last;
            for (int ch = 0; ch < 3; ch++) out_color[(size_t)ch * H * W + pid] = C[ch] + T * bg[ch];
            out_depth[pid] = Dp; out_unc[pid] = U;
        }
}

/* Diagnostic twin of gso_render_forward (same walk, forward.cu:493-554): how close did each pixel come to flipping one
 * of the walk's three discontinuous decisions?  Used by tools/flip_report.py / full_size_oracle_check.py and bench.py's
 * parity_check to CLASSIFY a build's outlier pixels (alpha = 1/255 flips vs T = 1e-4 flips).  Per pixel:
 *   m_alpha = min over evaluated instances of |alpha * 255 - 1| (alpha unclamped: opacity * exp(power)), g_alpha = that Gaussian
 *   m_T     = min over instances that reach the transmittance test of |test_T / 1e-4 - 1|,               g_T     = that Gaussian
 *   m_pow   = min |power| over the instances walked (the `power > 0` skip, forward.cu:529)
 * (1e30 / 0xffffffff where the pixel never reaches the test). */
void gso_render_margins(int W, int H, const uint32_t *ranges, const uint32_t *point_list, const float *means2D,
                        const float *conic_opacity, float *m_alpha, uint32_t *g_alpha, float *m_T, uint32_t *g_T,
                        float *m_pow, int nthreads)
{
    const int gx = (W + TILE - 1) / TILE;
    (void)nthreads;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 4) num_threads(nthreads > 0 ? nthreads : 1)
#endif
    for (int py = 0; py < H; py++)
        for (int px = 0; px < W; px++) {
            const int tile = (py / TILE) * gx + (px / TILE);
            const uint32_t r0 = ranges[2 * tile], r1 = ranges[2 * tile + 1];
            const float pixfx = (float)px, pixfy = (float)py;
            float T = 1.0f, ma = 1e30f, mt = 1e30f, mp = 1e30f;
            uint32_t ga = 0xffffffffu, gt = 0xffffffffu;
            for (uint32_t k = r0; k < r1; k++) {
                const uint32_t g = point_list[k];
                float dx = means2D[2 * g] - pixfx, dy = means2D[2 * g + 1] - pixfy;
                const float *co = conic_opacity + 4 * g;
                float power = -0.5f * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
                if (fabsf(power) < mp) mp = fabsf(power);
                if (power > 0.0f) continue;
                const float raw = co[3] * expf(power);
                const float da = fabsf(raw * 255.0f - 1.0f);
                if (da < ma) { ma = da; ga = g; }
                float alpha = fminf(0.99f, raw);
                if (alpha < 1.0f / 255.0f) continue;
                float test_T = T * (1 - alpha);
                const float dt = fabsf(test_T * 10000.0f - 1.0f);
                if (dt < mt) { mt = dt; gt = g; }
                if (test_T < 0.0001f) break;
                T = test_T;
            }
            const size_t pid = (size_t)py * W + px;
            m_alpha[pid] = ma; g_alpha[pid] = ga; m_T[pid] = mt; g_T[pid] = gt; m_pow[pid] = mp;
        }
}

/* backward.cu:409-604 renderCUDA (bwd), restated per pixel.  Outputs are double accumulators
 * (P-sized, caller zero-fills): mean2D[2P] (x,y), conic[3P] (the .x,.y,.w slots of the
 * reference float4), opacity[P], colors[3P], depth[P], unc[P]. */
void gso_render_backward(int W, int H, const uint32_t *ranges, const uint32_t *point_list, const float *bg,
                         const float *means2D, const float *conic_opacity, const float *colors, const float *depths,
                         const float *unc, const float *final_T, const uint32_t *n_contrib, const float *dL_dpix,
                         const float *dL_ddepthpix, const float *dL_duncpix, double *g_mean2D, double *g_conic,
                         double *g_opacity, double *g_colors, double *g_depth, double *g_unc, int nthreads)
{
    const int gx = (W + TILE - 1) / TILE;
    const float ddelx_dx = (float)(0.5 * W), ddely_dy = (float)(0.5 * H); /* :488-489 */
    (void)nthreads;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 4) num_threads(nthreads > 0 ? nthreads : 1)
#endif
    for (int py = 0; py < H; py++)
        for (int px = 0; px < W; px++) {
            const int tile = (py / TILE) * gx + (px / TILE);
            const uint32_t r0 = ranges[2 * tile], r1 = ranges[2 * tile + 1];
            const size_t pid = (size_t)py * W + px;
            const float pixfx = (float)px, pixfy = (float)py;
            const float T_final = final_T[pid];
            float T = T_final;
            const uint32_t last_contributor = n_contrib[pid];
            float accum_rec[3] = { 0, 0, 0 }, accum_depth_rec = 0, accum_unc_rec = 0;
            float dL_dpixel[3] = { dL_dpix[pid], dL_dpix[(size_t)H * W + pid], dL_dpix[(size_t)2 * H * W + pid] };
            const float dL_dpd = dL_ddepthpix[pid], dL_dunc = dL_duncpix[pid];
            float last_alpha = 0, last_color[3] = { 0, 0, 0 }, last_depth = 0, last_unc = 0;
            uint32_t contributor = r1 - r0;
            for (uint32_t kk = 0; kk < r1 - r0; kk++) {
                contributor--;
                if (contributor >= last_contributor) continue;
                const uint32_t g = point_list[r1 - 1 - kk];
                const float dx = means2D[2 * g] - pixfx, dy = means2D[2 * g + 1] - pixfy;
                const float *co = conic_opacity + 4 * g;
                const float power = -0.5f * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
                if (power > 0.0f) continue;
                const float G = expf(power);
                const float alpha = fminf(0.99f, co[3] * G);
                if (alpha < 1.0f / 255.0f) continue;
                T = T / (1.f - alpha);
                const float dchannel_dcolor = alpha * T;
                float dL_dalpha = 0.0f;
                for (int ch = 0; ch < 3; ch++) {
                    const float c = colors[3 * g + ch];
                    accum_rec[ch] = last_alpha * last_color[ch] + (1.f - last_alpha) * accum_rec[ch];
                    last_color[ch] = c;
                    dL_dalpha += (c - accum_rec[ch]) * dL_dpixel[ch];
                    const float v = dchannel_dcolor * dL_dpixel[ch];
#ifdef _OPENMP
#pragma omp atomic
#endif
                    g_colors[3 * g + ch] += v;
                }
                const float c_d = depths[g];
                accum_depth_rec = last_alpha * last_depth + (1.f - last_alpha) * accum_depth_rec;
                last_depth = c_d;
                dL_dalpha += (c_d - accum_depth_rec) * dL_dpd;
                const float c_unc = unc[g];
                accum_unc_rec = last_alpha * last_unc + (1.f - last_alpha) * accum_unc_rec;
                last_unc = c_unc;
                dL_dalpha += (c_unc - accum_unc_rec) * dL_dunc;
                dL_dalpha *= T;
                last_alpha = alpha;
                float bg_dot = 0;
                for (int i = 0; i < 3; i++) bg_dot += bg[i] * dL_dpixel[i];
                dL_dalpha += (-T_final / (1.f - alpha)) * bg_dot;
                const float dL_dG = co[3] * dL_dalpha;
                const float gdx = G * dx, gdy = G * dy;
                const float dG_ddelx = -gdx * co[0] - gdy * co[1];
                const float dG_ddely = -gdy * co[2] - gdx * co[1];
                const float v_depth = dchannel_dcolor * dL_dpd, v_unc = dchannel_dcolor * dL_dunc;
                const float v_mx = dL_dG * dG_ddelx * ddelx_dx, v_my = dL_dG * dG_ddely * ddely_dy;
                const float v_cx = -0.5f * gdx * dx * dL_dG, v_cy = -0.5f * gdx * dy * dL_dG, v_cw = -0.5f * gdy * dy * dL_dG;
                const float v_op = G * dL_dalpha;
#ifdef _OPENMP
#pragma omp atomic
#endif
                g_depth[g] += v_depth;
#ifdef _OPENMP
#pragma omp atomic
#endif
                g_unc[g] += v_unc;
#ifdef _OPENMP
#pragma omp atomic
#endif
                g_mean2D[2 * g] += v_mx;
#ifdef _OPENMP
#pragma omp atomic
#endif
                g_mean2D[2 * g + 1] += v_my;
#ifdef _OPENMP
#pragma omp atomic
#endif
                g_conic[3 * g] += v_cx;
#ifdef _OPENMP
#pragma omp atomic
#endif
                g_conic[3 * g + 1] += v_cy;
#ifdef _OPENMP
#pragma omp atomic
#endif
                g_conic[3 * g + 2] += v_cw;
#ifdef _OPENMP
#pragma omp atomic
#endif
                g_opacity[g] += v_op;
            }
        }
}

/* ORDER-NOISE ENVELOPE of the reference's own backward blend (round 5).  backward.cu:554-601 scatters the 11 per-contribution
 * terms with unordered fp32 atomicAdd: the reference's result for a Gaussian is whatever order the hardware served its pixels
 * in, so two runs of the REFERENCE differ by the rounding of a re-ordered fp32 sum.  gso_render_backward accumulates in double
 * (order-free); this function measures, for `ng` chosen Gaussians, how far fp32 accumulation in K random contribution orders
 * spreads: it collects every (pixel -> Gaussian) contribution exactly as gso_render_backward forms it (same walk, same fp32
 * expressions), then adds them up K times in fp32, each time in another random permutation.
 *   sums[ng][K][11]: mean2D.x, mean2D.y, conic.x, conic.y, conic.w, opacity, colour r, g, b, depth, feature  (slot order of g_*).
 * A Gaussian's tiles are the tiles of its rectangle (getRect, the oracle's un-culled lists).  Returns the largest number of
 * contributions one Gaussian had.  Test infrastructure like the rest of the file: parity_check uses it to say whether a gradient
 * element that differs from the double-accumulated value by more than 1e-3 lies inside the reference's own order noise. */
static uint64_t env_rng(uint64_t *s) { *s ^= *s << 13; *s ^= *s >> 7; *s ^= *s << 17; return *s; }
int gso_backward_envelope(int W, int H, const uint32_t *ranges, const uint32_t *point_list, const float *bg,
                          const float *means2D, const int *radii, const float *conic_opacity, const float *colors, const float *depths,
                          const float *unc, const float *final_T, const uint32_t *n_contrib, const float *dL_dpix,
                          const float *dL_ddepthpix, const float *dL_duncpix, int ng, const int *gids, int K, uint64_t seed,
                          float *sums)
{
    const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE;
    const float ddelx_dx = (float)(0.5 * W), ddely_dy = (float)(0.5 * H);
    int most = 0;
    for (int q = 0; q < ng; q++) {
        const int target = gids[q];
        int x0, y0, x1, y1;
        get_rect(means2D[2 * target], means2D[2 * target + 1], radii[target], gx, gy, &x0, &y0, &x1, &y1);
        const int cap = imax(1, (x1 - x0) * (y1 - y0) * TILE * TILE);
        float *con = (float *)malloc((size_t)cap * 11 * sizeof(float));
        int nc = 0;
        for (int ty = y0; ty < y1; ty++)
            for (int tx = x0; tx < x1; tx++)
                for (int py = ty * TILE; py < imin(H, (ty + 1) * TILE); py++)
                    for (int px = tx * TILE; px < imin(W, (tx + 1) * TILE); px++) {
                        /* the walk of gso_render_backward for this pixel, up to the target (backward.cu:492-603) */
                        const int tile = ty * gx + tx;
                        const uint32_t r0 = ranges[2 * tile], r1 = ranges[2 * tile + 1];
                        const size_t pid = (size_t)py * W + px;
                        const float pixfx = (float)px, pixfy = (float)py;
                        const float T_final = final_T[pid];
                        float T = T_final;
                        const uint32_t last_contributor = n_contrib[pid];
                        float accum_rec[3] = { 0, 0, 0 }, accum_depth_rec = 0, accum_unc_rec = 0;
                        const float dL_dpixel[3] = { dL_dpix[pid], dL_dpix[(size_t)H * W + pid], dL_dpix[(size_t)2 * H * W + pid] };
                        const float dL_dpd = dL_ddepthpix[pid], dL_dunc = dL_duncpix[pid];
                        float last_alpha = 0, last_color[3] = { 0, 0, 0 }, last_depth = 0, last_unc = 0;
                        uint32_t contributor = r1 - r0;
                        for (uint32_t kk = 0; kk < r1 - r0; kk++) {
                            contributor--;
                            if (contributor >= last_contributor) continue;
                            const uint32_t g = point_list[r1 - 1 - kk];
                            const float dx = means2D[2 * g] - pixfx, dy = means2D[2 * g + 1] - pixfy;
                            const float *co = conic_opacity + 4 * g;
                            const float power = -0.5f * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
                            if (power > 0.0f) { if ((int)g == target) break; continue; }
                            const float G = expf(power);
                            const float alpha = fminf(0.99f, co[3] * G);
                            if (alpha < 1.0f / 255.0f) { if ((int)g == target) break; continue; }
                            T = T / (1.f - alpha);
                            const float dchannel_dcolor = alpha * T;
                            float dL_dalpha = 0.0f;
                            float vcol[3];
                            for (int ch = 0; ch < 3; ch++) {
                                const float c = colors[3 * g + ch];
                                accum_rec[ch] = last_alpha * last_color[ch] + (1.f - last_alpha) * accum_rec[ch];
                                last_color[ch] = c;
                                dL_dalpha += (c - accum_rec[ch]) * dL_dpixel[ch];
                                vcol[ch] = dchannel_dcolor * dL_dpixel[ch];
                            }
                            const float c_d = depths[g];
                            accum_depth_rec = last_alpha * last_depth + (1.f - last_alpha) * accum_depth_rec;
                            last_depth = c_d;
                            dL_dalpha += (c_d - accum_depth_rec) * dL_dpd;
                            const float c_unc = unc[g];
                            accum_unc_rec = last_alpha * last_unc + (1.f - last_alpha) * accum_unc_rec;
                            last_unc = c_unc;
                            dL_dalpha += (c_unc - accum_unc_rec) * dL_dunc;
                            dL_dalpha *= T;
                            last_alpha = alpha;
                            float bg_dot = 0;
                            for (int i = 0; i < 3; i++) bg_dot += bg[i] * dL_dpixel[i];
                            dL_dalpha += (-T_final / (1.f - alpha)) * bg_dot;
                            if ((int)g != target) continue;
                            const float dL_dG = co[3] * dL_dalpha;
                            const float gdx = G * dx, gdy = G * dy;
                            const float dG_ddelx = -gdx * co[0] - gdy * co[1];
                            const float dG_ddely = -gdy * co[2] - gdx * co[1];
                            float *v = con + (size_t)nc * 11;
                            v[0] = dL_dG * dG_ddelx * ddelx_dx; v[1] = dL_dG * dG_ddely * ddely_dy;
                            v[2] = -0.5f * gdx * dx * dL_dG; v[3] = -0.5f * gdx * dy * dL_dG; v[4] = -0.5f * gdy * dy * dL_dG;
                            v[5] = G * dL_dalpha;
                            v[6] = vcol[0]; v[7] = vcol[1]; v[8] = vcol[2];
                            v[9] = dchannel_dcolor * dL_dpd; v[10] = dchannel_dcolor * dL_dunc;
                            nc++;
                            break;
                        }
                    }
        if (nc > most) most = nc;
        int *perm = (int *)malloc((size_t)imax(nc, 1) * sizeof(int));
        uint64_t st = seed * 0x9E3779B97F4A7C15ull + (uint64_t)target * 0xD1B54A32D192ED03ull + 1ull;
        for (int k = 0; k < K; k++) {
            for (int i = 0; i < nc; i++) perm[i] = i;
            for (int i = nc - 1; i > 0; i--) { const int j = (int)(env_rng(&st) % (uint64_t)(i + 1)); const int t = perm[i]; perm[i] = perm[j]; perm[j] = t; }
            float acc[11] = { 0 };
            for (int i = 0; i < nc; i++)
                for (int c = 0; c < 11; c++) acc[c] += con[(size_t)perm[i] * 11 + c];  /* fp32, this order: what one atomicAdd sequence gives */
            memcpy(sums + ((size_t)q * K + k) * 11, acc, sizeof(acc));
        }
        free(perm);
        free(con);
    }
    return most;
}

/* backward.cu:20-139 computeColorFromSH (bwd) */
static void sh_backward(int idx, int deg, int M, const float *means, const float *campos, const float *shs,
                        const unsigned char *clamped, const float *dL_dcolor, float *dL_dmeans, float *dL_dshs)
{
    float dox = means[3 * idx] - campos[0], doy = means[3 * idx + 1] - campos[1], doz = means[3 * idx + 2] - campos[2];
    float len = sqrtf(dox * dox + doy * doy + doz * doz);
    float x = dox / len, y = doy / len, z = doz / len;
    const float *sh = shs + (size_t)idx * M * 3;
    float *dsh = dL_dshs + (size_t)idx * M * 3;
    float dRGB[3];
    for (int k = 0; k < 3; k++) dRGB[k] = dL_dcolor[3 * idx + k] * (clamped[3 * idx + k] ? 0.f : 1.f);
    float dRGBdx[3] = { 0, 0, 0 }, dRGBdy[3] = { 0, 0, 0 }, dRGBdz[3] = { 0, 0, 0 };
#define SH(i, k) sh[(i) * 3 + (k)]
#define DSH(i, k) dsh[(i) * 3 + (k)]
    for (int k = 0; k < 3; k++) DSH(0, k) = SH_C0 * dRGB[k];
    if (deg > 0) {
        float dRGBdsh1 = -SH_C1 * y, dRGBdsh2 = SH_C1 * z, dRGBdsh3 = -SH_C1 * x;
        for (int k = 0; k < 3; k++) {
            DSH(1, k) = dRGBdsh1 * dRGB[k]; DSH(2, k) = dRGBdsh2 * dRGB[k]; DSH(3, k) = dRGBdsh3 * dRGB[k];
            dRGBdx[k] = -SH_C1 * SH(3, k); dRGBdy[k] = -SH_C1 * SH(1, k); dRGBdz[k] = SH_C1 * SH(2, k);
        }
        if (deg > 1) {
            float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
            float d4 = SH_C2[0] * xy, d5 = SH_C2[1] * yz, d6 = SH_C2[2] * (2.f * zz - xx - yy), d7 = SH_C2[3] * xz, d8 = SH_C2[4] * (xx - yy);
            for (int k = 0; k < 3; k++) {
                DSH(4, k) = d4 * dRGB[k]; DSH(5, k) = d5 * dRGB[k]; DSH(6, k) = d6 * dRGB[k]; DSH(7, k) = d7 * dRGB[k]; DSH(8, k) = d8 * dRGB[k];
                dRGBdx[k] += SH_C2[0] * y * SH(4, k) + SH_C2[2] * 2.f * -x * SH(6, k) + SH_C2[3] * z * SH(7, k) + SH_C2[4] * 2.f * x * SH(8, k);
                dRGBdy[k] += SH_C2[0] * x * SH(4, k) + SH_C2[1] * z * SH(5, k) + SH_C2[2] * 2.f * -y * SH(6, k) + SH_C2[4] * 2.f * -y * SH(8, k);
                dRGBdz[k] += SH_C2[1] * y * SH(5, k) + SH_C2[2] * 2.f * 2.f * z * SH(6, k) + SH_C2[3] * x * SH(7, k);
            }
            if (deg > 2) {
                float d9 = SH_C3[0] * y * (3.f * xx - yy), d10 = SH_C3[1] * xy * z, d11 = SH_C3[2] * y * (4.f * zz - xx - yy);
                float d12 = SH_C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy), d13 = SH_C3[4] * x * (4.f * zz - xx - yy);
                float d14 = SH_C3[5] * z * (xx - yy), d15 = SH_C3[6] * x * (xx - 3.f * yy);
                for (int k = 0; k < 3; k++) {
                    DSH(9, k) = d9 * dRGB[k]; DSH(10, k) = d10 * dRGB[k]; DSH(11, k) = d11 * dRGB[k]; DSH(12, k) = d12 * dRGB[k];
                    DSH(13, k) = d13 * dRGB[k]; DSH(14, k) = d14 * dRGB[k]; DSH(15, k) = d15 * dRGB[k];
                    dRGBdx[k] += (SH_C3[0] * SH(9, k) * 3.f * 2.f * xy + SH_C3[1] * SH(10, k) * yz + SH_C3[2] * SH(11, k) * -2.f * xy
                                  + SH_C3[3] * SH(12, k) * -3.f * 2.f * xz + SH_C3[4] * SH(13, k) * (-3.f * xx + 4.f * zz - yy)
                                  + SH_C3[5] * SH(14, k) * 2.f * xz + SH_C3[6] * SH(15, k) * 3.f * (xx - yy));
                    dRGBdy[k] += (SH_C3[0] * SH(9, k) * 3.f * (xx - yy) + SH_C3[1] * SH(10, k) * xz + SH_C3[2] * SH(11, k) * (-3.f * yy + 4.f * zz - xx)
                                  + SH_C3[3] * SH(12, k) * -3.f * 2.f * yz + SH_C3[4] * SH(13, k) * -2.f * xy
                                  + SH_C3[5] * SH(14, k) * -2.f * yz + SH_C3[6] * SH(15, k) * -3.f * 2.f * xy);
                    dRGBdz[k] += (SH_C3[1] * SH(10, k) * xy + SH_C3[2] * SH(11, k) * 4.f * 2.f * yz + SH_C3[3] * SH(12, k) * 3.f * (2.f * zz - xx - yy)
                                  + SH_C3[4] * SH(13, k) * 4.f * 2.f * xz + SH_C3[5] * SH(14, k) * (xx - yy));
                }
            }
        }
    }
#undef SH
#undef DSH
    /* backward.cu:131-138: dL_ddir then through the normalisation (auxiliary.h:107-118 dnormvdv) */
    float ddx = 0, ddy = 0, ddz = 0;
    ddx = dRGBdx[0] * dRGB[0] + dRGBdx[1] * dRGB[1] + dRGBdx[2] * dRGB[2];
    ddy = dRGBdy[0] * dRGB[0] + dRGBdy[1] * dRGB[1] + dRGBdy[2] * dRGB[2];
    ddz = dRGBdz[0] * dRGB[0] + dRGBdz[1] * dRGB[1] + dRGBdz[2] * dRGB[2];
    float sum2 = dox * dox + doy * doy + doz * doz;
    float invsum32 = 1.0f / sqrtf(sum2 * sum2 * sum2);
    float mx = ((+sum2 - dox * dox) * ddx - doy * dox * ddy - doz * dox * ddz) * invsum32;
    float my = (-dox * doy * ddx + (sum2 - doy * doy) * ddy - doz * doy * ddz) * invsum32;
    float mz = (-dox * doz * ddx - doy * doz * ddy + (sum2 - doz * doz) * ddz) * invsum32;
    dL_dmeans[3 * idx] += mx; dL_dmeans[3 * idx + 1] += my; dL_dmeans[3 * idx + 2] += mz;
}

/* backward.cu:144-274 computeCov2DCUDA, then :346-406 preprocessCUDA (bwd) with
 * :278-341 computeCov3D (bwd).  dL_dconic is [P,3] (x,y,w slots); dL_dmean2D is [P,2].
 * All outputs are P-sized float arrays, zero-filled by the caller (rows with radii<=0 stay 0,
 * :156,:369). */
void gso_preprocess_backward(int P, int D, int M, const float *means3D, const int *radii, const float *shs,
                             const unsigned char *clamped, const float *scales, const float *rotations,
                             float scale_modifier, const float *cov3Ds, const float *viewmatrix, const float *projmatrix,
                             int W, int H, float tanfovx, float tanfovy, const float *campos, const float *dL_dmean2D,
                             const float *dL_dconic, const float *dL_dcolor, const float *dL_ddepth, float *dL_dmeans,
                             float *dL_dcov, float *dL_dsh, float *dL_dscale, float *dL_drot)
{
    const float h_y = H / (2.0f * tanfovy), h_x = W / (2.0f * tanfovx);
    const float *vm = viewmatrix, *proj = projmatrix;
    for (int idx = 0; idx < P; idx++) {
        if (!(radii[idx] > 0)) continue;
        const float *cov3D = cov3Ds + 6 * idx;
        const float *m = means3D + 3 * idx;
        cov2d_t q;
        cov2d_eval(m, h_x, h_y, tanfovx, tanfovy, cov3D, vm, &q);
        const float dcx = dL_dconic[3 * idx], dcy = dL_dconic[3 * idx + 1], dcz = dL_dconic[3 * idx + 2];
        const float x_grad_mul = (q.txtz < -q.limx || q.txtz > q.limx) ? 0.f : 1.f;
        const float y_grad_mul = (q.tytz < -q.limy || q.tytz > q.limy) ? 0.f : 1.f;
        const float a = q.a, b = q.b, c = q.c;
        const float *T0 = q.A0, *T1 = q.A1;
        const float V[3][3] = { { cov3D[0], cov3D[1], cov3D[2] }, { cov3D[1], cov3D[3], cov3D[4] }, { cov3D[2], cov3D[4], cov3D[5] } };
        float denom = a * c - b * b;
        float dL_da = 0, dL_db = 0, dL_dc = 0;
        float denom2inv = 1.0f / ((denom * denom) + 0.0000001f);
        float *dcov = dL_dcov + 6 * idx;
        if (denom2inv != 0) {
            dL_da = denom2inv * (-c * c * dcx + 2 * b * c * dcy + (denom - a * c) * dcz);
            dL_dc = denom2inv * (-a * a * dcz + 2 * a * b * dcy + (denom - a * c) * dcx);
            dL_db = denom2inv * 2 * (b * c * dcx - (denom + 2 * b * b) * dcy + a * b * dcz);
            dcov[0] = (T0[0] * T0[0] * dL_da + T0[0] * T1[0] * dL_db + T1[0] * T1[0] * dL_dc);
            dcov[3] = (T0[1] * T0[1] * dL_da + T0[1] * T1[1] * dL_db + T1[1] * T1[1] * dL_dc);
            dcov[5] = (T0[2] * T0[2] * dL_da + T0[2] * T1[2] * dL_db + T1[2] * T1[2] * dL_dc);
            dcov[1] = 2 * T0[0] * T0[1] * dL_da + (T0[0] * T1[1] + T0[1] * T1[0]) * dL_db + 2 * T1[0] * T1[1] * dL_dc;
            dcov[2] = 2 * T0[0] * T0[2] * dL_da + (T0[0] * T1[2] + T0[2] * T1[0]) * dL_db + 2 * T1[0] * T1[2] * dL_dc;
            dcov[4] = 2 * T0[2] * T0[1] * dL_da + (T0[1] * T1[2] + T0[2] * T1[1]) * dL_db + 2 * T1[1] * T1[2] * dL_dc;
        } else {
            for (int i = 0; i < 6; i++) dcov[i] = 0;
        }
        float dL_dT00 = 2 * (T0[0] * V[0][0] + T0[1] * V[0][1] + T0[2] * V[0][2]) * dL_da + (T1[0] * V[0][0] + T1[1] * V[0][1] + T1[2] * V[0][2]) * dL_db;
        float dL_dT01 = 2 * (T0[0] * V[1][0] + T0[1] * V[1][1] + T0[2] * V[1][2]) * dL_da + (T1[0] * V[1][0] + T1[1] * V[1][1] + T1[2] * V[1][2]) * dL_db;
        float dL_dT02 = 2 * (T0[0] * V[2][0] + T0[1] * V[2][1] + T0[2] * V[2][2]) * dL_da + (T1[0] * V[2][0] + T1[1] * V[2][1] + T1[2] * V[2][2]) * dL_db;
        float dL_dT10 = 2 * (T1[0] * V[0][0] + T1[1] * V[0][1] + T1[2] * V[0][2]) * dL_dc + (T0[0] * V[0][0] + T0[1] * V[0][1] + T0[2] * V[0][2]) * dL_db;
        float dL_dT11 = 2 * (T1[0] * V[1][0] + T1[1] * V[1][1] + T1[2] * V[1][2]) * dL_dc + (T0[0] * V[1][0] + T0[1] * V[1][1] + T0[2] * V[1][2]) * dL_db;
        float dL_dT12 = 2 * (T1[0] * V[2][0] + T1[1] * V[2][1] + T1[2] * V[2][2]) * dL_dc + (T0[0] * V[2][0] + T0[1] * V[2][1] + T0[2] * V[2][2]) * dL_db;
        /* W[c][r] at backward.cu:183-186: W[0]=(vm0,vm4,vm8) W[1]=(vm1,vm5,vm9) W[2]=(vm2,vm6,vm10) */
        float dL_dJ00 = vm[0] * dL_dT00 + vm[4] * dL_dT01 + vm[8] * dL_dT02;
        float dL_dJ02 = vm[2] * dL_dT00 + vm[6] * dL_dT01 + vm[10] * dL_dT02;
        float dL_dJ11 = vm[1] * dL_dT10 + vm[5] * dL_dT11 + vm[9] * dL_dT12;
        float dL_dJ12 = vm[2] * dL_dT10 + vm[6] * dL_dT11 + vm[10] * dL_dT12;
        float tz = 1.f / q.tz, tz2 = tz * tz, tz3 = tz2 * tz;
        float dL_dtx = x_grad_mul * -h_x * tz2 * dL_dJ02;
        float dL_dty = y_grad_mul * -h_y * tz2 * dL_dJ12;
        float dL_dtz = -h_x * tz2 * dL_dJ00 - h_y * tz2 * dL_dJ11 + (2 * h_x * q.tx) * tz3 * dL_dJ02 + (2 * h_y * q.ty) * tz3 * dL_dJ12;
        /* auxiliary.h:89-97 transformVec4x3Transpose; ASSIGNED (backward.cu:273) */
        float dmx = vm[0] * dL_dtx + vm[1] * dL_dty + vm[2] * dL_dtz;
        float dmy = vm[4] * dL_dtx + vm[5] * dL_dty + vm[6] * dL_dtz;
        float dmz = vm[8] * dL_dtx + vm[9] * dL_dty + vm[10] * dL_dtz;

        /* ---- preprocessCUDA (bwd), backward.cu:371-396 ---- */
        float hw = proj[3] * m[0] + proj[7] * m[1] + proj[11] * m[2] + proj[15];
        float m_w = 1.0f / (hw + 0.0000001f);
        float mul1 = (proj[0] * m[0] + proj[4] * m[1] + proj[8] * m[2] + proj[12]) * m_w * m_w;
        float mul2 = (proj[1] * m[0] + proj[5] * m[1] + proj[9] * m[2] + proj[13]) * m_w * m_w;
        const float g2x = dL_dmean2D[2 * idx], g2y = dL_dmean2D[2 * idx + 1];
        float ax = (proj[0] * m_w - proj[3] * mul1) * g2x + (proj[1] * m_w - proj[3] * mul2) * g2y;
        float ay = (proj[4] * m_w - proj[7] * mul1) * g2x + (proj[5] * m_w - proj[7] * mul2) * g2y;
        float az = (proj[8] * m_w - proj[11] * mul1) * g2x + (proj[9] * m_w - proj[11] * mul2) * g2y;
        dmx += ax; dmy += ay; dmz += az;
        dmx += vm[2] * dL_ddepth[idx]; dmy += vm[6] * dL_ddepth[idx]; dmz += vm[10] * dL_ddepth[idx];
        dL_dmeans[3 * idx] = dmx; dL_dmeans[3 * idx + 1] = dmy; dL_dmeans[3 * idx + 2] = dmz;
        if (shs) sh_backward(idx, D, M, means3D, campos, shs, clamped, dL_dcolor, dL_dmeans, dL_dsh);
        if (scales) {
            /* backward.cu:278-341 */
            const float *rot = rotations + 4 * idx;
            float r = rot[0], x = rot[1], y = rot[2], z = rot[3];
            float R[3][3];
            R[0][0] = 1.f - 2.f * (y * y + z * z); R[0][1] = 2.f * (x * y - r * z); R[0][2] = 2.f * (x * z + r * y);
            R[1][0] = 2.f * (x * y + r * z); R[1][1] = 1.f - 2.f * (x * x + z * z); R[1][2] = 2.f * (y * z - r * x);
            R[2][0] = 2.f * (x * z - r * y); R[2][1] = 2.f * (y * z + r * x); R[2][2] = 1.f - 2.f * (x * x + y * y);
            float s[3] = { scale_modifier * scales[3 * idx], scale_modifier * scales[3 * idx + 1], scale_modifier * scales[3 * idx + 2] };
            float M2[3][3]; /* 2.0f * M, M[c][r] = s_r*R[c][r] */
            for (int cc = 0; cc < 3; cc++) for (int rr = 0; rr < 3; rr++) M2[cc][rr] = (s[rr] * R[cc][rr]) * 2.0f;
            float Dm[3][3] = { { dcov[0], 0.5f * dcov[1], 0.5f * dcov[2] }, { 0.5f * dcov[1], dcov[3], 0.5f * dcov[4] }, { 0.5f * dcov[2], 0.5f * dcov[4], dcov[5] } };
            float dM[3][3]; /* dL_dM = (2M) * dL_dSigma : dM[c][r] = sum_k M2[k][r]*Dm[c][k] */
            for (int cc = 0; cc < 3; cc++) for (int rr = 0; rr < 3; rr++)
                dM[cc][rr] = M2[0][rr] * Dm[cc][0] + M2[1][rr] * Dm[cc][1] + M2[2][rr] * Dm[cc][2];
            /* Rt[c][r] = R[r][c]; dMt[c][r] = dM[r][c]; dL_dscale_i = dot(Rt[i], dMt[i]) */
            float dMt[3][3];
            for (int cc = 0; cc < 3; cc++) for (int rr = 0; rr < 3; rr++) dMt[cc][rr] = dM[rr][cc];
            for (int i = 0; i < 3; i++)
                dL_dscale[3 * idx + i] = R[0][i] * dMt[i][0] + R[1][i] * dMt[i][1] + R[2][i] * dMt[i][2];
            for (int i = 0; i < 3; i++) for (int rr = 0; rr < 3; rr++) dMt[i][rr] *= s[i];
            float *dq = dL_drot + 4 * idx;
            dq[0] = 2 * z * (dMt[0][1] - dMt[1][0]) + 2 * y * (dMt[2][0] - dMt[0][2]) + 2 * x * (dMt[1][2] - dMt[2][1]);
            dq[1] = 2 * y * (dMt[1][0] + dMt[0][1]) + 2 * z * (dMt[2][0] + dMt[0][2]) + 2 * r * (dMt[1][2] - dMt[2][1]) - 4 * x * (dMt[2][2] + dMt[1][1]);
            dq[2] = 2 * x * (dMt[1][0] + dMt[0][1]) + 2 * r * (dMt[2][0] - dMt[0][2]) + 2 * z * (dMt[1][2] + dMt[2][1]) - 4 * y * (dMt[2][2] + dMt[0][0]);
            dq[3] = 2 * r * (dMt[0][1] - dMt[1][0]) + 2 * x * (dMt[2][0] + dMt[0][2]) + 2 * y * (dMt[1][2] + dMt[2][1]) - 4 * z * (dMt[1][1] + dMt[0][0]);
        }
    }
}

int gso_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
