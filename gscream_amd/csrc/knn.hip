// knn.hip -- mean squared distance to the 3 nearest neighbours of every point (init-only).
//
// Replaces simple_knn._C.distCUDA2 (GScream submodules/simple-knn: simple_knn.cu:185-220 SimpleKNN::knn, :134-183
// boxMeanDist / updateKBest, spatial.cu distCUDA2), which scene/gaussian_model.py uses once, at initialisation, to
// size the anchors.  Semantics kept exactly: for every point the three smallest squared fp32 distances to OTHER
// points (index != own; coincident points count with distance 0), averaged; fewer than four points leave FLT_MAX
// terms in the average, as in the reference.
//
// The reference sorts the points along a Morton curve, boxes them in runs of 1024 and lets every thread scan all boxes
// with an exact prune.  Same idea here, shaped for CDNA4:
//   * Morton order WITHOUT a library (round 4; rocPRIM's radix sort before): the 30-bit codes are counted into 4096 buckets by
//     their top 12 bits, a one-block scan turns the counts into segments, the 64-bit keys (code << 32 | index) are dropped into
//     their bucket's segment, and every segment is put in order by the rasterizer's own per-segment sort (binning.hip:
//     gsr_launch_tile_sort -- LDS bucket / bitonic sort up to 16384 keys, the global network beyond, so a cloud that
//     collapses into one Morton cell is slow, not wrong).  Keys are unique, so the order is the stable order of the codes.
//   * the sorted coordinates are GATHERED once into a contiguous float4 array, so the search streams coalesced
//     16-byte loads instead of chasing `points[indices[i]]`,
//   * boxes of 64 points (one wavefront's worth: finer pruning than 1024) under super-boxes of 64 boxes,
//   * one 256-thread workgroup per 256 consecutive points: candidate boxes are tested against the bounding box of the
//     whole group and the group's largest reject radius (wave-uniform control flow), then opened cooperatively: the
//     64 points of a box are staged in LDS once and read by all threads as broadcasts.
// Exact: a box is skipped only if its distance to the group's bounding box exceeds the largest 3rd-best distance in
// the group, which bounds every member's own test.
#include <cstring>
#include <hip/hip_runtime.h>
#include <float.h>
#include <stdint.h>

#include "gsr_math.h"

#define GSK_BOX 64
#define GSK_GROUP 256

struct GskBox { float lo[3], hi[3]; };

// ---- bounding box of the cloud (simple_knn.cu:190-199; the reference's reduction starts from {0,0,0}, which only
// moves the Morton grid, never the result -- kept for identical ordering of equal codes) ----
__global__ void __launch_bounds__(256) gsk_minmax_kernel(int P, const float* __restrict__ pts, float* __restrict__ part)
{
    __shared__ float red[6][256];
    float lo[3] = { 0.f, 0.f, 0.f }, hi[3] = { 0.f, 0.f, 0.f };
    for (int i = blockIdx.x * 256 + threadIdx.x; i < P; i += gridDim.x * 256)
#pragma unroll
        for (int k = 0; k < 3; k++) { const float v = pts[3 * (size_t)i + k]; lo[k] = fminf(lo[k], v); hi[k] = fmaxf(hi[k], v); }
#pragma unroll
    for (int k = 0; k < 3; k++) { red[k][threadIdx.x] = lo[k]; red[3 + k][threadIdx.x] = hi[k]; }
    __syncthreads();
    for (int s = 128; s >= 1; s >>= 1) {
        if ((int)threadIdx.x < s)
#pragma unroll
            for (int k = 0; k < 3; k++) {
                red[k][threadIdx.x] = fminf(red[k][threadIdx.x], red[k][threadIdx.x + s]);
                red[3 + k][threadIdx.x] = fmaxf(red[3 + k][threadIdx.x], red[3 + k][threadIdx.x + s]);
            }
        __syncthreads();
    }
    if (threadIdx.x < 6) part[blockIdx.x * 6 + threadIdx.x] = red[threadIdx.x][0];
}

__device__ __forceinline__ uint32_t gsk_prep_morton(uint32_t x)  // simple_knn.cu:40-47
{
    x = (x | (x << 16)) & 0x030000FF;
    x = (x | (x << 8)) & 0x0300F00F;
    x = (x | (x << 4)) & 0x030C30C3;
    x = (x | (x << 2)) & 0x09249249;
    return x;
}

#define GSK_BUCKET_BITS 12
#define GSK_BUCKETS (1 << GSK_BUCKET_BITS)
__global__ void __launch_bounds__(256) gsk_morton_kernel(int P, int nparts, const float* __restrict__ pts,
                                                         const float* __restrict__ part, uint32_t* __restrict__ codes,
                                                         uint32_t* __restrict__ bucket_count)
{
    __shared__ float bb[6];
    if (threadIdx.x < 6) {
        float v = part[threadIdx.x];
        for (int b = 1; b < nparts; b++) v = threadIdx.x < 3 ? fminf(v, part[b * 6 + threadIdx.x]) : fmaxf(v, part[b * 6 + threadIdx.x]);
        bb[threadIdx.x] = v;
    }
    __syncthreads();
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    uint32_t c[3];
#pragma unroll
    for (int k = 0; k < 3; k++)  // simple_knn.cu:49-61
        c[k] = gsk_prep_morton((uint32_t)(((pts[3 * (size_t)i + k] - bb[k]) / (bb[3 + k] - bb[k])) * ((1 << 10) - 1)));
    const uint32_t code = c[0] | (c[1] << 1) | (c[2] << 2);
    codes[i] = code;
    // (a degenerate axis gives NaN -> 0 or garbage bits above bit 29: the bucket index is masked, the order stays total)
    atomicAdd(&bucket_count[(code >> (30 - GSK_BUCKET_BITS)) & (GSK_BUCKETS - 1)], 1u);
}

// one block: exclusive scan of the 4096 bucket counts -> segments [start, end) + the scatter cursors
__global__ void __launch_bounds__(1024) gsk_bucket_scan_kernel(const uint32_t* __restrict__ bucket_count, uint2* __restrict__ ranges,
                                                               uint32_t* __restrict__ cursor)
{
    __shared__ uint32_t wsum[16];
    constexpr int PER = GSK_BUCKETS / 1024;
    uint32_t v[PER], sum = 0;
#pragma unroll
    for (int k = 0; k < PER; k++) { v[k] = bucket_count[threadIdx.x * PER + k]; sum += v[k]; }
    const uint32_t incl = gsr_wave_scan_add(sum);
    if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = incl;
    __syncthreads();
    uint32_t run = incl - sum;
    for (int w = 0; w < (int)(threadIdx.x >> 6); w++) run += wsum[w];
#pragma unroll
    for (int k = 0; k < PER; k++) {
        ranges[threadIdx.x * PER + k] = make_uint2(run, run + v[k]);
        cursor[threadIdx.x * PER + k] = run;
        run += v[k];
    }
}

__global__ void __launch_bounds__(256) gsk_bucket_scatter_kernel(int P, const uint32_t* __restrict__ codes, uint32_t* __restrict__ cursor,
                                                                 unsigned long long* __restrict__ seg_keys)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    const uint32_t code = codes[i];
    const uint32_t slot = atomicAdd(&cursor[(code >> (30 - GSK_BUCKET_BITS)) & (GSK_BUCKETS - 1)], 1u);
    seg_keys[slot] = ((unsigned long long)code << 32) | (uint32_t)i;  // unique: ties of the code -> ascending index
}

// sorted coordinates, contiguous: sp[i] = {x, y, z, original index}
__global__ void __launch_bounds__(256) gsk_gather_kernel(int P, const float* __restrict__ pts,
                                                         const uint32_t* __restrict__ ids_sorted, float4* __restrict__ sp)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    const uint32_t id = ids_sorted[i];
    sp[i] = make_float4(pts[3 * (size_t)id], pts[3 * (size_t)id + 1], pts[3 * (size_t)id + 2], __uint_as_float(id));
}

__global__ void __launch_bounds__(GSK_BOX) gsk_box_kernel(int P, const float4* __restrict__ sp, GskBox* __restrict__ boxes)
{
    const int i = blockIdx.x * GSK_BOX + threadIdx.x;
    float lo[3] = { FLT_MAX, FLT_MAX, FLT_MAX }, hi[3] = { -FLT_MAX, -FLT_MAX, -FLT_MAX };
    if (i < P) {
        const float4 p = sp[i];
        lo[0] = hi[0] = p.x; lo[1] = hi[1] = p.y; lo[2] = hi[2] = p.z;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1)
#pragma unroll
        for (int k = 0; k < 3; k++) { lo[k] = fminf(lo[k], __shfl_xor(lo[k], d, 64)); hi[k] = fmaxf(hi[k], __shfl_xor(hi[k], d, 64)); }
    if (threadIdx.x == 0) {
        GskBox b;
#pragma unroll
        for (int k = 0; k < 3; k++) { b.lo[k] = lo[k]; b.hi[k] = hi[k]; }
        boxes[blockIdx.x] = b;
    }
}

// bounding boxes of runs of 64 boxes (4096 points): a coarse level the search rejects 64 boxes at a time with
__global__ void __launch_bounds__(GSK_BOX) gsk_superbox_kernel(int nboxes, const GskBox* __restrict__ boxes, GskBox* __restrict__ sboxes)
{
    const int i = blockIdx.x * GSK_BOX + threadIdx.x;
    float lo[3] = { FLT_MAX, FLT_MAX, FLT_MAX }, hi[3] = { -FLT_MAX, -FLT_MAX, -FLT_MAX };
    if (i < nboxes) {
        const GskBox b = boxes[i];
#pragma unroll
        for (int k = 0; k < 3; k++) { lo[k] = b.lo[k]; hi[k] = b.hi[k]; }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1)
#pragma unroll
        for (int k = 0; k < 3; k++) { lo[k] = fminf(lo[k], __shfl_xor(lo[k], d, 64)); hi[k] = fmaxf(hi[k], __shfl_xor(hi[k], d, 64)); }
    if (threadIdx.x == 0) {
        GskBox b;
#pragma unroll
        for (int k = 0; k < 3; k++) { b.lo[k] = lo[k]; b.hi[k] = hi[k]; }
        sboxes[blockIdx.x] = b;
    }
}

__device__ __forceinline__ void gsk_update3(float dist, float best[3])  // simple_knn.cu:120-132 updateKBest<3>
{
#pragma unroll
    for (int j = 0; j < 3; j++)
        if (best[j] > dist) { const float t = best[j]; best[j] = dist; dist = t; }
}

__device__ __forceinline__ float gsk_dist2(const float4 a, const float4 b)
{
    const float dx = b.x - a.x, dy = b.y - a.y, dz = b.z - a.z;
    return dx * dx + dy * dy + dz * dz;
}

// squared distance between two axis-aligned boxes (0 if they overlap)
__device__ __forceinline__ float gsk_box_box(const float lo[3], const float hi[3], const GskBox& b)
{
    float d2 = 0.f;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const float g = fmaxf(fmaxf(b.lo[k] - hi[k], lo[k] - b.hi[k]), 0.f);
        d2 += g * g;
    }
    return d2;
}

__global__ void __launch_bounds__(GSK_GROUP) gsk_search_kernel(int P, int nboxes, const float4* __restrict__ sp,
                                                               const GskBox* __restrict__ boxes,
                                                               const GskBox* __restrict__ sboxes,
                                                               float* __restrict__ mean_dist2)
{
    __shared__ float4 cand[GSK_BOX];
    __shared__ float red[7][GSK_GROUP / 64];
    __shared__ float grp[7];  // group bounding box lo[3], hi[3], largest reject
    const int t = threadIdx.x, i = blockIdx.x * GSK_GROUP + t;
    const bool live = i < P;
    const float4 me = live ? sp[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    float best[3] = { FLT_MAX, FLT_MAX, FLT_MAX };
    // seed: the three neighbours on each side along the curve (simple_knn.cu:148-153) -> a first reject radius
    if (live)
        for (int j = max(0, i - 3); j <= min(P - 1, i + 3); j++)
            if (j != i) gsk_update3(gsk_dist2(me, sp[j]), best);
    const float reject = best[2];
    best[0] = best[1] = best[2] = FLT_MAX;

    // group bounding box and largest reject radius
    float v[7] = { live ? me.x : FLT_MAX, live ? me.y : FLT_MAX, live ? me.z : FLT_MAX,
                   live ? me.x : -FLT_MAX, live ? me.y : -FLT_MAX, live ? me.z : -FLT_MAX, live ? reject : 0.f };
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1)
#pragma unroll
        for (int k = 0; k < 7; k++) {
            const float o = __shfl_xor(v[k], d, 64);
            v[k] = k < 3 ? fminf(v[k], o) : fmaxf(v[k], o);
        }
    if ((t & 63) == 0)
#pragma unroll
        for (int k = 0; k < 7; k++) red[k][t >> 6] = v[k];
    __syncthreads();
    if (t < 7) {
        float r = red[t][0];
        for (int w = 1; w < GSK_GROUP / 64; w++) r = t < 3 ? fminf(r, red[t][w]) : fmaxf(r, red[t][w]);
        grp[t] = r;
    }
    __syncthreads();
    const float glo[3] = { grp[0], grp[1], grp[2] }, ghi[3] = { grp[3], grp[4], grp[5] };
    const float greject = grp[6];

    for (int b = 0; b < nboxes; b++) {
        if ((b & (GSK_BOX - 1)) == 0 && gsk_box_box(glo, ghi, sboxes[b / GSK_BOX]) > greject) {  // 64 boxes rejected at once
            b += GSK_BOX - 1;
            continue;
        }
        const GskBox box = boxes[b];  // uniform address: scalar loads
        if (gsk_box_box(glo, ghi, box) > greject) continue;  // workgroup-uniform: no member can need this box
        __syncthreads();  // previous candidates consumed
        if (t < GSK_BOX) {
            const int j = b * GSK_BOX + t;
            cand[t] = j < P ? sp[j] : make_float4(FLT_MAX, FLT_MAX, FLT_MAX, 0.f);
        }
        __syncthreads();
        // the member's own exact test (simple_knn.cu:165-168)
        float d2 = 0.f;
        {
            const float p[3] = { me.x, me.y, me.z };
#pragma unroll
            for (int k = 0; k < 3; k++) {
                float g = 0.f;
                if (p[k] < box.lo[k] || p[k] > box.hi[k]) g = fminf(fabsf(p[k] - box.lo[k]), fabsf(p[k] - box.hi[k]));
                d2 += g * g;
            }
        }
        if (!live || d2 > reject || d2 > best[2]) continue;
        const int n = min(GSK_BOX, P - b * GSK_BOX);
        for (int c = 0; c < n; c++) {
            if (b * GSK_BOX + c == i) continue;
            gsk_update3(gsk_dist2(me, cand[c]), best);
        }
    }
    if (live) mean_dist2[__float_as_uint(me.w)] = (best[0] + best[1] + best[2]) / 3.0f;
}

// ---- host side -------------------------------------------------------------------------------------------------
struct GskWorkspace {
    uint32_t *codes, *ids_sorted, *bucket_count, *cursor;
    unsigned long long* seg_keys;
    uint2* ranges;
    float4* sp;
    GskBox *boxes, *sboxes;
    float* part;
    size_t bytes;
};
#define GSK_MINMAX_BLOCKS 256

static GskWorkspace gsk_carve(void* base, int P)
{
    GskWorkspace w;
    char* b = (char*)base;
    size_t off = 0;
    const size_t p = (size_t)(P > 0 ? P : 1);
    w.codes = (uint32_t*)(b + off); off += gsr_align(p * 4);
    w.ids_sorted = (uint32_t*)(b + off); off += gsr_align(p * 4);
    w.seg_keys = (unsigned long long*)(b + off); off += gsr_align(p * 8);
    w.ranges = (uint2*)(b + off); off += gsr_align((size_t)GSK_BUCKETS * sizeof(uint2));
    w.bucket_count = (uint32_t*)(b + off); off += gsr_align((size_t)GSK_BUCKETS * 4);
    w.cursor = (uint32_t*)(b + off); off += gsr_align((size_t)GSK_BUCKETS * 4);
    w.sp = (float4*)(b + off); off += gsr_align(p * 16);
    w.boxes = (GskBox*)(b + off); off += gsr_align(((p + GSK_BOX - 1) / GSK_BOX) * sizeof(GskBox));
    w.sboxes = (GskBox*)(b + off); off += gsr_align(((p + GSK_BOX * GSK_BOX - 1) / (GSK_BOX * GSK_BOX)) * sizeof(GskBox));
    w.part = (float*)(b + off); off += gsr_align(GSK_MINMAX_BLOCKS * 6 * 4);
    w.bytes = off;
    return w;
}

size_t gsk_workspace_bytes(int P) { return gsk_carve(nullptr, P).bytes; }

hipError_t gsk_launch(int P, const float* points, float* mean_dist2, void* workspace, hipStream_t stream, const char** why)
{
    *why = "";
    if (P <= 0) return hipSuccess;
    const GskWorkspace w = gsk_carve(workspace, P);
    const int nb256 = (P + 255) / 256, nparts = nb256 < GSK_MINMAX_BLOCKS ? nb256 : GSK_MINMAX_BLOCKS;
    hipLaunchKernelGGL(gsk_minmax_kernel, dim3(nparts), dim3(256), 0, stream, P, points, w.part);
    hipError_t e = hipMemsetAsync(w.bucket_count, 0, (size_t)GSK_BUCKETS * 4, stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(gsk_morton_kernel, dim3(nb256), dim3(256), 0, stream, P, nparts, points, w.part, w.codes, w.bucket_count);
    hipLaunchKernelGGL(gsk_bucket_scan_kernel, dim3(1), dim3(1024), 0, stream, w.bucket_count, w.ranges, w.cursor);
    hipLaunchKernelGGL(gsk_bucket_scatter_kernel, dim3(nb256), dim3(256), 0, stream, P, w.codes, w.cursor, w.seg_keys);
    {   // every bucket's segment in key order, by the rasterizer's per-segment sort (longest segment unknown on the host: all
        // size classes are launched, workgroups of the wrong class leave at once); ids_sorted = the keys' low words in order
        GsrGeom geom{};
        GsrImage image{};
        GsrBinning bin{};
        image.ranges = w.ranges;
        bin.seg_keys = w.seg_keys;
        bin.point_list = w.ids_sorted;
        e = gsr_launch_tile_sort(GSK_BUCKETS, P, -1, false, false, true, geom, image, bin, stream);
        if (e != hipSuccess) { *why = "bucket sort"; return e; }
    }
    const int nboxes = (P + GSK_BOX - 1) / GSK_BOX;
    hipLaunchKernelGGL(gsk_gather_kernel, dim3(nb256), dim3(256), 0, stream, P, points, w.ids_sorted, w.sp);
    hipLaunchKernelGGL(gsk_box_kernel, dim3(nboxes), dim3(GSK_BOX), 0, stream, P, w.sp, w.boxes);
    hipLaunchKernelGGL(gsk_superbox_kernel, dim3((nboxes + GSK_BOX - 1) / GSK_BOX), dim3(GSK_BOX), 0, stream, nboxes, w.boxes, w.sboxes);
    hipLaunchKernelGGL(gsk_search_kernel, dim3((P + GSK_GROUP - 1) / GSK_GROUP), dim3(GSK_GROUP), 0, stream, P, nboxes, w.sp,
                       w.boxes, w.sboxes, mean_dist2);
    return hipGetLastError();
}
