// gsr_common.h -- internal layout of the workspaces shared by all translation units.
//
// HBM layout (P Gaussians, T = gx*gy 16x16 tiles, N = W*H pixels, R instances):
//
//   geom  : rec[P]       64 B  one cache line per Gaussian, gathered by the blend kernels
//                              a = {px, py, conA, conB}   b = {conC, opacity, depth, feature}
//                              the raw conic; the blend kernels scale it once per staged instance into
//                              (hA,hB,hC) = -log2(e) * (conA/2, conB, conC/2): exponent of the Gaussian in
//                              base 2 = dx (hA dx + hB dy) + hC dy^2, fed straight to v_exp_f32
//                              c = {r, g, b, cull threshold ln(255 opacity) + margins}  d = {rect width, x0|y0<<16, tile mask lo, hi}
//           rect[P]       8 B  {x0|x1<<16, y0|y1<<16} tile rectangle (0,0 = culled)
//           depthkey[P]   4 B  float bits of view-space depth (positive floats sort as uints)
//           tiles[P]      4 B  surviving tiles of the rectangle = gradient slots of the Gaussian
//           offsets[P]    4 B  exclusive scan of tiles = first gradient slot of the Gaussian
//           tmask[P]      8 B  which tiles of the rectangle the Gaussian can actually change (tile culling)
//           clamped[P*3]  1 B  SH clamp flags (only with SH colours)
//           scan block sums
//   image : ranges[T]     8 B  [start,end) of each tile in the sorted list
//           final_T[N], n_contrib[N]
//           table[NB*T]   4 B  per-(chunk, tile) instance counts -> scatter offsets
//           tile_count[T] 4 B, tile_work[T] 4 B (max n_contrib per tile)
//           sorted_len[T] 4 B  length of the depth-sorted prefix of the tile's list (partial sort of long lists),
//           need_full[T] 4 B   1 = a pixel of the tile was still blending at the end of that prefix
//           ckpt[GSR_SEG_MAX] slots of {float4[N'], float2[N']} (N' = N rounded up to 4): slot k = the checkpoint at list position
//                              gsr_ckpt_pos(k) (k = 0..GSR_SEG_MAX-2): {T in front of it, r, g, b}, {depth, feature}
//                              sums over the segment that ends there; last slot: {checkpoints passed, sums behind the
//                              last one}.  Lets the backward start in the middle of a list (independent depth segments)
//           info           16 B {R, max tile count, instances the occlusion cut-off dropped, quadrant walks that entered depth tier 2}
//           qresume[4T]    4 B  forward blend: where a quadrant ran off its tile's sorted prefix (resume point of the fix-up)
//           qorder[~4T]    4 B  forward blend: dispatch order of the quadrant tasks when the caller keeps per-view walk depths
//   binning: point_list[R] 4 B Gaussian ids per tile segment; bit 31 = "a pixel met this instance inside the alpha = 1/255 guard
//            band" (set by the forward blend, read by the backward blend) (unsorted after the scatter, sorted in place by the
//            tile sort)   seg_keys[R] 8 B key scratch, touched only for lists longer than the LDS sort capacity
//   scratch (backward): slots[R] 48 B  per-instance partial gradients (12 floats)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "../../include/gsraster.h"

#define GSR_TILE 16
#ifndef GSR_MAX_CHUNKS
#define GSR_MAX_CHUNKS 256       // NB: rows of the (chunk, tile) count table = workgroups of the histogram / scatter launches: one per CU
#endif                           // (measured at 1M Gaussians: 2048 x 128 threads 47 + 47 us, 1024 x 256: 36 + 44, 512 x 512: 32 + 41, 256 x 1024: 29 + 38)
#ifndef GSR_CHUNK_GAUSS
#define GSR_CHUNK_GAUSS 1024     // Gaussians per chunk at least (small clouds get fewer chunks, large ones GSR_MAX_CHUNKS bigger chunks)
#endif
#ifndef GSR_HIST_THREADS
#define GSR_HIST_THREADS 1024
#endif
#define GSR_MAX_TILES_LDS 36864  // tiles whose histogram fits one LDS allocation (144 KiB)
#define GSR_SORT_CAP_SMALL 4096  // per-tile list length sorted in 32 KiB of LDS
#define GSR_SORT_CAP_LARGE 16384 // ... in 128 KiB of LDS; longer lists use the global-memory path
#define GSR_NEAR_CAP 2048        // longer lists are sorted only up to (at most) this many nearest instances first

// Tile -> XCD.  Workgroup b of a launch runs on XCD b % 8 whatever it does, so the blend kernels choose which TILES an XCD gets:
// chunks of GSR_XCD_CHUNK consecutive tiles (raster order) dealt round-robin, XCD x's i-th tile = chunk x + 8 (i / c), tile i % c of it.
// Rounds 1-4 gave every XCD one contiguous band of T / 8 tiles (neighbouring tiles share Gaussian records: one L2 fetches them); on a
// scene whose density varies over the image the bands' work differs and the launch lasts as long as the heaviest band --
// tools/wave_trace.py on the init-state frame: the first XCD ran dry at 233 us of a 506 us backward, the last three at 380 / 430 / 506.
// Dealt in chunks of four (profiles/r05_xcd_mapping_ab.txt): that backward 480 -> 349 us, `surfaces` forward 54 -> 50 / backward 167 ->
// 164, config 3 -1 %, config 2 +-0, config 4 +0.3 % (its uniform slab had nothing to balance and loses some L2 sharing between
// rows); chunks of 1 / 16 / one row and 4x4-tile blocks on a skewed XCD pattern measured the same or worse.  Index arithmetic with
// compile-time divisors only: every workgroup of the grid runs it, the empty ones too.
#ifndef GSR_XCD_CHUNK
#define GSR_XCD_CHUNK 4
#endif
static __host__ __device__ __forceinline__ int gsr_xcd_tiles(int T)  // tile slots per XCD (the last chunks may be partly or wholly beyond T)
{
    return (((T + GSR_XCD_CHUNK - 1) / GSR_XCD_CHUNK + 7) / 8) * GSR_XCD_CHUNK;
}
static __host__ __device__ __forceinline__ int gsr_xcd_tile(int xcd, int i, int T)  // the i-th tile of XCD `xcd`, -1 = none
{
    const int t = (xcd + 8 * (i / GSR_XCD_CHUNK)) * GSR_XCD_CHUNK + i % GSR_XCD_CHUNK;
    return t < T ? t : -1;
}

// Forward blend, deepest walks first (gsr_tuning.walk_depths; blend.hip has the why).  One workgroup of NT threads orders the 4 xt
// quadrant-task slots of XCD x (slot i = quadrant i & 3 of tile gsr_xcd_tile(x, i >> 2)) by what each walked at the view's previous visit:
// counting sort on depth / 8 (256 classes, deepest first, slots without a tile last).  Every slot's class is read ONCE into a register and
// used for both passes, so `order` is a permutation of the slots whatever the array holds (garbage, or another stream writing it).
// Runs as eight extra workgroups of the column-scan launch (one-call forward: off the critical path) or as its own launch (blend.hip).
#define GSR_ORDER_MAX_SLOTS 8192  // 4 * gsr_xcd_tiles(T) an ordering workgroup takes (16 k tiles); beyond: natural order
template <int NT>
__device__ __forceinline__ void gsr_fwd_order_block(int x, int T, int xt, const uint32_t* __restrict__ walk_depths, uint32_t* __restrict__ order,
                                                    uint32_t* hist /* LDS [256] */, uint32_t* start /* LDS [256] */)
{
#ifdef __HIP_DEVICE_COMPILE__
    constexpr int PER = GSR_ORDER_MAX_SLOTS / NT;
    const int size = 4 * xt, t = (int)threadIdx.x;
    for (int i = t; i < 256; i += NT) hist[i] = 0;
    __syncthreads();
    uint32_t cls[PER];
#pragma unroll
    for (int k = 0; k < PER; k++) {
        const int i = t + k * NT;
        cls[k] = 255u;
        if (i < size) {
            const int tile = gsr_xcd_tile(x, i >> 2, T);
            if (tile >= 0) { const uint32_t d = walk_depths[4 * tile + (i & 3)] >> 3; cls[k] = 255u - (d < 255u ? d : 255u); }
            atomicAdd(&hist[cls[k]], 1u);
        }
    }
    __syncthreads();
    if (t < 64) {  // exclusive scan of the 256 counts by one wave
        uint32_t v[4], run = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) { v[k] = hist[t * 4 + k]; run += v[k]; }
        uint32_t incl = run;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint32_t o = (uint32_t)__shfl_up((int)incl, d, 64); if (t >= d) incl += o; }
        uint32_t base = incl - run;
#pragma unroll
        for (int k = 0; k < 4; k++) { start[t * 4 + k] = base; base += v[k]; }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < PER; k++) {
        const int i = t + k * NT;
        if (i < size) order[(size_t)x * size + atomicAdd(&start[cls[k]], 1u)] = (uint32_t)i;
    }
#endif
}
#define GSR_SLOT_FLOATS 12
#define GSR_SEG_LEN 128          // longest depth segment of a tile list = instances per backward task (LDS provision)
// Depth segments of a tile's list (= backward tasks; the forward leaves a checkpoint at every boundary), in THREE TIERS at FIXED list
// positions (rounds 5, 6):
//   tier 1: GSR_SEG1 = 7 segments of the launch's segment length L (64 / 128) -- the fine cut the bench-like frames live in (their
//           pixels saturate within a few hundred instances);
//   tier 2: GSR_T2_N = 6 segments of GSR_T2_LEN = 3 L behind position 7 L (with L = 64: boundaries at 640, 832, ..., 1 600);
//   tier 3: segments of GSR_SEG3_LEN = 8 L behind that (2 112, 2 624, ... 8 256), then whatever is left.
// History.  Rounds 1-4 had ONE segment behind tier 1: on frames whose pixels do not saturate -- the faint splats of an initialised,
// untrained scene walk every list to its end -- a 3000-entry list left a 2500-instance task to a single workgroup that ended up alone on
// its SIMD (init-state frame: backward blend 1.17 ms for 1.2 M instances).  Round 5 cut the tail into segments of 1, 2, 3, 4, 6, 8, 12 x L
// and one rest.  Round 6 (gscream_amd/fit.py: frames of a scene early in an optimisation run, tiles walked 4 000 - 6 000 deep): the rest
// behind 43 L -- up to 3 300 instances on one workgroup -- WAS the backward blend (900 us of a 1.61 ms frame), so the tiers went on; and
// the per-wave trace of those frames showed the launch ending with the 8 L / 12 L / 16 L tasks of a few deep tiles in which ONE strip has
// all the stragglers -- 770 iterations x 320 ns on a wave alone on its SIMD = 250 us whatever else the GPU does -- hence segments of at most
// 8 L, and 3 L where most pixels are still alive (profiles/r06_segment_layout_ab.txt: backward blend of the frame after 400 optimiser steps
// 256 -> 180 us; 2 L / 4 L / 6 L cuts: the same within noise, more checkpoint slots).
// The boundaries are FIXED list positions: the first version of tier 2 cut the tail into equal parts of a length chosen from the tile's
// list length, and the occlusion cut-off -- which only removes instances behind everything that blends -- then moved the boundaries,
// i.e. changed how the forward's segment sums associate: images and gradients differed in the last bit with the knob
// (tools/fuzz_parity.py, case 5049).
#define GSR_SEG1 7
#ifndef GSR_T2_LEN
#define GSR_T2_LEN 3     // length of a second-tier segment in units of L
#endif
#ifndef GSR_T2_N
#define GSR_T2_N 6       // second-tier segments
#endif
#ifndef GSR_SEG3_LEN
#define GSR_SEG3_LEN 8   // length of a third-tier segment in units of L
#endif
#ifndef GSR_SEG2
#define GSR_SEG2 20      // segments behind the first tier: GSR_T2_N of the second tier + 13 of the third + the rest
#endif
#define GSR_SEG_MAX (GSR_SEG1 + GSR_SEG2)   // segments per tile = checkpoint slots (GSR_SEG_MAX - 1 checkpoints + the "last" slot)
#define GSR_CKPT_PLANES (GSR_SEG_MAX * 6)
// (kept for the call sites: the unit of the boundaries behind tier 1 is the launch's segment length, whatever the list length)
__host__ __device__ static inline int gsr_seg2_len(int /* n */, int L) { return L; }
// list position of checkpoint k (k = 0 .. GSR_SEG_MAX-2) = end of segment k = start of segment k + 1
__host__ __device__ static inline int gsr_ckpt_pos(int k, int L, int unit)
{
    static_assert(GSR_SEG2 > GSR_T2_N + 1, "tiers");
    return k < GSR_SEG1 ? (k + 1) * L
         : k < GSR_SEG1 + GSR_T2_N ? (GSR_SEG1 + GSR_T2_LEN * (k - GSR_SEG1 + 1)) * unit
         : (GSR_SEG1 + GSR_T2_LEN * GSR_T2_N + GSR_SEG3_LEN * (k - GSR_SEG1 - GSR_T2_N + 1)) * unit;
}
// segments a backward launch has to cover for lists of up to `longest` entries (< 0: unknown): the one that holds position longest - 1
// and everything in front of it (the workgroups of segments no list reaches would only be dispatched to leave at once)
static inline int gsr_segments_for(int longest, int L)
{
    if (longest < 0) return GSR_SEG_MAX;
    int k = 0;
    while (k < GSR_SEG_MAX - 1 && gsr_ckpt_pos(k, L, L) < longest) k++;
    return k + 1;
}
// Segment length of a launch: small images have few tiles, so their lists are cut finer to get enough tasks for the
// 5120 wavefront slots; large ones already have them and shorter tasks would only add fixed costs (measured both ways).
#ifndef GSR_SEG64_MAX_TILES
#define GSR_SEG64_MAX_TILES 4096
#endif
static inline int gsr_seg_len(int T) { return T <= GSR_SEG64_MAX_TILES ? 64 : 128; }
#define GSR_LOG2E 1.4426950408889634f

struct GsrRec {
    float4 a, b, c;
    uint4 d;
};
static_assert(sizeof(GsrRec) == 64, "one record per 64-byte line");

static inline size_t gsr_align(size_t x) { return (x + 255) & ~(size_t)255; }

struct GsrGeom {
    GsrRec* rec;
    uint2* rect;
    uint32_t* depthkey;
    uint32_t* tiles;
    uint32_t* offsets;
    unsigned long long* tmask;  // survivor bit per tile of the rectangle (row-major, first 64 tiles)
    uint8_t* clamped;
    uint32_t* scan_sums;
    size_t bytes;
};

struct GsrImage {
    uint2* ranges;
    float* final_T;
    uint32_t* n_contrib;
    uint32_t* table;
    uint32_t* tile_count;
    uint32_t* tile_work;   // per tile: deepest n_contrib of its pixels = instances the backward must traverse
    uint32_t* sorted_len;  // per tile: its list is depth-sorted up to here (= list length unless partially sorted)
    uint32_t* need_full;   // per tile: 1 = the forward ran off the sorted prefix with pixels still blending
    float* ckpt;           // [GSR_CKPT_PLANES][N], see the header comment
    size_t N;
    uint32_t* info;  // [0] = R, [1] = max tile count, [2] = dropped by the occlusion cut-off, [3] = quadrant walks that entered the second tier of depth segments (forward blend)
    uint32_t* qresume;  // [4 T] per 8x8 quadrant: list position at which the forward ran off the sorted prefix (0 = it did not)
    uint32_t* qorder;   // [8][4 gsr_xcd_tiles(T)] forward blend: the quadrant-task slots of every XCD, deepest previous walk first (gsr_tuning.walk_depths)
    // conservative occlusion cut-off (gsr_tuning.occlusion_cut; preprocess.hip / binning.hip)
    uint32_t* occ_mass;  // [GSR_OCC_COPIES][T][GSR_OCC_BUCKETS] fixed-point (2^-12) sums of -log2(1 - alpha_min) of the whole-tile
                         // instances, one copy per XCD (its workgroups add with L2-local atomics), summed by the cut-off kernel
    uint32_t* occ_cut;   // [T] last depth bucket whose instances are binned (GSR_OCC_BUCKETS = no cut-off)
    uint32_t* occ_drop;  // [GSR_MAX_CHUNKS] instances each histogram chunk dropped
    uint32_t* tile_group;  // [ceil(T / 64)] totals of the column scan's tile groups (banded scatter: a band's first list position)
    size_t bytes;
};
#define GSR_OCC_BUCKETS 160           // 16 per octave of view depth from 2^-3 on (10 octaves; deeper: the last bucket)
#define GSR_OCC_FIXED 4096.0f         // fixed-point scale of the masses
#define GSR_OCC_MAX_TILES 8192        // (8 copies of the mass table: 42 MB at this size; the histogram keeps the cut-off table in LDS)
#define GSR_OCC_COPIES 8              // one mass table per XCD
// depth bucket of a Gaussian from the float bits of its view depth (> 0.2): exponent + 4 mantissa bits, monotone in the depth
__host__ __device__ static inline uint32_t gsr_occ_bucket(uint32_t depth_bits)
{
    const int b = (int)(depth_bits >> 19) - ((127 - 3) << 4);
    return (uint32_t)(b < 0 ? 0 : b >= GSR_OCC_BUCKETS ? GSR_OCC_BUCKETS - 1 : b);
}

struct GsrBinning {
    uint32_t* point_list;
    unsigned long long* seg_keys;
    uint8_t* slot_written;  // [R] 1 = the backward blend wrote this gradient slot; all zero between calls
    size_t bytes;
};

#define GSR_SCAN_ITEMS 2048  // elements per block in the prefix scan

static inline int gsr_scan_blocks(int P) { return (P + GSR_SCAN_ITEMS - 1) / GSR_SCAN_ITEMS; }

static inline int gsr_num_chunks(int P)
{
    int nb = (P + GSR_CHUNK_GAUSS - 1) / GSR_CHUNK_GAUSS;
    if (nb < 1) nb = 1;
    if (nb > GSR_MAX_CHUNKS) nb = GSR_MAX_CHUNKS;
    return nb;
}

static inline GsrGeom gsr_carve_geom(void* base, int P)
{
    GsrGeom g;
    size_t off = 0;
    char* b = (char*)base;
    size_t p = (size_t)(P > 0 ? P : 1);
    g.rec = (GsrRec*)(b + off); off += gsr_align(p * sizeof(GsrRec));
    g.rect = (uint2*)(b + off); off += gsr_align(p * sizeof(uint2));
    g.depthkey = (uint32_t*)(b + off); off += gsr_align(p * 4);
    g.tiles = (uint32_t*)(b + off); off += gsr_align(p * 4);
    g.offsets = (uint32_t*)(b + off); off += gsr_align(p * 4);
    g.tmask = (unsigned long long*)(b + off); off += gsr_align(p * 8);
    g.clamped = (uint8_t*)(b + off); off += gsr_align(p * 3);
    g.scan_sums = (uint32_t*)(b + off); off += gsr_align((size_t)(GSR_MAX_CHUNKS + 1) * 4);  // per-chunk instance totals (gsr_num_chunks(P) <= GSR_MAX_CHUNKS)
    g.bytes = off;
    return g;
}

static inline GsrImage gsr_carve_image(void* base, int P, int W, int H)
{
    GsrImage im;
    size_t off = 0;
    char* b = (char*)base;
    size_t gx = (W + GSR_TILE - 1) / GSR_TILE, gy = (H + GSR_TILE - 1) / GSR_TILE;
    size_t T = gx * gy > 0 ? gx * gy : 1, N = (size_t)W * H > 0 ? (size_t)W * H : 1;
    im.ranges = (uint2*)(b + off); off += gsr_align(T * sizeof(uint2));
    im.final_T = (float*)(b + off); off += gsr_align(N * 4);
    im.n_contrib = (uint32_t*)(b + off); off += gsr_align(N * 4);
    // [chunks][T] count table; beyond the LDS tile limit (global-counter fallback) only T cursor words are needed
    im.table = (uint32_t*)(b + off); off += gsr_align((T > GSR_MAX_TILES_LDS ? (size_t)1 : (size_t)gsr_num_chunks(P)) * T * 4);
    im.tile_count = (uint32_t*)(b + off); off += gsr_align(T * 4);
    im.tile_work = (uint32_t*)(b + off); off += gsr_align(T * 4);
    im.sorted_len = (uint32_t*)(b + off); off += gsr_align(T * 4);
    im.need_full = (uint32_t*)(b + off); off += gsr_align(T * 4);
    im.N = N;
    {
        // The occlusion cut-off's mass tables (GSR_OCC_COPIES * T * GSR_OCC_BUCKETS words = 5 KB per tile) live only from the preprocess
        // kernel to the cut-off kernel of the same forward; the checkpoint planes (GSR_CKPT_PLANES * 4 = 648 B per pixel = 162 KB per full tile: 370 MB at 1008x567) are first
        // written by the forward blend, later on the same stream: the tables ALIAS the checkpoint area (sized for the larger of the
        // two: images of a few pixels) instead of adding 11.6 MB (1008x567) / 42 MB (1920x1080) to every image workspace autograd
        // keeps alive (round 5).
        const size_t ck = (size_t)GSR_CKPT_PLANES * ((N + 3) & ~(size_t)3) * 4;
        const size_t oc = T <= GSR_OCC_MAX_TILES ? (size_t)GSR_OCC_COPIES * T * GSR_OCC_BUCKETS * 4 : 4;
        im.ckpt = (float*)(b + off);
        im.occ_mass = (uint32_t*)(b + off);
        off += gsr_align(ck > oc ? ck : oc);
    }
    im.info = (uint32_t*)(b + off); off += gsr_align(16);
    im.qresume = (uint32_t*)(b + off); off += gsr_align(4 * T * 4);
    im.qorder = (uint32_t*)(b + off); off += gsr_align((size_t)32 * gsr_xcd_tiles((int)T) * 4);
    im.occ_cut = (uint32_t*)(b + off); off += gsr_align(T * 4);
    im.occ_drop = (uint32_t*)(b + off); off += gsr_align((size_t)GSR_MAX_CHUNKS * 4);
    im.tile_group = (uint32_t*)(b + off); off += gsr_align((T / 64 + 1) * 4);
    im.bytes = off;
    return im;
}

static inline GsrBinning gsr_carve_binning(void* base, int R)
{
    GsrBinning bn;
    size_t off = 0;
    char* b = (char*)base;
    size_t r = (size_t)(R > 0 ? R : 1);
    bn.seg_keys = (unsigned long long*)(b + off); off += gsr_align(r * 8);
    bn.point_list = (uint32_t*)(b + off); off += gsr_align(r * 4);
    bn.slot_written = (uint8_t*)(b + off); off += gsr_align(r);
    bn.bytes = off;
    return bn;
}

// ---- stage launchers implemented in the .hip files (host functions, return hipError_t) ----
struct GsrCam {
    // device pointers: wave-uniform addresses, so the kernels fetch them with scalar loads (no host
    // round trip, unlike copying the matrices into the kernel arguments would need)
    const float* view;
    const float* proj;
    const float* campos;
    float tan_fovx, tan_fovy, focal_x, focal_y, scale_modifier;
    int W, H, gx, gy;
};

hipError_t gsr_launch_preprocess(int mode, int P, int D, int M, const GsrCam& cam, const float* means3D,
                                 const float* scales, const float* rotations, const float* opacities,
                                 const float* features, const float* shs, const float* cov3D_precomp,
                                 const float* colors_precomp, const GsrGeom* geom, int32_t* radii, float* px, float* py,
                                 int tile_cull, uint32_t* occ_mass /* NULL = no occlusion masses */, hipStream_t stream);
hipError_t gsr_launch_prefiltered_check(int P, const float* means3D, const float* viewmatrix, uint32_t* culled, hipStream_t stream);
hipError_t gsr_launch_mark_visible(int P, const float* means3D, const float* viewmatrix, uint8_t* present,
                                   hipStream_t stream);
hipError_t gsr_launch_count(int P, int T, int gx, const GsrGeom& geom, const GsrImage& image, uint32_t* info_host_mapped,
                            bool defer_tile_scan, bool occlusion_cut, const uint32_t* walk_depths /* order the forward's tasks by them, or NULL */,
                            bool* ordered /* out: image.qorder was written by this launch */, hipStream_t stream);
int gsr_scatter_bands(int P, int T, int gx, int expected_instances, int forced);  // bands of tile rows per chunk in the scatter launch (binning.hip)
hipError_t gsr_launch_scatter(int P, int T, int gx, const GsrGeom& geom, const GsrImage& image, const GsrBinning& bin,
                              int capacity, int expected_instances, int forced_bands, bool fused_tile_scan, uint32_t* fused_info_host, bool inference,
                              bool occlusion_cut, hipStream_t stream);
hipError_t gsr_launch_tile_sort(int T, int capacity, int max_tile_count, int partial /* 0 complete, 1 prefix bet, 2 complete with the long lists apart */, bool speculative, bool inference, const GsrGeom& geom, const GsrImage& image,
                                const GsrBinning& bin, hipStream_t stream);
hipError_t gsr_launch_blend_forward(int W, int H, int gx, int T, const float* bg, const GsrGeom& geom,
                                    const GsrImage& image, const GsrBinning& bin, float* out_color, float* out_depth,
                                    float* out_feature, int capacity, int max_tile_count, bool only_flagged, bool inference,
                                    uint32_t* walk_depths, bool walk_depths_valid, bool already_ordered, uint32_t* ranoff_report /* fix-up pass: host-mapped word */,
                                    uint32_t serial, hipStream_t stream);
hipError_t gsr_launch_sort_fixup(int T, int capacity, int max_tile_count, const GsrImage& image, const GsrBinning& bin,
                                 bool inference, hipStream_t stream);
hipError_t gsr_launch_blend_backward(int W, int H, int gx, int T, const float* bg, const GsrGeom& geom,
                                     const GsrImage& image, const GsrBinning& bin, const float* dL_dcolor,
                                     const float* dL_ddepth, const float* dL_dfeature, float* slots,
                                     uint8_t* slot_written, uint32_t* heavy_groups, int max_tile_count /* < 0: unknown */, hipStream_t stream);
hipError_t gsr_launch_gauss_backward(int P, int D, int M, const GsrCam& cam, const float* means3D, const int32_t* radii,
                                     const float* shs, const float* scales, const float* rotations,
                                     const float* cov3D_precomp, const GsrGeom& geom, const float* slots,
                                     uint8_t* slot_written, uint32_t* heavy_groups, uint32_t* heavy_seen_mapped, bool heavy_expected, int num_slots,
                                     float* dL_dmeans2D, float* dL_dcolors, float* dL_dopacity, float* dL_dfeatures,
                                     float* dL_dmeans3D, float* dL_dcov3D, float* dL_dsh, float* dL_dscales,
                                     float* dL_drotations, hipStream_t stream);

// ---- loss.hip ----
size_t gsl_workspace_bytes(int C, int H, int W);
hipError_t gsl_launch_forward(int C, int H, int W, const float* img, const float* gt, const float* weight, float a_l1,
                              float a_ssim, void* workspace, float* out, int keep_state, int window_size, hipStream_t stream);
hipError_t gsl_launch_backward(int C, int H, int W, const float* img, const float* gt, const float* weight, float a_l1,
                               float a_ssim, const void* workspace, const float* upstream, float* dL_dimg,
                               int window_size, hipStream_t stream);

// ---- knn.hip ----
size_t gsk_workspace_bytes(int P);
hipError_t gsk_launch(int P, const float* points, float* mean_dist2, void* workspace, hipStream_t stream, const char** why);

// ---- decode.hip ----
hipError_t gsd_launch_count(int N, int K, const float* const* weights, const int32_t* vis, const uint32_t* vis_count, const float* feat, const float* anchor,
                            const float* campos, float* neural_opacity, uint8_t* mask, uint8_t* count, uint32_t* first,
                            uint32_t* total, uint32_t* block_scratch, hipStream_t stream);
hipError_t gsd_launch_emit(int N, int K, const float* const* weights, const int32_t* vis, const uint32_t* vis_count, const float* feat, const float* anchor,
                           const float* offsets, const float* gscale, const float* campos, const float* neural_opacity,
                           const uint8_t* mask, const uint32_t* first, float* xyz, float* color, float* opacity,
                           float* uncertainty, float* scaling, float* rot, hipStream_t stream);
hipError_t gsd_launch_visible_rows(int N, const uint8_t* visible_mask, int32_t* rows, uint32_t* count, uint32_t* block_scratch, hipStream_t stream);
hipError_t gsd_launch_zero_hidden(int N, int K, const uint8_t* visible_mask, float* d_feat, float* d_anchor, float* d_off, float* d_gs,
                                  hipStream_t stream);
hipError_t gsd_launch_backward(int N, int K, const float* const* weights, const int32_t* vis, const float* feat, const float* anchor,
                                     const float* offsets, const float* gscale, const float* campos, const uint8_t* mask,
                                     const uint32_t* first, const float* g_xyz, const float* g_color, const float* g_opacity,
                                     const float* g_unc, const float* g_scaling, const float* g_rot, float* d_feat,
                                     float* d_anchor, float* d_offsets, float* d_gscale, void* workspace, float* const* grads16,
                                     hipStream_t stream);
size_t gsd_weight_grad_workspace_bytes();

// ---- depth_loss.hip (depth terms) ----
size_t gdl_workspace_bytes(int H, int W);
hipError_t gdl_launch_forward(int H, int W, const float* depth, const float* target, const float* lsq_mask,
                              const float* l1_weight, const float* grad_mask, float lambda_l1, float lambda_smooth,
                              void* workspace, float* out5, hipStream_t stream);
hipError_t gdl_launch_backward(int H, int W, const float* depth, const float* target, const float* lsq_mask,
                               const void* workspace, const float* upstream, float* dL_ddepth, hipStream_t stream);

// ---- stats.hip (densification statistics) ----
hipError_t gst_launch_training_stats(int Nv, int K, int M, const int32_t* visible, const float* neural_opacity,
                                     const uint8_t* selection, const uint32_t* first, const uint8_t* update_filter,
                                     const float* viewspace_grad, float* opacity_accum, float* anchor_demon,
                                     float* offset_gradient_accum, float* offset_denom, hipStream_t stream);
