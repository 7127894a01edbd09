// binning.hip -- Gaussian -> tile binning and the per-tile depth sort.
//
// Replaces, in DGR rasterizer_impl.cu: cub::DeviceScan::InclusiveSum (:283), duplicateWithKeys
// (:70-111), the 44/45-bit global cub::DeviceRadixSort::SortPairs (:306-314) and identifyTileRanges
// (:116-138).  The reference sorts all R instances globally by (tile | depth): 6 radix passes over
// 12-byte pairs.  Here the tile part of the key is handled by ONE counting pass (a T-bin histogram
// per chunk of Gaussians, kept in LDS), which yields each tile's segment directly (= `ranges`), and
// the depth part by an independent LDS sort per tile.  Order inside a tile is (depth bits, Gaussian
// id) ascending: identical to the reference's stable sort, whose ties are broken by emission order
// = ascending id (SURVEY A-9).  Keys are unique, so the result does not depend on the order in
// which the scatter claims slots.
//
// HBM traffic: hist reads rect + mask (16 B/G); table NB*T*4 B written, scanned, read; scatter reads 24 B/G and
// writes 4 B/instance (the id) + 16 B/G (record tail); sort reads 4 B + a 4-B depth gather and writes 4 B per instance.
#include "gsr_math.h"

typedef unsigned long long u64;

// ---------------------------------------------------------------------------------------------
// Tile histogram per chunk of Gaussians  (one counting-sort pass on the tile id, in LDS)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void gsr_chunk_bounds(int P, int nchunks, int chunk, int& lo, int& hi)
{
    const int per = (P + nchunks - 1) / nchunks;
    lo = chunk * per;
    hi = min(P, lo + per);
}

// GLOBAL = true: image with more tiles than one LDS allocation holds (> GSR_MAX_TILES_LDS, ~3000x3000 px): the
// counters are the global tile_count[] (zeroed by the launcher), one agent-scope atomic per instance -- slow but
// unbounded; no table is produced and the column scan is skipped.
// OCC (gsr_tuning.occlusion_cut): occ_cut[tile] = the last depth bucket whose instances can still blend in the tile
// (gsr_occ_cut_kernel).  An instance behind it is not counted, and the Gaussian's survivor mask -- tmask[], the copy in its
// record, its slot count tiles[] -- loses the bit, so that scatter, slot numbering and both backward kernels see the smaller
// footprint without knowing why.  (Rectangle positions beyond 64 have no mask bit and stay.)
struct GsrOcclusion {
    const uint32_t* cut;        // [T]
    const uint32_t* depthkey;   // [P]
    u64* tmask_rw;              // [P]
    uint32_t* tiles;            // [P]
    GsrRec* rec;                // [P]
    uint32_t* chunk_drop;       // [nchunks]
};
template <bool GLOBAL, bool OCC>
__global__ void __launch_bounds__(GSR_HIST_THREADS) gsr_tile_hist_kernel(int P, int T, int gx, int nchunks,
                                                                        const uint2* __restrict__ rect,
                                                                        const u64* __restrict__ tmask,
                                                                        uint32_t* __restrict__ table,
                                                                        uint32_t* __restrict__ chunk_sum, const GsrOcclusion oc)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t hist_lds[];
    uint32_t* hist = GLOBAL ? table /* = tile_count[T] */ : hist_lds;
    __shared__ uint32_t heads_all[GSR_HIST_THREADS];
    __shared__ u64 drop_all[OCC ? GSR_HIST_THREADS : 1];
    uint8_t* cutb = reinterpret_cast<uint8_t*>(hist_lds + T);  // OCC: the cut-off table, one byte per tile, behind the counters
    volatile uint32_t* heads = heads_all + (threadIdx.x & ~63u);
    u64* drops = drop_all + (OCC ? (threadIdx.x & ~63u) : 0u);
    uint32_t dropped = 0;
    if (!GLOBAL) {
        for (int t = threadIdx.x; t < T; t += blockDim.x) {
            hist[t] = 0;
            if (OCC) cutb[t] = (uint8_t)min(oc.cut[t], 255u);
        }
        __syncthreads();
    }
    uint32_t mine = 0;
    int lo, hi;
    gsr_chunk_bounds(P, nchunks, blockIdx.x, lo, hi);
    // four Gaussians per thread per trip, loads issued together (the kernel is latency-bound: 8 waves per CU)
    constexpr int U = 4;
    for (int gw = lo + (int)(threadIdx.x & ~63u); gw < hi; gw += blockDim.x * U) {  // wave-uniform trip count
        const int gb = gw + (int)(threadIdx.x & 63);
        uint2 rcs[U];
        u64 mks[U];
        uint32_t bks[U];
#pragma unroll
        for (int k = 0; k < U; k++) {
            const int g = gb + k * blockDim.x;
            rcs[k] = g < hi ? rect[g] : make_uint2(0u, 0u);
            // (OCC: this kernel also STORES the masks, through oc.tmask_rw -- the same buffer -- so it reads them through that pointer
            // too: a load through the __restrict__ const view may legally be re-issued after the store)
            mks[k] = g < hi ? (OCC ? oc.tmask_rw[g] : tmask[g]) : 0ull;
            bks[k] = (OCC && g < hi) ? gsr_occ_bucket(oc.depthkey[g]) : 0u;
        }
#pragma unroll
        for (int k = 0; k < U; k++) {
            if (!OCC) {
                gsr_wave_for_each_instance(rcs[k], mks[k], 0u, heads, [&](int, int x, int y, uint32_t) { atomicAdd(&hist[y * gx + x], 1u); if (GLOBAL) mine++; });
            } else {
                const int lane = (int)(threadIdx.x & 63u);
                drops[lane] = 0ull;
                gsr_wave_for_each_instance_t<true>(rcs[k], mks[k], bks[k], heads, [&](int owner, int x, int y, uint32_t obk, uint32_t pos) {
                    const int t = y * gx + x;
                    if (pos < 64u && obk > (uint32_t)cutb[t]) atomicOr(&drops[owner], 1ull << pos);  // behind the tile's cut-off: never binned
                    else atomicAdd(&hist[t], 1u);
                });
                const u64 d = *reinterpret_cast<volatile u64*>(&drops[lane]);  // (wave-private LDS, program order: the atomics above are done)
                if (d) {
                    const int g = gb + k * blockDim.x;
                    const u64 nm = mks[k] & ~d;
                    dropped += (uint32_t)__popcll(mks[k] & d);
                    oc.tmask_rw[g] = nm;
                    oc.tiles[g] = gsr_rect_count(rcs[k], nm);
                    uint2* rd = reinterpret_cast<uint2*>(&oc.rec[g].d.z);
                    *rd = make_uint2((uint32_t)nm, (uint32_t)(nm >> 32));
                }
            }
        }
    }
    __syncthreads();
    if (!GLOBAL) {
        uint32_t* row = table + (size_t)blockIdx.x * T;
        for (int t = threadIdx.x; t < T; t += blockDim.x) { const uint32_t v = hist[t]; row[t] = v; mine += v; }
    }
    // instances of the chunk = gradient slots of its Gaussians: the scatter turns these into slot offsets
    if (threadIdx.x == 0) heads_all[0] = 0u;
    __syncthreads();
    mine = gsr_wave_scan_add(mine);
    if ((threadIdx.x & 63) == 63) atomicAdd(&heads_all[0], mine);
    __syncthreads();
    if (threadIdx.x == 0) chunk_sum[blockIdx.x] = heads_all[0];
    if (OCC) {  // instances this chunk dropped (reported to the caller: does the pass pay on this kind of frame?)
        __syncthreads();
        if (threadIdx.x == 0) heads_all[0] = 0u;
        __syncthreads();
        dropped = gsr_wave_scan_add(dropped);
        if ((threadIdx.x & 63) == 63 && dropped) atomicAdd(&heads_all[0], dropped);
        __syncthreads();
        if (threadIdx.x == 0) oc.chunk_drop[blockIdx.x] = heads_all[0];
    }
}

// Occlusion cut-off per tile: along the depth buckets, the first one at which the accumulated whole-tile masses say that every
// pixel's transmittance is below 1e-4 (with a binade of margin): nothing behind that bucket can blend.  One wavefront per tile.
__global__ void __launch_bounds__(256) gsr_occ_cut_kernel(int T, const uint32_t* __restrict__ mass, uint32_t* __restrict__ cut)
{
    const int tile = blockIdx.x * 4 + (int)(threadIdx.x >> 6), lane = (int)(threadIdx.x & 63u);
    if (tile >= T) return;
    const uint32_t thr = (uint32_t)((13.2877f + 1.0f) * GSR_OCC_FIXED);  // -log2(1e-4) + one binade
    const uint32_t* m = mass + (size_t)tile * GSR_OCC_BUCKETS;
    uint32_t run = 0, first = GSR_OCC_BUCKETS;
    for (int b0 = 0; b0 < GSR_OCC_BUCKETS; b0 += 64) {
        const int b = b0 + lane;
        uint32_t v = 0u;
        if (b < GSR_OCC_BUCKETS)
#pragma unroll
            for (int c = 0; c < GSR_OCC_COPIES; c++) v += m[(size_t)c * T * GSR_OCC_BUCKETS + b];  // the XCDs' copies
        const uint32_t incl = gsr_wave_scan_add(v) + run;
        const unsigned long long hit = __ballot(b < GSR_OCC_BUCKETS && incl >= thr);
        if (hit) { first = (uint32_t)(b0 + __builtin_ctzll(hit)); break; }
        run = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    }
    if (lane == 0) cut[tile] = first;  // instances of buckets <= first stay; GSR_OCC_BUCKETS = no cut-off
}

// Column pass over table[nchunks][T]: for every tile, exclusive prefix over chunks (in place) and the
// tile total.  Block = 64 tiles x 8 chunk groups.
#ifndef GSR_COLSCAN_GROUPS
#define GSR_COLSCAN_GROUPS 8  // chunk groups per workgroup: GSR_COLSCAN_TILES tiles x 8 groups = 512 threads, <= 64 rows per thread
#endif
#ifndef GSR_COLSCAN_TILES
#define GSR_COLSCAN_TILES 64
#endif
__global__ void __launch_bounds__(GSR_COLSCAN_TILES * GSR_COLSCAN_GROUPS) gsr_table_colscan_kernel(int T, int nchunks, uint32_t* __restrict__ table,
                                                                                    uint32_t* __restrict__ tile_count,
                                                                                    uint32_t* __restrict__ group_total, int nscan, int xt,
                                                                                    const uint32_t* __restrict__ walk_depths,
                                                                                    uint32_t* __restrict__ qorder)
{
    __shared__ uint32_t part[GSR_COLSCAN_GROUPS][GSR_COLSCAN_TILES];
    // workgroups behind the scan's own: the forward blend's dispatch order, one XCD's each (gsr_common.h gsr_fwd_order_block) -- a
    // launch of its own in front of the blend cost the forward 4.6 us; here it runs beside the scan
    if ((int)blockIdx.x >= nscan) {
        uint32_t* l = &part[0][0];
        static_assert(GSR_COLSCAN_GROUPS * GSR_COLSCAN_TILES >= 512, "LDS for the ordering's two 256-word arrays");
        gsr_fwd_order_block<GSR_COLSCAN_TILES * GSR_COLSCAN_GROUPS>((int)blockIdx.x - nscan, T, xt, walk_depths, qorder, l, l + 256);
        return;
    }
    const int tl = threadIdx.x % GSR_COLSCAN_TILES, grp = threadIdx.x / GSR_COLSCAN_TILES;
    const int tile = blockIdx.x * GSR_COLSCAN_TILES + tl;
    const int per = (nchunks + GSR_COLSCAN_GROUPS - 1) / GSR_COLSCAN_GROUPS;
    const int c0 = grp * per, c1 = min(nchunks, c0 + per);
    // each thread owns <= 64 table rows of one tile: keep them in registers between the two passes (the loads of
    // the first pass are independent, so they are all in flight together; the second pass needs no re-read)
    constexpr int MAXR = GSR_MAX_CHUNKS / GSR_COLSCAN_GROUPS;
    uint32_t v[MAXR];
    uint32_t s = 0;
#pragma unroll
    for (int r = 0; r < MAXR; r++) {
        const int c = c0 + r;
        v[r] = (tile < T && c < c1) ? table[(size_t)c * T + tile] : 0u;
    }
#pragma unroll
    for (int r = 0; r < MAXR; r++) s += v[r];
    part[grp][tl] = s;
    __syncthreads();
    uint32_t run = 0, total = 0;
#pragma unroll
    for (int g2 = 0; g2 < GSR_COLSCAN_GROUPS; g2++) {
        const uint32_t pv = part[g2][tl];
        run += g2 < grp ? pv : 0u;
        total += pv;
    }
    if (tile < T) {
#pragma unroll
        for (int r = 0; r < MAXR; r++) {
            const int c = c0 + r;
            if (c < c1) table[(size_t)c * T + tile] = run;
            run += v[r];
        }
        if (grp == 0) tile_count[tile] = total;
    }
    // + the total of this workgroup's GSR_COLSCAN_TILES tiles (wave 0 = chunk group 0 holds one tile total per lane): lets a banded
    // scatter workgroup find its band's first list position from T / 64 words instead of scanning all T tile totals
    static_assert(GSR_COLSCAN_TILES == 64, "one wavefront per tile group");
    if (grp == 0) {
        const uint32_t incl = gsr_wave_scan_add(tile < T ? total : 0u);
        if (tl == 63) group_total[blockIdx.x] = incl;
    }
}

// Single block of 1024: exclusive scan of the tile totals -> ranges[t] = [start, end); info = {R, max count}.
// Thread i owns ceil(T/1024) consecutive tiles; one DPP wave scan + 16 wave totals in LDS.  The counts are loaded ONCE, all loads
// in flight together (PER = compile-time bound on the tiles per thread): the first version read them in two dependent loops and
// took 20 us at 8160 tiles -- a chain of memory round trips on a single workgroup (round 4: the banded scatter needs this kernel).
// (PER = 0: no register copy, the counts are read again in the second loop -- images beyond the LDS tile limit, any T)
template <int PER>
__global__ void __launch_bounds__(1024) gsr_tile_scan_kernel(int T, const uint32_t* __restrict__ tile_count,
                                                             uint2* __restrict__ ranges, uint32_t* __restrict__ info,
                                                             uint32_t* __restrict__ tile_work, uint32_t* __restrict__ sorted_len,
                                                             uint32_t* __restrict__ need_full, uint32_t* __restrict__ info_host,
                                                             const uint32_t* __restrict__ occ_drop, int ndrop)
{
    __shared__ uint32_t wsum[16], wmax[16];
    __shared__ uint32_t drop_total;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) drop_total = 0u;
    __syncthreads();
    if (occ_drop && (int)threadIdx.x < ndrop) { const uint32_t d = occ_drop[threadIdx.x]; if (d) atomicAdd(&drop_total, d); }  // ndrop <= 256
    const int per = (T + 1023) / 1024, t0 = threadIdx.x * per;  // per <= PER
    uint32_t v[PER > 0 ? PER : 1];
    uint32_t sum = 0, mx = 0;
    if (PER > 0) {
#pragma unroll
        for (int i = 0; i < PER; i++) v[i] = (i < per && t0 + i < T) ? tile_count[t0 + i] : 0u;
#pragma unroll
        for (int i = 0; i < PER; i++) { sum += v[i]; mx = max(mx, v[i]); }
    } else {
        for (int i = 0; i < per; i++) {
            const uint32_t c = t0 + i < T ? tile_count[t0 + i] : 0u;
            sum += c;
            mx = max(mx, c);
        }
    }
    const uint32_t incl = gsr_wave_scan_add(sum);
    mx = gsr_wave_scan_max(mx);
    if (lane == 63) { wsum[wave] = incl; wmax[wave] = mx; }
    __syncthreads();
    uint32_t run = incl - sum, total = 0, gmax = 0;
#pragma unroll
    for (int w = 0; w < 16; w++) {
        const uint32_t sw = wsum[w];
        run += w < wave ? sw : 0u;
        total += sw;
        gmax = max(gmax, wmax[w]);
    }
    auto emit = [&](const int t, const uint32_t c) {
        ranges[t] = make_uint2(run, run + c);
        tile_work[t] = 0u;  // the forward blend's quadrant wavefronts atomicMax their traversal depth into it
        sorted_len[t] = c;  // fully sorted unless the partial sort of long lists says otherwise
        need_full[t] = 0u;
        run += c;
    };
    if (PER > 0) {
#pragma unroll
        for (int i = 0; i < PER; i++)
            if (i < per && t0 + i < T) emit(t0 + i, v[i]);
    } else {
        for (int i = 0; i < per && t0 + i < T; i++) emit(t0 + i, tile_count[t0 + i]);
    }
    if (threadIdx.x == 0) {
        info[0] = total; info[1] = gmax; info[2] = drop_total;  // (the barrier between the atomics and here: the scan's __syncthreads)
        info[3] = 0u;  // quadrant walks that go beyond the first tier of depth segments: counted by the forward blend, read by the backward blend
        // the host's copy, written straight into its pinned (device-mapped) buffer: no copy kernel in the stream
        if (info_host) { info_host[0] = total; info_host[1] = gmax; info_host[2] = drop_total; }
    }
}

// ---------------------------------------------------------------------------------------------
// Scatter: every Gaussian writes (depth bits << 32 | id) into the segment of each tile of its
// rectangle (slot claimed with an LDS atomic on the chunk's cursor row), and completes its record.
// ---------------------------------------------------------------------------------------------
// GLOBAL = true (more tiles than LDS holds): the per-tile cursors are a global array initialised to the segment
// starts (`table` then points at it), claimed with agent-scope atomics.
// When the tile scan is folded into this kernel (the one-call forward: the host does not need R before stage 2 is enqueued):
// every workgroup scans the tile totals for its own cursors, workgroup 0 also writes what gsr_tile_scan_kernel writes.
struct GsrFusedScan {
    const uint32_t* tile_count;  // null = not folded in: `ranges` was written by gsr_tile_scan_kernel
    uint2* ranges;
    uint32_t *info, *tile_work, *sorted_len, *need_full, *info_host;
    const uint32_t* occ_drop;  // per-chunk counts of the occlusion cut-off's dropped instances (NULL = off): summed into info[2]
    const uint32_t* tile_group;  // totals of the tile groups of GSR_COLSCAN_TILES tiles (column scan)
};
// BANDED = false compiles the band arithmetic away (nbands = 1: the plain launch, e.g. config 2, keeps its round-3 code: with the
// run-time form it was 2 us slower).
template <bool GLOBAL, bool BANDED>
__global__ void __launch_bounds__(GSR_HIST_THREADS) gsr_scatter_kernel(
    int P, int T, int gx, int nchunks, const uint2* __restrict__ rect, const u64* __restrict__ tmask,
    const uint32_t* __restrict__ depthkey, const uint32_t* __restrict__ table, const uint32_t* __restrict__ chunk_sum,
    const uint2* __restrict__ ranges, uint32_t* __restrict__ offsets, u64* __restrict__ seg_keys, uint32_t capacity, uint32_t stage_cap,
    const GsrFusedScan fs, int nbands_arg, int gy)
{
    const int nbands = BANDED ? nbands_arg : 1;
    extern __shared__ __attribute__((aligned(16))) uint32_t cursor_lds[];
    uint32_t* cursor = GLOBAL ? const_cast<uint32_t*>(table) : cursor_lds;
    __shared__ uint32_t heads_all[GSR_HIST_THREADS];
    __shared__ uint32_t wsum[GSR_HIST_THREADS / 64];
    __shared__ uint32_t wsum4[4][GSR_HIST_THREADS / 64];
    __shared__ uint32_t chunk_first;
    volatile uint32_t* heads = heads_all + (threadIdx.x & ~63u);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // workgroup b runs on XCD b % 8: consecutive CHUNKS go to one XCD, so that the runs two neighbouring chunks write into a
    // tile segment (they share a 32-byte sector at their border) meet in the same L2
    // BANDED form (round 4; nbands > 1): images with so many tiles / instances that a chunk's keys do not fit the staging buffer
    // (config 4: 8160 tiles x 12 B of cursors leave room for 5k keys, a chunk has 50k) wrote every key as a lone 32-byte sector
    // (526 MB moved for 195 MB).  The launch then has nbands workgroups per chunk; workgroup (chunk, band) walks the chunk's
    // Gaussians but emits only the instances of its band of tile rows: the per-tile arrays shrink nbands-fold, the band's share
    // of the chunk fits the staging buffer, and the keys leave in runs again.  The (chunk, tile) table is unchanged.  All bands
    // of a chunk run on the same XCD (the chunk's rect / mask / depth stream stays in that L2).
    const int blk = BANDED ? (int)(blockIdx.x % (uint32_t)nchunks) : (int)blockIdx.x, band = BANDED ? (int)(blockIdx.x / (uint32_t)nchunks) : 0;
    const int chunk = (nchunks & 7) == 0 ? (blk & 7) * (nchunks >> 3) + (blk >> 3) : blk;
    const int by0 = BANDED ? (int)(((long long)band * gy) / nbands) : 0, by1 = BANDED ? (int)(((long long)(band + 1) * gy) / nbands) : gy;
    const int t_lo = BANDED ? by0 * gx : 0, TB = BANDED ? (by1 - by0) * gx : T;  // this workgroup's tiles: [t_lo, t_lo + TB)
    // STAGED form (round 3).  Written straight to its tile segment, every 8-byte key costs a 32-byte sector write (86 MB of
    // write requests for 21 MB of keys; with the stores cut out the launch takes 24 instead of 38 us).  When the chunk's
    // instances fit the rest of the LDS they are first placed TILE-MAJOR in a staging buffer -- the chunk's own count per tile
    // is the difference of two rows of the scanned table, an LDS scan of those gives the local offsets -- and then streamed
    // out: neighbouring staging entries of one tile are neighbouring keys of that tile's segment, so a wave's store covers
    // runs (4.6 keys on the bench scene) instead of 64 lone sectors.  Which slot of its (chunk, tile) run a key gets is as
    // arbitrary as before; the per-tile sort does not care (keys are unique).
    //   dynamic LDS: cursor[T] | loff[T] | gbase[T] | keys[cap] (8 B) | tile[cap] (2 B)
    uint32_t* loff = cursor_lds + TB;
    uint32_t* gbase = cursor_lds + 2 * (size_t)TB;
    u64* skey = reinterpret_cast<u64*>(cursor_lds + 3 * (size_t)TB + ((3 * (size_t)TB) & 1));
    uint16_t* stile = reinterpret_cast<uint16_t*>(skey + stage_cap);
    __shared__ uint32_t band_total;  // instances of this (chunk, band): known after the scan of the per-tile counts below
    const bool may_stage = !GLOBAL && stage_cap > 0u && TB <= 65535;  // block-uniform
    bool staged = false;
    const bool fused = !GLOBAL && fs.tile_count != nullptr;  // block-uniform
    if (BANDED && fused && blockIdx.x != 0) {
        // banded: only this band's tile starts are needed.  First list position of the band = the totals of the tile groups in front
        // of it (T / 64 words from the column scan) + an exclusive scan over the tiles from the band's group boundary on.
        const int g0 = t_lo / GSR_COLSCAN_TILES, ts = g0 * GSR_COLSCAN_TILES, n = t_lo + TB - ts;
        uint32_t pv = 0u;
        for (int g = threadIdx.x; g < g0; g += blockDim.x) pv += fs.tile_group[g];
        const int per = (n + (int)blockDim.x - 1) / (int)blockDim.x, t0 = ts + (int)threadIdx.x * per;
        uint32_t sum = 0;
        for (int i = 0; i < per; i++) sum += t0 + i < ts + n ? fs.tile_count[t0 + i] : 0u;
        const uint32_t incl = gsr_wave_scan_add(sum), gincl = gsr_wave_scan_add(pv);
        if (lane == 63) { wsum[wave] = incl; wsum4[0][wave] = gincl; }
        __syncthreads();
        uint32_t run = incl - sum;
        for (int w = 0; w < (int)(blockDim.x >> 6); w++) run += (w < wave ? wsum[w] : 0u) + wsum4[0][w];
        for (int i = 0; i < per; i++) {
            if (t0 + i >= ts + n) break;
            if (t0 + i >= t_lo) cursor[t0 + i - t_lo] = run;
            run += fs.tile_count[t0 + i];
        }
        __syncthreads();
    } else if (fused) {
        // tile starts = exclusive scan of the tile totals, left in cursor[] for the initialisation below (thread i owns
        // ceil(T / blockDim) consecutive tiles, like the staged scan further down)
        const int per = (T + (int)blockDim.x - 1) / (int)blockDim.x, t0 = (int)threadIdx.x * per;
        uint32_t sum = 0, mx = 0;
        for (int i = 0; i < per; i++) {
            const uint32_t v = t0 + i < T ? fs.tile_count[t0 + i] : 0u;
            sum += v;
            mx = max(mx, v);
        }
        const uint32_t incl = gsr_wave_scan_add(sum);
        mx = gsr_wave_scan_max(mx);
        if (lane == 63) { wsum[wave] = incl; wsum4[0][wave] = mx; }
        __syncthreads();
        uint32_t run = incl - sum, total = 0, gmax = 0;
        for (int w = 0; w < (int)(blockDim.x >> 6); w++) {
            const uint32_t sw = wsum[w];
            run += w < wave ? sw : 0u;
            total += sw;
            gmax = max(gmax, wsum4[0][w]);
        }
        for (int i = 0; i < per; i++) {
            if (t0 + i >= T) break;
            const uint32_t v = fs.tile_count[t0 + i];
            if (t0 + i >= t_lo && t0 + i < t_lo + TB) cursor[t0 + i - t_lo] = run;  // (this workgroup's band of tiles)
            if (blockIdx.x == 0) {  // what gsr_tile_scan_kernel writes
                fs.ranges[t0 + i] = make_uint2(run, run + v);
                fs.tile_work[t0 + i] = 0u;
                fs.sorted_len[t0 + i] = v;
                fs.need_full[t0 + i] = 0u;
            }
            run += v;
        }
        if (blockIdx.x == 0) {
            if (threadIdx.x == 0) chunk_first = 0u;  // (borrowed: reset below before its real use)
            __syncthreads();
            if (fs.occ_drop && (int)threadIdx.x < nchunks) { const uint32_t d = fs.occ_drop[threadIdx.x]; if (d) atomicAdd(&chunk_first, d); }
            __syncthreads();
            if (threadIdx.x == 0) {
                const uint32_t dropped = chunk_first;
                fs.info[0] = total; fs.info[1] = gmax; fs.info[2] = dropped; fs.info[3] = 0u;  // ([3]: see gsr_tile_scan_kernel)
                if (fs.info_host) { fs.info_host[0] = total; fs.info_host[1] = gmax; fs.info_host[2] = dropped; }
            }
        }
        __syncthreads();
    }
    if (!GLOBAL) {
        // cursor = tile start + this chunk's offset inside the tile; eight entries per thread per trip with all loads
        // issued before the first LDS store (one memory round trip per trip instead of one per entry)
        const uint32_t* row = table + (size_t)chunk * T;
        const uint32_t* nrow = row + T;  // the next chunk's offsets (the last chunk ends at the tile's end)
        const bool last_chunk = chunk + 1 >= nchunks;
        for (int t0 = threadIdx.x; t0 < TB; t0 += blockDim.x * 8) {
            uint32_t a[8], b[8], c[8];
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const int tl = t0 + k * (int)blockDim.x, t = t_lo + tl;  // local / global tile index
                uint2 rg = make_uint2(0u, 0u);
                if (tl < TB) {
                    if (fused) { rg.x = cursor[tl]; rg.y = rg.x + fs.tile_count[t]; }  // (read before the stores below overwrite it)
                    else rg = ranges[t];
                }
                a[k] = rg.x;
                b[k] = tl < TB ? row[t] : 0u;
                c[k] = !may_stage ? 0u : last_chunk ? rg.y - rg.x : (tl < TB ? nrow[t] : 0u);
            }
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const int tl = t0 + k * (int)blockDim.x;
                if (tl < TB) {
                    if (may_stage) { gbase[tl] = a[k] + b[k]; loff[tl] = c[k] - b[k]; }  // loff: count for now, offset after the scan
                    else cursor[tl] = a[k] + b[k];
                }
            }
        }
    }
    if (threadIdx.x == 0) chunk_first = 0u;
    __syncthreads();
    if (may_stage) {
        // exclusive scan of the (chunk, band)'s per-tile counts -> local offsets (thread i owns ceil(TB / blockDim) consecutive tiles);
        // the total decides whether the keys fit the staging buffer
        const int per = (TB + (int)blockDim.x - 1) / (int)blockDim.x, t0 = (int)threadIdx.x * per;
        uint32_t sum = 0;
        for (int i = 0; i < per; i++) sum += t0 + i < TB ? loff[t0 + i] : 0u;
        const uint32_t incl = gsr_wave_scan_add(sum);
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        uint32_t run = incl - sum, tot = 0;
        for (int w = 0; w < (int)(blockDim.x >> 6); w++) { const uint32_t sw = wsum[w]; tot += sw; run += w < wave ? sw : 0u; }
        if (threadIdx.x == 0) band_total = tot;
        staged = tot <= stage_cap;  // block-uniform
        for (int i = 0; i < per; i++) {
            if (t0 + i >= TB) break;
            const uint32_t v = loff[t0 + i];
            loff[t0 + i] = run;
            cursor[t0 + i] = staged ? run : gbase[t0 + i];  // the staging cursor of the tile, or (too many keys) its global cursor
            run += v;
        }
        __syncthreads();
    }
    // first gradient slot of the chunk = instances of all earlier chunks
    {
        uint32_t pv = 0u;
        for (int i = threadIdx.x; i < chunk; i += blockDim.x) pv += chunk_sum[i];
        pv = gsr_wave_scan_add(pv);
        if (lane == 63 && pv != 0u) atomicAdd(&chunk_first, pv);
    }
    __syncthreads();
    uint32_t carry = chunk_first;
    int lo, hi;
    gsr_chunk_bounds(P, nchunks, chunk, lo, hi);
    constexpr int U = 4;
    for (int tb = lo; tb < hi; tb += blockDim.x * U) {  // block-uniform trip count (barriers inside)
        const int gb = tb + (int)threadIdx.x;
        uint2 rcs[U];
        u64 mks[U];
        uint32_t dks[U];
#pragma unroll
        for (int k = 0; k < U; k++) {
            const int g = gb + k * blockDim.x;
            const bool v = g < hi;
            rcs[k] = v ? rect[g] : make_uint2(0u, 0u);
            mks[k] = v ? tmask[g] : 0ull;
            dks[k] = v ? depthkey[g] : 0u;
        }
        // offsets[] = exclusive scan of the per-Gaussian instance counts in index order: the Gaussian's first gradient slot
        // (blend backward, gauss_bwd).  Same count as the enumeration below by construction.  The four sub-trips' block scans
        // share ONE pair of barriers (wave totals of all four in LDS at once) instead of taking a pair each.
        if (offsets && band == 0) {  // (block-uniform; NULL = inference forward: nobody will ask for gradient slots)
            uint32_t cnt[U], incl[U];
#pragma unroll
            for (int k = 0; k < U; k++) {
                cnt[k] = gsr_rect_count(rcs[k], mks[k]);
                incl[k] = gsr_wave_scan_add(cnt[k]);
                if (lane == 63) wsum4[k][wave] = incl[k];
            }
            __syncthreads();
#pragma unroll
            for (int k = 0; k < U; k++) {
                const int g = gb + k * blockDim.x;
                uint32_t before = 0, tot = 0;
                for (int w = 0; w < (int)(blockDim.x >> 6); w++) {
                    const uint32_t sw = wsum4[k][w];
                    tot += sw;
                    before += w < wave ? sw : 0u;
                }
                if (g < hi) offsets[g] = carry + before + incl[k] - cnt[k];
                carry += tot;
            }
            __syncthreads();  // (the next trip rewrites the wave totals)
        }
#pragma unroll
        for (int k = 0; k < U; k++) {
            const int g = gb + k * blockDim.x;
            const int g_lane0 = g - lane;  // lanes of a wave hold consecutive Gaussians
            // the instance's 64-bit sort key (depth bits, id) goes straight into its tile's segment: a scattered store
            // costs one 32-byte sector whether it carries 4 or 8 bytes, and the sort then reads its keys coalesced
            // instead of gathering 4-byte depths
            uint2 rcb = rcs[k];
            u64 mkb = mks[k];
            if (BANDED) {  // this band's rows of the rectangle; the survivor mask follows (positions >= 64 always survive)
                const int ry0 = (int)(rcb.y & 0xffff), ry1 = (int)(rcb.y >> 16), rw = (int)(rcb.x >> 16) - (int)(rcb.x & 0xffff);
                const int cy0 = max(ry0, by0), cy1 = min(ry1, by1);
                if (cy0 >= cy1 || rw <= 0) { rcb = make_uint2(0u, 0u); mkb = 0ull; }
                else {
                    const long long sh = (long long)(cy0 - ry0) * rw;
                    mkb = sh >= 64 ? ~0ull : sh == 0 ? mkb : ((mkb >> sh) | (~0ull << (64 - sh)));
                    rcb.y = (uint32_t)cy0 | ((uint32_t)cy1 << 16);
                }
            }
            gsr_wave_for_each_instance(rcb, mkb, dks[k], heads, [&](int owner, int x, int y, uint32_t odk) {
                const uint32_t slot = atomicAdd(&cursor[y * gx + x - t_lo], 1u);
#ifdef GSR_SCATTER_NOSTORE  // diagnostic: everything but the key stores (are the 32-byte-sector writes what the kernel waits for?)
                if (slot == 0xffffffffu) seg_keys[0] = odk;
#else
                const u64 key = ((u64)odk << 32) | (uint32_t)(g_lane0 + owner);
                if (staged) {
                    if (slot < stage_cap) { skey[slot] = key; stile[slot] = (uint16_t)(y * gx + x - t_lo); }  // (always true: the counts are exact)
                } else if (slot < capacity) seg_keys[slot] = key;  // capacity < R only in a speculative launch that is redone
#endif
            });
        }
    }
    if (staged) {
        __syncthreads();
        const uint32_t chunk_total = band_total;
        for (uint32_t i = threadIdx.x; i < chunk_total; i += blockDim.x) {
            const uint32_t t = stile[i];
            const uint32_t dst = gbase[t] + (i - loff[t]);
            if (dst < capacity) seg_keys[dst] = skey[i];
        }
    }
}

__global__ void __launch_bounds__(256) gsr_cursor_init_kernel(int T, const uint2* __restrict__ ranges, uint32_t* __restrict__ cursor)
{
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t < T) cursor[t] = ranges[t].x;
}

// ---------------------------------------------------------------------------------------------
// Per-tile sort.  Bitonic network in the "flip" formulation: every compare-exchange moves the
// smaller key to the lower index, so indices >= n behave as +infinity padding and are simply
// skipped -- no padding writes, any n.
// ---------------------------------------------------------------------------------------------
template <typename KeyPtr>
__device__ __forceinline__ void gsr_bitonic(KeyPtr k, const uint32_t n, const int nthreads)
{
    uint32_t lm = 0;  // log2 of the padded length m
    while ((1u << lm) < n) lm++;
    const uint32_t npairs = (1u << lm) >> 1;
    for (uint32_t ls = 1; ls <= lm; ls++) {  // merge blocks of size 2^ls
        const uint32_t lh = ls - 1, half = 1u << lh;
        for (uint32_t t = threadIdx.x; t < npairs; t += nthreads) {
            const uint32_t blk = (t >> lh) << ls, l = t & (half - 1);
            const uint32_t i = blk + l, j = blk + (2u << lh) - 1 - l;  // "flip": i <-> mirror
            if (j < n) {
                const u64 a = k[i], b = k[j];
                if (a > b) { k[i] = b; k[j] = a; }
            }
        }
        __syncthreads();
        for (int lst = (int)lh - 1; lst >= 0; lst--) {  // half-cleaners, stride 2^lst
            const uint32_t stride = 1u << lst;
            for (uint32_t t = threadIdx.x; t < npairs; t += nthreads) {
                const uint32_t i = ((t >> lst) << (lst + 1)) + (t & (stride - 1)), j = i + stride;
                if (j < n) {
                    const u64 a = k[i], b = k[j];
                    if (a > b) { k[i] = b; k[j] = a; }
                }
            }
            __syncthreads();
        }
    }
}

// ---- LDS version with fused strides: 8 keys per thread stay in registers for three consecutive compare-exchange
// distances, so a 2048-key list needs 29 LDS round trips (+ barriers) instead of 66.  Same network, same result.
#define GSR_KEY_INF 0xffffffffffffffffull
// LDS index padding: one spare key after every 16.  The small-stride rounds give each thread 8 keys that are 1, 2 or
// 4 apart, so the lanes of a wave are 64/128/256 bytes apart -- without padding they all fall on the same few banks
// (up to 32-way conflicts); with it consecutive groups rotate through the banks.
#define GSR_PAD(i) ((i) + ((i) >> 4))
__device__ __forceinline__ void gsr_cex(u64& a, u64& b)
{
    const u64 lo = a < b ? a : b, hi = a < b ? b : a;
    a = lo; b = hi;
}
// half-cleaners at strides q*2^(L-1) ... q on the 2^L keys {base + t*q}; q = 2^lq
template <int L>
__device__ __forceinline__ void gsr_fused_round(u64* k, uint32_t n, uint32_t m, int lq, int nthreads, uint32_t vt)
{
    constexpr int E = 1 << L;
    // groups whose keys all lie in the virtual +inf padding (index >= n) have nothing to do: whole waves drop out
    const uint32_t span = (uint32_t)E << lq, qm = (1u << lq) - 1u;
    const uint32_t groups = min(m >> L, ((n + span - 1u) / span) << lq);
    for (uint32_t g = vt; g < groups; g += nthreads) {
        const uint32_t base = ((g >> lq) << (L + lq)) + (g & qm);
        u64 v[E];
#pragma unroll
        for (int t = 0; t < E; t++) {
            const uint32_t i = base + ((uint32_t)t << lq);
            v[t] = i < n ? k[GSR_PAD(i)] : GSR_KEY_INF;
        }
#pragma unroll
        for (int h = E >> 1; h >= 1; h >>= 1)
#pragma unroll
            for (int t = 0; t < E; t++)
                if ((t & h) == 0) gsr_cex(v[t], v[t + h]);
#pragma unroll
        for (int t = 0; t < E; t++) {
            const uint32_t i = base + ((uint32_t)t << lq);
            if (i < n) k[GSR_PAD(i)] = v[t];
        }
    }
}

__device__ __forceinline__ void gsr_sort_lds_fused(u64* k, const uint32_t n, const int nthreads)
{
    // Work is dealt to wavefronts in 64-thread quanta from the low indices up, so with virtual padding the high
    // wavefronts idle.  Wave w of every workgroup tends to sit on SIMD w: rotate the wave <-> quantum map per
    // workgroup so that the idle slots of the CU's resident workgroups fall on different SIMDs.
    const uint32_t vt = (threadIdx.x + 64u * ((blockIdx.x >> 3) & 3u)) & (uint32_t)(nthreads - 1);

    uint32_t lm = 3;
    while ((1u << lm) < n) lm++;
    const uint32_t m = 1u << lm;
    // phase 0: every run of 8 consecutive keys sorted in registers (the merges of size 2, 4, 8)
    for (uint32_t c = vt; c < ((n + 7u) >> 3); c += nthreads) {
        u64 v[8];
#pragma unroll
        for (int t = 0; t < 8; t++) v[t] = 8 * c + t < n ? k[GSR_PAD(8 * c + t)] : GSR_KEY_INF;
        gsr_cex(v[0], v[1]); gsr_cex(v[2], v[3]); gsr_cex(v[4], v[5]); gsr_cex(v[6], v[7]);   // size 2
        gsr_cex(v[0], v[3]); gsr_cex(v[1], v[2]); gsr_cex(v[4], v[7]); gsr_cex(v[5], v[6]);   // size 4: flip
        gsr_cex(v[0], v[1]); gsr_cex(v[2], v[3]); gsr_cex(v[4], v[5]); gsr_cex(v[6], v[7]);   //         stride 1
        gsr_cex(v[0], v[7]); gsr_cex(v[1], v[6]); gsr_cex(v[2], v[5]); gsr_cex(v[3], v[4]);   // size 8: flip
        gsr_cex(v[0], v[2]); gsr_cex(v[1], v[3]); gsr_cex(v[4], v[6]); gsr_cex(v[5], v[7]);   //         stride 2
        gsr_cex(v[0], v[1]); gsr_cex(v[2], v[3]); gsr_cex(v[4], v[5]); gsr_cex(v[6], v[7]);   //         stride 1
#pragma unroll
        for (int t = 0; t < 8; t++)
            if (8 * c + t < n) k[GSR_PAD(8 * c + t)] = v[t];
    }
    __syncthreads();
    for (uint32_t ls = 4; ls <= lm; ls++) {
        const uint32_t lh = ls - 1, half = 1u << lh;
        const uint32_t npairs = min(m >> 1, ((n + (1u << ls) - 1u) >> ls) << lh);  // blocks that start below n
        for (uint32_t t = vt; t < npairs; t += nthreads) {  // flip: i <-> mirror inside the 2^ls block
            const uint32_t blk = (t >> lh) << ls, l = t & (half - 1);
            const uint32_t i = blk + l, j = blk + (2u << lh) - 1 - l;
            if (j < n) {
                const u64 a = k[GSR_PAD(i)], b = k[GSR_PAD(j)];
                if (a > b) { k[GSR_PAD(i)] = b; k[GSR_PAD(j)] = a; }
            }
        }
        __syncthreads();
        for (int hs = (int)ls - 2; hs >= 0; hs -= 3) {  // half-cleaner strides 2^hs ... 1, three per round
            if (hs >= 2) gsr_fused_round<3>(k, n, m, hs - 2, nthreads, vt);
            else if (hs == 1) gsr_fused_round<2>(k, n, m, 0, nthreads, vt);
            else gsr_fused_round<1>(k, n, m, 0, nthreads, vt);
            __syncthreads();
        }
    }
}

// Counting sort of one tile list held in registers (n <= NT * KPT keys, thread t owns keys t, t + NT, ...), two levels, O(n)
// instead of the bitonic network's n log^2 n / 2 compare-exchanges of 5 VALU operations each (2048 keys: 66 stages).
// Level 1: the 64-bit keys (depth bits, id) are spread over 4 * NT equal-width buckets of their own range (LDS atomics, one scan).
// Level 2: a bucket of c keys is cut into 2^floor(log2 c) equal-width sub-buckets (the next key bits), numbered inside the bucket's
// own slice [start, start + c) -- so ONE n-entry counter array serves all buckets (16-bit counters packed into offs[], which is free
// once every key knows its sub-bucket) and its exclusive scan is the placement.  A key's position inside its sub-bucket -- one or two
// keys when the cluster is smooth at that resolution -- is its rank by counting: independent LDS reads, no serial chain.
// Exact for any input; its speed no longer depends on the keys being spread out.  Depth keys are not, in general: surfaces, and the
// layers of an anchor grid, put dozens of buckets of 20 - 70 keys into every list of the large-splat stand-in frame
// (tools/sort_buckets_probe.py).  Rounds 1 - 3 finished the level-1 buckets with an insertion sort by the owning thread (serial,
// dependent LDS round trips: fine for 1 - 2 keys, the slowest thread's 10 - 13-key bucket set the pace of every tile even on the
// uniform bench scene) and sent clustered tiles to the network: tile sort 32 -> 22 us on the bench scene, 108 -> 66 us at config 4,
// 134 -> 35 us on the large-splat frame (round 4).  Only keys that are equal in every bit the two levels look at (thousands of
// Gaussians on exact depth planes: the ids differ) defeat it; the sum of the squared sub-bucket counts measures that, and beyond
// 24 n the tile falls back to the fused network.  Returns false in that case, with the keys stored at k[GSR_PAD(i)] for it; true
// with the sorted keys at k[i].
// NT threads, KPT keys per thread in registers (n <= NT * KPT), 4 * NT buckets (thread t owns buckets 4t .. 4t+3).
template <int NT, int KPT>
__device__ __forceinline__ bool gsr_sort_buckets(const u64 (&v)[KPT], const uint32_t n, u64* k, uint32_t* offs /*[4 NT]*/,
                                                 u64* red /*[2 NT / 64]*/, uint32_t* wtot /*[NT / 64 + 2]*/)
{
    constexpr int NW = NT / 64, NB = 4 * NT, LOGNB = NT == 256 ? 10 : 12;
    static_assert(NT == 256 || NT == 1024, "bucket count = 4 NT must match LOGNB");
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    u64 mn = ~0ull, mx = 0ull;
#pragma unroll
    for (int j = 0; j < KPT; j++)
        if ((uint32_t)(t + NT * j) < n) { mn = min(mn, v[j]); mx = max(mx, v[j]); }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        mn = min(mn, (u64)__shfl_xor((unsigned long long)mn, d, 64));
        mx = max(mx, (u64)__shfl_xor((unsigned long long)mx, d, 64));
    }
    if (lane == 0) { red[wave] = mn; red[NW + wave] = mx; }
    for (int i = t; i < NB; i += NT) offs[i] = 0u;
    __syncthreads();
    u64 kmin = red[0], kmax = red[NW];
#pragma unroll
    for (int w = 1; w < NW; w++) { kmin = min(kmin, red[w]); kmax = max(kmax, red[NW + w]); }
    const u64 span = kmax - kmin;
    const int shift = span < (u64)NB ? 0 : (64 - __builtin_clzll(span)) - LOGNB;  // (key - kmin) >> shift < NB
    uint32_t b[KPT];
#pragma unroll
    for (int j = 0; j < KPT; j++) {
        b[j] = (uint32_t)((v[j] - kmin) >> shift);
        if ((uint32_t)(t + NT * j) < n) atomicAdd(&offs[b[j]], 1u);
    }
    __syncthreads();
    // exclusive scan of the bucket counts (thread t owns buckets 4t .. 4t+3)
    const uint32_t h0 = offs[4 * t], h1 = offs[4 * t + 1], h2 = offs[4 * t + 2], h3 = offs[4 * t + 3];
    const uint32_t mine = h0 + h1 + h2 + h3;
    const uint32_t incl = gsr_wave_scan_add(mine);
    if (lane == 63) { wtot[wave] = incl; wtot[NW] = 0u; wtot[NW + 1] = 0u; }
    __syncthreads();
    uint32_t run = incl - mine;
    for (int w = 0; w < wave; w++) run += wtot[w];
    offs[4 * t] = run; offs[4 * t + 1] = run + h0; offs[4 * t + 2] = run + h0 + h1; offs[4 * t + 3] = run + h0 + h1 + h2;
    __syncthreads();
    // ---- second level: 16-bit sub-bucket counters, two per word of offs[] (free once every key knows its sub-bucket) ----
    uint32_t idx[KPT];
    const bool fits2 = n <= 2u * (uint32_t)NB;  // (block-uniform; only the 1024-thread class can exceed it)
#pragma unroll
    for (int j = 0; j < KPT; j++) {
        idx[j] = 0u;
        if ((uint32_t)(t + NT * j) < n) {
            const uint32_t s0 = offs[b[j]], c = (b[j] + 1u < (uint32_t)NB ? offs[b[j] + 1u] : n) - s0;
            const int L = min(31 - __builtin_clz(c), shift);  // c >= 1 (this key); 2^L <= c sub-buckets, never finer than the key bits left
            idx[j] = s0 + (uint32_t)(((v[j] - kmin) >> (shift - L)) & (u64)((1u << L) - 1u));
        }
    }
    __syncthreads();
    for (int i = t; i < NB; i += NT) offs[i] = 0u;
    __syncthreads();
    if (fits2) {
#pragma unroll
        for (int j = 0; j < KPT; j++)
            if ((uint32_t)(t + NT * j) < n) atomicAdd(&offs[idx[j] >> 1], 1u << ((idx[j] & 1u) * 16u));
    }
    __syncthreads();
    {   // exclusive scan of the counters in place (thread t owns words 4t .. 4t+3 = entries 8t .. 8t+7), and the sum of their squares
        uint32_t c2[8], loc = 0u, sq = 0u;
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const uint32_t w = offs[4 * t + e];
            c2[2 * e] = w & 0xffffu; c2[2 * e + 1] = w >> 16;
        }
#pragma unroll
        for (int e = 0; e < 8; e++) { loc += c2[e]; sq += c2[e] * c2[e]; }
        const uint32_t incl2 = gsr_wave_scan_add(loc);
        const uint32_t sqw = gsr_wave_scan_add(sq);
        if (lane == 63) { wtot[wave] = incl2; atomicAdd(&wtot[NW + 1], sqw); }
        __syncthreads();
        if (!fits2 || wtot[NW + 1] > 24u * n) {  // keys equal in every bit both levels look at: the network (block-uniform)
#pragma unroll
            for (int j = 0; j < KPT; j++)
                if ((uint32_t)(t + NT * j) < n) k[GSR_PAD(t + NT * j)] = v[j];
            __syncthreads();
            return false;
        }
        uint32_t run2 = incl2 - loc;
        for (int w = 0; w < wave; w++) run2 += wtot[w];
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const uint32_t lo16 = run2, hi16 = run2 + c2[2 * e];
            offs[4 * t + e] = lo16 | (hi16 << 16);
            run2 = hi16 + c2[2 * e + 1];
        }
    }
    __syncthreads();
    // placement: a counter is the next free position of its sub-bucket's slice (and the slice's end once every key is placed)
#pragma unroll
    for (int j = 0; j < KPT; j++)
        if ((uint32_t)(t + NT * j) < n) {
            const uint32_t sh = (idx[j] & 1u) * 16u;
            k[(atomicAdd(&offs[idx[j] >> 1], 1u << sh) >> sh) & 0xffffu] = v[j];
        }
    __syncthreads();
    // a key's place inside its sub-bucket (one key for most): its rank by counting; a slice starts where its predecessor ends
#pragma unroll
    for (int j = 0; j < KPT; j++) {
        uint32_t r = 0u;
        if ((uint32_t)(t + NT * j) < n) {
            const uint32_t en = (offs[idx[j] >> 1] >> ((idx[j] & 1u) * 16u)) & 0xffffu;
            const uint32_t st = idx[j] ? (offs[(idx[j] - 1u) >> 1] >> (((idx[j] - 1u) & 1u) * 16u)) & 0xffffu : 0u;
            r = st;
            for (uint32_t i = st; i < en; i++) r += k[i] < v[j] ? 1u : 0u;
        }
        idx[j] = r;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < KPT; j++)
        if ((uint32_t)(t + NT * j) < n) k[idx[j]] = v[j];
    __syncthreads();
    return true;
}

// LDS variant for lo < n <= hi (dynamic LDS = 8 * GSR_PAD(hi) bytes).  NT = 256 threads (8 keys each) for the lists up
// to 2048, 1024 threads (16 keys each) for the class up to 16384.
template <int NT>
__global__ void __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(NT == 256 ? 8 : 4))) gsr_tile_sort_lds_kernel(const uint2* __restrict__ ranges,
                                                               const u64* __restrict__ seg_keys,
                                                               uint32_t* __restrict__ point_list,
                                                               uint8_t* __restrict__ slot_written, uint32_t lo,
                                                               uint32_t hi, uint32_t fits, uint32_t capacity,
                                                               const uint32_t* __restrict__ only_flagged,
                                                               uint32_t* __restrict__ sorted_len)
{
    constexpr int KPT = NT == 256 ? 8 : 16;
    extern __shared__ __attribute__((aligned(16))) u64 keys[];
    __shared__ uint32_t offs[4 * NT];
    __shared__ u64 red[2 * NT / 64];
    __shared__ uint32_t wtot[NT / 64 + 2];
    if (only_flagged && !only_flagged[blockIdx.x]) return;  // fix-up pass: only the tiles whose sorted prefix ran out
    const uint2 rg = ranges[blockIdx.x];
    const uint32_t n = rg.y - rg.x;
    // n > fits: the LDS was provisioned from a stale hint of the longest list (speculative launch); the host sees the
    // true maximum after the fact and redoes stage 2
    if (n <= lo || n > hi || n > fits || rg.y > capacity) return;
    // the tile ranges partition [0, R): each block clears its share of the written-slot flags for the backward
    if (slot_written) for (uint32_t i = threadIdx.x; i < n; i += NT) slot_written[rg.x + i] = 0;
    // the scatter left the tile's 64-bit keys (depth bits, Gaussian id) in its segment of seg_keys
    if (n <= (uint32_t)(NT * KPT)) {
        u64 v[KPT];
#pragma unroll
        for (int j = 0; j < KPT; j++) {
            const uint32_t i = threadIdx.x + (uint32_t)NT * j;
            v[j] = i < n ? seg_keys[rg.x + i] : 0ull;
        }
        if (gsr_sort_buckets<NT, KPT>(v, n, keys, offs, red, wtot)) {
            for (uint32_t i = threadIdx.x; i < n; i += NT) point_list[rg.x + i] = (uint32_t)keys[i];
            if (sorted_len && threadIdx.x == 0) sorted_len[blockIdx.x] = n;
            return;
        }
    } else {
        for (uint32_t i = threadIdx.x; i < n; i += NT) keys[GSR_PAD(i)] = seg_keys[rg.x + i];
        __syncthreads();
    }
    gsr_sort_lds_fused(keys, n, NT);
    for (uint32_t i = threadIdx.x; i < n; i += NT) point_list[rg.x + i] = (uint32_t)keys[GSR_PAD(i)];
    if (sorted_len && threadIdx.x == 0) sorted_len[blockIdx.x] = n;
}

// Partial sort of a long list (n > GSR_NEAR_CAP).  The blend stops at the depth where the tile's pixels saturate --
// typically a small fraction of a long list -- so only the nearest instances need to be in order: three coalesced
// passes over the tile's keys find their range, histogram them into 1024 equal-width key buckets and split the list
// at the last bucket boundary that keeps <= GSR_NEAR_CAP keys in front: those are sorted in LDS and written first
// (sorted_len[tile] = their count), the others follow unsorted.  If the forward runs off the sorted prefix with pixels
// still blending it says so (need_full[tile]) and the tile is redone after a full sort (gsr_launch_sort_fixup).
// Fixed LDS (21 KiB) whatever the list length; a full bitonic sort of a 14 000-entry list needs 123 KiB and ~9x the
// compare-exchanges.
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(6))) gsr_tile_sort_near_kernel(const uint2* __restrict__ ranges,
                                                                 const u64* __restrict__ seg_keys,
                                                                 uint32_t* __restrict__ point_list,
                                                                 uint8_t* __restrict__ slot_written,
                                                                 uint32_t* __restrict__ sorted_len, uint32_t capacity)
{
    __shared__ __attribute__((aligned(16))) u64 keys[GSR_PAD(GSR_NEAR_CAP) + 1];
    __shared__ uint32_t hist[1024];
    __shared__ u64 red[8];
    __shared__ uint32_t s_pick, s_near, s_far;
    const uint2 rg = ranges[blockIdx.x];
    const uint32_t n = rg.y - rg.x;
    if (rg.y > capacity) return;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const u64* src = seg_keys + rg.x;
    if (slot_written) for (uint32_t i = t; i < n; i += 256) slot_written[rg.x + i] = 0;
    if (n <= GSR_NEAR_CAP) {
        // A list short enough for the complete LDS sort is done right here, in the same launch (this kernel's LDS covers it):
        // as a launch of their own behind this one these few tiles were a 90 us tail of single workgroups on a scene whose
        // other lists are long; here they overlap with the long lists' passes.
        u64 v[8];
#pragma unroll
        for (int j = 0; j < 8; j++) v[j] = (uint32_t)(t + 256 * j) < n ? src[t + 256 * j] : 0ull;
        __shared__ uint32_t wtot0[6];
        if (gsr_sort_buckets<256, 8>(v, n, keys, hist, red, wtot0)) {
            for (uint32_t i = t; i < n; i += 256) point_list[rg.x + i] = (uint32_t)keys[i];
        } else {
            if (n > 1) gsr_sort_lds_fused(keys, n, 256);
            for (uint32_t i = t; i < n; i += 256) point_list[rg.x + i] = (uint32_t)keys[GSR_PAD(i)];
        }
        return;  // sorted_len[tile] = n already (tile scan)
    }
    // pass A: key range
    u64 mn = ~0ull, mx = 0ull;
    for (uint32_t i = t; i < n; i += 256) { const u64 k = src[i]; mn = min(mn, k); mx = max(mx, k); }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        mn = min(mn, (u64)__shfl_xor((unsigned long long)mn, d, 64));
        mx = max(mx, (u64)__shfl_xor((unsigned long long)mx, d, 64));
    }
    if (lane == 0) { red[wave] = mn; red[4 + wave] = mx; }
    for (int i = t; i < 1024; i += 256) hist[i] = 0u;
    if (t == 0) { s_pick = 0u; s_near = 0u; s_far = 0u; }
    __syncthreads();
    const u64 kmin = min(min(red[0], red[1]), min(red[2], red[3])), kmax = max(max(red[4], red[5]), max(red[6], red[7]));
    const u64 span = kmax - kmin;
    const int shift = span < 1024ull ? 0 : (64 - __builtin_clzll(span)) - 10;  // (k - kmin) >> shift < 1024
    // pass B: histogram of the key buckets
    for (uint32_t i = t; i < n; i += 256) atomicAdd(&hist[(uint32_t)((src[i] - kmin) >> shift)], 1u);
    __syncthreads();
    // the last bucket boundary with <= GSR_NEAR_CAP keys in front of it: thread t owns buckets 4t .. 4t+3
    {
        const uint32_t h0 = hist[4 * t], h1 = hist[4 * t + 1], h2 = hist[4 * t + 2], h3 = hist[4 * t + 3];
        const uint32_t mine = h0 + h1 + h2 + h3;
        const uint32_t incl = gsr_wave_scan_add(mine);
        __shared__ uint32_t wtot[4];
        if (lane == 63) wtot[wave] = incl;
        __syncthreads();
        uint32_t run = incl - mine;
        for (int w = 0; w < wave; w++) run += wtot[w];
        const uint32_t hh[4] = { h0, h1, h2, h3 };
        uint32_t best = 0u;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            run += hh[j];
            if (run <= GSR_NEAR_CAP) best = ((uint32_t)(4 * t + j + 1) << 12) | run;  // (bucket + 1, keys up to it)
        }
        if (best) atomicMax(&s_pick, best);
    }
    __syncthreads();
    const uint32_t nb = s_pick >> 12, m = s_pick & 0xfffu;  // buckets < nb form the near set of m keys (0: none fits)
    // pass C: near keys into LDS, the others behind the sorted prefix (any order); one LDS atomic per wave and side
    for (uint32_t i0 = 0; i0 < n; i0 += 256) {
        const uint32_t i = i0 + t;
        const u64 k = i < n ? src[i] : 0ull;
        const bool near = i < n && (uint32_t)((k - kmin) >> shift) < nb, far = i < n && !near;
        const unsigned long long bn = __ballot(near), bf = __ballot(far);
        uint32_t basen = 0u, basef = 0u;
        if (lane == 0) { basen = atomicAdd(&s_near, (uint32_t)__popcll(bn)); basef = atomicAdd(&s_far, (uint32_t)__popcll(bf)); }
        basen = (uint32_t)__shfl((int)basen, 0, 64); basef = (uint32_t)__shfl((int)basef, 0, 64);
        if (near) keys[GSR_PAD(basen + __builtin_amdgcn_mbcnt_hi((uint32_t)(bn >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bn, 0u)))] = k;
        if (far) point_list[rg.x + m + basef + __builtin_amdgcn_mbcnt_hi((uint32_t)(bf >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bf, 0u))] = (uint32_t)k;
    }
    __syncthreads();
    // the near set, sorted: bucket sort from registers like the short lists, the network if its keys are clustered
    {
        u64 v[8];
#pragma unroll
        for (int j = 0; j < 8; j++) v[j] = (uint32_t)(t + 256 * j) < m ? keys[GSR_PAD(t + 256 * j)] : 0ull;
        __syncthreads();
        __shared__ uint32_t wtot2[6];
        if (gsr_sort_buckets<256, 8>(v, m, keys, hist, red, wtot2)) {  // m is block-uniform
            for (uint32_t i = t; i < m; i += 256) point_list[rg.x + i] = (uint32_t)keys[i];
        } else {
            if (m > 1) gsr_sort_lds_fused(keys, m, 256);
            for (uint32_t i = t; i < m; i += 256) point_list[rg.x + i] = (uint32_t)keys[GSR_PAD(i)];
        }
    }
    if (t == 0) sorted_len[blockIdx.x] = m;
}

// Global-memory variant for lists longer than the LDS capacity (degenerate inputs: e.g. a tiny
// image with a huge cloud).  Same network, in place on seg_keys, one 1024-thread block per tile.
__global__ void __launch_bounds__(1024) gsr_tile_sort_global_kernel(const uint2* __restrict__ ranges,
                                                                    u64* __restrict__ seg_keys,
                                                                    uint32_t* __restrict__ point_list,
                                                                    uint8_t* __restrict__ slot_written, uint32_t lo,
                                                                    uint32_t capacity, const uint32_t* __restrict__ only_flagged,
                                                                    uint32_t* __restrict__ sorted_len)
{
    if (only_flagged && !only_flagged[blockIdx.x]) return;
    const uint2 rg = ranges[blockIdx.x];
    const uint32_t n = rg.y - rg.x;
    if (n <= lo || rg.y > capacity) return;
    if (slot_written) for (uint32_t i = threadIdx.x; i < n; i += 1024) slot_written[rg.x + i] = 0;
    u64* k = seg_keys + rg.x;  // the scatter's keys, sorted in place
    __syncthreads();
    gsr_bitonic(k, n, 1024);
    for (uint32_t i = threadIdx.x; i < n; i += 1024) point_list[rg.x + i] = (uint32_t)k[i];
    if (sorted_len && threadIdx.x == 0) sorted_len[blockIdx.x] = n;
}

// ---------------------------------------------------------------------------------------------
// Host launchers
// ---------------------------------------------------------------------------------------------
// Dynamic LDS beyond 64 KiB needs a one-time opt-in per kernel and device (kept out of the per-call path: a
// hipFuncSetAttribute between the stage-1 read-back and the stage-2 launches is exposed GPU idle time).
static hipError_t gsr_allow_big_lds()
{
    static thread_local uint64_t done_mask = 0;  // one bit per device: switching devices re-issues nothing
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev >= 0 && dev < 64 && ((done_mask >> dev) & 1)) return hipSuccess;
    const int big = 160 * 1024 - 8192;  // static LDS: hist / scatter 2 KiB, tile sort 4.2 KiB (bucket offsets)
    e = hipFuncSetAttribute((const void*)gsr_tile_hist_kernel<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, big);
    // (the occlusion variant has 8 KiB more static LDS -- the per-lane dropped-tile masks -- and needs 5 B per tile up to GSR_OCC_MAX_TILES)
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)gsr_tile_hist_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, big - 16384);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)gsr_scatter_kernel<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, big);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)gsr_scatter_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, big);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)gsr_tile_sort_lds_kernel<1024>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 17408);  // 16.9 KiB static: 4096 bucket offsets + scan scratch
    if (e == hipSuccess && dev >= 0 && dev < 64) done_mask |= 1ull << dev;
    return e;
}

hipError_t gsr_launch_count(int P, int T, int gx, const GsrGeom& geom, const GsrImage& image, uint32_t* info_host_mapped,
                            bool defer_tile_scan, bool occlusion_cut, const uint32_t* walk_depths, bool* ordered, hipStream_t stream)
{
    const int nchunks = gsr_num_chunks(P);
    if (ordered) *ordered = false;
    hipError_t e;
    const GsrOcclusion oc = { image.occ_cut, geom.depthkey, geom.tmask, geom.tiles, geom.rec, image.occ_drop };
    if (T > GSR_MAX_TILES_LDS) {
        // fallback for very large images: global counters (see gsr_tile_hist_kernel<true>)
        e = hipMemsetAsync(image.tile_count, 0, (size_t)T * 4, stream);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((gsr_tile_hist_kernel<true, false>), dim3(nchunks), dim3(GSR_HIST_THREADS), 0, stream, P, T, gx, nchunks,
                           geom.rect, geom.tmask, image.tile_count, geom.scan_sums, oc);
    } else {
        // (1) per-chunk tile histogram -> table, per-chunk instance totals
        e = gsr_allow_big_lds();
        if (e != hipSuccess) return e;
        if (occlusion_cut) {  // (api.hip enables it only up to GSR_OCC_MAX_TILES: the cut-off table sits behind the counters in LDS)
            hipLaunchKernelGGL(gsr_occ_cut_kernel, dim3((T + 3) / 4), dim3(256), 0, stream, T, image.occ_mass, image.occ_cut);
            hipLaunchKernelGGL((gsr_tile_hist_kernel<false, true>), dim3(nchunks), dim3(GSR_HIST_THREADS), gsr_align((size_t)T * 5), stream, P, T, gx,
                               nchunks, geom.rect, geom.tmask, image.table, geom.scan_sums, oc);
        } else
        hipLaunchKernelGGL((gsr_tile_hist_kernel<false, false>), dim3(nchunks), dim3(GSR_HIST_THREADS), (size_t)T * 4, stream, P, T, gx,
                           nchunks, geom.rect, geom.tmask, image.table, geom.scan_sums, oc);
        // (2) column scan -> per-(chunk, tile) offsets + tile totals
        const int nscan = (T + GSR_COLSCAN_TILES - 1) / GSR_COLSCAN_TILES, xt = gsr_xcd_tiles(T);
        const bool order = walk_depths != nullptr && ordered != nullptr && 4 * xt <= GSR_ORDER_MAX_SLOTS;  // + 8 workgroups: the forward's dispatch order
        hipLaunchKernelGGL(gsr_table_colscan_kernel, dim3(nscan + (order ? 8 : 0)), dim3(GSR_COLSCAN_TILES * GSR_COLSCAN_GROUPS), 0, stream, T, nchunks, image.table,
                           image.tile_count, image.tile_group, nscan, xt, walk_depths, image.qorder);
        if (order) *ordered = true;
    }
    // (3) tile scan -> ranges, info.  The one-call forward folds it into the scatter kernel (gsr_launch_scatter with
    // fused_info_host): one launch less on the critical path; the host then learns R when the scatter has started.
    if (!(defer_tile_scan && T <= GSR_MAX_TILES_LDS))
    {
        const int per = (T + 1023) / 1024;
#define GSR_TILE_SCAN(N)                                                                                                       \
        hipLaunchKernelGGL(gsr_tile_scan_kernel<N>, dim3(1), dim3(1024), 0, stream, T, image.tile_count, image.ranges, image.info, \
                           image.tile_work, image.sorted_len, image.need_full, info_host_mapped,                                   \
                           occlusion_cut ? (const uint32_t*)image.occ_drop : (const uint32_t*)nullptr, nchunks)
        if (per <= 4) GSR_TILE_SCAN(4);
        else if (per <= 12) GSR_TILE_SCAN(12);
        else if (per <= 36) GSR_TILE_SCAN(36);
        else GSR_TILE_SCAN(0);  // (images beyond the LDS tile limit: <= 2^24 tiles)
#undef GSR_TILE_SCAN
    }
    return hipGetLastError();
}

// Bands of tile rows per chunk in the scatter launch (1 = the plain form).  More than one when a chunk's expected keys
// (expected instances / chunks; = R, or what the speculative forward expects it to be) exceed what the LDS left beside the per-tile
// arrays can stage: the smallest power of two whose average (chunk, band) share fits beside the band's own arrays.
int gsr_scatter_bands(int P, int T, int gx, int expected_instances, int forced)
{
    if (T > GSR_MAX_TILES_LDS || T > 65535 * 8 || gx <= 0) return 1;
    const int nchunks = gsr_num_chunks(P), gy = T / gx;
    if (forced > 0) return forced < gy ? forced : (gy > 0 ? gy : 1);  // gsr_tuning.scatter_bands
    const double per_chunk = (double)(expected_instances > 0 ? expected_instances : 0) / nchunks;
    const size_t budget = 160 * 1024 - 8192 - 1024;
    for (int nb = 1; nb <= 16 && nb <= gy; nb *= 2) {
        const size_t tb = (size_t)((gy + nb - 1) / nb) * gx, fixed = gsr_align(tb * 12 + 8);
        // (the AVERAGE share has to fit: a workgroup whose share does not takes the direct path on its own.  Measured at config 4,
        // 2M Gaussians @1920x1080, scatter launch: 1 band 203 us, 2: 185, 4: 125, 8: 172, 16: 269 -- every band walks all the
        // chunk's Gaussians again, so more bands than needed cost more than they save; config 2 with 2 forced bands: 34 -> 42 us)
        if (budget > fixed + 4096 && per_chunk / nb <= (double)((budget - fixed) / 10)) return nb;
    }
    return 1;  // nothing fits (huge image): the plain form with its direct stores
}

hipError_t gsr_launch_scatter(int P, int T, int gx, const GsrGeom& geom, const GsrImage& image, const GsrBinning& bin,
                              int capacity, int expected_instances, int forced_bands, bool fused_tile_scan, uint32_t* fused_info_host, bool inference,
                              bool occlusion_cut, hipStream_t stream)
{
    const int nchunks = gsr_num_chunks(P);
    uint32_t* const offsets = inference ? nullptr : geom.offsets;  // gradient-slot numbering: only a backward reads it
    const int gy = gx > 0 ? T / gx : 0;
    const int nbands = gsr_scatter_bands(P, T, gx, expected_instances, forced_bands);
    GsrFusedScan fs = { nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr };
    if (fused_tile_scan && T <= GSR_MAX_TILES_LDS)
        fs = GsrFusedScan{ image.tile_count, image.ranges, image.info, image.tile_work, image.sorted_len, image.need_full, fused_info_host,
                           occlusion_cut ? (const uint32_t*)image.occ_drop : (const uint32_t*)nullptr, image.tile_group };
    if (T > GSR_MAX_TILES_LDS) {
        hipLaunchKernelGGL(gsr_cursor_init_kernel, dim3((T + 255) / 256), dim3(256), 0, stream, T, image.ranges, image.table);
        hipLaunchKernelGGL((gsr_scatter_kernel<true, false>), dim3(nchunks), dim3(GSR_HIST_THREADS), 0, stream, P, T, gx, nchunks,
                           geom.rect, geom.tmask, geom.depthkey, image.table, geom.scan_sums, image.ranges, offsets,
                           bin.seg_keys, (uint32_t)capacity, 0u, fs, 1, gy);
        return hipGetLastError();
    }
    hipError_t e = gsr_allow_big_lds();
    if (e != hipSuccess) return e;
    // dynamic LDS: the three per-tile arrays of the workgroup's band of tile rows + the staging buffer (10 B per instance).  Plain
    // form: as much staging as the CU has left (a chunk with more instances takes the direct path inside the same launch).  Banded
    // form: 1.3 x the expected (chunk, band) share -- a smaller footprint lets two workgroups share a CU.
    const size_t budget = 160 * 1024 - 8192 - 1024;  // static arrays of the kernel: ~4.3 KiB
    const size_t tb = (size_t)((gy + nbands - 1) / nbands) * gx;
    const size_t fixed = gsr_align(tb * 12 + 8);
    size_t stage_cap = budget > fixed + 4096 ? (budget - fixed) / 10 : 0;
    if (nbands > 1) {
        const size_t want = (size_t)(1.3 * (double)(expected_instances > 0 ? expected_instances : 0) / nchunks / nbands) + 256;
        if (want < stage_cap) stage_cap = want;
    }
    stage_cap &= ~(size_t)63;
#ifdef GSR_SCATTER_NO_STAGING
    stage_cap = 0;
#endif
    const size_t lds = stage_cap ? fixed + stage_cap * 10 : tb * 4;
    if (nbands > 1)
        hipLaunchKernelGGL((gsr_scatter_kernel<false, true>), dim3(nchunks * nbands), dim3(GSR_HIST_THREADS), lds, stream, P, T, gx, nchunks,
                           geom.rect, geom.tmask, geom.depthkey, image.table, geom.scan_sums, image.ranges, offsets,
                           bin.seg_keys, (uint32_t)capacity, (uint32_t)stage_cap, fs, nbands, gy);
    else
        hipLaunchKernelGGL((gsr_scatter_kernel<false, false>), dim3(nchunks), dim3(GSR_HIST_THREADS), lds, stream, P, T, gx, nchunks,
                           geom.rect, geom.tmask, geom.depthkey, image.table, geom.scan_sums, image.ranges, offsets,
                           bin.seg_keys, (uint32_t)capacity, (uint32_t)stage_cap, fs, 1, gy);
    return hipGetLastError();
}

// Full sort of the lists in (lo0, ...] by size class: (.., SMALL] and (SMALL, LARGE] in LDS, longer in global memory.
// The network is a chain of ~30 LDS round trips + barriers per workgroup and runs at the speed occupancy allows, and
// virtual padding is never stored: the dynamic LDS is sized for the longest list that exists (8.5 B per key), not for
// the class limit -- 11 KiB instead of 34 KiB on the bench scene, twice the resident workgroups.
static hipError_t gsr_launch_full_sorts(int T, int capacity, uint32_t lo0, uint32_t max_tile_count, const GsrImage& image,
                                        const GsrBinning& bin_in, const uint32_t* only_flagged, uint32_t* sorted_len,
                                        bool inference, hipStream_t stream, bool split_at_near_cap = false)
{
    GsrBinning bin = bin_in;
    if (inference) bin.slot_written = nullptr;  // the written-slot flags belong to the backward
    // split_at_near_cap (the partial sort's bet is off, api.hip gsr_partial_bet): the many lists up to GSR_NEAR_CAP keep their 256-thread
    // kernel and only the few longer ones take the 1024-thread one (one class for both gives every 400-entry list 1024 threads)
    const uint32_t caps[] = { split_at_near_cap ? (uint32_t)GSR_NEAR_CAP : 0u, (uint32_t)GSR_SORT_CAP_SMALL, (uint32_t)GSR_SORT_CAP_LARGE };
    uint32_t lo = lo0;
    for (uint32_t cap : caps) {
        if (cap > lo && max_tile_count > lo) {
            if (cap > GSR_SORT_CAP_SMALL) {
                const hipError_t e = gsr_allow_big_lds();
                if (e != hipSuccess) return e;
            }
            const uint32_t longest = min(cap, max_tile_count);
            const size_t lds = gsr_align((size_t)GSR_PAD(longest) * 8 + 8);
            if (longest <= 2048u)
                hipLaunchKernelGGL(gsr_tile_sort_lds_kernel<256>, dim3(T), dim3(256), lds, stream, image.ranges, bin.seg_keys,
                                   bin.point_list, bin.slot_written, lo, cap, longest, (uint32_t)capacity, only_flagged, sorted_len);
            else
                hipLaunchKernelGGL(gsr_tile_sort_lds_kernel<1024>, dim3(T), dim3(1024), lds, stream, image.ranges, bin.seg_keys,
                                   bin.point_list, bin.slot_written, lo, cap, longest, (uint32_t)capacity, only_flagged, sorted_len);
        }
        lo = max(lo, cap);
    }
    if (max_tile_count > GSR_SORT_CAP_LARGE)
        hipLaunchKernelGGL(gsr_tile_sort_global_kernel, dim3(T), dim3(1024), 0, stream, image.ranges, bin.seg_keys,
                           bin.point_list, bin.slot_written, (uint32_t)GSR_SORT_CAP_LARGE, (uint32_t)capacity, only_flagged,
                           sorted_len);
    return hipGetLastError();
}

hipError_t gsr_launch_tile_sort(int T, int capacity, int max_tile_count, int partial /* 0 = complete sorts, 1 = long lists bet on a sorted
                                prefix, 2 = complete sorts, lists beyond GSR_NEAR_CAP in a launch of their own */, bool speculative, bool inference, const GsrGeom& geom,
                                const GsrImage& image, const GsrBinning& bin_in, hipStream_t stream)
{
    (void)geom;
    GsrBinning bin = bin_in;
    if (inference) bin.slot_written = nullptr;
    // max_tile_count < 0: not known -> run every variant, blocks exit on mismatch.  speculative: max_tile_count is the
    // caller's guess (it sizes the LDS; the host checks it against the truth afterwards)
    if (capacity <= 0) return hipSuccess;
    const uint32_t mx = max_tile_count < 0 ? 0x7fffffffu : (uint32_t)max_tile_count;
    if (partial != 1) return gsr_launch_full_sorts(T, capacity, 0u, mx, image, bin, nullptr, nullptr, inference, stream, partial == 2 && mx > GSR_NEAR_CAP);
    // lists up to GSR_NEAR_CAP: full sort in LDS; longer ones: sorted prefix only (gsr_tile_sort_near_kernel)
    // (a guess below the cap that turns out too small fails the host's check anyway and stage 2 is redone)
    // (round 4 measured the alternative for frames whose longest list is in (2048, 4096] -- complete sorts, 256-thread kernel for the
    // lists up to 2048 + 1024-thread kernel for the longer ones: config 4 108 -> 122 us, large-splat frame after the occlusion
    // cut-off 133 -> 239 us: the prefix-sort kernel is not the slower one, the cost follows the number of keys)
    if (mx > GSR_NEAR_CAP) {  // long lists exist: one launch does both classes (fixed 21 KiB of LDS)
        hipLaunchKernelGGL(gsr_tile_sort_near_kernel, dim3(T), dim3(256), 0, stream, image.ranges, bin.seg_keys, bin.point_list,
                           bin.slot_written, image.sorted_len, (uint32_t)capacity);
    } else {  // short lists only: LDS sized for the longest one (twice the resident workgroups on the bench scene)
        const uint32_t longest = min((uint32_t)GSR_NEAR_CAP, mx);
        const size_t lds = gsr_align((size_t)GSR_PAD(longest) * 8 + 8);
        hipLaunchKernelGGL(gsr_tile_sort_lds_kernel<256>, dim3(T), dim3(256), lds, stream, image.ranges, bin.seg_keys, bin.point_list,
                           bin.slot_written, 0u, (uint32_t)GSR_NEAR_CAP, longest, (uint32_t)capacity, (const uint32_t*)nullptr,
                           (uint32_t*)nullptr);
    }
    return hipGetLastError();
}

// After a forward over partially sorted lists: full sort of the tiles that ran off their sorted prefix (need_full).
hipError_t gsr_launch_sort_fixup(int T, int capacity, int max_tile_count, const GsrImage& image, const GsrBinning& bin,
                                 bool inference, hipStream_t stream)
{
    if (capacity <= 0 || max_tile_count <= GSR_NEAR_CAP) return hipSuccess;
    return gsr_launch_full_sorts(T, capacity, (uint32_t)GSR_NEAR_CAP, (uint32_t)max_tile_count, image, bin, image.need_full,
                                 image.sorted_len, inference, stream);
}
