// Microbenchmark (round 6): does the fp32 matrix pipe run BESIDE a VALU-bound loop on gfx950, at the occupancy a blend
// kernel has?  Behind the backward blend's MFMA reduction: the per-instance sums over the pixels of a wave are a
// contraction over the pixel index, so the wave writes its per-(pixel, instance) values to LDS, reads them back as the B
// operand of v_mfma_f32_16x16x4_f32 (k = pixel, column = (instance, kind)) against a per-pixel constant A operand, and the
// butterfly + partial products (150 of the loop's 296 issue cycles) become 4 matrix instructions per iteration.  That only
// pays if the 128 matrix-pipe cycles per iteration hide behind the other waves' VALU work.
//
// Every wave runs ITER iterations of: VALU block (V packed fma + 4 exp + 8 cmp/cndmask: ~190 issue cycles at V = 30),
// two LDS stores of its values; every 8th iteration: 8 ds_read_b128 + 32 MFMAs on two accumulators (the flush of 8 staged
// instances).  Modes: 1 = VALU only, 2 = LDS + MFMA only, 3 = both.  Occupancy is set by dynamic LDS.
// Prints SIMD cycles per wave-iteration = launch time x 2.4 GHz x 1024 SIMDs / (waves x ITER).
//
// build: hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_overlap tools/microbench/mfma_overlap.hip && /tmp/mfma_overlap
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

typedef float v4f __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
#define ITER 2048
#define ROWS 16
#define STRIDE 144  // floats per staged row (128 pixels + pad: the b128 reads of 16 rows x 4 k-groups are conflict-free)

template <int MODE, int V>
__global__ __launch_bounds__(128) void k_overlap(float* __restrict__ out, float seed)
{
    extern __shared__ float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* stage = lds + wave * ROWS * STRIDE;
    f2 p0 = {seed + lane, seed}, p1 = {seed + 1, seed + 2}, p2 = {seed + 3, seed}, p3 = {seed, seed + 5};
    const f2 pm = {0.99999f, 0.99998f}, pc = {1e-7f, 2e-7f};
    float e0 = seed * 0.01f, e1 = seed * 0.02f;
    float cnst[32];
#pragma unroll
    for (int i = 0; i < 32; i++) cnst[i] = seed + (float)(i * 64 + lane);
    v4f acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
    for (int it = 0; it < ITER; it++) {
        if (MODE & 1) {
#pragma unroll
            for (int v = 0; v < V / 4; v++)
                asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                             : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pm), "v"(pc));
            asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n"
                         "v_cmp_gt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc\n v_cmp_gt_f32 vcc, %1, %0\n v_cndmask_b32 %1, %1, %0, vcc\n"
                         "v_cmp_gt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc\n v_cmp_gt_f32 vcc, %1, %0\n v_cndmask_b32 %1, %1, %0, vcc\n"
                         : "+v"(e0), "+v"(e1) : : "vcc");
        }
        if (MODE & 2) {
            const int row = 2 * (it & 7);
            stage[row * STRIDE + lane] = p0.x;
            stage[row * STRIDE + 64 + lane] = p0.y;
            stage[(row + 1) * STRIDE + lane] = p1.x;
            stage[(row + 1) * STRIDE + 64 + lane] = p1.y;
            if ((it & 7) == 7) {
                __builtin_amdgcn_wave_barrier();
                const int a = lane & 15, g = lane >> 4;
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const v4f b = *reinterpret_cast<const v4f*>(&stage[a * STRIDE + 16 * u + 4 * g]);
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(cnst[4 * u + 0], b.x, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(cnst[4 * u + 1], b.y, acc1, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(cnst[4 * u + 2], b.z, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(cnst[4 * u + 3], b.w, acc1, 0, 0, 0);
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
    }
    const v4f acc = acc0 + acc1;
    const float r = p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y + e0 + e1 + acc.x + acc.y + acc.z + acc.w;
    if (r == 12345.678f) out[threadIdx.x] = r;
}

// MODE 7: as MODE 3, but the matrix instructions are INTERLEAVED with the VALU block: at a group boundary the 8 b128 reads go into 32
// registers, and each of the next 8 iterations issues 4 of the 32 MFMAs, one after every quarter of its VALU block.
template <int V>
__global__ __launch_bounds__(128) void k_interleaved(float* __restrict__ out, float seed)
{
    extern __shared__ float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* stage = lds + wave * ROWS * STRIDE;
    f2 p0 = {seed + lane, seed}, p1 = {seed + 1, seed + 2}, p2 = {seed + 3, seed}, p3 = {seed, seed + 5};
    const f2 pm = {0.99999f, 0.99998f}, pc = {1e-7f, 2e-7f};
    float e0 = seed * 0.01f, e1 = seed * 0.02f;
    float cnst[32];
#pragma unroll
    for (int i = 0; i < 32; i++) cnst[i] = seed + (float)(i * 64 + lane);
    v4f acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
    v4f breg[8];
#pragma unroll
    for (int u = 0; u < 8; u++) breg[u] = acc0;
    const int a = lane & 15, g = lane >> 4;
    for (int it0 = 0; it0 < ITER; it0 += 8) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
#pragma unroll
            for (int q = 0; q < 4; q++) {
#pragma unroll
                for (int v = 0; v < V / 16; v++)
                    asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                                 : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pm), "v"(pc));
                if (q == 0) asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n" : "+v"(e0), "+v"(e1));
                if (q == 1) asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n" : "+v"(e0), "+v"(e1));
                if (q >= 2) asm volatile("v_cmp_gt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc\n v_cmp_gt_f32 vcc, %1, %0\n v_cndmask_b32 %1, %1, %0, vcc\n"
                                         : "+v"(e0), "+v"(e1) : : "vcc");
                const int m = 4 * i + q;
                if (m & 1) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(cnst[m], breg[m >> 2][m & 3], acc1, 0, 0, 0);
                else acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(cnst[m], breg[m >> 2][m & 3], acc0, 0, 0, 0);
            }
            const int row = 2 * i;
            stage[row * STRIDE + lane] = p0.x;
            stage[row * STRIDE + 64 + lane] = p0.y;
            stage[(row + 1) * STRIDE + lane] = p1.x;
            stage[(row + 1) * STRIDE + 64 + lane] = p1.y;
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int u = 0; u < 8; u++) breg[u] = *reinterpret_cast<const v4f*>(&stage[a * STRIDE + 16 * u + 4 * g]);
        __builtin_amdgcn_wave_barrier();
    }
    const v4f acc = acc0 + acc1;
    const float r = p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y + e0 + e1 + acc.x + acc.y + acc.z + acc.w;
    if (r == 12345.678f) out[threadIdx.x] = r;
}

template <int MODE, int V>
__global__ __launch_bounds__(128) void k_dispatch(float* __restrict__ out, float seed);

template <int MODE, int V>
static void run(const char* name, int waves_per_simd, float* dout)
{
    // workgroups of 2 waves; W waves per SIMD = 2 W workgroups per CU: LDS per workgroup = 160 KB / (2 W) (rounded down)
    const int wg_per_cu = 2 * waves_per_simd;
    size_t lds = (160 * 1024 / wg_per_cu) & ~(size_t)511;
    if (lds > 64 * 1024) lds = 64 * 1024;
    if (lds < 2 * ROWS * STRIDE * 4) lds = 2 * ROWS * STRIDE * 4;
    hipFuncSetAttribute((const void*)k_overlap<MODE, V>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const int rounds = 4, grid = 256 * wg_per_cu * rounds;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    if (MODE == 7) hipFuncSetAttribute((const void*)k_interleaved<V>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
#define LAUNCH() do { if (MODE == 7) hipLaunchKernelGGL((k_interleaved<V>), dim3(grid), dim3(128), lds, 0, dout, 1.0f); \
                      else hipLaunchKernelGGL((k_overlap<MODE, V>), dim3(grid), dim3(128), lds, 0, dout, 1.0f); } while (0)
    LAUNCH();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    LAUNCH();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    const double waves = (double)grid * 2, cyc = ms * 1e-3 * 2.4e9 * 1024.0 / (waves * ITER);
    printf("%-28s V=%2d waves/SIMD=%d lds/wg=%6zu  %8.3f ms  %7.1f SIMD cycles per wave-iteration\n", name, V, waves_per_simd, lds, ms, cyc);
}

int main()
{
    float* dout;
    hipMalloc(&dout, 4096);
    for (int w : {1, 2, 3, 4, 5}) {
        run<1, 32>("VALU only", w, dout);
        run<2, 32>("LDS + MFMA only", w, dout);
        run<3, 32>("both", w, dout);
        run<7, 32>("both, interleaved", w, dout);
        run<1, 16>("VALU only", w, dout);
        run<3, 16>("both", w, dout);
        run<7, 16>("both, interleaved", w, dout);
    }
    hipFree(dout);
    return 0;
}
