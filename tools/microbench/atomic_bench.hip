// Microbenchmark: throughput of global (agent-scope) integer atomics on MI355X with a tile-binning access pattern.
// build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o tools/microbench/libatomic_bench.so tools/microbench/atomic_bench.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
__global__ void k_noret(int n, const uint32_t* __restrict__ idx, uint32_t* cnt)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) atomicAdd(&cnt[idx[i]], 1u);
}
__global__ void k_ret(int n, const uint32_t* __restrict__ idx, uint32_t* cnt, uint32_t* out)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = atomicAdd(&cnt[idx[i]], 1u);
}
extern "C" int run(int n, const uint32_t* idx, uint32_t* cnt, uint32_t* out, int nbins, int reps, float* ms_noret, float* ms_ret)
{
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    dim3 g((n + 255) / 256), bl(256);
    for (int mode = 0; mode < 2; mode++) {
        hipMemset(cnt, 0, nbins * 4);
        for (int w = 0; w < 3; w++) { if (mode) hipLaunchKernelGGL(k_ret, g, bl, 0, 0, n, idx, cnt, out); else hipLaunchKernelGGL(k_noret, g, bl, 0, 0, n, idx, cnt); }
        hipEventRecord(a, 0);
        for (int r = 0; r < reps; r++) { if (mode) hipLaunchKernelGGL(k_ret, g, bl, 0, 0, n, idx, cnt, out); else hipLaunchKernelGGL(k_noret, g, bl, 0, 0, n, idx, cnt); }
        hipEventRecord(b, 0); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        *(mode ? ms_ret : ms_noret) = ms / reps;
    }
    return (int)hipGetLastError();
}
