// Microbenchmark: what do per-(instance, tile) gradient flushes cost as global float atomics (no return) into a dense
// [P][12] accumulator, instead of private 48-byte slots?  620k flushes x 12 floats, ids uniform in [0, P).
// build: hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics -o /tmp/fatomic_bench tools/microbench/fatomic_bench.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
__global__ void k_atomic(int n, const uint32_t* __restrict__ ids, float* acc)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float* a = acc + (size_t)ids[i] * 12;
#pragma unroll
    for (int f = 0; f < 12; f++) atomicAdd(a + f, 1.0f);
}
__global__ void k_store(int n, const uint32_t* __restrict__ ids, float4* slots)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float4* s = slots + (size_t)i * 3;
    s[0] = make_float4(1, 1, 1, 1); s[1] = s[0]; s[2] = s[0];
}
int main()
{
    const int n = 620000, P = 1000000;
    uint32_t* h = (uint32_t*)malloc(n * 4);
    srand(1);
    for (int i = 0; i < n; i++) h[i] = (uint32_t)(((uint64_t)rand() * 32768 + rand()) % P);
    uint32_t* ids; float* acc; float4* slots;
    hipMalloc(&ids, n * 4); hipMalloc(&acc, (size_t)P * 48); hipMalloc(&slots, (size_t)n * 48);
    hipMemcpy(ids, h, n * 4, hipMemcpyHostToDevice);
    hipMemset(acc, 0, (size_t)P * 48);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](const char* name, auto launch) {
        for (int i = 0; i < 3; i++) launch();
        hipEventRecord(e0);
        for (int i = 0; i < 20; i++) launch();
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%s: %.1f us per launch\n", name, ms / 20 * 1e3);
    };
    run("620k x 12 global_atomic_add_f32 (random Gaussians)", [&] { hipLaunchKernelGGL(k_atomic, dim3((n + 255) / 256), dim3(256), 0, 0, n, ids, acc); });
    run("620k x 48-byte private slot stores                ", [&] { hipLaunchKernelGGL(k_store, dim3((n + 255) / 256), dim3(256), 0, 0, n, ids, slots); });
    run("memset 48 MB                                      ", [&] { hipMemsetAsync(acc, 0, (size_t)P * 48, 0); });
    return 0;
}
