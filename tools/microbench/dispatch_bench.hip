// How fast can the GPU launch many tiny workgroups?  (diagnostic for the one-wave-per-workgroup kernels)
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void __launch_bounds__(64) k64(int* out) { if (threadIdx.x == 0 && blockIdx.x == 0x7fffffff) out[0] = 1; }
__global__ void __launch_bounds__(256) k256(int* out) { if (threadIdx.x == 0 && blockIdx.x == 0x7fffffff) out[0] = 1; }
__global__ void __launch_bounds__(64) k64lds(int* out) { __shared__ int s[1818]; s[threadIdx.x] = threadIdx.x; __syncthreads(); if (s[63 - threadIdx.x] == 12345) out[0] = 1; }
int main() {
    int* d; hipMalloc(&d, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](const char* name, auto launch) {
        for (int i = 0; i < 3; i++) launch();
        hipEventRecord(e0);
        for (int i = 0; i < 20; i++) launch();
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%s: %.2f us per launch\n", name, ms / 20 * 1e3);
    };
    run("15625 x 64 threads        ", [&] { hipLaunchKernelGGL(k64, dim3(15625), dim3(64), 0, 0, d); });
    run("15625 x 64 threads + 7 KB LDS", [&] { hipLaunchKernelGGL(k64lds, dim3(15625), dim3(64), 0, 0, d); });
    run("3907 x 256 threads        ", [&] { hipLaunchKernelGGL(k256, dim3(3907), dim3(256), 0, 0, d); });
    return 0;
}
