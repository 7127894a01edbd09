// stats.hip -- densification statistics of one training iteration (second half).
//
// Replaces the body of GScream's scene/gaussian_model.py:730-757 GaussianModel.training_statis: ~15 boolean-mask
// indexing ops (each a nonzero + gather/scatter with a host sync) become one kernel:
//   opacity_accum[a]        += sum_k max(neural_opacity[n, k], 0)                                   (:733-737)
//   anchor_demon[a]         += 1                                                                    (:746)
//   for every offset k the decode kept (selection mask) whose Gaussian passed the update filter (radii > 0):
//     offset_gradient_accum[a*K + k] += |viewspace_grad[row, :2]|,   offset_denom[a*K + k] += 1     (:749-757)
// with a = visible[n] the anchor's row in the model and row = first[n] + rank of k among the kept offsets -- the same
// bookkeeping the decode produced (gsr_decode_count), so nothing is re-derived on the host.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gsr_common.h"

// One thread per (visible anchor, offset): neighbouring lanes touch neighbouring words of every [N*K] array (the first version
// had one thread per anchor walk its K offsets: every access strided by 4 K bytes, 39 us for ~50 MB; this one is bandwidth-
// bound).  The thread of offset 0 also sums the anchor's opacities, in offset order like the reference's sum(dim=1).
__global__ void __launch_bounds__(256) gst_training_stats_kernel(
    int Nv, int K, int M, const int32_t* __restrict__ visible, const float* __restrict__ neural_opacity,
    const uint8_t* __restrict__ selection, const uint32_t* __restrict__ first, const uint8_t* __restrict__ update_filter,
    const float* __restrict__ viewspace_grad /*[M,3]*/, float* __restrict__ opacity_accum, float* __restrict__ anchor_demon,
    float* __restrict__ offset_gradient_accum, float* __restrict__ offset_denom)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const bool live = i < (size_t)Nv * K;
    // (32-bit division wherever the index fits: the 64-bit one is a ~100-instruction routine, and this kernel is otherwise a stream)
    const int n = !live ? 0 : (size_t)Nv * K <= 0xffffffffull ? (int)((uint32_t)i / (uint32_t)K) : (int)(i / (size_t)K);
    const int k = (int)(i - (size_t)n * K);
    const bool kept = live && selection[i] != 0;
    // rank of k among the offsets the decode kept: the anchor's offsets sit in consecutive lanes, so it is a population count over
    // the wave's ballot -- except for the offsets an anchor has in the previous wavefront, which are counted from memory
    const unsigned long long bal = __ballot(kept);
    if (!live) return;
    const int a = visible ? visible[n] : n;
    const uint8_t* sel = selection + (size_t)n * K;
    if (kept) {
        const int lane = (int)(threadIdx.x & 63u), lane0 = lane - k;  // lane of the anchor's offset 0 (negative: in the previous wave)
        uint32_t row = first[n];
        if (lane0 >= 0) row += (uint32_t)__popcll(bal & ((1ull << lane) - 1ull) & ~((1ull << lane0) - 1ull));
        else {
            for (int j = 0; j < -lane0; j++) row += sel[j] ? 1u : 0u;
            row += (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
        }
        if (row < (uint32_t)M && update_filter[row]) {  // row >= M: a selection mask of another render; never read out of bounds
            const float gx = viewspace_grad[3 * (size_t)row], gy = viewspace_grad[3 * (size_t)row + 1];
            offset_gradient_accum[(size_t)a * K + k] += sqrtf(gx * gx + gy * gy);
            offset_denom[(size_t)a * K + k] += 1.0f;
        }
    }
    if (k == 0) {
        float osum = 0.f;
        for (int j = 0; j < K; j++) {
            const float op = neural_opacity[(size_t)n * K + j];
            osum += op < 0.f ? 0.f : op;
        }
        opacity_accum[a] += osum;
        anchor_demon[a] += 1.0f;
    }
}

hipError_t gst_launch_training_stats(int Nv, int K, int M, const int32_t* visible, const float* neural_opacity,
                                     const uint8_t* selection, const uint32_t* first, const uint8_t* update_filter,
                                     const float* viewspace_grad, float* opacity_accum, float* anchor_demon,
                                     float* offset_gradient_accum, float* offset_denom, hipStream_t stream)
{
    if (Nv <= 0) return hipSuccess;
    hipLaunchKernelGGL(gst_training_stats_kernel, dim3((unsigned)(((size_t)Nv * K + 255) / 256)), dim3(256), 0, stream, Nv, K, M, visible, neural_opacity,
                       selection, first, update_filter, viewspace_grad, opacity_accum, anchor_demon, offset_gradient_accum,
                       offset_denom);
    return hipGetLastError();
}
