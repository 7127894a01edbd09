// Microbenchmark: how many cycles does one wave64 VALU instruction occupy its SIMD on gfx950, per instruction kind
// and per number of resident waves?  Settles the "2 or 4 issue cycles per wave64 VALU instruction" question behind
// DESIGN.md's VALU bound of the blend kernels (MI355X_MICROARCH.md: SIMD-32, v_fma_f32 = 2 cycles).
//
// Each wave runs LOOPS x 64 instructions of one kind (8 independent register streams, or one dependent chain) between
// two s_memtime reads; W waves per SIMD are launched (grid = 256 CUs x W workgroups of 256 threads).  Reported per kind
// and W: what one wave sees per instruction, and the SIMD's issue cost per wave-instruction from the launch duration.
// s_memtime counts shader cycles on gfx950 (MI355X_MICROARCH.md, "s_memtime tick = shader cycle"); s_memrealtime is the
// constant 100 MHz clock, so cycles / realtime = the shader clock the loop really ran at (DVFS), printed per row.
//
// build: hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_issue tools/microbench/valu_issue.hip && /tmp/valu_issue
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#define LOOPS 1024
#define PER_LOOP 64

#define REP8(S0, S1, S2, S3, S4, S5, S6, S7) S0 S1 S2 S3 S4 S5 S6 S7
// 8 instructions on the 8 independent streams (registers %0..%7 scalar floats, %8..%15 as float2 pairs)
#define STREAM8(OP) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)
#define X8(B) B B B B B B B B

#define FMA(i)    "v_fma_f32 %" #i ", %" #i ", %16, %17\n"
#define MUL(i)    "v_mul_f32 %" #i ", %" #i ", %16\n"
#define ADD(i)    "v_add_f32 %" #i ", %" #i ", %17\n"
#define PKFMA(i)  "v_pk_fma_f32 %" #i ", %" #i ", %18, %19\n"
#define PKMUL(i)  "v_pk_mul_f32 %" #i ", %" #i ", %18\n"
#define PKADD(i)  "v_pk_add_f32 %" #i ", %" #i ", %19\n"
#define EXP(i)    "v_exp_f32 %" #i ", %" #i "\n"
#define RCP(i)    "v_rcp_f32 %" #i ", %" #i "\n"
#define DPPADD(i) "v_add_f32_dpp %" #i ", %" #i ", %" #i " row_ror:4 row_mask:0xf bank_mask:0xf\n"
#define DPPMOV(i) "v_mov_b32_dpp %" #i ", %" #i " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
#define CNDMASK(i) "v_cndmask_b32 %" #i ", %" #i ", %16, vcc\n"
#define CMP(i)    "v_cmp_gt_f32 vcc, %" #i ", %16\n"
#define MAXF(i)   "v_max_f32 %" #i ", %" #i ", %16\n"
#define SWAP32(i) "v_permlane32_swap_b32 %" #i ", %" #i "\n"
#define CVT(i)    "v_cvt_f32_i32 %" #i ", %" #i "\n"
#define FMAC(i)   "v_fmac_f32 %" #i ", %16, %17\n"
#define FMAK(i)   "v_fma_f32 %" #i ", %" #i ", %16, 1.0\n"
#define CNDS(i)   "v_cndmask_b32_e64 %" #i ", %" #i ", %16, s[20:21]\n"
#define CMPS(i)   "v_cmp_gt_f32_e64 s[22:23], %" #i ", %16\n"
#define MOV64(i)  "v_mov_b64 %" #i ", %18\n"
#define RDLANE(i) "v_readlane_b32 s24, %" #i ", 3\n"
#define MINF(i)   "v_min_f32 %" #i ", 0x3f7d70a4, %" #i "\n"
#define MADU64(i) "v_mad_u64_u32 %" #i ", s[22:23], s25, 48, %" #i "\n"
#define CMPCNDV(i) "v_cmp_gt_f32 vcc, %" #i ", %16\nv_cndmask_b32 %" #i ", %" #i ", %17, vcc\n"
#define CMPCNDS(i) "v_cmp_gt_f32_e64 s[22:23], %" #i ", %16\nv_cndmask_b32_e64 %" #i ", %" #i ", %17, s[22:23]\n"
#define CNDVE64(i) "v_cndmask_b32_e64 %" #i ", %" #i ", %16, vcc\n"
#define CNDV0(i)   "v_cndmask_b32 %" #i ", 0, %" #i ", vcc\n"
#define FMADEP(i) "v_fma_f32 %0, %0, %16, %17\n"
#define PKFMADEP(i) "v_pk_fma_f32 %0, %0, %18, %19\n"
#define MIXA(i)   "v_fma_f32 %" #i ", %" #i ", %16, %17\nv_pk_fma_f32 %" #i ", %" #i ", %18, %19\n"

enum Kind { K_FMA, K_MUL, K_ADD, K_PKFMA, K_PKMUL, K_PKADD, K_EXP, K_RCP, K_DPPADD, K_DPPMOV, K_CNDMASK, K_CMP, K_MAX,
            K_SWAP32, K_CVT, K_FMADEP, K_PKFMADEP, K_FMAC, K_FMAK, K_CNDS, K_CMPS, K_MOV64, K_RDLANE, K_MINF, K_MADU64, K_SWAP16, K_CMPCNDV, K_CMPCNDS, K_CNDVE64, K_CNDV0, K_COUNT };
static const char* kNames[K_COUNT] = {
    "v_fma_f32", "v_mul_f32", "v_add_f32", "v_pk_fma_f32", "v_pk_mul_f32", "v_pk_add_f32", "v_exp_f32", "v_rcp_f32",
    "v_add_f32 dpp row_ror", "v_mov_b32 dpp quad_perm", "v_cndmask_b32", "v_cmp_gt_f32", "v_max_f32",
    "v_permlane32_swap", "v_cvt_f32_i32", "v_fma_f32 dependent chain", "v_pk_fma_f32 dependent chain", "v_fmac_f32 (2 VGPR sources + dst)",
    "v_fma_f32 (2 VGPR sources + inline constant)", "v_cndmask_b32_e64 (SGPR-pair mask)", "v_cmp_gt_f32_e64 (to SGPR pair)", "v_mov_b64",
    "v_readlane_b32", "v_min_f32 (literal)", "v_mad_u64_u32", "v_permlane16_swap", "pair: v_cmp_gt_f32 vcc + v_cndmask_b32 vcc (per instruction)",
    "pair: v_cmp_gt_f32_e64 s[22:23] + v_cndmask_b32_e64 s[22:23] (per instruction)", "v_cndmask_b32_e64 with vcc as the mask pair", "v_cndmask_b32 dst, 0, src, vcc" };

template <int KIND>
__global__ __launch_bounds__(256) void k_issue(uint64_t* __restrict__ out, float seed)
{
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a1, a0}, p5 = {a3, a2}, p6 = {a5, a4}, p7 = {a7, a6};
    float m = 0.99999f, c = 1e-7f;
    f2 pm = {m, m}, pc = {c, c};
    uint64_t r0 = __builtin_amdgcn_s_memrealtime();  // 100 MHz
    uint64_t t0 = __builtin_readcyclecounter();      // s_memtime: shader cycles
    for (int it = 0; it < LOOPS; it++) {
#define BODY(OPS) asm volatile(OPS : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), \
                                     "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) \
                                   : "v"(m), "v"(c), "v"(pm), "v"(pc) : "vcc", "s20", "s21", "s22", "s23", "s24", "s25")
        if (KIND == K_FMA) BODY(X8(STREAM8(FMA)));
        if (KIND == K_MUL) BODY(X8(STREAM8(MUL)));
        if (KIND == K_ADD) BODY(X8(STREAM8(ADD)));
#define P(i) #i
        if (KIND == K_PKFMA) BODY(X8("v_pk_fma_f32 %8, %8, %18, %19\nv_pk_fma_f32 %9, %9, %18, %19\nv_pk_fma_f32 %10, %10, %18, %19\nv_pk_fma_f32 %11, %11, %18, %19\n"
                                     "v_pk_fma_f32 %12, %12, %18, %19\nv_pk_fma_f32 %13, %13, %18, %19\nv_pk_fma_f32 %14, %14, %18, %19\nv_pk_fma_f32 %15, %15, %18, %19\n"));
        if (KIND == K_PKMUL) BODY(X8("v_pk_mul_f32 %8, %8, %18\nv_pk_mul_f32 %9, %9, %18\nv_pk_mul_f32 %10, %10, %18\nv_pk_mul_f32 %11, %11, %18\n"
                                     "v_pk_mul_f32 %12, %12, %18\nv_pk_mul_f32 %13, %13, %18\nv_pk_mul_f32 %14, %14, %18\nv_pk_mul_f32 %15, %15, %18\n"));
        if (KIND == K_PKADD) BODY(X8("v_pk_add_f32 %8, %8, %19\nv_pk_add_f32 %9, %9, %19\nv_pk_add_f32 %10, %10, %19\nv_pk_add_f32 %11, %11, %19\n"
                                     "v_pk_add_f32 %12, %12, %19\nv_pk_add_f32 %13, %13, %19\nv_pk_add_f32 %14, %14, %19\nv_pk_add_f32 %15, %15, %19\n"));
        if (KIND == K_EXP) BODY(X8(STREAM8(EXP)));
        if (KIND == K_RCP) BODY(X8(STREAM8(RCP)));
        if (KIND == K_DPPADD) BODY(X8(STREAM8(DPPADD)));
        if (KIND == K_DPPMOV) BODY(X8(STREAM8(DPPMOV)));
        if (KIND == K_CNDMASK) BODY(X8(STREAM8(CNDMASK)));
        if (KIND == K_CMP) BODY(X8(STREAM8(CMP)));
        if (KIND == K_MAX) BODY(X8(STREAM8(MAXF)));
        if (KIND == K_SWAP32) BODY(X8("v_permlane32_swap_b32 %0, %1\nv_permlane32_swap_b32 %2, %3\nv_permlane32_swap_b32 %4, %5\nv_permlane32_swap_b32 %6, %7\n"
                                      "v_permlane32_swap_b32 %1, %2\nv_permlane32_swap_b32 %3, %4\nv_permlane32_swap_b32 %5, %6\nv_permlane32_swap_b32 %7, %0\n"));
        if (KIND == K_CVT) BODY(X8(STREAM8(CVT)));
        if (KIND == K_FMADEP) BODY(X8(STREAM8(FMADEP)));
        if (KIND == K_FMAC) BODY(X8(STREAM8(FMAC)));
        if (KIND == K_FMAK) BODY(X8(STREAM8(FMAK)));
        if (KIND == K_CNDS) BODY(X8(STREAM8(CNDS)));
        if (KIND == K_CMPS) BODY(X8(STREAM8(CMPS)));
        if (KIND == K_MOV64) BODY(X8("v_mov_b64 %8, %18\nv_mov_b64 %9, %18\nv_mov_b64 %10, %18\nv_mov_b64 %11, %18\nv_mov_b64 %12, %18\nv_mov_b64 %13, %18\nv_mov_b64 %14, %18\nv_mov_b64 %15, %18\n"));
        if (KIND == K_RDLANE) BODY(X8(STREAM8(RDLANE)));
        if (KIND == K_MINF) BODY(X8(STREAM8(MINF)));
        if (KIND == K_MADU64) BODY(X8("v_mad_u64_u32 %8, s[22:23], s25, 48, %8\nv_mad_u64_u32 %9, s[22:23], s25, 48, %9\nv_mad_u64_u32 %10, s[22:23], s25, 48, %10\nv_mad_u64_u32 %11, s[22:23], s25, 48, %11\n"
                                     "v_mad_u64_u32 %12, s[22:23], s25, 48, %12\nv_mad_u64_u32 %13, s[22:23], s25, 48, %13\nv_mad_u64_u32 %14, s[22:23], s25, 48, %14\nv_mad_u64_u32 %15, s[22:23], s25, 48, %15\n"));
        if (KIND == K_SWAP16) BODY(X8("v_permlane16_swap_b32 %0, %1\nv_permlane16_swap_b32 %2, %3\nv_permlane16_swap_b32 %4, %5\nv_permlane16_swap_b32 %6, %7\n"
                                      "v_permlane16_swap_b32 %1, %2\nv_permlane16_swap_b32 %3, %4\nv_permlane16_swap_b32 %5, %6\nv_permlane16_swap_b32 %7, %0\n"));
        if (KIND == K_CMPCNDV) BODY(X8(CMPCNDV(0) CMPCNDV(1) CMPCNDV(2) CMPCNDV(3)));
        if (KIND == K_CMPCNDS) BODY(X8(CMPCNDS(0) CMPCNDS(1) CMPCNDS(2) CMPCNDS(3)));
        if (KIND == K_CNDVE64) BODY(X8(STREAM8(CNDVE64)));
        if (KIND == K_CNDV0) BODY(X8(STREAM8(CNDV0)));
        if (KIND == K_PKFMADEP) BODY(X8("v_pk_fma_f32 %8, %8, %18, %19\nv_pk_fma_f32 %8, %8, %18, %19\nv_pk_fma_f32 %8, %8, %18, %19\nv_pk_fma_f32 %8, %8, %18, %19\n"
                                        "v_pk_fma_f32 %8, %8, %18, %19\nv_pk_fma_f32 %8, %8, %18, %19\nv_pk_fma_f32 %8, %8, %18, %19\nv_pk_fma_f32 %8, %8, %18, %19\n"));
    }
    uint64_t t1 = __builtin_readcyclecounter();
    uint64_t r1 = __builtin_amdgcn_s_memrealtime();
    float s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.x + p2.x + p3.x + p4.x + p5.x + p6.x + p7.x + p0.y + p1.y + p2.y + p3.y;
    if ((threadIdx.x & 63) == 0) {
        int w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
        out[4 * w + 0] = t0;
        out[4 * w + 1] = t1;
        out[4 * w + 2] = r1 - r0;
        out[4 * w + 3] = (uint64_t)(s == 12345.678f);
    }
}

typedef void (*KernelFn)(uint64_t*, float);
template <int K> static KernelFn pick() { return k_issue<K>; }
static KernelFn kTable[K_COUNT] = {
    k_issue<K_FMA>, k_issue<K_MUL>, k_issue<K_ADD>, k_issue<K_PKFMA>, k_issue<K_PKMUL>, k_issue<K_PKADD>, k_issue<K_EXP>,
    k_issue<K_RCP>, k_issue<K_DPPADD>, k_issue<K_DPPMOV>, k_issue<K_CNDMASK>, k_issue<K_CMP>, k_issue<K_MAX>, k_issue<K_SWAP32>,
    k_issue<K_CVT>, k_issue<K_FMADEP>, k_issue<K_PKFMADEP>, k_issue<K_FMAC>, k_issue<K_FMAK>, k_issue<K_CNDS>, k_issue<K_CMPS>,
    k_issue<K_MOV64>, k_issue<K_RDLANE>, k_issue<K_MINF>, k_issue<K_MADU64>, k_issue<K_SWAP16>, k_issue<K_CMPCNDV>, k_issue<K_CMPCNDS>,
    k_issue<K_CNDVE64>, k_issue<K_CNDV0> };

int main(int argc, char** argv)
{
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    printf("# %s, %d CUs, nominal clockRate %d kHz\n", prop.name, cus, prop.clockRate);
    uint64_t* d;
    const int max_waves = cus * 4 * 8;
    hipMalloc(&d, (size_t)max_waves * 4 * sizeof(uint64_t));
    std::vector<uint64_t> h((size_t)max_waves * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const double n_instr = (double)LOOPS * PER_LOOP;
    printf("| instruction | waves/SIMD launched | us per launch | shader MHz | cyc/instr one wave sees | cyc/instr/SIMD (launch time x clock x SIMDs / all wave-instructions) |\n");
    printf("|---|---:|---:|---:|---:|---:|\n");
    const int Ws[] = {1, 2, 4, 8};
    for (int k = 0; k < K_COUNT; k++) {
        for (int wi = 0; wi < 4; wi++) {
            const int W = Ws[wi];
            const int blocks = cus * W;  // 256-thread blocks: 4 waves each, one per SIMD
            for (int r = 0; r < 2; r++) hipLaunchKernelGGL(kTable[k], dim3(blocks), dim3(256), 0, 0, d, 1.0f);
            hipEventRecord(e0);
            hipLaunchKernelGGL(kTable[k], dim3(blocks), dim3(256), 0, 0, d, 1.0f);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            const int waves = blocks * 4;
            hipMemcpy(h.data(), d, (size_t)waves * 4 * sizeof(uint64_t), hipMemcpyDeviceToHost);
            double mean = 0, real = 0;
            for (int w = 0; w < waves; w++) {
                mean += (double)(h[4 * w + 1] - h[4 * w + 0]);
                real += (double)h[4 * w + 2];
            }
            mean /= waves;
            real /= waves;
            const double mhz = mean / (real / 100.0);  // shader cycles per microsecond while the loop ran
            // The SIMD arbitrates oldest-first: with 4 or 8 waves per SIMD the older ones finish first and every wave sees
            // fewer competitors than are resident, so the per-wave figure understates the cost; the launch-time figure
            // (all wave-instructions of a SIMD over the cycles the launch lasted) is the SIMD's real issue cost.
            const double simd_cycles = ms * 1e3 * mhz;
            printf("| %s | %d | %.1f | %.0f | %.2f | %.2f |\n", kNames[k], W, ms * 1e3, mhz, mean / n_instr, simd_cycles / (W * n_instr));
        }
    }
    return 0;
}
