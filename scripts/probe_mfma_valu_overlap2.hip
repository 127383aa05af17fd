// Round 5 redo of the MFMA || VALU overlap probe (the round-4 probe used the 4-pass v_mfma_f32_16x16x32_bf16 and let the compiler
// reorder / SLP-pack its streams).  Here every stream is ONE asm volatile block per loop iteration, on the instruction conv_dma3 /
// conv_direct3 issue (v_mfma_f32_32x32x16_bf16, 8 passes = 32 clk per SIMD), so that the ISA is what the source says
// (profiles/r05_probe_mfma_valu_overlap2_isa.s is the hipcc -S of this file).
//
// 256 workgroups x 512 threads: waves w and w + 4 of a workgroup share SIMD w; waves 0-3 are the OLDER half, 4-7 the YOUNGER half.
// Roles per half:   0 idle (exit)
//                   1 MFMA stream: 8 MFMAs on 8 independent accumulators per iteration (256 clk of matrix pipe)
//                   2 VALU stream, plain: 32 v_fma_f32 on 8 independent chains per iteration
//                   3 VALU stream, epilogue mix: per iteration 4 x (v_cvt_pk_bf16_f32, v_lshlrev_b32, v_and_b32, 2 v_sub_f32, v_exp_f32, v_max_f32,
//                     v_fma_f32) = 32 instructions, the operations of bias + ELU + three-way split
//                   4 / 5 / 6 ONE wave interleaving 1 MFMA : 2 / 4 / 6 plain VALU in program order (8 MFMAs per iteration)
//                   7 ONE wave interleaving 1 MFMA : 4 epilogue-mix VALU
// s_setprio of each half is a launch argument.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define MF(n) "v_mfma_f32_32x32x16_bf16 %" #n ", %8, %9, %" #n "\n\t"
// plain VALU: chains x0..x7 are operands 10..17, constants 18, 19
#define VF(n) "v_fma_f32 %" #n ", %" #n ", %18, %19\n\t"
// epilogue mix on (value, scratch) register pairs: (10,11) (12,13) (14,15) (16,17)
#define VM(a, b) "v_cvt_pk_bf16_f32 %" #b ", %" #a ", %" #a "\n\t" "v_lshlrev_b32 %" #b ", 16, %" #b "\n\t" "v_sub_f32 %" #b ", %" #a ", %" #b "\n\t" \
                 "v_exp_f32 %" #b ", %" #b "\n\t" "v_max_f32 %" #a ", %" #a ", %18\n\t" "v_and_b32 %" #b ", 0xffff0000, %" #b "\n\t" \
                 "v_sub_f32 %" #a ", %" #a ", %" #b "\n\t" "v_fma_f32 %" #a ", %" #a ", %18, %19\n\t"
#define OPS : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) \
            : "v"(wa), "v"(xb), "v"(x0), "v"(x1), "v"(x2), "v"(x3), "v"(x4), "v"(x5), "v"(x6), "v"(x7), "v"(c0), "v"(c1)
// the VALU registers are read-write in the asm but declared as inputs on purpose: the values are garbage either way and an input
// constraint keeps the compiler from adding copies; "memory" keeps the blocks in order

__global__ __launch_bounds__(512, 1) void k(float* out, long long* cyc, int iters, int role_old, int role_young, int prio_old, int prio_young) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int role = wave < 4 ? role_old : role_young, prio = wave < 4 ? prio_old : prio_young;
    f32x16 a0, a1, a2, a3, a4, a5, a6, a7;
    for (int r = 0; r < 16; ++r) { a0[r] = 0.f; a1[r] = 0.f; a2[r] = 0.f; a3[r] = 0.f; a4[r] = 0.f; a5[r] = 0.f; a6[r] = 0.f; a7[r] = 0.f; }
    // operands that are hi planes of dense values (random mantissas), as in the engine
    unsigned s = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return s; };
    auto pair = [&]() { return (0x3f00u | ((rnd() >> 9) & 0xffu)) | ((0x3f00u | ((rnd() >> 11) & 0xffu)) << 16); };
    u32x4 wa = {pair(), pair(), pair(), pair()}, xb = {pair(), pair(), pair(), pair()};
    float x0 = threadIdx.x * 1e-3f, x1 = 0.1f, x2 = 0.2f, x3 = 0.3f, x4 = 0.4f, x5 = 0.5f, x6 = 0.6f, x7 = 0.7f, c0 = 0.999f, c1 = 1e-3f;
    if (role == 0) return;
    if (prio) __builtin_amdgcn_s_setprio(1);
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    if (role == 1) {
        for (int i = 0; i < iters; ++i) asm volatile(MF(0) MF(1) MF(2) MF(3) MF(4) MF(5) MF(6) MF(7) OPS : "memory");
    } else if (role == 2) {
        for (int i = 0; i < iters; ++i)
            asm volatile(VF(10) VF(11) VF(12) VF(13) VF(14) VF(15) VF(16) VF(17) VF(10) VF(11) VF(12) VF(13) VF(14) VF(15) VF(16) VF(17)
                         VF(10) VF(11) VF(12) VF(13) VF(14) VF(15) VF(16) VF(17) VF(10) VF(11) VF(12) VF(13) VF(14) VF(15) VF(16) VF(17) OPS : "memory");
    } else if (role == 3) {
        for (int i = 0; i < iters; ++i) asm volatile(VM(10, 11) VM(12, 13) VM(14, 15) VM(16, 17) OPS : "memory");
    } else if (role == 4) {
        for (int i = 0; i < iters; ++i)
            asm volatile(MF(0) VF(10) VF(11) MF(1) VF(12) VF(13) MF(2) VF(14) VF(15) MF(3) VF(16) VF(17)
                         MF(4) VF(10) VF(11) MF(5) VF(12) VF(13) MF(6) VF(14) VF(15) MF(7) VF(16) VF(17) OPS : "memory");
    } else if (role == 5) {
        for (int i = 0; i < iters; ++i)
            asm volatile(MF(0) VF(10) VF(11) VF(12) VF(13) MF(1) VF(14) VF(15) VF(16) VF(17) MF(2) VF(10) VF(11) VF(12) VF(13) MF(3) VF(14) VF(15) VF(16) VF(17)
                         MF(4) VF(10) VF(11) VF(12) VF(13) MF(5) VF(14) VF(15) VF(16) VF(17) MF(6) VF(10) VF(11) VF(12) VF(13) MF(7) VF(14) VF(15) VF(16) VF(17) OPS : "memory");
    } else if (role == 6) {
        for (int i = 0; i < iters; ++i)
            asm volatile(MF(0) VF(10) VF(11) VF(12) VF(13) VF(14) VF(15) MF(1) VF(16) VF(17) VF(10) VF(11) VF(12) VF(13) MF(2) VF(14) VF(15) VF(16) VF(17) VF(10) VF(11)
                         MF(3) VF(12) VF(13) VF(14) VF(15) VF(16) VF(17) MF(4) VF(10) VF(11) VF(12) VF(13) VF(14) VF(15) MF(5) VF(16) VF(17) VF(10) VF(11) VF(12) VF(13)
                         MF(6) VF(14) VF(15) VF(16) VF(17) VF(10) VF(11) MF(7) VF(12) VF(13) VF(14) VF(15) VF(16) VF(17) OPS : "memory");
    } else if (role == 7) {
        // 1 MFMA : 4 epilogue-mix instructions: each VM block (8 instructions) is cut in two halves behind two MFMAs
        for (int i = 0; i < iters; ++i)
            asm volatile(MF(0) "v_cvt_pk_bf16_f32 %11, %10, %10\n\tv_lshlrev_b32 %11, 16, %11\n\tv_sub_f32 %11, %10, %11\n\tv_exp_f32 %11, %11\n\t"
                         MF(1) "v_max_f32 %10, %10, %18\n\tv_and_b32 %11, 0xffff0000, %11\n\tv_sub_f32 %10, %10, %11\n\tv_fma_f32 %10, %10, %18, %19\n\t"
                         MF(2) "v_cvt_pk_bf16_f32 %13, %12, %12\n\tv_lshlrev_b32 %13, 16, %13\n\tv_sub_f32 %13, %12, %13\n\tv_exp_f32 %13, %13\n\t"
                         MF(3) "v_max_f32 %12, %12, %18\n\tv_and_b32 %13, 0xffff0000, %13\n\tv_sub_f32 %12, %12, %13\n\tv_fma_f32 %12, %12, %18, %19\n\t"
                         MF(4) "v_cvt_pk_bf16_f32 %15, %14, %14\n\tv_lshlrev_b32 %15, 16, %15\n\tv_sub_f32 %15, %14, %15\n\tv_exp_f32 %15, %15\n\t"
                         MF(5) "v_max_f32 %14, %14, %18\n\tv_and_b32 %15, 0xffff0000, %15\n\tv_sub_f32 %14, %14, %15\n\tv_fma_f32 %14, %14, %18, %19\n\t"
                         MF(6) "v_cvt_pk_bf16_f32 %17, %16, %16\n\tv_lshlrev_b32 %17, 16, %17\n\tv_sub_f32 %17, %16, %17\n\tv_exp_f32 %17, %17\n\t"
                         MF(7) "v_max_f32 %16, %16, %18\n\tv_and_b32 %17, 0xffff0000, %17\n\tv_sub_f32 %16, %16, %17\n\tv_fma_f32 %16, %16, %18, %19\n\t" OPS : "memory");
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) cyc[wave] = t1 - t0;
    float acc = 0.f;
    for (int r = 0; r < 16; ++r) acc += a0[r] + a1[r] + a2[r] + a3[r] + a4[r] + a5[r] + a6[r] + a7[r];
    out[blockIdx.x * 512 + threadIdx.x] = acc + x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}

int main() {
    float* out; (void)hipMalloc(&out, 4 * 512 * 256);
    long long* cyc; (void)hipMalloc(&cyc, 8 * 8);
    hipEvent_t ea, eb; (void)hipEventCreate(&ea); (void)hipEventCreate(&eb);
    const int iters = 20000;
    struct Case { int ro, ry, po, py; const char* what; };
    const Case cases[] = {
        {1, 0, 0, 0, "MFMA stream alone (older half; 8 MFMAs / iteration)"},
        {0, 1, 0, 0, "MFMA stream alone (younger half)"},
        {2, 0, 0, 0, "plain VALU stream alone (32 v_fma / iteration)"},
        {3, 0, 0, 0, "epilogue-mix VALU stream alone (32 instr / iteration)"},
        {1, 2, 0, 0, "MFMA older || plain VALU younger, prio 0/0"},
        {2, 1, 0, 0, "plain VALU older || MFMA younger, prio 0/0"},
        {1, 2, 1, 0, "MFMA older prio 1 || plain VALU younger"},
        {1, 2, 0, 1, "MFMA older || plain VALU younger prio 1"},
        {2, 1, 1, 0, "plain VALU older prio 1 || MFMA younger"},
        {2, 1, 0, 1, "plain VALU older || MFMA younger prio 1"},
        {1, 3, 0, 0, "MFMA older || epilogue-mix younger, prio 0/0"},
        {3, 1, 0, 0, "epilogue-mix older || MFMA younger, prio 0/0"},
        {1, 3, 1, 0, "MFMA older prio 1 || epilogue-mix younger"},
        {1, 3, 0, 1, "MFMA older || epilogue-mix younger prio 1"},
        {1, 1, 0, 0, "MFMA || MFMA (two streams on one SIMD: must be the sum)"},
        {2, 2, 0, 0, "plain VALU || plain VALU"},
        {4, 0, 0, 0, "ONE wave, 1 MFMA : 2 plain VALU interleaved"},
        {5, 0, 0, 0, "ONE wave, 1 MFMA : 4 plain VALU interleaved"},
        {6, 0, 0, 0, "ONE wave, 1 MFMA : 6 plain VALU interleaved"},
        {7, 0, 0, 0, "ONE wave, 1 MFMA : 4 epilogue-mix interleaved"},
        {5, 5, 0, 0, "TWO waves, each 1 MFMA : 4 plain VALU interleaved"},
        {7, 7, 0, 0, "TWO waves, each 1 MFMA : 4 epilogue-mix interleaved"},
    };
    for (int rep = 0; rep < 2; ++rep)
        for (const Case& c : cases) {
            hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, out, cyc, 200, c.ro, c.ry, c.po, c.py);
            (void)hipDeviceSynchronize();
            (void)hipMemset(cyc, 0, 64);
            (void)hipEventRecord(ea);
            hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, out, cyc, iters, c.ro, c.ry, c.po, c.py);
            (void)hipEventRecord(eb); (void)hipEventSynchronize(eb);
            float ms; (void)hipEventElapsedTime(&ms, ea, eb);
            long long h[8]; (void)hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
            printf("rep %d roles %d/%d prio %d/%d  %-62s %8.3f ms = %7.1f ns/iter   s_memtime ticks/iter: older wave %7.1f  younger wave %7.1f\n", rep, c.ro, c.ry, c.po, c.py, c.what, ms,
                   ms * 1e6 / iters, (double)h[0] / iters, (double)h[4] / iters);
        }
    return 0;
}
