// do MFMA work of one wave and VALU work of ANOTHER wave on the same SIMD overlap?  512 threads: waves 0-3 run a chain of
// v_mfma_f32_16x16x32_bf16, waves 4-7 a chain of v_fma_f32 (mode 1: only the MFMA waves work, 2: only the VALU waves, 3: both);
// also mode 4: ONE wave per SIMD alternating 1 MFMA : 4 independent VALU in program order
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(512) void k(float* out, int iters, int mode) {
    const int wave = threadIdx.x >> 6;
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
    u32x4 ux = {0x3f803f80u + threadIdx.x, 0x3f813f80u, 0x3f823f80u, 0x3f833f80u}, uw = {0x3e803e80u, 0x3e813e80u, 0x3e823e80u, 0x3e833e80u + threadIdx.x};
    bf16x8 x = __builtin_bit_cast(bf16x8, ux), w = __builtin_bit_cast(bf16x8, uw);
    float a = threadIdx.x * 1e-3f, b = 1.0001f, c = 0.5f, d = 0.25f, e = 0.125f;
    if (mode == 4) {
        if (wave < 4)
            for (int i = 0; i < iters; ++i) {
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, x, acc0, 0, 0, 0);
                    a = a * b + c; c = c * b + d; d = d * b + e; e = e * b + a;
                }
            }
    } else if (wave < 4) {
        if (mode & 1)
            for (int i = 0; i < iters; ++i) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, x, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, x, acc1, 0, 0, 0);
                }
            }
    } else {
        if (mode & 2)
            for (int i = 0; i < iters; ++i) {
#pragma unroll
                for (int u = 0; u < 8; ++u) { a = a * b + c; c = c * b + d; d = d * b + e; e = e * b + a; }
            }
    }
    out[blockIdx.x * 512 + threadIdx.x] = acc0[0] + acc1[1] + a + c + d + e;
}
int main() {
    float* out; (void)hipMalloc(&out, 4 * 512 * 256);
    hipEvent_t ea, eb; (void)hipEventCreate(&ea); (void)hipEventCreate(&eb);
    const int iters = 20000;
    const char* nm[5] = {"", "MFMA waves only (8 MFMAs per iteration)", "VALU waves only (32 FMAs per iteration)", "both", "one wave: 1 MFMA : 4 FMA interleaved"};
    for (int rep = 0; rep < 2; ++rep)
        for (int mode = 1; mode <= 4; ++mode) {
            hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, out, 100, mode);
            (void)hipDeviceSynchronize();
            (void)hipEventRecord(ea);
            hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, out, iters, mode);
            (void)hipEventRecord(eb); (void)hipEventSynchronize(eb);
            float ms; (void)hipEventElapsedTime(&ms, ea, eb);
            printf("mode %d (%s): %.3f ms = %.1f ns per iteration\n", mode, nm[mode], ms, ms * 1e6 / iters);
        }
    return 0;
}
