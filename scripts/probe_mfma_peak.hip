// sustained v_mfma_f32_32x32x16_bf16 rate of the chip: 8 waves per CU, 4 independent accumulators per wave, no memory
// traffic.  Calibrates what fraction of the 2.5 PFLOP/s nominal dense peak a real kernel can reach under power limits.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
__global__ __launch_bounds__(512) void k(float* out, int iters, float seed) {
    f32x16 acc[4];
    for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    bf16x8 x, w;
    for (int e = 0; e < 8; ++e) { x[e] = (__bf16)(seed + threadIdx.x * 1e-3f + e); w[e] = (__bf16)(seed * 0.5f - e); }
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int a = 0; a < 4; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w, x, acc[a], 0, 0, 0);
    }
    float s = 0.f;
    for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}
int main() {
    float* out; hipMalloc(&out, 4 * 512 * 2048);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int blocks : {256, 512}) {
        for (int iters : {2000, 20000, 100000}) {
            hipLaunchKernelGGL(k, dim3(blocks), dim3(512), 0, 0, out, 100, 1.0f);
            hipDeviceSynchronize();
            hipEventRecord(a);
            hipLaunchKernelGGL(k, dim3(blocks), dim3(512), 0, 0, out, iters, 1.0f);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            const double flops = (double)blocks * 8 * iters * 32 * 32768.0;
            printf("blocks %d iters %d: %.3f ms  %.1f TFLOP/s\n", blocks, iters, ms, flops / ms / 1e9);
        }
    }
    return 0;
}
