// sustained v_mfma_f32_32x32x16_bf16 rate of the chip: 8 waves per CU, 4 independent accumulators per wave, no memory
// traffic.  mode 0: constant operands; mode 1: random operands (8 different register sets cycled), i.e. realistic
// switching activity -- what fraction of the 2.5 PFLOP/s nominal dense peak survives the power limit.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ unsigned hash(unsigned x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
__global__ __launch_bounds__(512) void k(float* out, int iters, int mode) {
    f32x16 acc[4];
    for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    bf16x8 x[8], w[8];
    for (int s = 0; s < 8; ++s) {
        u32x4 ux, uw;
        for (int e = 0; e < 4; ++e) {
            unsigned hx = mode ? hash(threadIdx.x * 131 + s * 17 + e + blockIdx.x * 7919) : 0x3f803f80u;
            unsigned hw = mode ? hash(threadIdx.x * 733 + s * 29 + e + 12345) : 0x3f003f00u;
            // keep exponents moderate: sign + 7-bit mantissa random, exponent ~ 2^-1..2^1
            ux[e] = (hx & 0x807f807fu) | 0x3f003f00u | ((hx >> 3) & 0x00800080u);
            uw[e] = (hw & 0x807f807fu) | 0x3e803e80u | ((hw >> 5) & 0x00800080u);
        }
        x[s] = __builtin_bit_cast(bf16x8, ux); w[s] = __builtin_bit_cast(bf16x8, uw);
    }
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int a = 0; a < 4; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[(u + a) & 7], x[u], acc[a], 0, 0, 0);
    }
    float s = 0.f;
    for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}
int main() {
    float* out; (void)hipMalloc(&out, 4 * 512 * 2048);
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    for (int mode : {0, 1, 0, 1}) {
        for (int iters : {20000, 100000}) {
            const int blocks = 512;
            hipLaunchKernelGGL(k, dim3(blocks), dim3(512), 0, 0, out, 100, mode);
            (void)hipDeviceSynchronize();
            (void)hipEventRecord(a);
            hipLaunchKernelGGL(k, dim3(blocks), dim3(512), 0, 0, out, iters, mode);
            (void)hipEventRecord(b); (void)hipEventSynchronize(b);
            float ms; (void)hipEventElapsedTime(&ms, a, b);
            const double flops = (double)blocks * 8 * iters * 32 * 32768.0;
            printf("mode %d iters %d: %.3f ms  %.1f TFLOP/s\n", mode, iters, ms, flops / ms / 1e9);
        }
    }
    return 0;
}
