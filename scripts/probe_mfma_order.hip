// does the ORDER of back-to-back MFMAs change the sustained (power-limited) rate?  Random bf16 operands, 8 register sets
// each for A and B, 4 independent accumulators; the variants differ only in how often an operand changes between
// consecutive v_mfma_f32_32x32x16_bf16 of a wave.
//   0: both operands change every MFMA      1: A changes every MFMA, B every 4th (probe_mfma_peak's mode 1)
//   2: gray order, one operand changes per MFMA (A0B0 A0B1 A1B1 A1B0 ...)      3: A every 4th, B every 4th (alternating)
//   4: both operands change every 8th MFMA   5: constant operands
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ unsigned hash(unsigned x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
template <int MODE>
__global__ __launch_bounds__(512) void k(float* out, int iters) {
    f32x16 acc[4];
    for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    bf16x8 x[8], w[8];
    for (int s = 0; s < 8; ++s) {
        u32x4 ux, uw;
        for (int e = 0; e < 4; ++e) {
            unsigned hx = hash(threadIdx.x * 131 + s * 17 + e + blockIdx.x * 7919);
            unsigned hw = hash(threadIdx.x * 733 + s * 29 + e + 12345);
            ux[e] = (hx & 0x807f807fu) | 0x3f003f00u | ((hx >> 3) & 0x00800080u);
            uw[e] = (hw & 0x807f807fu) | 0x3e803e80u | ((hw >> 5) & 0x00800080u);
        }
        x[s] = __builtin_bit_cast(bf16x8, ux); w[s] = __builtin_bit_cast(bf16x8, uw);
    }
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int n = 0; n < 32; ++n) {
            int ia, ib;
            if (MODE == 0) { ia = n & 7; ib = (n * 3 + 1) & 7; }
            else if (MODE == 1) { ia = ((n >> 2) + (n & 3)) & 7; ib = (n >> 2) & 7; }
            else if (MODE == 2) { ia = ((n + 1) >> 1) & 7; ib = (n >> 1) & 7; }
            else if (MODE == 3) { ia = (n >> 2) & 7; ib = ((n + 2) >> 2) & 7; }
            else if (MODE == 4) { ia = (n >> 3) & 7; ib = (n >> 3) & 7; }
            else { ia = 0; ib = 0; }
            acc[n & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[ia], x[ib], acc[n & 3], 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}
template <int MODE> void run(float* out) {
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    const int blocks = 512, iters = 100000;
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(512), 0, 0, out, 100);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(a);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(512), 0, 0, out, iters);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    const double flops = (double)blocks * 8 * iters * 32 * 32768.0;
    printf("mode %d: %.3f ms  %.1f TFLOP/s\n", MODE, ms, flops / ms / 1e9);
}
int main() {
    float* out; (void)hipMalloc(&out, 4 * 512 * 2048);
    for (int rep = 0; rep < 2; ++rep) { run<0>(out); run<1>(out); run<2>(out); run<3>(out); run<4>(out); run<5>(out); }
    return 0;
}
