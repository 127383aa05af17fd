// does a chain of DEPENDENT v_mfma_f32_16x16x32_bf16 (same accumulator) issue at the full rate?  1 wave per SIMD (256 threads), NCH independent
// accumulator chains per wave; reports clocks per MFMA (16 = the 4-pass rate)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <int NCH> __global__ __launch_bounds__(256) void k(float* out, int iters, long long* clk) {
    f32x4 acc[NCH];
    for (int a = 0; a < NCH; ++a) acc[a] = f32x4{0.f, 0.f, 0.f, 0.f};
    u32x4 ux = {0x3f803f80u + threadIdx.x, 0x3f813f80u, 0x3f823f80u, 0x3f833f80u}, uw = {0x3e803e80u, 0x3e813e80u, 0x3e823e80u, 0x3e833e80u + threadIdx.x};
    bf16x8 x = __builtin_bit_cast(bf16x8, ux), w = __builtin_bit_cast(bf16x8, uw);
    const long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int a = 0; a < NCH; ++a) acc[a] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, x, acc[a], 0, 0, 0);
    }
    const long long t1 = clock64();
    float s = 0.f;
    for (int a = 0; a < NCH; ++a) s += acc[a][0] + acc[a][1] + acc[a][2] + acc[a][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *clk = t1 - t0;
}
template <int NCH> void run(float* out, long long* clk) {
    const int iters = 2000;
    hipLaunchKernelGGL(k<NCH>, dim3(256), dim3(256), 0, 0, out, 10, clk);
    hipLaunchKernelGGL(k<NCH>, dim3(256), dim3(256), 0, 0, out, iters, clk);
    (void)hipDeviceSynchronize();
    long long c; (void)hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost);
    printf("%d independent chains: %.2f clock64 ticks per MFMA (%lld ticks, %d MFMAs)\n", NCH, (double)c / (iters * 8.0 * NCH), c, iters * 8 * NCH);
}
int main() {
    float* out; long long* clk;
    (void)hipMalloc(&out, 4 * 256 * 256); (void)hipMalloc(&clk, 8);
    run<1>(out, clk); run<2>(out, clk); run<3>(out, clk); run<4>(out, clk); run<6>(out, clk);
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    for (int n : {1, 2, 4}) {
        (void)hipEventRecord(a);
        const int iters = 20000;
        if (n == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(256), 0, 0, out, iters, clk);
        if (n == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(256), 0, 0, out, iters, clk);
        if (n == 4) hipLaunchKernelGGL(k<4>, dim3(256), dim3(256), 0, 0, out, iters, clk);
        (void)hipEventRecord(b); (void)hipEventSynchronize(b);
        float ms; (void)hipEventElapsedTime(&ms, a, b);
        printf("%d chains: %.3f ms for %d MFMAs per wave -> %.1f ns per MFMA\n", n, ms, iters * 8 * n, ms * 1e6 / (iters * 8.0 * n));
    }
    return 0;
}
