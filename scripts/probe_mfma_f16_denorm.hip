// does v_mfma_f32_32x32x16_f16 keep fp16 subnormal inputs?  A = all lanes 2^-20 (subnormal in fp16), B = 1.0:
// each output = 16 * 2^-20 = 1.52587890625e-05 if subnormals are honoured, 0 if flushed.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
__global__ void k(float* out) {
    f16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (_Float16)9.5367431640625e-07f; b[e] = (_Float16)1.0f; }
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
    out[threadIdx.x] = acc[0];
}
int main() {
    float* d; (void)hipMalloc(&d, 256);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    float h[64]; (void)hipMemcpy(h, d, 256, hipMemcpyDeviceToHost);
    printf("mfma f16 with subnormal A: out = %.10g (expected 1.52587890625e-05 if kept, 0 if flushed)\n", h[0]);
    return 0;
}
