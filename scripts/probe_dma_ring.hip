#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))
// each wave copies tiles of 64x16B through a 3-stage LDS ring with global_load_lds, then sums
__global__ __launch_bounds__(256) void k(const u32x4* __restrict__ src, unsigned* out, int ntiles) {
    __shared__ __attribute__((aligned(16))) u32x4 ring[3][256];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    unsigned acc = 0;
    auto issue = [&](int tile, int stage) {
        const u32x4* g = src + (size_t)tile * 256 + wave * 64 + lane;          // per-lane source
        u32x4* l = &ring[stage][wave * 64];                                     // wave-uniform LDS base (+ lane*16 by hardware)
        __builtin_amdgcn_global_load_lds(GLB_PTR(g), LDS_PTR(l), 16, 0, 0);
    };
    issue(0, 0);
    for (int kt = 0; kt < ntiles; ++kt) {
        if (kt + 1 < ntiles) { issue(kt + 1, (kt + 1) % 3); asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); }
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        // read a DIFFERENT wave's part to prove cross-wave visibility
        const u32x4 v = ring[kt % 3][((wave + 1) & 3) * 64 + lane];
        acc += v[0] + v[1] + v[2] + v[3];
    }
    out[blockIdx.x * 256 + t] = acc;
}
int main() {
    const int ntiles = 50;
    u32x4* src; unsigned* out;
    hipMalloc(&src, sizeof(u32x4) * 256 * ntiles); hipMalloc(&out, 4 * 256);
    unsigned* h = new unsigned[256 * 4 * ntiles];
    for (int i = 0; i < 256 * 4 * ntiles; ++i) h[i] = i * 2654435761u;
    hipMemcpy(src, h, sizeof(u32x4) * 256 * ntiles, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, src, out, ntiles);
    unsigned o[256]; hipMemcpy(o, out, 1024, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int t = 0; t < 256; ++t) {
        int lane = t & 63, wave = t >> 6; unsigned ref = 0;
        for (int kt = 0; kt < ntiles; ++kt) { int u = kt * 256 + ((wave + 1) & 3) * 64 + lane; for (int j = 0; j < 4; ++j) ref += h[u * 4 + j]; }
        bad += ref != o[t];
    }
    printf("dma ring test: %d mismatches\n", bad);
    return bad != 0;
}
