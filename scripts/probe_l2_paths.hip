// Round 5: how many bytes per clock and CU do the two L2 -> CU paths deliver, alone and together?
//   (a) LDS-DMA (global_load_lds_dwordx4: 1 KiB per wave-instruction, lands in LDS)           -- what the GEMM kernels stage BOTH operands with
//   (b) plain vector loads (global_load_dwordx4 into VGPRs, 1 KiB per wave-instruction)       -- a weight fragment could come this way, bypassing LDS
// 256 workgroups x 512 threads (one per CU); every workgroup walks the SAME L2-resident buffer (weight-panel-like sharing), `iters` rounds;
// per round a wave issues NA LDS-DMA instructions and NB vector loads (addresses advance through the buffer), then waits for all of them.
// Modes: NA/NB = 8/0 (64 KB per round per CU by DMA: the HS GEMM k-tile today), 4/0 (32 KB), 0/8 (64 KB by vector loads), 4/8 (32 KB DMA + 64 KB
// vector: X by DMA, W fragments straight into registers, each fragment fetched by the two waves that share it), 4/4.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void dma16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

template <int NA, int NB>
__global__ __launch_bounds__(512, 1) void k(const u32x4* __restrict__ buf, unsigned nunits, unsigned* out, long long* cyc, int iters) {
    __shared__ __attribute__((aligned(16))) u32x4 lds[8 * 8 * 64];          // 64 KB
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) void*)lds;
    unsigned acc = 0;
    unsigned pos = (blockIdx.x * 977u) % (nunits / 4096u) * 4096u;          // workgroups start at different places of the shared buffer
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        u32x4 v[NB > 0 ? NB : 1];
#pragma unroll
        for (int i = 0; i < NA; ++i) dma16(buf + ((pos + (wave * NA + i) * 64 + lane) % nunits), lds0 + (unsigned)((wave * 8 + i) * 1024));
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const u32x4* g = buf + ((pos + 2048 + ((wave >> 1) * NB + i) * 64 + lane) % nunits);      // waves 2j, 2j + 1 fetch the same fragment
            v[i] = *g;              // (a compiler-visible load: hipcc counts it itself; an asm load's destination registers are not protected while it is in flight)
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 0; i < NB; ++i) acc += v[i][0] ^ v[i][1] ^ v[i][2] ^ v[i][3];      // (every element: a dead element's register would be reused while the load is in flight)
        pos = (pos + 4096u) % nunits;
        __builtin_amdgcn_s_barrier();
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    if (NA) acc += lds[(t * 7) % (8 * 8 * 64)][0];
    out[blockIdx.x * 512 + t] = acc;
    if (blockIdx.x == 0 && t == 0) cyc[0] = t1 - t0;
}

template <int NA, int NB>
void run(const u32x4* buf, unsigned nunits, unsigned* out, long long* cyc, const char* what) {
    const int iters = 4000;
    hipEvent_t ea, eb; (void)hipEventCreate(&ea); (void)hipEventCreate(&eb);
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL((k<NA, NB>), dim3(256), dim3(512), 0, 0, buf, nunits, out, cyc, 100);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(ea);
        hipLaunchKernelGGL((k<NA, NB>), dim3(256), dim3(512), 0, 0, buf, nunits, out, cyc, iters);
        (void)hipEventRecord(eb); (void)hipEventSynchronize(eb);
        float ms; (void)hipEventElapsedTime(&ms, ea, eb);
        long long c; (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        const double bytes = (double)(NA + NB) * 8 * 1024;          // per round and CU (8 waves)
        printf("%-64s %7.3f ms  %8.1f ticks / round  %6.2f B / tick / CU  (%5.2f TB/s chip-wide; DMA %d KB + vector %d KB per round)\n", what, ms, (double)c / iters,
               bytes * iters / (double)c, bytes * iters * 256 / (ms * 1e-3) / 1e12, NA * 8, NB * 8);
    }
}

int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    const unsigned nunits = 1u << 20;            // 16 MB: L2 / MALL resident, shared by every workgroup
    u32x4* buf; unsigned* out; long long* cyc;
    (void)hipMalloc(&buf, (size_t)nunits * 16); (void)hipMalloc(&out, 4 * 512 * 256); (void)hipMalloc(&cyc, 8);
    (void)hipMemset(buf, 1, (size_t)nunits * 16);
    run<8, 0>(buf, nunits, out, cyc, "LDS-DMA 64 KB per round (the HS GEMM k-tile today)");
    run<4, 0>(buf, nunits, out, cyc, "LDS-DMA 32 KB");
    run<0, 8>(buf, nunits, out, cyc, "vector loads 64 KB (pairs of waves fetch the same KB)");
    run<0, 4>(buf, nunits, out, cyc, "vector loads 32 KB");
    run<4, 8>(buf, nunits, out, cyc, "LDS-DMA 32 KB + vector 64 KB (X by DMA, W fragments into VGPRs)");
    run<4, 4>(buf, nunits, out, cyc, "LDS-DMA 32 KB + vector 32 KB");
    run<6, 0>(buf, nunits, out, cyc, "LDS-DMA 48 KB");
    return 0;
}
