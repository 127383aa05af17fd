// Round 5: under the power cap the useful MFMA rate is (P_cap - P_other) / (energy per MFMA) -- does the MFMA SHAPE change the energy per product?
// The same plane data as scripts/probe_mfma_planes.hip (modes 2 / 3: hi / mid / lo planes of dense / post-ReLU values in the six-product mix; HS: fp16 hi + scaled-lo
// planes in the three-product mix), no memory traffic, 8 waves per CU, through
//     shape 0: v_mfma_f32_32x32x16_{bf16,f16}   (8 passes, 16 MACs per accumulator update, 1024 operand elements per 16384 MACs: what every kernel of the engine issues)
//     shape 1: v_mfma_f32_16x16x32_{bf16,f16}   (4 passes, 32 MACs per accumulator update, 1024 operand elements per  8192 MACs)
//   hipcc --offload-arch=gfx950 -O3 scripts/probe_mfma_shapes.hip -o /tmp/probe_mfma_shapes && /tmp/probe_mfma_shapes
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ unsigned hash(unsigned x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
__device__ unsigned short bf16_rne(float v) { unsigned u = __float_as_uint(v); u += 0x7FFFu + ((u >> 16) & 1u); return (unsigned short)(u >> 16); }
__device__ float bf16_f(unsigned short h) { return __uint_as_float((unsigned)h << 16); }
__device__ float gauss(unsigned seed) {
    float s = 0.f;
    for (int i = 0; i < 4; ++i) s += (float)(hash(seed * 4 + i) >> 8) * (1.0f / 16777216.0f) - 0.5f;
    return s * 1.7320508f;
}
constexpr int SETS = 4;
template <int SHAPE> struct Acc { typedef f32x16 t; static constexpr int N = 16, NACC = 4; };
template <> struct Acc<1> { typedef f32x4 t; static constexpr int N = 4, NACC = 8; };
template <int SHAPE, bool F16> __device__ __forceinline__ typename Acc<SHAPE>::t mma(u32x4 a, u32x4 b, typename Acc<SHAPE>::t c) {
    if constexpr (SHAPE == 0) {
        if constexpr (F16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
        else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    } else {
        if constexpr (F16) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
        else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    }
}
// mode 2: dense values, 3: post-ReLU activations.  F16: HS planes (hi = fp16(v), lo = fp16((v - hi) 2^11); weights fp16 hi + lo of w 2^12), three products
template <int SHAPE, bool F16>
__global__ __launch_bounds__(512) void k(float* out, int iters, int mode) {
    typedef typename Acc<SHAPE>::t acc_t;
    constexpr int NA = Acc<SHAPE>::NACC;
    acc_t acc[NA];
    for (int a = 0; a < NA; ++a) for (int r = 0; r < Acc<SHAPE>::N; ++r) acc[a][r] = 0.f;
    u32x4 x[3][SETS], w[3][SETS];
    for (int s = 0; s < SETS; ++s) {
        unsigned short xp[3][8], wp[3][8];
        for (int e = 0; e < 8; ++e) {
            const unsigned sx = (blockIdx.x * 512 + threadIdx.x) * 64 + s * 8 + e, sw = 0x9e3779b9u + threadIdx.x * 64 + s * 8 + e;
            float xv = gauss(sx), wv = 0.05f * gauss(sw);
            if (mode == 3 && xv < 0.f) xv = 0.f;
            if constexpr (F16) {
                _Float16 h = (_Float16)xv, l = (_Float16)((xv - (float)h) * 2048.f);
                xp[0][e] = __builtin_bit_cast(unsigned short, h); xp[1][e] = __builtin_bit_cast(unsigned short, l); xp[2][e] = 0;
                const float ws = wv * 4096.f;
                _Float16 wh = (_Float16)ws, wl = (_Float16)(ws - (float)wh), wh2 = (_Float16)((float)wh * (1.f / 2048.f));
                wp[0][e] = __builtin_bit_cast(unsigned short, wh); wp[1][e] = __builtin_bit_cast(unsigned short, wl); wp[2][e] = __builtin_bit_cast(unsigned short, wh2);
            } else {
                xp[0][e] = bf16_rne(xv); float r = xv - bf16_f(xp[0][e]); xp[1][e] = bf16_rne(r); xp[2][e] = bf16_rne(r - bf16_f(xp[1][e]));
                wp[0][e] = bf16_rne(wv); r = wv - bf16_f(wp[0][e]); wp[1][e] = bf16_rne(r); wp[2][e] = bf16_rne(r - bf16_f(wp[1][e]));
            }
        }
        for (int pl = 0; pl < 3; ++pl)
            for (int d = 0; d < 4; ++d) { x[pl][s][d] = xp[pl][2 * d] | ((unsigned)xp[pl][2 * d + 1] << 16); w[pl][s][d] = wp[pl][2 * d] | ((unsigned)wp[pl][2 * d + 1] << 16); }
    }
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < SETS; ++u) {
            if constexpr (F16) {        // W_lo x X_hi | W_hi x (X_lo [against w_hi 2^-11], X_hi)
#pragma unroll
                for (int a = 0; a < NA; ++a) acc[a] = mma<SHAPE, true>(w[1][(u + a) % SETS], x[0][u], acc[a]);
#pragma unroll
                for (int a = 0; a < NA; ++a) acc[a] = mma<SHAPE, true>(w[2][(u + a) % SETS], x[1][u], acc[a]);
#pragma unroll
                for (int a = 0; a < NA; ++a) acc[a] = mma<SHAPE, true>(w[0][(u + a) % SETS], x[0][u], acc[a]);
            } else {
#pragma unroll
                for (int a = 0; a < NA; ++a) acc[a] = mma<SHAPE, false>(w[2][(u + a) % SETS], x[0][u], acc[a]);
#pragma unroll
                for (int a = 0; a < NA; ++a) acc[a] = mma<SHAPE, false>(w[1][(u + a) % SETS], x[1][u], acc[a]);
#pragma unroll
                for (int a = 0; a < NA; ++a) acc[a] = mma<SHAPE, false>(w[1][(u + a) % SETS], x[0][u], acc[a]);
#pragma unroll
                for (int a = 0; a < NA; ++a) acc[a] = mma<SHAPE, false>(w[0][(u + a) % SETS], x[2][u], acc[a]);
#pragma unroll
                for (int a = 0; a < NA; ++a) acc[a] = mma<SHAPE, false>(w[0][(u + a) % SETS], x[1][u], acc[a]);
#pragma unroll
                for (int a = 0; a < NA; ++a) acc[a] = mma<SHAPE, false>(w[0][(u + a) % SETS], x[0][u], acc[a]);
            }
        }
    }
    float s = 0.f;
    for (int a = 0; a < NA; ++a) for (int r = 0; r < Acc<SHAPE>::N; ++r) s += acc[a][r];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}
template <int SHAPE, bool F16>
void run(float* out, int mode, const char* what) {
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    const int blocks = 512, iters = 12000;
    hipLaunchKernelGGL((k<SHAPE, F16>), dim3(blocks), dim3(512), 0, 0, out, 100, mode);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(a);
    hipLaunchKernelGGL((k<SHAPE, F16>), dim3(blocks), dim3(512), 0, 0, out, iters, mode);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    const int prods = F16 ? 3 : 6;
    const double flops = (double)blocks * 8 * iters * (SETS * prods * Acc<SHAPE>::NACC) * (SHAPE == 0 ? 32768.0 : 16384.0);
    printf("%-28s mode %d: %8.3f ms  %7.1f TFLOP/s of MFMA products = %6.1f TFLOP/s of fp32-grade work\n", what, mode, ms, flops / ms / 1e9, flops / ms / 1e9 / prods);
}
int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    float* out; (void)hipMalloc(&out, 4 * 512 * 2048);
    for (int rep = 0; rep < 2; ++rep)
        for (int mode = 2; mode <= 3; ++mode) {
            run<0, false>(out, mode, "v_mfma_f32_32x32x16_bf16");
            run<1, false>(out, mode, "v_mfma_f32_16x16x32_bf16");
            run<0, true>(out, mode, "v_mfma_f32_32x32x16_f16 (HS)");
            run<1, true>(out, mode, "v_mfma_f32_16x16x32_f16 (HS)");
        }
    return 0;
}
