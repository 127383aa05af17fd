// Input stage: cv2.resize(frame, (W, H), INTER_CUBIC) for uint8 HWC frames on the GPU.
// The per-axis tap indices and the int16 fixed-point weights (cvRound(w * 2048), A = -0.75) are computed on the host
// exactly like OpenCV's scalar path (capi.cpp: resize_tables) and uploaded; a thread produces one output pixel (all
// channels) as sum_ky beta[ky] * (sum_kx alpha[kx] * S) in int32 -- the same integers as OpenCV's horizontal + vertical
// passes -- and (sum + 2^21) >> 22, saturated to u8.  HBM-bound: 16 source pixels (L2-resident) per output pixel.
#include "kernels.hpp"

namespace sd {

__global__ __launch_bounds__(256) void resize_cubic_u8_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, int B, int sh, int sw,
                                                              int dh, int dw, int C, const int* __restrict__ xi, const int* __restrict__ xa,
                                                              const int* __restrict__ yi, const int* __restrict__ ya) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    const long total = (long)B * dh * dw;
    if (i >= total) return;
    const int x = (int)(i % dw);
    const long r = i / dw;
    const int y = (int)(r % dh);
    const int b = (int)(r / dh);
    int ix[4], ax[4], iy[4], ay[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { ix[k] = xi[x * 4 + k]; ax[k] = xa[x * 4 + k]; iy[k] = yi[y * 4 + k]; ay[k] = ya[y * 4 + k]; }
    const uint8_t* s = src + (size_t)b * sh * sw * C;
    for (int c = 0; c < C; ++c) {
        int acc = 0;
#pragma unroll
        for (int ky = 0; ky < 4; ++ky) {
            const uint8_t* row = s + (size_t)iy[ky] * sw * C + c;
            int h = 0;
#pragma unroll
            for (int kx = 0; kx < 4; ++kx) h += (int)row[(size_t)ix[kx] * C] * ax[kx];
            acc += h * ay[ky];
        }
        int v = (acc + (1 << 21)) >> 22;
        v = v < 0 ? 0 : (v > 255 ? 255 : v);
        dst[(size_t)i * C + c] = (uint8_t)v;
    }
}

hipError_t launch_resize_cubic_u8(const uint8_t* src, uint8_t* dst, int B, int sh, int sw, int dh, int dw, int C, const int* xi, const int* xa,
                                  const int* yi, const int* ya, hipStream_t s) {
    const long total = (long)B * dh * dw;
    hipLaunchKernelGGL(resize_cubic_u8_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, src, dst, B, sh, sw, dh, dw, C, xi, xa, yi, ya);
    return hipGetLastError();
}

}  // namespace sd
