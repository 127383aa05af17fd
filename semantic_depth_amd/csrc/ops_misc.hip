// HBM-bound network ops around the conv engine (gfx950): input pre-processing, pools, small-N convs
// (score layers, disparity heads), the FCN-8s transposed-conv ladder and its softmax/threshold/argmax head.
// SURVEY.md §2.2 rows K1, K3, K6-K9, K11, K15.
#include <cstdlib>
#include "kernels.hpp"
#include "split_fmt.hpp"

namespace sd {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// Every op below exists for the activation formats f32 NHWC (0), split-bf16 planes (1), ONE fp16 plane (2), [read side: fp16 hi + lo (3)]
// , bf16 x 3 planes (4) and fp16 hi + scaled lo (5, "HS": SD_PREC_F16X2) of split_fmt.hpp (template parameter SPLIT; `plane` = element offset between planes).
// ---------------------------------------------------------------------------------------------
// K1: VGG 'Processing' block [UPSTREAM Udacity vgg]: split (c0,c1,c2), subtract means, concat reversed.
// 4 stored channels (the 4th is zero and meets zero weight rows): conv1_1 gathers whole channel quads.
// ---------------------------------------------------------------------------------------------
template <int SPLIT>
__device__ __forceinline__ void store4(float* base, size_t plane, long quad_index, f32x4 v) {
    if (SPLIT == 5) {              // fp16 hi + scaled lo
        uint2 h, l;
        split4_hs(v, h, l, (sat_ptr_t) nullptr);
        reinterpret_cast<uint2*>(base)[quad_index] = h;
        reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(base) + plane)[quad_index] = l;
    } else if (SPLIT == 4) {              // bf16 x 3: exact
        uint2 h, m, l;
        split4_x3(v, h, m, l);
        uint16_t* b16 = reinterpret_cast<uint16_t*>(base);
        reinterpret_cast<uint2*>(b16)[quad_index] = h;
        reinterpret_cast<uint2*>(b16 + plane)[quad_index] = m;
        reinterpret_cast<uint2*>(b16 + 2 * plane)[quad_index] = l;
    } else if (SPLIT) {
        uint2 h, l;
        split4_t<SPLIT == 2>(v, h, l);
        uint2* hp = reinterpret_cast<uint2*>(base);             // bf16 plane: one uint2 per channel quad
        hp[quad_index] = h;
        if (SPLIT != 2) reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(base) + plane)[quad_index] = l;      // fp16: ONE plane
    } else {
        reinterpret_cast<f32x4*>(base)[quad_index] = v;
    }
}
template <int SPLIT>
__device__ __forceinline__ f32x4 load4(const float* base, size_t plane, long quad_index) {
    if (SPLIT == 5)
        return recon4_hs(reinterpret_cast<const uint2*>(base)[quad_index], reinterpret_cast<const uint2*>(reinterpret_cast<const uint16_t*>(base) + plane)[quad_index]);
    if (SPLIT == 4) {
        const uint16_t* b16 = reinterpret_cast<const uint16_t*>(base);
        return recon4_x3(reinterpret_cast<const uint2*>(b16)[quad_index], reinterpret_cast<const uint2*>(b16 + plane)[quad_index],
                         reinterpret_cast<const uint2*>(b16 + 2 * plane)[quad_index]);
    }
    if (SPLIT) {
        const uint2 h = reinterpret_cast<const uint2*>(base)[quad_index];
        uint2 l = {0u, 0u};
        if (SPLIT != 2) l = reinterpret_cast<const uint2*>(reinterpret_cast<const uint16_t*>(base) + plane)[quad_index];
        if (SPLIT == 3) {          // fp16 hi + lo planes (read side only: the score heads on pool3 / pool4)
            const f32x2_t a = recon2_f16x2(h.x, l.x), b = recon2_f16x2(h.y, l.y);
            return f32x4{a[0], a[1], b[0], b[1]};
        }
        return recon4_t<SPLIT == 2>(h, l);
    }
    return reinterpret_cast<const f32x4*>(base)[quad_index];
}

template <int SPLIT>
__global__ __launch_bounds__(256) void pre_vgg_kernel(const uint8_t* __restrict__ in, float* __restrict__ out, long npix, size_t plane) {
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= npix) return;
    const uint8_t* p = in + i * 3;
    store4<SPLIT>(out, plane, i, (f32x4){(float)p[2] - 103.939f, (float)p[1] - 116.779f, (float)p[0] - 123.68f, 0.f});
}
hipError_t launch_pre_vgg(const uint8_t* frames, float* out, long npix, int split, size_t plane, hipStream_t s) {
    const dim3 grid((unsigned)((npix + 255) / 256));
    if (split == 5) hipLaunchKernelGGL(pre_vgg_kernel<5>, grid, dim3(256), 0, s, frames, out, npix, plane);
    else if (split == 4) hipLaunchKernelGGL(pre_vgg_kernel<4>, grid, dim3(256), 0, s, frames, out, npix, plane);
    else if (split == 2) hipLaunchKernelGGL(pre_vgg_kernel<2>, grid, dim3(256), 0, s, frames, out, npix, plane);
    else if (split) hipLaunchKernelGGL(pre_vgg_kernel<1>, grid, dim3(256), 0, s, frames, out, npix, plane);
    else hipLaunchKernelGGL(pre_vgg_kernel<0>, grid, dim3(256), 0, s, frames, out, npix, plane);
    return hipGetLastError();
}

// monodepth input: frame.astype(f32)/255 and its fliplr, stacked per frame (semantic_depth.py:671-672)
template <int SPLIT>
__global__ __launch_bounds__(256) void pre_mono_kernel(const uint8_t* __restrict__ in, float* __restrict__ out, int B, int H, int W, size_t plane, int raw) {
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    long npix = (long)B * H * W;
    if (i >= npix) return;
    int x = (int)(i % W);
    long row = i / W;               // b*H + y
    int b = (int)(row / H);
    int y = (int)(row - (long)b * H);
    const uint8_t* p = in + i * 3;
    // raw: the pixel values themselves (exact in fp16; the stem's weights carry the 1/255: NetPlan::input_scale)
    const f32x4 v = raw ? f32x4{(float)p[0], (float)p[1], (float)p[2], 0.f} : f32x4{(float)p[0] / 255.0f, (float)p[1] / 255.0f, (float)p[2] / 255.0f, 0.f};
    store4<SPLIT>(out, plane, ((long)(2 * b) * H + y) * W + x, v);
    store4<SPLIT>(out, plane, ((long)(2 * b + 1) * H + y) * W + (W - 1 - x), v);
}
hipError_t launch_pre_mono(const uint8_t* frames, float* out, int B, int H, int W, int split, size_t plane, int raw, hipStream_t s) {
    long npix = (long)B * H * W;
    const dim3 grid((unsigned)((npix + 255) / 256));
    if (split == 5) hipLaunchKernelGGL(pre_mono_kernel<5>, grid, dim3(256), 0, s, frames, out, B, H, W, plane, raw);
    else if (split == 4) hipLaunchKernelGGL(pre_mono_kernel<4>, grid, dim3(256), 0, s, frames, out, B, H, W, plane, raw);
    else if (split == 2) hipLaunchKernelGGL(pre_mono_kernel<2>, grid, dim3(256), 0, s, frames, out, B, H, W, plane, raw);
    else if (split) hipLaunchKernelGGL(pre_mono_kernel<1>, grid, dim3(256), 0, s, frames, out, B, H, W, plane, raw);
    else hipLaunchKernelGGL(pre_mono_kernel<0>, grid, dim3(256), 0, s, frames, out, B, H, W, plane, raw);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// K3: max-pool 2x2 stride 2 (TF SAME on even dims = no padding); K11: zero-pad 1 then 3x3 stride 2 VALID
// ---------------------------------------------------------------------------------------------
template <int SPLIT>
__global__ __launch_bounds__(256) void maxpool2_kernel(const float* __restrict__ x, float* __restrict__ y, int N, int H, int W, int C4,
                                                       size_t plane_in, size_t plane_out) {
    const int Ho = H / 2, Wo = W / 2;
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    long total = (long)N * Ho * Wo * C4;
    if (i >= total) return;
    int c = (int)(i % C4);
    long r = i / C4;
    int ox = (int)(r % Wo); r /= Wo;
    int oy = (int)(r % Ho);
    int n = (int)(r / Ho);
    const long q = (((long)n * H + 2 * oy) * W + 2 * ox) * C4 + c;
    const f32x4 a = load4<SPLIT>(x, plane_in, q), b = load4<SPLIT>(x, plane_in, q + C4);
    const f32x4 d = load4<SPLIT>(x, plane_in, q + (long)W * C4), e = load4<SPLIT>(x, plane_in, q + (long)W * C4 + C4);
    f32x4 m;
#pragma unroll
    for (int j = 0; j < 4; ++j) m[j] = fmaxf(fmaxf(a[j], b[j]), fmaxf(d[j], e[j]));
    store4<SPLIT>(y, plane_out, i, m);
}
hipError_t launch_maxpool2(const float* x, float* y, int N, int H, int W, int C, int split, size_t plane_in, size_t plane_out, hipStream_t s) {
    long total = (long)N * (H / 2) * (W / 2) * (C / 4);
    const dim3 grid((unsigned)((total + 255) / 256));
    if (split == 5) hipLaunchKernelGGL(maxpool2_kernel<5>, grid, dim3(256), 0, s, x, y, N, H, W, C / 4, plane_in, plane_out);
    else if (split == 4) hipLaunchKernelGGL(maxpool2_kernel<4>, grid, dim3(256), 0, s, x, y, N, H, W, C / 4, plane_in, plane_out);
    else if (split == 2) hipLaunchKernelGGL(maxpool2_kernel<2>, grid, dim3(256), 0, s, x, y, N, H, W, C / 4, plane_in, plane_out);
    else if (split) hipLaunchKernelGGL(maxpool2_kernel<1>, grid, dim3(256), 0, s, x, y, N, H, W, C / 4, plane_in, plane_out);
    else hipLaunchKernelGGL(maxpool2_kernel<0>, grid, dim3(256), 0, s, x, y, N, H, W, C / 4, plane_in, plane_out);
    return hipGetLastError();
}

template <int SPLIT>
__global__ __launch_bounds__(256) void maxpool3z_kernel(const float* __restrict__ x, float* __restrict__ y, int N, int H, int W, int C4,
                                                        size_t plane_in, size_t plane_out) {
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    long total = (long)N * Ho * Wo * C4;
    if (i >= total) return;
    int c = (int)(i % C4);
    long r = i / C4;
    int ox = (int)(r % Wo); r /= Wo;
    int oy = (int)(r % Ho);
    int n = (int)(r / Ho);
    f32x4 m = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            int iy = 2 * oy - 1 + dy, ix = 2 * ox - 1 + dx;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};   // the padding is ZERO and takes part in the max (upstream quirk)
            if (iy >= 0 && iy < H && ix >= 0 && ix < W) v = load4<SPLIT>(x, plane_in, (((long)n * H + iy) * W + ix) * C4 + c);
#pragma unroll
            for (int j = 0; j < 4; ++j) m[j] = fmaxf(m[j], v[j]);
        }
    store4<SPLIT>(y, plane_out, i, m);
}
// split planes with C % 8 == 0: one thread per (output pixel, channel octet), 16-byte loads per plane and tap.
// NPL = planes: 1 ONE fp16 plane, 2 bf16 hi + lo, 3 bf16 hi + mid + lo (exact: the max of exact values, split exactly again)
// HS (NPL = 2): the planes are fp16 hi + scaled lo (the max of 22-bit values, split again without loss)
// sub_nmax > 0: the input is stored as 16-channel sub-planes [C / 16][sub_nmax][H][W][16] (TensorDesc::planar16; the output never is)
template <int NPL, bool HS = false>
__global__ __launch_bounds__(256) void maxpool3z_oct_kernel(const float* __restrict__ x, float* __restrict__ y, int N, int H, int W, int C8,
                                                            size_t plane_in, size_t plane_out, int sub_nmax) {
    constexpr bool F16 = NPL == 1;
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    long total = (long)N * Ho * Wo * C8;
    if (i >= total) return;
    int c = (int)(i % C8);
    long r = i / C8;
    int ox = (int)(r % Wo); r /= Wo;
    int oy = (int)(r % Ho);
    int n = (int)(r / Ho);
    const u32x4_t* const xh = reinterpret_cast<const u32x4_t*>(x);
    const u32x4_t* const xl = reinterpret_cast<const u32x4_t*>(reinterpret_cast<const uint16_t*>(x) + plane_in);
    const u32x4_t* const x3 = reinterpret_cast<const u32x4_t*>(reinterpret_cast<const uint16_t*>(x) + 2 * plane_in);
    f32x4 m0 = {-INFINITY, -INFINITY, -INFINITY, -INFINITY}, m1 = m0;
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            int iy = 2 * oy - 1 + dy, ix = 2 * ox - 1 + dx;
            f32x4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = v0;   // the padding is ZERO and takes part in the max (upstream quirk)
            if (iy >= 0 && iy < H && ix >= 0 && ix < W) {
                const long q = sub_nmax ? ((((long)(c >> 1) * sub_nmax + n) * H + iy) * W + ix) * 2 + (c & 1) : (((long)n * H + iy) * W + ix) * C8 + c;
                const u32x4_t h = xh[q], l = F16 ? h : xl[q];
                if constexpr (NPL == 3) {
                    const u32x4_t t = x3[q];
                    v0 = recon4_x3(uint2{h[0], h[1]}, uint2{l[0], l[1]}, uint2{t[0], t[1]});
                    v1 = recon4_x3(uint2{h[2], h[3]}, uint2{l[2], l[3]}, uint2{t[2], t[3]});
                } else if constexpr (HS) {
                    v0 = recon4_hs(uint2{h[0], h[1]}, uint2{l[0], l[1]});
                    v1 = recon4_hs(uint2{h[2], h[3]}, uint2{l[2], l[3]});
                } else {
                    v0 = recon4_t<F16>(uint2{h[0], h[1]}, uint2{l[0], l[1]});
                    v1 = recon4_t<F16>(uint2{h[2], h[3]}, uint2{l[2], l[3]});
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) { m0[j] = fmaxf(m0[j], v0[j]); m1[j] = fmaxf(m1[j], v1[j]); }
        }
    uint2 h0, l0, h1, l1;
    if constexpr (NPL == 3) {
        uint2 t0, t1;
        split4_x3(m0, h0, l0, t0);
        split4_x3(m1, h1, l1, t1);
        reinterpret_cast<u32x4_t*>(reinterpret_cast<uint16_t*>(y) + 2 * plane_out)[i] = (u32x4_t){t0.x, t0.y, t1.x, t1.y};
    } else if constexpr (HS) {
        split4_hs(m0, h0, l0, (sat_ptr_t) nullptr);
        split4_hs(m1, h1, l1, (sat_ptr_t) nullptr);
    } else {
        split4_t<F16>(m0, h0, l0);
        split4_t<F16>(m1, h1, l1);
    }
    reinterpret_cast<u32x4_t*>(y)[i] = (u32x4_t){h0.x, h0.y, h1.x, h1.y};
    if (!F16) reinterpret_cast<u32x4_t*>(reinterpret_cast<uint16_t*>(y) + plane_out)[i] = (u32x4_t){l0.x, l0.y, l1.x, l1.y};
}
hipError_t launch_maxpool3z(const float* x, float* y, int N, int H, int W, int C, int split, size_t plane_in, size_t plane_out, int sub_nmax, hipStream_t s) {
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    if (sub_nmax && !(split && C % 16 == 0)) return hipErrorInvalidValue;      // (sub-planar inputs: the octet kernel only)
    if (split && C % 8 == 0) {
        const long tot8 = (long)N * Ho * Wo * (C / 8);
        const dim3 g8((unsigned)((tot8 + 255) / 256));
        if (split == 5) hipLaunchKernelGGL((maxpool3z_oct_kernel<2, true>), g8, dim3(256), 0, s, x, y, N, H, W, C / 8, plane_in, plane_out, sub_nmax);
        else if (split == 4) hipLaunchKernelGGL(maxpool3z_oct_kernel<3>, g8, dim3(256), 0, s, x, y, N, H, W, C / 8, plane_in, plane_out, sub_nmax);
        else if (split == 2) hipLaunchKernelGGL(maxpool3z_oct_kernel<1>, g8, dim3(256), 0, s, x, y, N, H, W, C / 8, plane_in, plane_out, sub_nmax);
        else hipLaunchKernelGGL(maxpool3z_oct_kernel<2>, g8, dim3(256), 0, s, x, y, N, H, W, C / 8, plane_in, plane_out, sub_nmax);
        return hipGetLastError();
    }
    long total = (long)N * Ho * Wo * (C / 4);
    const dim3 grid((unsigned)((total + 255) / 256));
    if (split == 5) hipLaunchKernelGGL(maxpool3z_kernel<5>, grid, dim3(256), 0, s, x, y, N, H, W, C / 4, plane_in, plane_out);
    else if (split == 4) hipLaunchKernelGGL(maxpool3z_kernel<4>, grid, dim3(256), 0, s, x, y, N, H, W, C / 4, plane_in, plane_out);
    else if (split == 2) hipLaunchKernelGGL(maxpool3z_kernel<2>, grid, dim3(256), 0, s, x, y, N, H, W, C / 4, plane_in, plane_out);
    else if (split) hipLaunchKernelGGL(maxpool3z_kernel<1>, grid, dim3(256), 0, s, x, y, N, H, W, C / 4, plane_in, plane_out);
    else hipLaunchKernelGGL(maxpool3z_kernel<0>, grid, dim3(256), 0, s, x, y, N, H, W, C / 4, plane_in, plane_out);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// K6 / K15: convolution with <= 4 output channels, k in {1,3}, stride 1, zero pad (k-1)/2.
//   THREAD variant: one thread per output pixel, weights [nout][K] staged in LDS
//   WAVE   variant: one wave per output pixel, lanes stride over the channel quads, shuffle reduction
// Input in either format; output f32, or split planes for the 2-channel disparity maps that feed the next iconv.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float smalln_act(float v, int act) {
    if (act == ACT_SIGMOID03) return 0.3f * (1.0f / (1.0f + expf(-v)));
    if (act == ACT_RELU) return v > 0.f ? v : 0.f;
    if (act == ACT_ELU) return fast_elu(v);
    return v;
}

// weights are laid out [nout][K] (K = k*k*C): one broadcast ds_read_b128 + 4 FMAs per output channel and input quad
template <int NOUT, int IN_SPLIT>
__global__ __launch_bounds__(256) void conv_smalln_thread_kernel(const SmallNParams p) {
    extern __shared__ __attribute__((aligned(16))) float wl[];   // [NOUT][K]
    const int K = p.k * p.k * p.C;
    for (int i = threadIdx.x; i < NOUT * K / 4; i += 256) reinterpret_cast<f32x4*>(wl)[i] = reinterpret_cast<const f32x4*>(p.wt)[i];
    __syncthreads();
    long pix = (long)blockIdx.x * 256 + threadIdx.x;
    long npix = (long)p.N * p.H * p.W;
    if (pix >= npix) return;
    int x = (int)(pix % p.W);
    long r = pix / p.W;
    int y = (int)(r % p.H);
    int n = (int)(r / p.H);
    const int pad = (p.k - 1) / 2;
    float acc[NOUT];
#pragma unroll
    for (int j = 0; j < NOUT; ++j) acc[j] = p.bias[j];
    const int C4 = p.C / 4;
    for (int ky = 0; ky < p.k; ++ky) {
        int iy = y + ky - pad;
        if (iy < 0 || iy >= p.H) continue;
        for (int kx = 0; kx < p.k; ++kx) {
            int ix = x + kx - pad;
            if (ix < 0 || ix >= p.W) continue;
            const long q0 = (((long)n * p.H + iy) * p.W + ix) * C4;
            const f32x4* wp = reinterpret_cast<const f32x4*>(wl) + (ky * p.k + kx) * C4;
#pragma unroll 4
            for (int c = 0; c < C4; ++c) {
                const f32x4 v = load4<IN_SPLIT>(p.x, p.in_plane, q0 + c);
#pragma unroll
                for (int j = 0; j < NOUT; ++j) {
                    const f32x4 w = wp[j * (K / 4) + c];
                    acc[j] += v[0] * w[0] + v[1] * w[1] + v[2] * w[2] + v[3] * w[3];
                }
            }
        }
    }
    if (p.out_split) {            // NOUT == 2: one bf16 pair per plane
        unsigned h, l, m = 0u;
        const f32x2_t v2 = {smalln_act(acc[0], p.act), smalln_act(acc[NOUT > 1 ? 1 : 0], p.act)};
        if (p.x3) split2_x3(v2, h, m, l);
        else if (p.out_f16 == 3) split2_hs(v2, h, l, (sat_ptr_t) nullptr);
        else if (p.out_f16) split2_t<true>(v2, h, l);
        else split2_t<false>(v2, h, l);
        const bool two = !p.out_f16 || p.out_f16 == 3;          // a second plane exists (bf16 hi + lo, fp16 hi + scaled lo)
        uint16_t* const o16 = reinterpret_cast<uint16_t*>(p.out);
        if (p.out_c == 8) {       // one zero-padded channel octet per pixel (source of the direct 3x3 kernel)
            reinterpret_cast<u32x4_t*>(o16)[pix] = (u32x4_t){h, 0u, 0u, 0u};
            if (p.x3) {
                reinterpret_cast<u32x4_t*>(o16 + p.out_plane)[pix] = (u32x4_t){m, 0u, 0u, 0u};
                reinterpret_cast<u32x4_t*>(o16 + 2 * p.out_plane)[pix] = (u32x4_t){l, 0u, 0u, 0u};
            } else if (two) reinterpret_cast<u32x4_t*>(o16 + p.out_plane)[pix] = (u32x4_t){l, 0u, 0u, 0u};
        } else {
            reinterpret_cast<unsigned*>(o16)[pix] = h;
            if (p.x3) {
                reinterpret_cast<unsigned*>(o16 + p.out_plane)[pix] = m;
                reinterpret_cast<unsigned*>(o16 + 2 * p.out_plane)[pix] = l;
            } else if (two) reinterpret_cast<unsigned*>(o16 + p.out_plane)[pix] = l;
        }
    } else {
        float* o = p.out + pix * NOUT;
#pragma unroll
        for (int j = 0; j < NOUT; ++j) o[j] = smalln_act(acc[j], p.act);
    }
}

template <int IN_SPLIT>
__global__ __launch_bounds__(256) void conv_smalln_wave_kernel(const SmallNParams p) {
    const int lane = threadIdx.x & 63;
    long pix = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    long npix = (long)p.N * p.H * p.W;
    if (pix >= npix) return;
    int x = (int)(pix % p.W);
    long r = pix / p.W;
    int y = (int)(r % p.H);
    int n = (int)(r / p.H);
    const int pad = (p.k - 1) / 2;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const int C4 = p.C / 4;
    for (int ky = 0; ky < p.k; ++ky) {
        int iy = y + ky - pad;
        if (iy < 0 || iy >= p.H) continue;
        for (int kx = 0; kx < p.k; ++kx) {
            int ix = x + kx - pad;
            if (ix < 0 || ix >= p.W) continue;
            const long q0 = (((long)n * p.H + iy) * p.W + ix) * C4;
            const f32x4* wp = reinterpret_cast<const f32x4*>(p.wt) + (long)(ky * p.k + kx) * C4;
            const int K4 = p.k * p.k * C4;
            for (int c = lane; c < C4; c += 64) {
                const f32x4 v = load4<IN_SPLIT>(p.x, p.in_plane, q0 + c);
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (j < p.nout) {
                        const f32x4 w = wp[(long)j * K4 + c];
                        acc[j] += v[0] * w[0] + v[1] * w[1] + v[2] * w[2] + v[3] * w[3];
                    }
            }
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] += __shfl_xor(acc[j], off);
    if (lane == 0) {
        float* o = p.out + pix * p.nout;
        for (int j = 0; j < p.nout; ++j) o[j] = smalln_act(acc[j] + p.bias[j], p.act);
    }
}

// TILED variant for the 3x3 heads on split-plane inputs (monodepth get_disp at every scale): a workgroup owns an 8 x 32
// pixel tile, DMAs the 10 x 34 halo of 16 channels at a time into LDS (the layout of conv_direct.hip: 32 B per pixel and
// plane, octet slot XOR-swizzled by (pixel >> 3) & 1) and every thread accumulates its pixel's nine taps from LDS; the
// thread kernel above re-reads every input nine times through L1/L2 with 8-byte accesses at pixel stride.
__device__ __forceinline__ void sn_dma16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
constexpr int SN_TH = 8, SN_TW = 32, SN_HW = SN_TW + 2, SN_HH = SN_TH + 2;
constexpr int SN_XI = (SN_HH * SN_HW * 2 + 63) / 64;        // 11 DMA instructions per plane
constexpr int SN_XUNITS = SN_XI * 64;
// NPL = planes of the input: 1 ONE fp16 plane, 2 bf16 hi + lo, 3 bf16 hi + mid + lo (SD_PREC_BF16X3: exact f32 values, f32 FMAs)
// HS (NPL = 2): fp16 hi + scaled lo planes (SD_PREC_F16X2)
template <int NOUT, int NPL, bool HS = false>
__global__ __launch_bounds__(256) void conv_smalln_tile_kernel(const SmallNParams p) {
    constexpr bool F16 = NPL == 1;
    __shared__ __attribute__((aligned(16))) u32x4_t X[NPL * SN_XUNITS];
    extern __shared__ __attribute__((aligned(16))) float wl[];   // [NOUT][9 C]
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int K = 9 * p.C;
    for (int i = t; i < NOUT * K / 4; i += 256) reinterpret_cast<f32x4*>(wl)[i] = reinterpret_cast<const f32x4*>(p.wt)[i];
    const int tiles_x = p.W / SN_TW, tiles_y = (p.H + SN_TH - 1) / SN_TH;
    int bid = blockIdx.x;
    const int tx0 = (bid % tiles_x) * SN_TW; bid /= tiles_x;
    const int ty0 = (bid % tiles_y) * SN_TH;
    const int img = bid / tiles_y;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) void*)X;
    const int pstride = p.in_sub ? 16 : p.C;                     // elements per pixel in one (sub-)plane
    const uint16_t* const img_hi = reinterpret_cast<const uint16_t*>(p.x) + (size_t)img * p.H * p.W * pstride;
    const int row = t >> 5, col = t & 31;
    float acc[NOUT];
#pragma unroll
    for (int j = 0; j < NOUT; ++j) acc[j] = p.bias[j];
    for (int c0 = 0; c0 < p.C; c0 += 16) {
        const int nvalid = (p.C - c0) >= 16 ? 2 : 1;
        __syncthreads();                                   // the previous chunk is consumed (and the weights are staged)
        for (int j = wave; j < NPL * SN_XI; j += 4) {        // wave-uniform: instruction j of [hi plane | lo plane ...]; fp16: ONE plane
            const int pl = j / SN_XI;
            const int u = (j - pl * SN_XI) * 64 + lane;
            const int pix = u >> 1;
            const int oct = (u & 1) ^ ((pix >> 3) & 1);
            const int ry = pix / SN_HW, rx = pix - ry * SN_HW;
            const int gy = ty0 - 1 + ry, gx = tx0 - 1 + rx;
            const bool ok = pix < SN_HH * SN_HW && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W && oct < nvalid;
            const uint16_t* src = img_hi + (size_t)pl * p.in_plane + ((size_t)gy * p.W + gx) * pstride +
                                  (p.in_sub ? (size_t)(c0 >> 4) * p.in_sub : (size_t)c0) + oct * 8;
            sn_dma16(ok ? (const void*)src : p.zero16, lds0 + (unsigned)(j * 1024));
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int lp = (row + tap / 3) * SN_HW + col + tap % 3;
            for (int oct = 0; oct < nvalid; ++oct) {
                const int idx = lp * 2 + (oct ^ ((lp >> 3) & 1));
                const u32x4_t h = X[idx], l = F16 ? h : X[SN_XUNITS + idx];
                f32x4 v0, v1;
                if constexpr (NPL == 3) {
                    const u32x4_t m3 = X[2 * SN_XUNITS + idx];            // (l = the mid plane, m3 = the lo plane: memory order)
                    v0 = recon4_x3(uint2{h[0], h[1]}, uint2{l[0], l[1]}, uint2{m3[0], m3[1]});
                    v1 = recon4_x3(uint2{h[2], h[3]}, uint2{l[2], l[3]}, uint2{m3[2], m3[3]});
                } else if constexpr (HS) {
                    v0 = recon4_hs(uint2{h[0], h[1]}, uint2{l[0], l[1]});
                    v1 = recon4_hs(uint2{h[2], h[3]}, uint2{l[2], l[3]});
                } else {
                    v0 = recon4_t<F16>(uint2{h[0], h[1]}, uint2{l[0], l[1]});
                    v1 = recon4_t<F16>(uint2{h[2], h[3]}, uint2{l[2], l[3]});
                }
#pragma unroll
                for (int j = 0; j < NOUT; ++j) {
                    const f32x4* wp = reinterpret_cast<const f32x4*>(wl + j * K + tap * p.C + c0 + oct * 8);
                    const f32x4 w0 = wp[0], w1 = wp[1];
                    acc[j] += v0[0] * w0[0] + v0[1] * w0[1] + v0[2] * w0[2] + v0[3] * w0[3];
                    acc[j] += v1[0] * w1[0] + v1[1] * w1[1] + v1[2] * w1[2] + v1[3] * w1[3];
                }
            }
        }
    }
    const int y = ty0 + row;
    if (y >= p.H) return;
    const long pix = ((long)img * p.H + y) * p.W + tx0 + col;
    if (p.out_split) {            // NOUT == 2: one bf16 pair per plane
        unsigned h, l, m = 0u;
        const f32x2_t v2 = {smalln_act(acc[0], p.act), smalln_act(acc[NOUT > 1 ? 1 : 0], p.act)};
        if (NPL == 3) split2_x3(v2, h, m, l);
        else if (HS) split2_hs(v2, h, l, (sat_ptr_t) nullptr);
        else if (p.out_f16) split2_t<true>(v2, h, l);
        else split2_t<false>(v2, h, l);
        const bool two = HS || !p.out_f16;
        uint16_t* const o16 = reinterpret_cast<uint16_t*>(p.out);
        if (p.out_c == 8) {
            reinterpret_cast<u32x4_t*>(o16)[pix] = (u32x4_t){h, 0u, 0u, 0u};
            if (NPL == 3) {
                reinterpret_cast<u32x4_t*>(o16 + p.out_plane)[pix] = (u32x4_t){m, 0u, 0u, 0u};
                reinterpret_cast<u32x4_t*>(o16 + 2 * p.out_plane)[pix] = (u32x4_t){l, 0u, 0u, 0u};
            } else if (two) reinterpret_cast<u32x4_t*>(o16 + p.out_plane)[pix] = (u32x4_t){l, 0u, 0u, 0u};
        } else {
            reinterpret_cast<unsigned*>(o16)[pix] = h;
            if (NPL == 3) {
                reinterpret_cast<unsigned*>(o16 + p.out_plane)[pix] = m;
                reinterpret_cast<unsigned*>(o16 + 2 * p.out_plane)[pix] = l;
            } else if (two) reinterpret_cast<unsigned*>(o16 + p.out_plane)[pix] = l;
        }
    } else {
        float* o = p.out + pix * NOUT;
#pragma unroll
        for (int j = 0; j < NOUT; ++j) o[j] = smalln_act(acc[j], p.act);
    }
}

template <int IN_SPLIT>
static void launch_smalln_t(const SmallNParams& p, hipStream_t s) {
    const long npix = (long)p.N * p.H * p.W;
    const int K = p.k * p.k * p.C;
    if (IN_SPLIT && p.k == 3 && p.W % SN_TW == 0 && p.C % 8 == 0 && p.nout <= 2 && p.zero16 && (size_t)K * 4 * p.nout <= 24576 &&
        !(p.sw & SW_NO_SMALLN_TILE)) {
        const dim3 grid((unsigned)((p.W / SN_TW) * ((p.H + SN_TH - 1) / SN_TH) * p.N));
        const size_t lds = (size_t)K * 4 * p.nout;
        if (IN_SPLIT == 5) {
            if (p.nout == 1) hipLaunchKernelGGL((conv_smalln_tile_kernel<1, 2, true>), grid, dim3(256), lds, s, p);
            else hipLaunchKernelGGL((conv_smalln_tile_kernel<2, 2, true>), grid, dim3(256), lds, s, p);
        } else if (IN_SPLIT == 4) {
            if (p.nout == 1) hipLaunchKernelGGL((conv_smalln_tile_kernel<1, 3>), grid, dim3(256), lds, s, p);
            else hipLaunchKernelGGL((conv_smalln_tile_kernel<2, 3>), grid, dim3(256), lds, s, p);
        } else if (IN_SPLIT == 2) {
            if (p.nout == 1) hipLaunchKernelGGL((conv_smalln_tile_kernel<1, 1>), grid, dim3(256), lds, s, p);
            else hipLaunchKernelGGL((conv_smalln_tile_kernel<2, 1>), grid, dim3(256), lds, s, p);
        } else {
            if (p.nout == 1) hipLaunchKernelGGL((conv_smalln_tile_kernel<1, 2>), grid, dim3(256), lds, s, p);
            else hipLaunchKernelGGL((conv_smalln_tile_kernel<2, 2>), grid, dim3(256), lds, s, p);
        }
        return;
    }
    if (K <= 2048) {
        const dim3 grid((unsigned)((npix + 255) / 256));
        const size_t lds = (size_t)K * 4 * p.nout;
        switch (p.nout) {
            case 1: hipLaunchKernelGGL((conv_smalln_thread_kernel<1, IN_SPLIT>), grid, dim3(256), lds, s, p); break;
            case 2: hipLaunchKernelGGL((conv_smalln_thread_kernel<2, IN_SPLIT>), grid, dim3(256), lds, s, p); break;
            case 3: hipLaunchKernelGGL((conv_smalln_thread_kernel<3, IN_SPLIT>), grid, dim3(256), lds, s, p); break;
            default: hipLaunchKernelGGL((conv_smalln_thread_kernel<4, IN_SPLIT>), grid, dim3(256), lds, s, p); break;
        }
    } else {
        hipLaunchKernelGGL(conv_smalln_wave_kernel<IN_SPLIT>, dim3((unsigned)((npix + 3) / 4)), dim3(256), 0, s, p);
    }
}
bool conv_smalln_tiled(int in_split, int k, int W, int C, int nout, unsigned sw) {
    return in_split && k == 3 && W % SN_TW == 0 && C % 8 == 0 && nout <= 2 && (size_t)9 * C * 4 * nout <= 24576 &&
           !(sw & SW_NO_SMALLN_TILE);
}
hipError_t launch_conv_smalln(const SmallNParams& p, hipStream_t s) {
    if (p.in_sub && !(conv_smalln_tiled(p.in_split, p.k, p.W, p.C, p.nout, p.sw) && p.zero16)) return hipErrorInvalidValue;   // sub-planes: tiled kernel only
    if (p.out_split && (p.nout != 2 || p.k * p.k * p.C > 2048)) return hipErrorInvalidValue;
    if (p.in_split && p.x3) {            // bf16 x 3 input: exact f32 arithmetic on the reconstructed values (tiled or per-thread / per-wave)
        if (p.in_sub) return hipErrorInvalidValue;
        launch_smalln_t<4>(p, s);
    } else if (p.in_split && p.f16 == 3) {      // fp16 hi + scaled lo planes (SD_PREC_F16X2): f32 arithmetic on the reconstructed values
        launch_smalln_t<5>(p, s);                // (sub-planar inputs: the tiled kernel, checked above)
    } else if (p.in_split && p.f16 == 2) {      // fp16 hi + lo input: the per-thread / per-wave kernels only
        if (p.in_sub || p.out_split) return hipErrorInvalidValue;
        SmallNParams q = p;
        q.sw |= SW_NO_SMALLN_TILE;
        launch_smalln_t<3>(q, s);
    } else if (p.in_split && p.f16) launch_smalln_t<2>(p, s);
    else if (p.in_split) launch_smalln_t<1>(p, s);
    else launch_smalln_t<0>(p, s);
    return hipGetLastError();
}

// split planes [npix][C] -> f32 [npix][Ctf] (introspection: sd_net_tensor); Ctf < C for zero-padded tensors
// (sub > 0: the tensor is stored as C/16 sub-planes of 16 channels, sub = elements per sub-plane, TensorDesc::planar16)
__global__ __launch_bounds__(256) void unsplit_kernel(const float* __restrict__ x, float* __restrict__ y, long total, int C, int Ctf, size_t plane, size_t sub, int f16) {
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const long pix = i / Ctf;
    const int c = (int)(i - pix * Ctf);
    const uint16_t* h = reinterpret_cast<const uint16_t*>(x) + (sub ? (size_t)(c >> 4) * sub + (size_t)pix * 16 + (c & 15) : (size_t)pix * C + c);
    if (f16 < 0) y[i] = (__uint_as_float((unsigned)h[0] << 16) + __uint_as_float((unsigned)h[plane] << 16)) + __uint_as_float((unsigned)h[2 * plane] << 16);   // bf16 x 3
    else if (f16 == 3) y[i] = (float)__builtin_bit_cast(_Float16, h[0]) + (float)__builtin_bit_cast(_Float16, h[plane]) * (1.f / 2048.f);      // fp16 hi + scaled lo
    else if (f16 == 2) y[i] = (float)__builtin_bit_cast(_Float16, h[0]) + (float)__builtin_bit_cast(_Float16, h[plane]);      // fp16 hi + lo
    else if (f16) y[i] = (float)__builtin_bit_cast(_Float16, h[0]);                   // ONE fp16 plane
    else y[i] = __uint_as_float((unsigned)h[0] << 16) + __uint_as_float((unsigned)h[plane] << 16);
}
hipError_t launch_unsplit(const float* x, float* y, long npix, int C, int Ctf, size_t plane, size_t sub, int f16, hipStream_t s) {
    const long total = npix * Ctf;
    hipLaunchKernelGGL(unsplit_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, x, y, total, C, Ctf, plane, sub, f16);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// K7: conv2d_transpose 4x4 stride 2 SAME, 3 -> 3 channels, + bias + skip (fcn8s/fcn.py:186-204)
//   y[n, 2i+ky-1, 2j+kx-1, o] += x[n,i,j,c] * w[ky,kx,o,c]     (TF kernel layout [kh,kw,out,in])
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void deconv4s2_add_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                            const float* __restrict__ bias, const float* __restrict__ skip,
                                                            float* __restrict__ y, int N, int H, int W) {
    __shared__ float wl[4 * 4 * 9];
    if (threadIdx.x < 144) wl[threadIdx.x] = w[threadIdx.x];
    __syncthreads();
    const int Ho = 2 * H, Wo = 2 * W;
    long pix = (long)blockIdx.x * 256 + threadIdx.x;
    if (pix >= (long)N * Ho * Wo) return;
    int ox = (int)(pix % Wo);
    long r = pix / Wo;
    int oy = (int)(r % Ho);
    int n = (int)(r / Ho);
    float acc[3] = {bias[0], bias[1], bias[2]};
    // oy + 1 = 2*iy + ky, ky in [0,4): iy = (oy+1)>>1 with ky = (oy+1)&1, and iy-1 with ky+2
    const int iyh = (oy + 1) >> 1, kyh = (oy + 1) & 1;
    const int ixh = (ox + 1) >> 1, kxh = (ox + 1) & 1;
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        int iy = iyh - a, ky = kyh + 2 * a;
        if (iy < 0 || iy >= H) continue;
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            int ix = ixh - b, kx = kxh + 2 * b;
            if (ix < 0 || ix >= W) continue;
            const float* xp = x + (((long)n * H + iy) * W + ix) * 3;
            const float* wp = wl + (ky * 4 + kx) * 9;
#pragma unroll
            for (int o = 0; o < 3; ++o)
#pragma unroll
                for (int c = 0; c < 3; ++c) acc[o] += xp[c] * wp[o * 3 + c];
        }
    }
    const float* sp = skip + pix * 3;
    float* yp = y + pix * 3;
#pragma unroll
    for (int o = 0; o < 3; ++o) yp[o] = acc[o] + sp[o];
}
hipError_t launch_deconv4s2_add(const float* x, const float* w, const float* bias, const float* skip, float* y,
                                int N, int H, int W, hipStream_t s) {
    long npix = (long)N * 4 * H * W;
    hipLaunchKernelGGL(deconv4s2_add_kernel, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, s, x, w, bias, skip, y, N, H, W);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// K8 + K9: conv2d_transpose 16x16 stride 8 SAME (3 -> 3) fused with the reference's consumers of 'logits':
//   tf.nn.softmax (semantic_depth.py:551), > 0.5 for road (class 0) / fence (class 1) (:555-556, :563-564),
//   argmax (fcn8s/fcn.py:218-224).  Gather form: each output pixel has 2x2 contributing inputs.
//   oy + 4 = 8*iy + ky, ky in [0,16).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void deconv16s8_head_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                              const float* __restrict__ bias, int N, int H, int W,
                                                              float* __restrict__ logits, uint8_t* __restrict__ road,
                                                              uint8_t* __restrict__ fence, uint8_t* __restrict__ amax) {
    __shared__ float wl[16 * 16 * 9];
    for (int i = threadIdx.x; i < 2304; i += 256) wl[i] = w[i];
    __syncthreads();
    const int Ho = 8 * H, Wo = 8 * W;
    long pix = (long)blockIdx.x * 256 + threadIdx.x;
    if (pix >= (long)N * Ho * Wo) return;
    int ox = (int)(pix % Wo);
    long r = pix / Wo;
    int oy = (int)(r % Ho);
    int n = (int)(r / Ho);
    float acc[3] = {bias[0], bias[1], bias[2]};
    const int iyh = (oy + 4) >> 3, kyh = (oy + 4) & 7;
    const int ixh = (ox + 4) >> 3, kxh = (ox + 4) & 7;
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        int iy = iyh - a, ky = kyh + 8 * a;
        if (iy < 0 || iy >= H) continue;
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            int ix = ixh - b, kx = kxh + 8 * b;
            if (ix < 0 || ix >= W) continue;
            const float* xp = x + (((long)n * H + iy) * W + ix) * 3;
            const float* wp = wl + (ky * 16 + kx) * 9;
#pragma unroll
            for (int o = 0; o < 3; ++o)
#pragma unroll
                for (int c = 0; c < 3; ++c) acc[o] += xp[c] * wp[o * 3 + c];
        }
    }
    if (logits) {
        float* lp = logits + pix * 3;
        lp[0] = acc[0]; lp[1] = acc[1]; lp[2] = acc[2];
    }
    const float mx = fmaxf(acc[0], fmaxf(acc[1], acc[2]));
    const float e0 = expf(acc[0] - mx), e1 = expf(acc[1] - mx), e2 = expf(acc[2] - mx);
    const float sum = e0 + e1 + e2;
    const float p0 = e0 / sum, p1 = e1 / sum, p2 = e2 / sum;
    if (road) road[pix] = p0 > 0.5f;
    if (fence) fence[pix] = p1 > 0.5f;
    if (amax) {
        int am = 0; float best = p0;
        if (p1 > best) { best = p1; am = 1; }
        if (p2 > best) { am = 2; }
        amax[pix] = (uint8_t)am;
    }
}
hipError_t launch_deconv16s8_head(const float* x, const float* w, const float* bias, int N, int H, int W,
                                  float* logits, uint8_t* road, uint8_t* fence, uint8_t* argmax, hipStream_t s) {
    long npix = (long)N * 64 * H * W;
    hipLaunchKernelGGL(deconv16s8_head_kernel, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, s, x, w, bias, N, H, W,
                       logits, road, fence, argmax);
    return hipGetLastError();
}

}  // namespace sd
