// Fusion stage (gfx950, HBM-bound): flip-pair post-processing, disparity scaling, back-projection,
// BGR->RGB, and ORDER-PRESERVING mask gather of the road / fence point clouds.
// SURVEY.md §2.2 rows K16-K19.  Compiled with -ffp-contract=off: the arithmetic below must round exactly
// like numpy (semantic_depth.py:656-664,676,145) and OpenCV's reprojectImageTo3D (semantic_depth.py:696).
//
// Compaction is three launches (no inter-workgroup hand-off inside a launch):
//   count  : one 256-pixel block per workgroup -> road / fence counts per block      (reads 2 B/pixel)
//   scan   : one workgroup per frame, exclusive scan over the frame's block counts   (tiny)
//   write  : recompute the per-pixel record, rank inside the block with wave ballots, write at base + rank
// Row-major order of points3D[mask] (semantic_depth.py:183-187) is preserved exactly.
#include <cstdlib>
#include "kernels.hpp"

namespace sd {

// l_mask of semantic_depth.py:660-661 for column x: np.linspace(0,1,W)[x] = x*step (last element exactly 1)
__device__ __forceinline__ double ramp_l(int x, int W, double step) {
    double l = (x == W - 1) ? 1.0 : (double)x * step;
    double t = 20.0 * (l - 0.05);
    t = t < 0.0 ? 0.0 : (t > 1.0 ? 1.0 : t);
    return 1.0 - t;
}

__global__ __launch_bounds__(256) void post_process_kernel(const float* __restrict__ raw, float* __restrict__ pp, int B, int H, int W) {
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    long total = (long)B * H * W;
    if (i >= total) return;
    int x = (int)(i % W);
    long row = i / W;
    int b = (int)(row / H);
    int y = (int)(row - (long)b * H);
    const float* Lp = raw + ((long)(2 * b) * H + y) * W;
    const float* Rp = raw + ((long)(2 * b + 1) * H + y) * W;
    const float L = Lp[x];
    const float R = Rp[W - 1 - x];                 // fliplr of the flipped frame's disparity
    const float m = 0.5f * (L + R);                // m_disp stays float32 in the reference
    const double step = 1.0 / (double)(W - 1);
    const double lm = ramp_l(x, W, step);
    const double rm = ramp_l(W - 1 - x, W, step);  // r_mask = fliplr(l_mask)
    const double v = (rm * (double)L + lm * (double)R) + ((1.0 - lm) - rm) * (double)m;
    pp[i] = (float)v;
}
hipError_t launch_post_process(const float* disp_raw, float* disp_pp, int B, int H, int W, hipStream_t s) {
    long total = (long)B * H * W;
    hipLaunchKernelGGL(post_process_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, disp_raw, disp_pp, B, H, W);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
size_t fuse_scratch_bytes(int B, int H, int W) {
    size_t nblk = ((size_t)H * W + 255) / 256;
    return (size_t)B * nblk * 2 * sizeof(int32_t) * 2;
}

__global__ __launch_bounds__(256) void fuse_count_kernel(const uint8_t* __restrict__ road, const uint8_t* __restrict__ fence,
                                                         int npix, int nblk, int32_t* __restrict__ blk_counts) {
    const int b = blockIdx.y, blk = blockIdx.x;
    const int i = blk * 256 + threadIdx.x;
    bool r = false, f = false;
    if (i < npix) {
        r = road && road[(size_t)b * npix + i] != 0;
        f = fence && fence[(size_t)b * npix + i] != 0;
    }
    __shared__ int wr[4], wf[4];
    const unsigned long long mr = __ballot(r), mf = __ballot(f);
    if ((threadIdx.x & 63) == 0) { wr[threadIdx.x >> 6] = __popcll(mr); wf[threadIdx.x >> 6] = __popcll(mf); }
    __syncthreads();
    if (threadIdx.x == 0) {
        int32_t* o = blk_counts + ((size_t)b * nblk + blk) * 2;
        o[0] = wr[0] + wr[1] + wr[2] + wr[3];
        o[1] = wf[0] + wf[1] + wf[2] + wf[3];
    }
}

// one workgroup (1024 threads) per frame: exclusive scan of the block counts of both classes
__global__ __launch_bounds__(1024) void fuse_scan_kernel(const int32_t* __restrict__ blk_counts, int32_t* __restrict__ blk_offsets,
                                                         int nblk, int32_t* __restrict__ n_road, int32_t* __restrict__ n_fence) {
    const int b = blockIdx.x, t = threadIdx.x;
    const int per = (nblk + 1023) / 1024;
    const int32_t* c = blk_counts + (size_t)b * nblk * 2;
    int32_t* o = blk_offsets + (size_t)b * nblk * 2;
    int sr = 0, sf = 0;
    for (int j = 0; j < per; ++j) {
        int k = t * per + j;
        if (k < nblk) { sr += c[k * 2]; sf += c[k * 2 + 1]; }
    }
    __shared__ int pr[1024], pf[1024];
    pr[t] = sr; pf[t] = sf;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {       // Hillis-Steele inclusive scan
        int ar = 0, af = 0;
        if (t >= off) { ar = pr[t - off]; af = pf[t - off]; }
        __syncthreads();
        pr[t] += ar; pf[t] += af;
        __syncthreads();
    }
    int br = pr[t] - sr, bf = pf[t] - sf;            // exclusive prefix of this thread's chunk
    for (int j = 0; j < per; ++j) {
        int k = t * per + j;
        if (k < nblk) {
            o[k * 2] = br; o[k * 2 + 1] = bf;
            br += c[k * 2]; bf += c[k * 2 + 1];
        }
    }
    if (t == 1023) {
        if (n_road) n_road[b] = pr[1023];
        if (n_fence) n_fence[b] = pf[1023];
    }
}

__global__ __launch_bounds__(256) void fuse_write_kernel(const FuseParams p, int npix, int nblk) {
    const int b = blockIdx.y, blk = blockIdx.x;
    const int i = blk * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool in = i < npix;
    bool r = false, f = false;
    float X = 0.f, Y = 0.f, Z = 0.f;
    uint8_t c0 = 0, c1 = 0, c2 = 0;
    if (in) {
        const size_t gi = (size_t)b * npix + i;
        if (p.road) r = p.road[gi] != 0;
        if (p.fence) f = p.fence[gi] != 0;
    }
    if (in && (p.dense || r || f)) {       // the point is only needed where it is stored (three f64 divisions per pixel)
        const size_t gi = (size_t)b * npix + i;
        const int y = i / p.W, x = i - y * p.W;
        const CamDev cam = p.cams[b];
        // disparity = disp_pp * multiplier in float32 (semantic_depth.py:145; seq:146)
        const float dpx = p.disp_pp[gi] * cam.mult;
        // cv2.reprojectImageTo3D [UPSTREAM OpenCV 4.x]: homg = Q*(x,y,d,1) in double, left to right;
        // numerators narrowed to float, multiplied by the double 1/W (matx.hpp operator/=), narrowed again.
        const double xd = (double)x, yd = (double)y, d = (double)dpx;
        const double* q = cam.q;
        const double Wh = ((q[12] * xd + q[13] * yd) + q[14] * d) + q[15];
        const double n0 = ((q[0] * xd + q[1] * yd) + q[2] * d) + q[3];
        const double n1 = ((q[4] * xd + q[5] * yd) + q[6] * d) + q[7];
        const double n2 = ((q[8] * xd + q[9] * yd) + q[10] * d) + q[11];
        const double iW = 1.0 / Wh;               // Vec3f /= double is a multiply by 1./alpha (core/matx.hpp)
        X = (float)((double)(float)n0 * iW);
        Y = (float)((double)(float)n1 * iW);
        Z = (float)((double)(float)n2 * iW);
        if (p.dense) {
            float* dp = p.dense + gi * 3;
            dp[0] = X; dp[1] = Y; dp[2] = Z;
        }
        if (p.frames && (r || f)) {               // colours = cv2.cvtColor(frame, BGR2RGB), semantic_depth.py:161
            const uint8_t* fp = p.frames + gi * 3;
            c0 = fp[2]; c1 = fp[1]; c2 = fp[0];
        }
    }
    if (!p.road_xyz && !p.fence_xyz) return;
    __shared__ int wr[4], wf[4];
    const unsigned long long mr = __ballot(r), mf = __ballot(f);
    if (lane == 0) { wr[wave] = __popcll(mr); wf[wave] = __popcll(mf); }
    __syncthreads();
    const unsigned long long below = (1ull << lane) - 1ull;
    const int32_t* off = p.blk_offsets + ((size_t)b * nblk + blk) * 2;
    if (r && p.road_xyz) {
        int pos = off[0] + __popcll(mr & below);
        for (int w = 0; w < wave; ++w) pos += wr[w];
        if (pos < p.cap) {
            float* o = p.road_xyz + ((size_t)b * p.cap + pos) * 3;
            o[0] = X; o[1] = Y; o[2] = Z;
            if (p.road_rgb) { uint8_t* c = p.road_rgb + ((size_t)b * p.cap + pos) * 3; c[0] = c0; c[1] = c1; c[2] = c2; }
        }
    }
    if (f && p.fence_xyz) {
        int pos = off[1] + __popcll(mf & below);
        for (int w = 0; w < wave; ++w) pos += wf[w];
        if (pos < p.cap) {
            float* o = p.fence_xyz + ((size_t)b * p.cap + pos) * 3;
            o[0] = X; o[1] = Y; o[2] = Z;
            if (p.fence_rgb) { uint8_t* c = p.fence_rgb + ((size_t)b * p.cap + pos) * 3; c[0] = c0; c[1] = c1; c[2] = c2; }
        }
    }
}

// ---- four pixels per thread (W % 4 == 0): 16-byte disparity loads, 4-byte mask loads, 12 bytes of frame per thread; a block
//      is 1024 consecutive pixels.  Same row-major output order: thread order = pixel order, and a thread's four pixels are
//      written in order.  Wave prefix of the 0..4 per-lane counts from three ballots (one per bit of the count).
__device__ __forceinline__ int nz_bytes(unsigned v) { return ((v & 0xffu) != 0) + ((v & 0xff00u) != 0) + ((v & 0xff0000u) != 0) + ((v & 0xff000000u) != 0); }
__device__ __forceinline__ void wave_prefix_total(int c, int lane, int& prefix, int& total) {
    const unsigned long long below = (1ull << lane) - 1ull;
    const unsigned long long b0 = __ballot(c & 1), b1 = __ballot(c & 2), b2 = __ballot(c & 4);
    prefix = __popcll(b0 & below) + 2 * __popcll(b1 & below) + 4 * __popcll(b2 & below);
    total = __popcll(b0) + 2 * __popcll(b1) + 4 * __popcll(b2);
}
__global__ __launch_bounds__(256) void fuse_count4_kernel(const uint8_t* __restrict__ road, const uint8_t* __restrict__ fence,
                                                          int npix, int nblk, int32_t* __restrict__ blk_counts) {
    const int b = blockIdx.y, blk = blockIdx.x;
    const int i4 = blk * 256 + threadIdx.x;                   // pixel quad
    int cr = 0, cf = 0;
    if (i4 * 4 < npix) {
        if (road) cr = nz_bytes(reinterpret_cast<const unsigned*>(road + (size_t)b * npix)[i4]);
        if (fence) cf = nz_bytes(reinterpret_cast<const unsigned*>(fence + (size_t)b * npix)[i4]);
    }
    __shared__ int wr[4], wf[4];
    int pr_, tr, pf_, tf;
    wave_prefix_total(cr, threadIdx.x & 63, pr_, tr);
    wave_prefix_total(cf, threadIdx.x & 63, pf_, tf);
    if ((threadIdx.x & 63) == 0) { wr[threadIdx.x >> 6] = tr; wf[threadIdx.x >> 6] = tf; }
    __syncthreads();
    if (threadIdx.x == 0) {
        int32_t* o = blk_counts + ((size_t)b * nblk + blk) * 2;
        o[0] = wr[0] + wr[1] + wr[2] + wr[3];
        o[1] = wf[0] + wf[1] + wf[2] + wf[3];
    }
}
__global__ __launch_bounds__(256) void fuse_write4_kernel(const FuseParams p, int npix, int nblk) {
    const int b = blockIdx.y, blk = blockIdx.x;
    const int i4 = blk * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool in = i4 * 4 < npix;
    unsigned mr4 = 0, mf4 = 0;
    if (in) {
        if (p.road) mr4 = reinterpret_cast<const unsigned*>(p.road + (size_t)b * npix)[i4];
        if (p.fence) mf4 = reinterpret_cast<const unsigned*>(p.fence + (size_t)b * npix)[i4];
    }
    const int cr = nz_bytes(mr4), cf = nz_bytes(mf4);
    int pre_r, tot_r, pre_f, tot_f;
    wave_prefix_total(cr, lane, pre_r, tot_r);
    wave_prefix_total(cf, lane, pre_f, tot_f);
    __shared__ int wr[4], wf[4];
    if (lane == 0) { wr[wave] = tot_r; wf[wave] = tot_f; }
    __syncthreads();
    if (!in || (!p.dense && !cr && !cf)) return;              // (no barrier below)
    const int32_t* off = p.blk_offsets + ((size_t)b * nblk + blk) * 2;
    int pos_r = off[0] + pre_r, pos_f = off[1] + pre_f;
    for (int w = 0; w < wave; ++w) { pos_r += wr[w]; pos_f += wf[w]; }
    const size_t g4 = (size_t)b * npix + (size_t)i4 * 4;
    const int i = i4 * 4, y = i / p.W, x0 = i - y * p.W;
    const CamDev cam = p.cams[b];
    const float4 dq = *reinterpret_cast<const float4*>(p.disp_pp + g4);
    const float dv[4] = {dq.x, dq.y, dq.z, dq.w};
    unsigned fw[3] = {0u, 0u, 0u};
    if (p.frames) {
        const unsigned* fp = reinterpret_cast<const unsigned*>(p.frames + g4 * 3);
        fw[0] = fp[0]; fw[1] = fp[1]; fw[2] = fp[2];
    }
    const double* q = cam.q;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const bool r = (mr4 >> (8 * k)) & 0xffu, f = (mf4 >> (8 * k)) & 0xffu;
        if (!p.dense && !r && !f) continue;
        // disparity = disp_pp * multiplier in float32 (semantic_depth.py:145; seq:146); cv2.reprojectImageTo3D [UPSTREAM OpenCV
        // 4.x]: homg = Q*(x,y,d,1) in double, left to right; numerators narrowed to float, multiplied by the double 1/W, narrowed again
        const float dpx = dv[k] * cam.mult;
        const double xd = (double)(x0 + k), yd = (double)y, d = (double)dpx;
        const double Wh = ((q[12] * xd + q[13] * yd) + q[14] * d) + q[15];
        const double n0 = ((q[0] * xd + q[1] * yd) + q[2] * d) + q[3];
        const double n1 = ((q[4] * xd + q[5] * yd) + q[6] * d) + q[7];
        const double n2 = ((q[8] * xd + q[9] * yd) + q[10] * d) + q[11];
        const double iW = 1.0 / Wh;                 // Vec3f /= double is a multiply by 1./alpha (core/matx.hpp)
        const float X = (float)((double)(float)n0 * iW), Y = (float)((double)(float)n1 * iW), Z = (float)((double)(float)n2 * iW);
        if (p.dense) { float* dp = p.dense + (g4 + k) * 3; dp[0] = X; dp[1] = Y; dp[2] = Z; }
        // bytes 3k .. 3k+2 of the 12 frame bytes: B, G, R -> colours = cv2.cvtColor(frame, BGR2RGB), semantic_depth.py:161
        const unsigned long long lo64 = ((unsigned long long)fw[1] << 32) | fw[0], hi64 = ((unsigned long long)fw[2] << 32) | fw[1];
        const unsigned bgr = k < 2 ? (unsigned)(lo64 >> (24 * k)) : (unsigned)(hi64 >> (24 * k - 32));
        const uint8_t c0 = (uint8_t)(bgr >> 16), c1 = (uint8_t)(bgr >> 8), c2 = (uint8_t)bgr;
        if (r && p.road_xyz) {
            if (pos_r < p.cap) {
                float* o = p.road_xyz + ((size_t)b * p.cap + pos_r) * 3;
                o[0] = X; o[1] = Y; o[2] = Z;
                if (p.road_rgb) { uint8_t* c = p.road_rgb + ((size_t)b * p.cap + pos_r) * 3; c[0] = c0; c[1] = c1; c[2] = c2; }
            }
            ++pos_r;
        }
        if (f && p.fence_xyz) {
            if (pos_f < p.cap) {
                float* o = p.fence_xyz + ((size_t)b * p.cap + pos_f) * 3;
                o[0] = X; o[1] = Y; o[2] = Z;
                if (p.fence_rgb) { uint8_t* c = p.fence_rgb + ((size_t)b * p.cap + pos_f) * 3; c[0] = c0; c[1] = c1; c[2] = c2; }
            }
            ++pos_f;
        }
    }
}

hipError_t launch_fuse(const FuseParams& p, hipStream_t s) {
    const int npix = p.H * p.W;
    const bool gather = p.road_xyz || p.fence_xyz;
    if (p.W % 4 == 0 && gather && !(p.sw & SW_NO_FUSE4)) {          // four pixels per thread, 1024 per block
        const int nblk4 = (npix + 1023) / 1024;
        hipLaunchKernelGGL(fuse_count4_kernel, dim3(nblk4, p.B), dim3(256), 0, s, p.road, p.fence, npix, nblk4, p.blk_counts);
        hipLaunchKernelGGL(fuse_scan_kernel, dim3(p.B), dim3(1024), 0, s, p.blk_counts, p.blk_offsets, nblk4, p.n_road, p.n_fence);
        hipLaunchKernelGGL(fuse_write4_kernel, dim3(nblk4, p.B), dim3(256), 0, s, p, npix, nblk4);
        return hipGetLastError();
    }
    const int nblk = (npix + 255) / 256;
    if (gather) {
        hipLaunchKernelGGL(fuse_count_kernel, dim3(nblk, p.B), dim3(256), 0, s, p.road, p.fence, npix, nblk, p.blk_counts);
        hipLaunchKernelGGL(fuse_scan_kernel, dim3(p.B), dim3(1024), 0, s, p.blk_counts, p.blk_offsets, nblk, p.n_road, p.n_fence);
    }
    hipLaunchKernelGGL(fuse_write_kernel, dim3(nblk, p.B), dim3(256), 0, s, p, npix, nblk);
    return hipGetLastError();
}

}  // namespace sd
