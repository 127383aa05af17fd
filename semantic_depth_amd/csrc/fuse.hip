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

// ---------------------------------------------------------------------------------------------
// ONE pass (SURVEY K16-K19 in a single launch): flip-pair post-processing (when the raw pair is given), back-projection, and
// the ordered gather of both clouds by decoupled look-back -- each block of 8 x 1024 pixels counts its masks first and publishes
// its (road, fence) counts in one 8-byte word, resolves its exclusive prefix from its predecessors' words while it post-processes
// its first sub-tile, and then writes every masked pixel at prefix + rank (thread order == pixel order, so a wave writes one
// contiguous run per class).  The next sub-tile's loads are in flight while the current one is processed.
//   * blocks take a TICKET per frame instead of trusting blockIdx order: every lower-numbered block is then already
//     running, so the look-back cannot wait on a block that was never scheduled (HIP promises no dispatch order)
//   * a look-back word is one naturally aligned 8-byte granule written by ONE relaxed agent-scope store and read by relaxed
//     agent-scope loads (L1-bypassing): data and flag cannot tear, no fence is needed (MI355X_MICROARCH.md, granule hand-off)
//   * words carry the launch's epoch, so the scratch is not cleared between launches; the last ticket holder re-arms the ticket
//   * measured (scripts/fuse_bench.py, 32 frames, 451 MB algorithmic): 173 us against 62 + 15 + 5 + 100 us of the four launches it
//     replaces.  PMC: 51 % of the wave cycles wait on memory, 23 % issue; the stage is bound by the ~640 VALU + 25 scattered
//     store instructions per pixel quad, not by HBM (2.6 of 8 TB/s).  Rejected variants, same speed within 5 %: staging each
//     sub-tile's points in LDS and writing them as coalesced dword runs (165 us), 1024- / 4096-pixel blocks (252 / 170 us: the
//     ticket + look-back cost per block shows below 4096 pixels), closed forms for the interior columns / the sparse Q (kept:
//     they are exact, but the f64 arithmetic was not the limiter).
#ifndef SD_F1_SUB
#define SD_F1_SUB 8
#endif
#ifndef SD_F1_WAVES
#define SD_F1_WAVES 4
#endif
constexpr int F1_SUB = SD_F1_SUB;                       // sub-tiles of 1024 pixels (256 threads x 4 pixels) per block
constexpr int F1_PIX = 1024 * F1_SUB;           // pixels per block: one ticket, one look-back, one published word
__device__ __forceinline__ unsigned long long f1_pack(unsigned epoch, unsigned flag, unsigned r, unsigned f) {
    return ((unsigned long long)epoch << 54) | ((unsigned long long)flag << 52) | ((unsigned long long)r << 26) | (unsigned long long)f;
}
size_t fuse_onepass_scratch_bytes(int B, int H, int W) {
    const size_t nblk = ((size_t)H * W + F1_PIX - 1) / F1_PIX;
    return (size_t)B * nblk * 8 + (size_t)B * 4 + 256;
}
bool fuse_onepass_eligible(const FuseParams& p) {
    return p.W % 4 == 0 && (p.H * p.W) % 4 == 0 && !p.dense && (p.road_xyz || p.fence_xyz) && p.lb_state && p.lb_ticket && p.epoch &&
           (size_t)p.H * p.W < (1u << 26) && !(p.sw & SW_NO_FUSE1);
}

struct F1Quad { float4 L, R; unsigned fw[3]; };
struct __attribute__((packed, aligned(4))) F1Vec3 { float x, y, z; };        // one point: a single 12-byte store
struct __attribute__((packed, aligned(1))) F1Vec3u { unsigned x, y, z; };     // the 12 colour bytes of four points: ONE store at any
                                                                              // byte alignment (gfx9+ global memory accesses may be unaligned)
__device__ __forceinline__ F1Quad f1_load(const FuseParams& p, int b, int npix, int i4, bool in, unsigned masks) {
    F1Quad qd;
    qd.L = qd.R = make_float4(0.f, 0.f, 0.f, 0.f);
    qd.fw[0] = qd.fw[1] = qd.fw[2] = 0u;
    if (!in) return qd;
    const int i = i4 * 4, y = i / p.W, x0 = i - y * p.W;
    if (p.disp_raw) {
        qd.L = *reinterpret_cast<const float4*>(p.disp_raw + ((size_t)(2 * b) * p.H + y) * p.W + x0);
        qd.R = *reinterpret_cast<const float4*>(p.disp_raw + ((size_t)(2 * b + 1) * p.H + y) * p.W + (p.W - 4 - x0));   // R[W-1-x0-3 .. W-1-x0]
    } else {
        qd.L = *reinterpret_cast<const float4*>(p.disp_pp + (size_t)b * npix + i);
    }
    if (p.frames && masks) {
        const unsigned* fp = reinterpret_cast<const unsigned*>(p.frames + ((size_t)b * npix + (size_t)i) * 3);
        qd.fw[0] = fp[0]; qd.fw[1] = fp[1]; qd.fw[2] = fp[2];
    }
    return qd;
}

__global__ __launch_bounds__(256, SD_F1_WAVES) void fuse_onepass_kernel(const FuseParams p, int npix, int nblk) {
    const int b = blockIdx.y, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    __shared__ int s_blk, wr[F1_SUB][4], wf[F1_SUB][4], s_excl[2];
    if (t == 0) s_blk = atomicAdd(&p.lb_ticket[b], 1);
    __syncthreads();
    const int blk = s_blk;
    const unsigned* const road4 = (p.road && p.road_xyz) ? reinterpret_cast<const unsigned*>(p.road + (size_t)b * npix) : nullptr;
    const unsigned* const fence4 = (p.fence && p.fence_xyz) ? reinterpret_cast<const unsigned*>(p.fence + (size_t)b * npix) : nullptr;
    const int q0 = blk * F1_SUB * 256 + t;                      // first pixel quad of this thread; sub-tile u: q0 + 256 u
    // ---- pass 1 over the masks only: the block's counts are what the successors wait for ----
    for (int u = 0; u < F1_SUB; ++u) {
        const int i4 = q0 + 256 * u;
        unsigned mr = 0, mf = 0;
        if (i4 * 4 < npix) { if (road4) mr = road4[i4]; if (fence4) mf = fence4[i4]; }
        int pr, tr, pf, tf;
        wave_prefix_total(nz_bytes(mr), lane, pr, tr);
        wave_prefix_total(nz_bytes(mf), lane, pf, tf);
        if (lane == 0) { wr[u][wave] = tr; wf[u][wave] = tf; }
    }
    __syncthreads();
    int blk_r = 0, blk_f = 0;
    for (int u = 0; u < F1_SUB; ++u) {
        blk_r += wr[u][0] + wr[u][1] + wr[u][2] + wr[u][3];
        blk_f += wf[u][0] + wf[u][1] + wf[u][2] + wf[u][3];
    }
    unsigned long long* const st = p.lb_state + (size_t)b * nblk;
    if (t == 0) {                                              // publish: block 0 knows its inclusive prefix at once
        __hip_atomic_store(&st[blk], f1_pack(p.epoch, blk == 0 ? 2u : 1u, (unsigned)blk_r, (unsigned)blk_f), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
        if (blk == 0) { s_excl[0] = 0; s_excl[1] = 0; }
    }
    const CamDev cam = p.cams[b];
    const double* q = cam.q;
    const bool sparse_q = q[0] == 1.0 && q[1] == 0.0 && q[2] == 0.0 && q[4] == 0.0 && q[5] == -1.0 && q[6] == 0.0 && q[8] == 0.0 && q[9] == 0.0 &&
                          q[10] == 0.0 && q[12] == 0.0 && q[13] == 0.0 && q[15] == 0.0;
    // columns x with l_mask(x) == 0 and r_mask(x) == 0 exactly (ramp_l saturates at 0 from 20 (l - 0.05) >= 1 on): the interior of the image
    int mid_lo = p.W, mid_hi = -1;
    {
        const double step = 1.0 / (double)(p.W - 1);
        int lo = (int)(0.1 * (p.W - 1)) - 2;
        lo = lo < 0 ? 0 : lo;
        while (lo < p.W && !(ramp_l(lo, p.W, step) == 0.0)) ++lo;          // first column whose left ramp is exactly 0
        mid_lo = lo; mid_hi = p.W - 1 - lo;                                   // r_mask is the mirror image
    }
    const float inv_w = 1.0f / (float)p.W;
    int ex_r = 0, ex_f = 0;
    // ---- pass 2, per sub-tile: post-process, back-project the masked pixels, stage them in output order, write contiguous runs;
    //      the next sub-tile's loads are in flight meanwhile.  The look-back is resolved after the first sub-tile's arithmetic. ----
    unsigned nmr = 0, nmf = 0;
    if (q0 * 4 < npix) { if (road4) nmr = road4[q0]; if (fence4) nmf = fence4[q0]; }
    F1Quad nxt = f1_load(p, b, npix, q0, q0 * 4 < npix, nmr | nmf);
    for (int u = 0; u < F1_SUB; ++u) {
        const int i4 = q0 + 256 * u;
        const bool in = i4 * 4 < npix;
        const int i = i4 * 4;
        int y = (int)((float)i * inv_w);                         // row of the quad: float estimate, then exact correction (i < 2^26)
        y -= (y * p.W > i); y += ((y + 1) * p.W <= i); y -= (y * p.W > i);
        if (!in) y = 0;
        const int x0 = i - y * p.W;
        const F1Quad cur = nxt;
        const unsigned mr = nmr, mf = nmf;
        if (u + 1 < F1_SUB) {
            const int j4 = i4 + 256;
            nmr = nmf = 0;
            if (j4 * 4 < npix) { if (road4) nmr = road4[j4]; if (fence4) nmf = fence4[j4]; }
            nxt = f1_load(p, b, npix, j4, j4 * 4 < npix, nmr | nmf);
        }
        float dv[4];
        if (p.disp_raw) {
            // DepthFrame.post_processing, semantic_depth.py:656-664 (as post_process_kernel): L at x, R at W-1-x of the flipped pass
            const float Lv[4] = {cur.L.x, cur.L.y, cur.L.z, cur.L.w}, Rv[4] = {cur.R.w, cur.R.z, cur.R.y, cur.R.x};
            const double step = 1.0 / (double)(p.W - 1);
            // both ramps are exactly 0 between the two 10 %-wide margins: the blend is then 0*L + 0*R + 1*m = m bit for bit (for
            // finite L, R; anything else takes the general expression)
            const bool interior = x0 >= mid_lo && x0 + 3 <= mid_hi;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int x = x0 + k;
                const float m = 0.5f * (Lv[k] + Rv[k]);
                if (interior && fabsf(m) < INFINITY) { dv[k] = m; continue; }
                const double lm = ramp_l(x, p.W, step), rm = ramp_l(p.W - 1 - x, p.W, step);
                dv[k] = (float)((rm * (double)Lv[k] + lm * (double)Rv[k]) + ((1.0 - lm) - rm) * (double)m);
            }
            if (in) *reinterpret_cast<float4*>(p.pp_out + (size_t)b * npix + i) = make_float4(dv[0], dv[1], dv[2], dv[3]);
        } else {
            dv[0] = cur.L.x; dv[1] = cur.L.y; dv[2] = cur.L.z; dv[3] = cur.L.w;
        }
        float X[4], Y[4], Z[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            X[k] = Y[k] = Z[k] = 0.f;
            if (!(((mr | mf) >> (8 * k)) & 0xffu)) continue;
            // disparity = disp_pp * multiplier in float32 (semantic_depth.py:145; seq:146); cv2.reprojectImageTo3D as fuse_write_kernel
            const float dpx = dv[k] * cam.mult;
            const double xd = (double)(x0 + k), yd = (double)y, d = (double)dpx;
            double Wh, n0, n1, n2;
            if (sparse_q && fabsf(dpx) < INFINITY) {
                // Q of make_cam ([1 0 0 -cx; 0 -1 0 cy; 0 0 0 -f; 0 0 1/b 0]) and a finite disparity: the zero products and the
                // additions of +0.0 of the general expression below drop out without changing a bit
                Wh = 0.0 + q[14] * d;
                n0 = xd + q[3];
                n1 = (0.0 - yd) + q[7];
                n2 = q[11];
            } else {
                Wh = ((q[12] * xd + q[13] * yd) + q[14] * d) + q[15];
                n0 = ((q[0] * xd + q[1] * yd) + q[2] * d) + q[3];
                n1 = ((q[4] * xd + q[5] * yd) + q[6] * d) + q[7];
                n2 = ((q[8] * xd + q[9] * yd) + q[10] * d) + q[11];
            }
            const double iW = 1.0 / Wh;                         // Vec3f /= double is a multiply by 1./alpha (core/matx.hpp)
            X[k] = (float)((double)(float)n0 * iW); Y[k] = (float)((double)(float)n1 * iW); Z[k] = (float)((double)(float)n2 * iW);
        }
        if (u == 0) {
            // ---- decoupled look-back by wave 0: exclusive prefix of this block ----
            if (wave == 0 && blk > 0) {
                int look = blk - 1;
                unsigned sum_r = 0, sum_f = 0;
                while (true) {
                    const int idx = look - lane;
                    unsigned long long w = idx >= 0 ? __hip_atomic_load(&st[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : f1_pack(p.epoch, 2u, 0u, 0u);
                    unsigned flag = (unsigned)(w >> 54) == p.epoch ? (unsigned)((w >> 52) & 3u) : 0u;
                    const unsigned long long incl = __ballot(flag == 2u), notready = __ballot(flag == 0u);
                    const int first_incl = incl ? __ffsll((long long)incl) - 1 : 64;
                    const unsigned long long needed = first_incl < 64 ? ((first_incl == 63 ? ~0ull : ((2ull << first_incl) - 1ull))) : ~0ull;
                    if (notready & needed) { __builtin_amdgcn_s_sleep(2); continue; }
                    unsigned r = lane <= first_incl ? (unsigned)((w >> 26) & 0x3ffffffu) : 0u, f = lane <= first_incl ? (unsigned)(w & 0x3ffffffu) : 0u;
#pragma unroll
                    for (int o = 32; o > 0; o >>= 1) { r += __shfl_xor(r, o); f += __shfl_xor(f, o); }
                    sum_r += r; sum_f += f;
                    if (first_incl < 64) break;
                    look -= 64;
                }
                if (lane == 0) {
                    s_excl[0] = (int)sum_r; s_excl[1] = (int)sum_f;
                    __hip_atomic_store(&st[blk], f1_pack(p.epoch, 2u, sum_r + (unsigned)blk_r, sum_f + (unsigned)blk_f), __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            __syncthreads();
            ex_r = s_excl[0]; ex_f = s_excl[1];
            if (blk == nblk - 1 && t == 0) {                    // the last ticket: totals, and re-arm the ticket for the next launch
                if (p.n_road) p.n_road[b] = ex_r + blk_r;
                if (p.n_fence) p.n_fence[b] = ex_f + blk_f;
                p.lb_ticket[b] = 0;
            }
        }
        int pre_r, pre_f, tr_, tf_;
        wave_prefix_total(nz_bytes(mr), lane, pre_r, tr_);
        wave_prefix_total(nz_bytes(mf), lane, pre_f, tf_);
        const int cnt_r = p.road_xyz ? wr[u][0] + wr[u][1] + wr[u][2] + wr[u][3] : 0;
        const int cnt_f = p.fence_xyz ? wf[u][0] + wf[u][1] + wf[u][2] + wf[u][3] : 0;
        if (cnt_r + cnt_f == 0) continue;                       // (uniform over the block)
        // write each masked pixel at (exclusive prefix of the block) + (sub-tiles before) + (rank inside the sub-tile): thread order ==
        // pixel order, so the lanes of a wave write one contiguous run per class.  A point is ONE 12-byte store; a quad whose four
        // pixels are all masked (the interior of a mask blob) writes its 12 colour bytes as one (unaligned) 12-byte store too,
        // B<->R swapped per pixel by four byte permutes.
#pragma unroll
        for (int cls = 0; cls < 2; ++cls) {
            const int cnt = cls ? cnt_f : cnt_r;
            if (cnt == 0) continue;
            F1Vec3* const oxyz = reinterpret_cast<F1Vec3*>(cls ? p.fence_xyz : p.road_xyz) + (size_t)b * p.cap;        // (uniform base)
            uint8_t* const orgb = (cls ? p.fence_rgb : p.road_rgb) ? (cls ? p.fence_rgb : p.road_rgb) + (size_t)b * p.cap * 3 : nullptr;
            const unsigned m4 = cls ? mf : mr;
            unsigned pos = (unsigned)(cls ? ex_f + pre_f : ex_r + pre_r);
            for (int w = 0; w < wave; ++w) pos += (unsigned)(cls ? wf[u][w] : wr[u][w]);
            const bool full = (m4 & 0xffu) && (m4 & 0xff00u) && (m4 & 0xff0000u) && (m4 & 0xff000000u);
            if (full && pos + 4u <= (unsigned)p.cap) {
#pragma unroll
                for (int k = 0; k < 4; ++k) oxyz[pos + k] = F1Vec3{X[k], Y[k], Z[k]};
                if (orgb) {
                    // frame bytes B0 G0 R0 B1 | G1 R1 B2 G2 | R2 B3 G3 R3  ->  R0 G0 B0 R1 | G1 B1 R2 G2 | B2 R3 G3 B3
                    F1Vec3u c;
                    c.x = __builtin_amdgcn_perm(cur.fw[1], cur.fw[0], 0x05000102u);
                    c.y = __builtin_amdgcn_perm(cur.fw[2], __builtin_amdgcn_perm(cur.fw[1], cur.fw[0], 0x07000304u), 0x03040100u);
                    c.z = __builtin_amdgcn_perm(cur.fw[2], cur.fw[1], 0x05060702u);
                    *reinterpret_cast<F1Vec3u*>(orgb + (size_t)pos * 3) = c;
                }
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (!((m4 >> (8 * k)) & 0xffu)) continue;
                    if (pos < (unsigned)p.cap) {
                        oxyz[pos] = F1Vec3{X[k], Y[k], Z[k]};
                        if (orgb) {
                            // bytes 3k .. 3k+2 of the 12 frame bytes: B, G, R -> colours = cv2.cvtColor(frame, BGR2RGB), semantic_depth.py:161
                            const unsigned long long lo64 = ((unsigned long long)cur.fw[1] << 32) | cur.fw[0], hi64 = ((unsigned long long)cur.fw[2] << 32) | cur.fw[1];
                            const unsigned bgr = k < 2 ? (unsigned)(lo64 >> (24 * k)) : (unsigned)(hi64 >> (24 * k - 32));
                            uint8_t* c = orgb + (size_t)pos * 3;
                            c[0] = (uint8_t)(bgr >> 16); c[1] = (uint8_t)(bgr >> 8); c[2] = (uint8_t)bgr;
                        }
                    }
                    ++pos;
                }
            }
            if (cls) ex_f += cnt; else ex_r += cnt;
        }
    }
}

hipError_t launch_fuse(const FuseParams& p, hipStream_t s) {
    const int npix = p.H * p.W;
    const bool gather = p.road_xyz || p.fence_xyz;
    if (fuse_onepass_eligible(p)) {
        const int nblk = (npix + F1_PIX - 1) / F1_PIX;
        hipLaunchKernelGGL(fuse_onepass_kernel, dim3(nblk, p.B), dim3(256), 0, s, p, npix, nblk);
        return hipGetLastError();
    }
    if (p.disp_raw) {           // the three-launch forms read a post-processed map
        hipError_t e = launch_post_process(p.disp_raw, p.pp_out, p.B, p.H, p.W, s);
        if (e != hipSuccess) return e;
        FuseParams q = p;
        q.disp_pp = p.pp_out; q.disp_raw = nullptr;
        return launch_fuse(q, s);
    }
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
