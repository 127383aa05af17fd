// Stem convolutions of the split engine: the layers that read the 3-channel network input (stored as 4 channels, 8 B per
// pixel and plane) -- VGG conv1_1 (3x3), monodepth conv1 (7x7, stride 2 for resnet50 / stride 1 and 2 for vgg).
//
// im2col over 4-channel pixels makes every 32-wide k-tile a gather of eight different taps; the generic register-staged
// kernel spends its time in that gather (conv1_1 43 TF/s, enc/conv1 105 TF/s).  Here the whole weight matrix (<= 57 KiB)
// stays in LDS for the life of a persistent workgroup, the input halo of an output tile (8 or 16 rows x 32 columns) is
// copied once into LDS, and a lane builds its MFMA B fragment (8 consecutive k = two taps x 4 channels) with two
// ds_read_b64 per plane straight from the halo.  The next tile's halo travels global -> registers under the MFMAs.
// Same weight image ([K/8][CoutPad][8] per plane, K = (tap, channel)), same epilogue as the other conv kernels.
#include <cstdlib>
#include "kernels.hpp"
#include "split_fmt.hpp"

namespace sd {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int ST_TW = 32;              // output tile width
constexpr int ST_MAXPIX = 1536;        // halo pixels: 21 x 69 (7x7 stride 2, 8 rows) = 1449, 22 x 38 (7x7 stride 1, 16 rows) = 836
// H2 (SD_PREC_F16X2): fp16 hi + scaled lo input planes x fp16 hi + lo weights, three fp16 products (the x_lo one against w_hi * 2^-11), the
// accumulator times ConvParams::alpha, HS output planes (split_fmt.hpp "HS")
template <int NB, int RW, bool F16, bool X3 = false, bool H2 = false>
__global__ __launch_bounds__(512, 1) void conv_stem_kernel(const ConvParams p) {
    static_assert(!H2 || (!F16 && !X3), "H2 stages like the bf16 x 2 form");
    constexpr int NPL = X3 ? 3 : 2;              // planes per operand (X3: bf16 hi, mid, lo -- SD_PREC_BF16X3, six products per product)
    constexpr int ST_PF = NPL * ST_MAXPIX / 512; // halo units (pixel, plane) prefetched per thread
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int TH = 8 * RW, CP = 32 * NB;                 // tile rows, padded output channels (= Cout)
    constexpr int ROW = 64 * NB + 16;                        // epilogue slab row (one plane of 32 pixels per wave)
    const int k = p.kh, s = p.stride, taps = k * k;
    const int IH = (TH - 1) * s + k, IW = (ST_TW - 1) * s + k, npix = IH * IW;
    const int wunits = (p.Kpad / 8) * CP;                    // 16-B units per weight plane
    uint2* const Xh = reinterpret_cast<uint2*>(smem);        // [plane][IH][IW] 4 channels
    uint2* const Xl = Xh + ST_MAXPIX;                        // (X3: the mid plane; the lo plane follows)
    u32x4* const Wh = reinterpret_cast<u32x4*>(Xh + NPL * ST_MAXPIX);
    u32x4* const Wl = Wh + wunits;
    unsigned char* const slab = reinterpret_cast<unsigned char*>(Wh + NPL * wunits);    // (every weight plane in either format)

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int frow = lane & 31, fk = lane >> 5;
    // halo offsets of the two taps of (k-step, k half), -1 = past the kernel (zero weights there): computed once, not with two runtime
    // divisions per k-step and tile (the VALU work beside the MFMAs is paid in full: profiles/r04_mfma_valu_overlap_probe.txt)
    __shared__ int2 otab[64];
    if (t < 64) {
        const int t0 = 4 * (t >> 1) + 2 * (t & 1), t1 = t0 + 1;
        otab[t] = int2{t0 < taps ? (t0 / k) * IW + (t0 % k) : -1, t1 < taps ? (t1 / k) * IW + (t1 % k) : -1};
    }
    {   // weights: the LDS image is the global image
        const u32x4* g = reinterpret_cast<const u32x4*>(p.wt);
        const size_t wplane = (size_t)(p.Kpad / 8) * p.CoutPad;
        for (int i = t; i < wunits; i += 512) {
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl) Wh[pl * wunits + i] = g[pl * wplane + i];
        }
    }
    f32x4 bias[4 * NB];
#pragma unroll
    for (int r4 = 0; r4 < 4 * NB; ++r4) bias[r4] = *reinterpret_cast<const f32x4*>(p.bias + 8 * r4 + 4 * fk);

    const int tiles_x = p.Wout / ST_TW, tiles_y = (p.Hout + TH - 1) / TH;
    const int total = tiles_x * tiles_y * p.N;
    const uint2* const src = reinterpret_cast<const uint2*>(p.src0);          // 4 channels = one uint2 per pixel and plane
    const size_t splane = p.src0_plane / 4;                                   // in pixels
    struct Tile { int img, ty0, tx0; };
    auto tile_of = [&](int tid) {
        if ((total & 7) == 0) tid = (tid & 7) * (total >> 3) + (tid >> 3);     // neighbouring tiles on one XCD
        Tile r;
        const int bx = tid % tiles_x; tid /= tiles_x;
        r.tx0 = bx * ST_TW; r.ty0 = (tid % tiles_y) * TH; r.img = tid / tiles_y;
        return r;
    };
    uint2 pf[ST_PF];
    int pgeo[ST_PF];            // halo units of this thread (the same for every tile): ry | rx << 8 | plane << 16 | valid << 20
#pragma unroll
    for (int i = 0; i < ST_PF; ++i) {
        const int u = t + 512 * i;
        const int pl = u >= 2 * npix ? 2 : u >= npix ? 1 : 0;
        const int px = u - pl * npix;
        const int ry = px / IW, rx = px - ry * IW;
        pgeo[i] = ry | (rx << 8) | (pl << 16) | ((u < (F16 ? 1 : NPL) * npix ? 1 : 0) << 20);      // fp16 input: ONE plane
    }
    auto prefetch = [&](const Tile& tl) {                    // halo unit u = plane * npix + pixel
        const int gy0 = tl.ty0 * s - p.pad, gx0 = tl.tx0 * s - p.pad;
        const uint2* const isrc = src + (size_t)tl.img * p.Hin * p.Win;
#pragma unroll
        for (int i = 0; i < ST_PF; ++i) {
            const int gy = gy0 + (pgeo[i] & 0xff), gx = gx0 + ((pgeo[i] >> 8) & 0xff), pl = (pgeo[i] >> 16) & 3;
            uint2 v = {0u, 0u};
            if (((pgeo[i] >> 20) & 1) && (unsigned)gy < (unsigned)p.Hin && (unsigned)gx < (unsigned)p.Win)
                v = isrc[(size_t)pl * splane + (size_t)(gy * p.Win + gx)];
            pf[i] = v;
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int i = 0; i < ST_PF; ++i) {
            const int u = t + 512 * i;
            if (u < npix) Xh[u] = pf[i];
            else if (u < 2 * npix) Xl[u - npix] = pf[i];
            else if (X3 && u < 3 * npix) Xh[2 * ST_MAXPIX + u - 2 * npix] = pf[i];
        }
    };

    int tid = blockIdx.x;
    if (tid >= total) return;
    Tile cur = tile_of(tid);
    prefetch(cur);
    const int ksteps = (taps * 4 + 15) / 16;
    for (; tid < total; tid += gridDim.x) {
        __syncthreads();                                     // previous tile: fragments and slab reads are done
        commit();
        __syncthreads();
        const bool has_next = tid + (int)gridDim.x < total;
        Tile nxt = cur;
        if (has_next) { nxt = tile_of(tid + gridDim.x); prefetch(nxt); }

        f32x16 acc[RW][NB];
#pragma unroll
        for (int a = 0; a < RW; ++a)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][nb][r] = 0.f;
        for (int ks = 0; ks < ksteps; ++ks) {
            // this lane's two taps (k = 16 ks + 8 fk .. + 7 = taps 4 ks + 2 fk, + 1); taps past the kernel meet zero weights
            const int2 ot = otab[2 * ks + fk];
            const bool has0 = ot.x >= 0, has1 = ot.y >= 0;
            const int o0 = has0 ? ot.x : 0, o1 = has1 ? ot.y : 0;
            u32x4 xh[RW], xl[RW], xm[RW];        // (X3: xl = the mid plane, xm = the lo plane -- named by their position in memory)
#pragma unroll
            for (int a = 0; a < RW; ++a) {
                const int base = ((RW * wave + a) * s) * IW + frow * s;
                const uint2 h0 = Xh[base + o0], h1 = Xh[base + o1], l0 = Xl[base + o0], l1 = Xl[base + o1];
                xh[a] = u32x4{h0.x, h0.y, has1 ? h1.x : 0u, has1 ? h1.y : 0u};
                xl[a] = u32x4{l0.x, l0.y, has1 ? l1.x : 0u, has1 ? l1.y : 0u};
                if constexpr (X3) {
                    const uint2 m0 = Xh[2 * ST_MAXPIX + base + o0], m1 = Xh[2 * ST_MAXPIX + base + o1];
                    xm[a] = u32x4{m0.x, m0.y, has1 ? m1.x : 0u, has1 ? m1.y : 0u};
                } else xm[a] = xl[a];
                if (!has0) { xh[a] = u32x4{0u, 0u, 0u, 0u}; xl[a] = xh[a]; xm[a] = xh[a]; }
            }
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                const int wi = (2 * ks + fk) * CP + nb * 32 + frow;
                const u32x4 wh = Wh[wi];
                const u32x4 wl = Wl[wi];
                if constexpr (X3) {
                    const u32x4 w3 = Wh[2 * wunits + wi];     // planes in memory order: wh = hi, wl = mid, w3 = lo; xh, xl (mid), xm (lo)
#pragma unroll
                    for (int pr = 0; pr < 6; ++pr) {          // hi*lo, lo*hi, mid*mid, hi*mid, mid*hi, hi*hi
#pragma unroll
                        for (int a = 0; a < RW; ++a) {
                            const u32x4 wv = pr == 0 ? w3 : (pr == 2 || pr == 3) ? wl : wh;
                            const u32x4 xv = pr == 1 ? xm[a] : (pr == 2 || pr == 4) ? xl[a] : xh[a];
                            acc[a][nb] = mfma_frag<false>(wv, xv, acc[a][nb]);
                        }
                    }
                } else {
                const u32x4 whs = H2 ? hs_wscaled(wh) : wh;   // the weight operand of the x_lo product
#pragma unroll
                for (int pr = 0; pr < 3; ++pr) {              // x_hi*w_lo, x_lo*w_hi, x_hi*w_hi; an fp16 input has no lo plane
                    if (F16 && pr == 1) continue;
#pragma unroll
                    for (int a = 0; a < RW; ++a)
                        acc[a][nb] = mfma_frag<F16 || H2>(pr == 0 ? wl : pr == 1 ? whs : wh, pr == 1 ? xl[a] : xh[a], acc[a][nb]);
                }
                }
            }
        }

        // ---- epilogue: bias + activation, split once, LDS transpose one plane at a time, 16-byte runs per pixel ----
        auto epilogue = [&](auto tag, auto otag) {
            constexpr int ACT = decltype(tag)::value;
            constexpr int OF = decltype(otag)::value;          // 0 bf16 hi + lo, 1 ONE fp16 plane, 3 fp16 hi + scaled lo
            constexpr bool O16 = OF == 1;
            constexpr int SEGS = 4 * NB, PPP = 64 / SEGS;
            unsigned char* sh = slab + wave * (32 * ROW);
            const int seg = lane % SEGS, prow = lane / SEGS;
            uint16_t* const out_hi = reinterpret_cast<uint16_t*>(p.out);
#pragma unroll
            for (int a = 0; a < RW; ++a) {
                const int y = cur.ty0 + RW * wave + a;
                uint2 hh[4 * NB], ll[4 * NB], mm[4 * NB];
#pragma unroll
                for (int r4 = 0; r4 < 4 * NB; ++r4) {
                    const int nb = r4 >> 2, q = r4 & 3;
                    f32x4 v = {acc[a][nb][4 * q], acc[a][nb][4 * q + 1], acc[a][nb][4 * q + 2], acc[a][nb][4 * q + 3]};
                    if constexpr (H2) v = v * p.alpha + bias[r4]; else v += bias[r4];
                    if constexpr (X3) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = act_x3<ACT>(v[r]);
                        split4_x3(v, hh[r4], ll[r4], mm[r4]);          // hh = hi, ll = mid, mm = lo (memory order)
                    } else {
                        v = act_split4<ACT>(v);
                        split4_fmt<OF>(v, hh[r4], ll[r4], p.sat);
                        mm[r4] = ll[r4];
                    }
                }
#pragma unroll
                for (int pl = 0; pl < (X3 ? 3 : O16 ? 1 : 2); ++pl) {      // fp16 outputs: the hi plane only
#pragma unroll
                    for (int r4 = 0; r4 < 4 * NB; ++r4)
                        *reinterpret_cast<uint2*>(sh + frow * ROW + (8 * r4 + 4 * fk) * 2) = pl == 2 ? mm[r4] : pl ? ll[r4] : hh[r4];
                    __builtin_amdgcn_wave_barrier();
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                    for (int ps = 0; ps < 32 / PPP; ++ps) {
                        const int pix = ps * PPP + prow;
                        const u32x4 v = *reinterpret_cast<const u32x4*>(sh + pix * ROW + seg * 16);
                        if (y < p.Hout) {
                            const size_t px = (size_t)(cur.img * p.Hout + y) * p.Wout + cur.tx0 + pix;
                            uint16_t* o = p.out_planar16 ? out_hi + ((size_t)(seg >> 1) * p.Nmax * p.Hout * p.Wout + px) * 16 + (seg & 1) * 8
                                                         : out_hi + px * p.Cout + seg * 8;
                            *reinterpret_cast<u32x4*>(o + pl * p.out_plane) = v;
                        }
                    }
                    __builtin_amdgcn_wave_barrier();
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                }
            }
        };
        auto ep = [&](auto tag) {
            if constexpr (H2) epilogue(tag, IntTag<3>{});
            else { if (p.out_f16) epilogue(tag, IntTag<1>{}); else epilogue(tag, IntTag<0>{}); }
        };
        if (p.act == ACT_RELU) ep(ActTag<ACT_RELU>{});
        else if (p.act == ACT_ELU) ep(ActTag<ACT_ELU>{});
        else ep(ActTag<ACT_NONE>{});
        cur = nxt;
    }
}

// eligible: split engine, one 4-channel source without upsample, odd k <= 7, stride 1 or 2, Cout 32 or 64 unpadded
bool conv_stem_eligible(const ConvParams& p) {
    if (!p.src0 || p.nsrc != 1 || p.Ctot != 4 || p.kh != p.kw || !(p.kh & 1) || p.kh > 7 || (p.stride != 1 && p.stride != 2)) return false;
    if ((p.Cout != 32 && p.Cout != 64) || p.CoutPad != p.Cout || p.Wout % ST_TW || p.residual || p.pool) return false;
    if (p.out_planar16 && p.Cout % 16) return false;
    const int rw = p.stride == 1 ? 2 : 1, th = 8 * rw;
    const int ih = (th - 1) * p.stride + p.kh, iw = (ST_TW - 1) * p.stride + p.kh;
    const int npl = p.x3 ? 3 : 2, nb = p.Cout / 32;
    const size_t lds = (size_t)npl * ST_MAXPIX * 8 + (size_t)(p.Kpad / 8) * p.Cout * 16 * npl + (size_t)8 * 32 * (64 * nb + 16);
    return ih * iw <= ST_MAXPIX && lds + 512 <= 160 * 1024 && (p.kh * p.kw * 4 + 15) / 16 <= 32 && !(p.sw & SW_NO_STEM);      // (+ the tap table)
}

hipError_t launch_conv_stem(const ConvParams& p, hipStream_t s) {
    if (!conv_stem_eligible(p)) return hipErrorInvalidValue;
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return hipErrorInvalidDevice;
        cus = prop.multiProcessorCount;
    }
    const int nb = p.Cout / 32, rw = p.stride == 1 ? 2 : 1;
    const int tiles = (p.Wout / ST_TW) * ((p.Hout + 8 * rw - 1) / (8 * rw)) * p.N;
    const int npl = p.x3 ? 3 : 2;
    const size_t wbytes = (size_t)(p.Kpad / 8) * p.Cout * 16 * npl;
    const size_t lds = (size_t)npl * ST_MAXPIX * 8 + wbytes + (size_t)8 * 32 * (64 * nb + 16);
    if (lds + 512 > 160 * 1024) return hipErrorInvalidValue;
    const int wgs = cus - p.reserve_cus > 0 ? cus - p.reserve_cus : 1;
    const dim3 grid((unsigned)(tiles < wgs ? tiles : wgs));
#define SD_STEM(NB_, RW_, F_, ...)                                                                                     \
    do {                                                                                                               \
        static bool attr = false;                                                                                      \
        if (!attr) { hipFuncSetAttribute((const void*)conv_stem_kernel<NB_, RW_, F_, ##__VA_ARGS__>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512); attr = true; } \
        hipLaunchKernelGGL((conv_stem_kernel<NB_, RW_, F_, ##__VA_ARGS__>), grid, dim3(512), lds, s, p);                \
    } while (0)
    if (p.x3) {
        if (nb == 1 && rw == 1) SD_STEM(1, 1, false, true); else if (nb == 1) SD_STEM(1, 2, false, true);
        else if (rw == 1) SD_STEM(2, 1, false, true); else SD_STEM(2, 2, false, true);
    } else if (p.f16 == 4) {
        if (nb == 1 && rw == 1) SD_STEM(1, 1, false, false, true); else if (nb == 1) SD_STEM(1, 2, false, false, true);
        else if (rw == 1) SD_STEM(2, 1, false, false, true); else SD_STEM(2, 2, false, false, true);
    } else if (p.f16) {
        if (nb == 1 && rw == 1) SD_STEM(1, 1, true); else if (nb == 1) SD_STEM(1, 2, true);
        else if (rw == 1) SD_STEM(2, 1, true); else SD_STEM(2, 2, true);
    } else {
        if (nb == 1 && rw == 1) SD_STEM(1, 1, false); else if (nb == 1) SD_STEM(1, 2, false);
        else if (rw == 1) SD_STEM(2, 1, false); else SD_STEM(2, 2, false);
    }
#undef SD_STEM
    return hipGetLastError();
}

}  // namespace sd
