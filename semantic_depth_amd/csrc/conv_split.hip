// Implicit-GEMM convolution on the bf16 MFMA with split-float operands (precision SD_PREC_BF16X2).
//
// Every f32 operand v is carried as two bf16 values  v = hi + lo  (hi = RNE_bf16(v), lo = RNE_bf16(v - hi), 16 mantissa
// bits together) and a product is formed from three MFMA products accumulated in f32:
//        x*w  ~=  x_hi*w_hi + x_hi*w_lo + x_lo*w_hi          (the dropped x_lo*w_lo term is ~2^-18 relative)
// i.e. 3 x v_mfma_f32_32x32x16_bf16 (2.5 PFLOP/s dense) per algorithmic product: an ~833 TFLOP/s ceiling, 5.3x the exact
// f32 MFMA, at ~1e-5 relative error per product (measured end-to-end in tests/test_gpu_nets.py against the 1e-3 budget
// of BASELINE.json).  gfx950 has no xf32/TF32 MFMA; this is the explicit, error-bounded substitute.
//
// Same gather tables and tile mapping as conv_igemm.hip; differences:
//   * activations live in HBM as split-bf16 planes (split_fmt.hpp): the producer's epilogue splits ONCE, consumers read
//     16-byte runs of 8 channels from the hi and the lo plane and stage them unchanged; weights are split offline
//     (relayout_weight) into two bf16 planes laid out exactly like their LDS image [k/8][n][8]
//   * LDS images are [k/8][row][8] bf16 (16 B per lane per fragment): conflict-free ds_read_b128 for the 32x32x16
//     fragments (lane = row & 31, k-octet = lane >> 5)
//   * wave tile = (MT*32) x (NT*32) from 32x32x16 MFMAs; weights are the A operand so that a lane owns runs of 4
//     consecutive output channels of one pixel (16-byte NHWC stores)
#include "kernels.hpp"
#include "split_fmt.hpp"

namespace sd {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));


// Output path shared by the split kernels.  A wave owns a (MT*32) x (NT*32) tile; per 32-pixel slab it applies bias +
// activation, splits ONCE into hi/lo, transposes through a wave-private LDS slab and writes 16-byte runs of 8 channels,
// so every pixel's NT*64 bytes per plane leave as one contiguous segment (the MFMA layout alone gives 8-byte fragments).
// OF = OUTPUT format: 0 bf16 hi + lo, 1 ONE fp16 plane, 3 fp16 hi + scaled lo (an H2 layer: the accumulator times ConvParams::alpha)
template <int ACT, int MT, int NT, int OF>
__device__ __forceinline__ void split_epilogue_act(f32x16 (&acc)[MT][NT], unsigned char* slab, const ConvParams& p, int m0, int n0, int M, int lane) {
    constexpr bool F16 = OF == 1;
    constexpr int ROW = NT * 64 + 16;
    unsigned char* sh = slab;
    unsigned char* sl = slab + 32 * ROW;
    constexpr int SEGS = NT * 4;                  // 16-B segments per pixel row
    constexpr int PPP = 64 / SEGS;                // pixels per read pass
    const int seg = lane % SEGS, prow = lane / SEGS;
    uint16_t* const out_hi = reinterpret_cast<uint16_t*>(p.out);
#pragma unroll
    for (int a = 0; a < MT; ++a) {
#pragma unroll
        for (int b = 0; b < NT; ++b)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const int nl = b * 32 + 8 * r4 + 4 * (lane >> 5);
                const int n = n0 + nl;
                f32x4 v = {acc[a][b][4 * r4], acc[a][b][4 * r4 + 1], acc[a][b][4 * r4 + 2], acc[a][b][4 * r4 + 3]};
                if constexpr (OF == 3) v *= p.alpha;
                if (n < p.Cout) v += *reinterpret_cast<const f32x4*>(p.bias + n);
                v = act_split4<ACT>(v);
                uint2 h, l;
                split4_fmt<OF>(v, h, l, p.sat);
                *reinterpret_cast<uint2*>(sh + (lane & 31) * ROW + nl * 2) = h;
                *reinterpret_cast<uint2*>(sl + (lane & 31) * ROW + nl * 2) = l;
            }
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int ps = 0; ps < 32 / PPP; ++ps) {
            const int pix = ps * PPP + prow;
            const int mo = m0 + a * 32 + pix;
            const u32x4 h = *reinterpret_cast<const u32x4*>(sh + pix * ROW + seg * 16);
            const u32x4 l = *reinterpret_cast<const u32x4*>(sl + pix * ROW + seg * 16);
            const int n = n0 + seg * 8;
            if (mo < M && n < p.Cout) {
                uint16_t* o = out_hi + (size_t)mo * p.Cout + n;
                *reinterpret_cast<u32x4*>(o) = h;
                if constexpr (!F16) *reinterpret_cast<u32x4*>(o + p.out_plane) = l;      // (F16 here = the OUTPUT format)
            }
        }
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
}
// bf16 x 3 output (SD_PREC_BF16X3): the exact three-way split, one slab per plane
template <int ACT, int MT, int NT>
__device__ __forceinline__ void split_epilogue_x3(f32x16 (&acc)[MT][NT], unsigned char* slab, const ConvParams& p, int m0, int n0, int M, int lane) {
    constexpr int ROW = NT * 64 + 16;
    constexpr int SEGS = NT * 4, PPP = 64 / SEGS;
    const int seg = lane % SEGS, prow = lane / SEGS;
    uint16_t* const out_hi = reinterpret_cast<uint16_t*>(p.out);
#pragma unroll
    for (int a = 0; a < MT; ++a) {
#pragma unroll
        for (int b = 0; b < NT; ++b)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const int nl = b * 32 + 8 * r4 + 4 * (lane >> 5);
                const int n = n0 + nl;
                f32x4 v = {acc[a][b][4 * r4], acc[a][b][4 * r4 + 1], acc[a][b][4 * r4 + 2], acc[a][b][4 * r4 + 3]};
                if (n < p.Cout) v += *reinterpret_cast<const f32x4*>(p.bias + n);
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = act_x3<ACT>(v[r]);
                uint2 h, m, l;
                split4_x3(v, h, m, l);
                *reinterpret_cast<uint2*>(slab + (lane & 31) * ROW + nl * 2) = h;
                *reinterpret_cast<uint2*>(slab + 32 * ROW + (lane & 31) * ROW + nl * 2) = m;
                *reinterpret_cast<uint2*>(slab + 64 * ROW + (lane & 31) * ROW + nl * 2) = l;
            }
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int ps = 0; ps < 32 / PPP; ++ps) {
            const int pix = ps * PPP + prow;
            const int mo = m0 + a * 32 + pix;
            const int n = n0 + seg * 8;
            if (mo < M && n < p.Cout) {
                uint16_t* o = out_hi + (size_t)mo * p.Cout + n;
#pragma unroll
                for (int pl = 0; pl < 3; ++pl)
                    *reinterpret_cast<u32x4*>(o + pl * p.out_plane) = *reinterpret_cast<const u32x4*>(slab + pl * 32 * ROW + pix * ROW + seg * 16);
            }
        }
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
}
template <int MT, int NT, bool F16, bool H2 = false>
__device__ __forceinline__ void split_epilogue(f32x16 (&acc)[MT][NT], unsigned char* slab, const ConvParams& p, int m0, int n0, int M, int lane) {
    // the template's F16 is the INPUT format of the kernel; the output planes follow p.out_f16 (the consumers' format)
    if constexpr (H2) {
        if (p.act == ACT_RELU) split_epilogue_act<ACT_RELU, MT, NT, 3>(acc, slab, p, m0, n0, M, lane);
        else if (p.act == ACT_ELU) split_epilogue_act<ACT_ELU, MT, NT, 3>(acc, slab, p, m0, n0, M, lane);
        else split_epilogue_act<ACT_NONE, MT, NT, 3>(acc, slab, p, m0, n0, M, lane);
        return;
    }
    if (p.out_f16) {
        if (p.act == ACT_RELU) split_epilogue_act<ACT_RELU, MT, NT, 1>(acc, slab, p, m0, n0, M, lane);
        else if (p.act == ACT_ELU) split_epilogue_act<ACT_ELU, MT, NT, 1>(acc, slab, p, m0, n0, M, lane);
        else split_epilogue_act<ACT_NONE, MT, NT, 1>(acc, slab, p, m0, n0, M, lane);
        return;
    }
    if (p.act == ACT_RELU) split_epilogue_act<ACT_RELU, MT, NT, 0>(acc, slab, p, m0, n0, M, lane);
    else if (p.act == ACT_ELU) split_epilogue_act<ACT_ELU, MT, NT, 0>(acc, slab, p, m0, n0, M, lane);
    else split_epilogue_act<ACT_NONE, MT, NT, 0>(acc, slab, p, m0, n0, M, lane);
}

template <int WAVES_M, int WAVES_N, int MT, int NT, int NPL = 2>
struct STile {
    static constexpr int BM = WAVES_M * MT * 32;
    static constexpr int BN = WAVES_N * NT * 32;
    static constexpr int NTHR = WAVES_M * WAVES_N * 64;
    static constexpr int A_IT = BM * 4 / NTHR;                // (pixel, k-octet) items per thread per k-tile
    static constexpr int B_LD = (BN * 4 + NTHR - 1) / NTHR;   // 16-B loads per thread per k-tile and plane
    static constexpr int STAGE_BYTES = (BM + BN) * 32 * 2 * NPL;                // NPL planes (hi + lo, or hi + mid + lo), bf16
    static constexpr int EPI_ROW = NT * 64 + 16;                                // bytes per pixel row of a wave's output slab (+16 pad)
    static constexpr int EPI_BYTES = WAVES_M * WAVES_N * NPL * 32 * EPI_ROW;    // per wave: one slab of 32 pixels per plane
    static constexpr int LDS_BYTES = STAGE_BYTES > EPI_BYTES ? STAGE_BYTES : EPI_BYTES;
};

// F16: ONE fp16 activation plane (the lo-plane reads below fetch unused bytes), two fp16 weight planes, two MFMA products per
// product (split_fmt.hpp).  X3: three bf16 planes per operand, six MFMA products per product (SD_PREC_BF16X3).
// H2 (SD_PREC_F16X2): the bf16 x 2 staging with fp16 hi + scaled lo activations and fp16 hi + lo weights: three fp16 products, the x_lo one
// against w_hi * 2^-11 formed in registers (split_fmt.hpp "HS")
template <int WAVES_M, int WAVES_N, int MT, int NT, bool VEC, bool F16 = false, bool X3 = false, bool H2 = false>
__global__ __launch_bounds__(WAVES_M * WAVES_N * 64) void conv_split_kernel(const ConvParams p, int M, int tilesM, int tilesN) {
    static_assert(!H2 || (!F16 && !X3), "H2 stages like the bf16 x 2 form");
    constexpr int NPL = X3 ? 3 : 2;
    using T = STile<WAVES_M, WAVES_N, MT, NT, NPL>;
    constexpr int BM = T::BM, BN = T::BN, NTHR = T::NTHR;
    __shared__ __attribute__((aligned(16))) unsigned char lds[T::LDS_BYTES];
    u32x4* const Xp = reinterpret_cast<u32x4*>(lds);       // [NPL][4][BM] 16-B units
    u32x4* const Wp = Xp + NPL * 4 * BM;                   // [NPL][4][BN]

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = t >> 6;
    const int wm0 = (wave % WAVES_M) * (MT * 32);
    const int wn0 = (wave / WAVES_M) * (NT * 32);

    // XCD-aware tile order: hardware places block b on XCD b % 8; give every XCD a contiguous run of tile ids so that
    // neighbouring pixel tiles (shared halo rows) and the blocks sharing a weight panel meet in the same L2
    int tid_;
    {
        const int nwg = tilesM * tilesN, bid = blockIdx.x, xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
        tid_ = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    int tm, tn;
    if (p.m_fastest) { tm = tid_ % tilesM; tn = tid_ / tilesM; }
    else             { tn = tid_ % tilesN; tm = tid_ / tilesN; }
    const int bm0 = tm * BM, bn0 = tn * BN;

    const int m_l = t % BM;
    const int kg0 = t / BM;                 // 0 (BM=256) or 0/1 (BM=128)
    constexpr int KG_STEP = NTHR / BM;      // 1, 2 or 4
    const int m = bm0 + m_l;
    const bool m_ok = m < M;
    int img, oy, ox;
    {
        const int hw = p.Hout * p.Wout;
        const int mm = m_ok ? m : 0;
        img = mm / hw;
        const int r = mm - img * hw;
        oy = r / p.Wout;
        ox = r - oy * p.Wout;
    }
    const KEntry* __restrict__ const ktab = p.ktab;
    const int CoutPad = p.CoutPad;
    const u32x4* __restrict__ const wt_hi = reinterpret_cast<const u32x4*>(p.wt);
    const size_t wplane = (size_t)(p.Kpad / 8) * CoutPad;

    u32x4 rx[NPL][T::A_IT];                // one k-octet of one pixel per plane
    const int Nmax = p.Nmax;
    u32x4 rw[NPL][T::B_LD];
    const int ktiles = p.Kpad / 32, vtiles = p.vtiles;

    auto load_tile = [&](int kt) {
#pragma unroll
        for (int i = 0; i < T::B_LD; ++i) {
            const int idx = t + NTHR * i;
            const int n_l = idx % BN, kg = idx / BN;
            if (BN * 4 >= NTHR || idx < BN * 4) {
                const size_t o = (size_t)(kt * 4 + kg) * CoutPad + bn0 + n_l;
#pragma unroll
                for (int pl = 0; pl < NPL; ++pl) rw[pl][i] = wt_hi[pl * wplane + o];
            }
        }
        if (VEC || kt < vtiles) {
            const KEntry e = ktab[kt];   // wave-uniform -> one s_load_dwordx8
            const int st = (e.flags >> 4) & 3, up = e.flags & 1;       // per-source stride / x2 upsample
            int iy = oy * st + e.dy, ix = ox * st + e.dx;
            const bool ok = (e.flags & 0x10000) && m_ok && iy >= 0 && ix >= 0 && iy < (e.H << up) && ix < (e.W << up);
            iy >>= up; ix >>= up;
            const uint16_t* base = reinterpret_cast<const uint16_t*>(e.base) + ((size_t)(img * e.H + iy) * e.W + ix) * e.C;
            const size_t plane = (size_t)Nmax * e.H * e.W * e.C;
#pragma unroll
            for (int i = 0; i < T::A_IT; ++i) {
                const int kg = kg0 + KG_STEP * i;
#pragma unroll
                for (int pl = 0; pl < NPL; ++pl) {
                    u32x4 v = {0u, 0u, 0u, 0u};
                    if (ok) v = *reinterpret_cast<const u32x4*>(base + pl * plane + kg * 8);
                    rx[pl][i] = v;
                }
            }
        } else {
            const KEntry* __restrict__ const qtab = ktab + ktiles + (kt - vtiles) * 8;   // this tile's 8 quad descriptors
#pragma unroll
            for (int i = 0; i < T::A_IT; ++i) {
                const int kg = __builtin_amdgcn_readfirstlane(kg0 + KG_STEP * i);
                u32x4 v[NPL];
#pragma unroll
                for (int pl = 0; pl < NPL; ++pl) v[pl] = u32x4{0u, 0u, 0u, 0u};
#pragma unroll
                for (int hq = 0; hq < 2; ++hq) {
                    const KEntry q4 = qtab[kg * 2 + hq];
                    const int st = (q4.flags >> 4) & 3, up = q4.flags & 1;
                    int iy = oy * st + q4.dy, ix = ox * st + q4.dx;
                    const bool ok = (q4.flags & 0x10000) && m_ok && iy >= 0 && ix >= 0 && iy < (q4.H << up) && ix < (q4.W << up);
                    iy >>= up; ix >>= up;
                    const int nv = (q4.flags >> 8) & 7;
                    const uint16_t* q = reinterpret_cast<const uint16_t*>(q4.base) + ((size_t)(img * q4.H + iy) * q4.W + ix) * q4.C;
                    const size_t plane = (size_t)Nmax * q4.H * q4.W * q4.C;
#pragma unroll
                    for (int pl = 0; pl < NPL; ++pl) {
                        unsigned w0 = 0, w1 = 0;
                        const uint16_t* qp = q + pl * plane;
                        if (ok) {
                            if (nv == 4) {
                                const uint2 a = *reinterpret_cast<const uint2*>(qp);
                                w0 = a.x; w1 = a.y;
                            } else if (nv == 2) {
                                w0 = *reinterpret_cast<const unsigned*>(qp);
                            } else {
                                for (int j = 0; j < nv; ++j) {
                                    const unsigned a = qp[j];
                                    if (j == 0) w0 |= a; else if (j == 1) w0 |= a << 16; else w1 |= a;
                                }
                            }
                        }
                        v[pl][2 * hq] = w0; v[pl][2 * hq + 1] = w1;
                    }
                }
#pragma unroll
                for (int pl = 0; pl < NPL; ++pl) rx[pl][i] = v[pl];
            }
        }
    };

    f32x16 acc[MT][NT];
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    load_tile(0);
    const int frow = lane & 31, fk = lane >> 5;
    for (int kt = 0; kt < ktiles; ++kt) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < T::A_IT; ++i) {
            const int kg = kg0 + KG_STEP * i;
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl) Xp[pl * 4 * BM + kg * BM + m_l] = rx[pl][i];
        }
#pragma unroll
        for (int i = 0; i < T::B_LD; ++i) {
            const int idx = t + NTHR * i;
            if (BN * 4 >= NTHR || idx < BN * 4) {
#pragma unroll
                for (int pl = 0; pl < NPL; ++pl) Wp[pl * 4 * BN + idx] = rw[pl][i];
            }
        }
        __syncthreads();
        if (kt + 1 < ktiles) load_tile(kt + 1);
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int kg = 2 * s + fk;
            u32x4 w[NPL][NT], x[NPL][MT];
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl) {
#pragma unroll
                for (int b = 0; b < NT; ++b) w[pl][b] = Wp[pl * 4 * BN + kg * BN + wn0 + b * 32 + frow];
#pragma unroll
                for (int a = 0; a < MT; ++a) x[pl][a] = Xp[pl * 4 * BM + kg * BM + wm0 + a * 32 + frow];
            }
            // the products of one accumulator are issued MT*NT MFMAs apart (no back-to-back dependent MFMAs), small terms first.
            // (s_setprio(1) around this cluster was measured: -25 %, the co-resident blocks' staging starves)
            // (x plane, w plane): bf16 x 3: hi*lo, lo*hi, mid*mid, hi*mid, mid*hi, hi*hi; bf16 x 2: hi*lo, lo*hi, hi*hi
            constexpr int NPR = X3 ? 6 : 3;
            constexpr int xp3[6] = {0, 2, 1, 0, 1, 0}, wp3[6] = {2, 0, 1, 1, 0, 0};
            constexpr int xp2[3] = {0, 1, 0}, wp2[3] = {1, 0, 0};
#pragma unroll
            for (int pr = 0; pr < NPR; ++pr) {
                if (!X3 && F16 && pr == 1) continue;                      // fp16 activations: no lo plane, products x*w_lo and x*w_hi
                if (!X3 && F16 && pr == 0 && p.f16 == 2) continue;        // one-product layer (':1'): x*w_hi only, as conv_dma / conv_direct W1
                const int xi = X3 ? xp3[pr] : xp2[pr], wi = X3 ? wp3[pr] : wp2[pr];
#pragma unroll
                for (int b = 0; b < NT; ++b) {
                    const u32x4 wv = (H2 && pr == 1) ? hs_wscaled(w[0][b]) : w[wi][b];      // (H2: x_lo multiplies w_hi * 2^-11)
#pragma unroll
                    for (int a = 0; a < MT; ++a)
                        acc[a][b] = mfma_frag<F16 || H2>(wv, x[xi][a], acc[a][b]);
                }
            }
        }
    }

    // ---- epilogue: D[row = channel (r&3) + 8*(r>>2) + 4*(lane>>5)][col = pixel lane&31] ----
    __syncthreads();                      // every wave is done with the stage memory: reuse it as output staging
    if constexpr (X3) {
        unsigned char* slab = lds + wave * (3 * 32 * T::EPI_ROW);
        if (p.act == ACT_RELU) split_epilogue_x3<ACT_RELU, MT, NT>(acc, slab, p, bm0 + wm0, bn0 + wn0, M, lane);
        else if (p.act == ACT_ELU) split_epilogue_x3<ACT_ELU, MT, NT>(acc, slab, p, bm0 + wm0, bn0 + wn0, M, lane);
        else split_epilogue_x3<ACT_NONE, MT, NT>(acc, slab, p, bm0 + wm0, bn0 + wn0, M, lane);
    } else {
        split_epilogue<MT, NT, F16, H2>(acc, lds + wave * (2 * 32 * T::EPI_ROW), p, bm0 + wm0, bn0 + wn0, M, lane);
    }
}

template <int WAVES_M, int WAVES_N, int MT, int NT>
static hipError_t launch_scfg(const ConvParams& p, hipStream_t s) {
    using T = STile<WAVES_M, WAVES_N, MT, NT>;
    const long M = (long)p.N * p.Hout * p.Wout;
    const int tilesM = (int)((M + T::BM - 1) / T::BM);
    const int tilesN = (p.Cout + T::BN - 1) / T::BN;
    dim3 grid((unsigned)(tilesM * tilesN));
    if (p.x3) {
        if (p.vec)
            hipLaunchKernelGGL((conv_split_kernel<WAVES_M, WAVES_N, MT, NT, true, false, true>), grid, dim3(T::NTHR), 0, s, p, (int)M, tilesM, tilesN);
        else
            hipLaunchKernelGGL((conv_split_kernel<WAVES_M, WAVES_N, MT, NT, false, false, true>), grid, dim3(T::NTHR), 0, s, p, (int)M, tilesM, tilesN);
    } else if (p.f16 == 4) {
        if (p.vec)
            hipLaunchKernelGGL((conv_split_kernel<WAVES_M, WAVES_N, MT, NT, true, false, false, true>), grid, dim3(T::NTHR), 0, s, p, (int)M, tilesM, tilesN);
        else
            hipLaunchKernelGGL((conv_split_kernel<WAVES_M, WAVES_N, MT, NT, false, false, false, true>), grid, dim3(T::NTHR), 0, s, p, (int)M, tilesM, tilesN);
    } else if (p.f16) {
        if (p.vec)
            hipLaunchKernelGGL((conv_split_kernel<WAVES_M, WAVES_N, MT, NT, true, true>), grid, dim3(T::NTHR), 0, s, p, (int)M, tilesM, tilesN);
        else
            hipLaunchKernelGGL((conv_split_kernel<WAVES_M, WAVES_N, MT, NT, false, true>), grid, dim3(T::NTHR), 0, s, p, (int)M, tilesM, tilesN);
    } else if (p.vec)
        hipLaunchKernelGGL((conv_split_kernel<WAVES_M, WAVES_N, MT, NT, true>), grid, dim3(T::NTHR), 0, s, p, (int)M, tilesM, tilesN);
    else
        hipLaunchKernelGGL((conv_split_kernel<WAVES_M, WAVES_N, MT, NT, false>), grid, dim3(T::NTHR), 0, s, p, (int)M, tilesM, tilesN);
    return hipGetLastError();
}

int conv_split_tile_n(int Cout) {
    if (Cout % 128 == 0) return 128;
    if (Cout % 64 == 0) return 64;
    return 32;
}

// big tiles (8 waves) when there are enough of them to fill the chip: a 128x256 / 256x128 tile moves 25 % fewer bytes
// from L2 per flop than 128x128 and halves the per-thread staging work on the wider side
static int split_variant(const ConvParams& p) {
    const long M = (long)p.N * p.Hout * p.Wout;
    const int tn = conv_split_tile_n(p.Cout);
    if (tn == 128) {
        if (p.Cout % 256 == 0 && ((M + 127) / 128) * (p.Cout / 256) >= 512) return 0;    // 128 x 256
        return 2;                                                                          // 128 x 128
    }
    return tn == 64 ? 3 : 4;
}

hipError_t launch_conv_split(const ConvParams& p, hipStream_t s) {
    switch (split_variant(p)) {
        case 0:  return launch_scfg<2, 4, 2, 2>(p, s);   // 128 x 256, 8 waves
        case 1:  return launch_scfg<4, 2, 2, 2>(p, s);   // 256 x 128, 8 waves
        case 2:  return launch_scfg<2, 2, 2, 2>(p, s);   // 128 x 128
        case 3:  return launch_scfg<4, 1, 2, 2>(p, s);   // 256 x 64
        default: return launch_scfg<4, 1, 2, 1>(p, s);   // 256 x 32 (Cout = 16 is padded to 32)
    }
}

const char* conv_split_kernel_name(const ConvParams& p) {
    if (p.x3) {
        switch (split_variant(p)) {
            case 0:  return "conv_split_x3_kernel<2,4,2,2>";
            case 1:  return "conv_split_x3_kernel<4,2,2,2>";
            case 2:  return "conv_split_x3_kernel<2,2,2,2>";
            case 3:  return "conv_split_x3_kernel<4,1,2,2>";
            default: return "conv_split_x3_kernel<4,1,2,1>";
        }
    }
    if (p.f16 == 4) return "conv_split_hs_kernel";
    if (p.f16) {
        switch (split_variant(p)) {
            case 0:  return "conv_split_f16w_kernel<2,4,2,2>";
            case 1:  return "conv_split_f16w_kernel<4,2,2,2>";
            case 2:  return "conv_split_f16w_kernel<2,2,2,2>";
            case 3:  return "conv_split_f16w_kernel<4,1,2,2>";
            default: return "conv_split_f16w_kernel<4,1,2,1>";
        }
    }
    switch (split_variant(p)) {
        case 0:  return p.vec ? "conv_split_kernel<2,4,2,2,true>" : "conv_split_kernel<2,4,2,2,false>";
        case 1:  return p.vec ? "conv_split_kernel<4,2,2,2,true>" : "conv_split_kernel<4,2,2,2,false>";
        case 2:  return p.vec ? "conv_split_kernel<2,2,2,2,true>" : "conv_split_kernel<2,2,2,2,false>";
        case 3:  return p.vec ? "conv_split_kernel<4,1,2,2,true>" : "conv_split_kernel<4,1,2,2,false>";
        default: return p.vec ? "conv_split_kernel<4,1,2,1,true>" : "conv_split_kernel<4,1,2,1,false>";
    }
}

}  // namespace sd
