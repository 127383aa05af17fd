// Split-bf16 implicit-GEMM convolution, LDS-DMA pipeline (vec layers of SD_PREC_BF16X2 with Cout % 128 == 0).
//
// Same math as conv_split.hip (3 x v_mfma_f32_32x32x16_bf16 per product on hi/lo bf16 planes), but nothing passes through
// VGPRs on the way in: activations are already split-bf16 planes in HBM (split_fmt.hpp) and weights are stored in their
// LDS image, so both operands are staged with global_load_lds_dwordx4 (16 B per lane, 1 KiB per wave-instruction) into a
// 3-stage LDS ring.  Per k-tile a wave issues 6 DMA instructions, 16 ds_read_b128 and 24 MFMAs, and the workgroup
// meets at ONE barrier:
//      wait(tile t landed) ; barrier ; issue DMA(tile t+2) ; 16 ds_read + 24 MFMA on stage t%3
// (stage (t+2)%3 was last read in iteration t-1, which every wave finished before this barrier).
// 8 waves; block 256x256 in TWO 64-KiB stages (layers with >= 512 such blocks: 170 B of DMA per MFMA, half the barriers), else
// 128x256 (Cout % 256 == 0), 256x128 (Cout % 128 == 0), 256x64 or 256x32 with 3 x 36-48 KiB stages; one workgroup per CU.
// The fp16 forms keep only the planes they read in a stage (48 / 32 KB for 256x256), which buys a deeper ring (one-product
// 256x256: 4 stages, three k-tiles in flight: fc7 +26 %, fc6 +5 %) or a second workgroup per CU (256x64; one-product 128x256, 256x128).
// (3x3 stride-1 layers of widths that are multiples of 32 never come here: conv_direct.hip.)
// Out-of-image taps and pixels beyond M read a 16-byte zero page, so zero padding costs no branch in the pipeline.
// Gather granularity: four consecutive lanes fetch the four 16-byte octets of ONE pixel (64 contiguous bytes per plane),
// so a wave-instruction touches 16 cache lines instead of 64; the LDS image is therefore [pixel][octet] and the octet
// slot is XOR-swizzled with (pixel >> 2) & 3 on the SOURCE side (LDS-DMA writes lane-linear), which makes the
// ds_read_b128 fragment reads bank-conflict free.
#pragma once
#include <cstdlib>
#include "kernels.hpp"
#include "split_fmt.hpp"

namespace sd {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// One LDS-DMA: 64 lanes x 16 B from per-lane global addresses to LDS [dst, dst + 1 KiB) (lane-linear).  Inline asm so
// that hipcc neither counts it nor drains it: the pipeline below waits with its own counted s_waitcnt vmcnt(N)
// (the builtin makes hipcc insert vmcnt(0) before every following DMA, i.e. a one-tile-deep pipeline).
__device__ __forceinline__ void dma16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}


// the gather descriptor of a k-tile through the scalar cache.  (A compiler-visible vector load inside the pipeline would
// bring a compiler-counted s_waitcnt vmcnt(0) with it, which also drains the LDS-DMAs it does not know about.)
typedef int i32x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ KEntry load_kentry(const KEntry* ptr) {
    i32x8 v;
    asm volatile("s_load_dwordx8 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(ptr) : "memory");
    KEntry e;
    e.base = reinterpret_cast<const float*>(((unsigned long long)(unsigned)v[1] << 32) | (unsigned)v[0]);
    e.H = v[2]; e.W = v[3]; e.C = v[4]; e.dy = v[5]; e.dx = v[6]; e.flags = v[7];
    return e;
}

// SIMPLE: one source, stride 1, no upsample (every VGG conv, fc6, the stride-1 ResNet convs): the per-tile gather descriptor
// is replaced by arithmetic -- a per-lane base pointer and in-bounds mask over the taps, computed once per workgroup, plus
// a wave-uniform (tap, channel block) offset per k-tile.
// STAGES = 3 with 8 waves and one workgroup per CU (two k-tiles in flight).  Measured and rejected on MI355X: a STAGES = 2,
// 4-wave, two-workgroups-per-CU 128 x 128 instantiation (4-20 % slower on every layer, short K included), and a six-stage
// ring of 16-channel k-steps whose fragments are read one barrier ahead (13-15 % slower: twice the barriers and 32-byte
// gather pieces cost more than the hidden LDS round trip gains).  With random operands the chip sustains 1.81 PFLOP/s of
// v_mfma_f32_32x32x16_bf16 (scripts/probe_mfma_peak.hip; 2.47 with constant operands): the power limit, not the issue
// rate, is the practical ceiling this kernel runs against.  A persistent variant (workgroups walking output tiles, the next
// tile's first k-tiles in flight under the epilogue) measured the same fps within noise (348.4 vs 347.4) and was dropped;
// issuing the DMAs of the two waves of a SIMD at different points of the k-tile (one before, one between the MFMA clusters)
// measured -2 % (337.7 vs 344.9 fps); four waves (one per SIMD) with 64 x 128 wave tiles on the same 128 x 256 block -8 %.
// F16: ONE fp16 activation plane, two fp16 weight planes, two MFMA products per product (split_fmt.hpp): a k-tile then carries half
// the activation DMAs and 16 instead of 24 MFMAs per wave.
// SIMPLE = 2: the two-source 1x1 GEMM of a ResNet block tail (conv3 over its 3x3 output + projection over the block input,
// each with its own stride): two base pointers per lane, the k-tile index selects the source.
// W1 (with F16): ONE MFMA product per product -- the w_lo plane is neither fetched nor multiplied (plain fp16 x fp16, f32 accumulate)
// X3 (SD_PREC_BF16X3): three bf16 planes per operand, six MFMA products per product, three output planes; two 72-KiB stages for the
// 128 x 256 / 256 x 128 blocks (a k-tile moves 72 KB through the 16 B/clk L2 -> LDS path against 48 MFMAs per wave: the DMA bounds it)
// H2 (SD_PREC_F16X2): the stage of the bf16 x 2 form (two planes per operand, three products), but the planes are fp16 hi + SCALED lo
// activations and fp16 hi + lo weights (split_fmt.hpp "HS"): fp16 MFMAs, the x_lo product against w_hi * 2^-11 formed in registers, the
// accumulator times ConvParams::alpha, HS output planes
template <int WAVES_M, int WAVES_N, int MT, int NT, int SIMPLE, int STAGES, bool F16 = false, bool W1 = false, bool X3 = false, bool H2 = false>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N, (F16 && MT * NT <= 4) ? 4 : 1) void conv_dma_kernel(const ConvParams p, int M, int tilesM, int tilesN) {
    static_assert(!X3 || (!F16 && !W1), "bf16 x 3 is a form of its own");
    static_assert(!H2 || (!F16 && !W1 && !X3), "H2 stages like the bf16 x 2 form");
    constexpr int NW = WAVES_M * WAVES_N;
    constexpr int BM = WAVES_M * MT * 32, BN = WAVES_N * NT * 32;
    constexpr int NPX = F16 ? 1 : X3 ? 3 : 2, NPW = W1 ? 1 : X3 ? 3 : 2;      // planes of a stage per operand
    // 16-B units of a stage: Xh [4][BM] | Xl [4][BM] | Wh [4][BN] | Wl [4][BN]; the fp16 forms drop the planes they do not read
    // (Xl; W1: Wl too), so the same LDS holds a deeper ring: more k-tiles in flight against the L2 -> LDS latency
    constexpr int X_UNITS = NPX * 4 * BM, W_UNITS = NPW * 4 * BN;
    constexpr int STAGE_UNITS = X_UNITS + W_UNITS;
    static_assert(STAGES * STAGE_UNITS * 16 <= 160 * 1024, "ring fits in the LDS of a CU");
    constexpr int XI = 8 * BM / 64 / NW;                      // activation DMA instructions per wave and tile for TWO planes (2 or 4)
    constexpr int WI = NPW * 4 * BN / 64;                     // weight instructions of a tile, every plane the form reads
    constexpr int WPW = (WI + NW - 1) / NW;                   // per wave (a short last round re-fetches earlier units: equal counts)
    constexpr int NDMA = NPX * (XI / 2) + WPW;                // DMA instructions per wave per tile (6; fp16 activations have ONE plane: 4 or 5; bf16 x 3: 9)
    constexpr int EPI_ROW = NT * 64 + 16;
    static_assert(NW * (X3 ? 3 : 2) * 32 * EPI_ROW <= STAGES * STAGE_UNITS * 16, "epilogue slabs fit in the ring");
    static_assert((8 * BM / 64) % (2 * NW) == 0 && ((8 * BN / 64) % NW == 0 || NW % (8 * BN / 64 / 2) == 0), "whole DMA instructions per wave and plane (fewer weight instructions than waves: duplicate fetches)");
    __shared__ __attribute__((aligned(16))) u32x4 ring[STAGES * STAGE_UNITS];

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm0 = (wave % WAVES_M) * (MT * 32), wn0 = (wave / WAVES_M) * (NT * 32);

    int tid_;
    {
        const int nwg = tilesM * tilesN * ((!X3 && p.fold) ? 4 : 1), bid = blockIdx.x, xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
        tid_ = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    // upsample-folded conv (ConvParams::fold; every two-plane form -- bf16 x 3 folds on conv_dma3.hip): the parity is the outermost tile index; parity q
    // reads weight rows [q Kpad, (q + 1) Kpad) and its own gather table, and writes source pixel (i, j) to output pixel (2 i + py, 2 j + px)
    int par = 0;
    if constexpr (!X3) { if (p.fold) { par = tid_ / (tilesM * tilesN); tid_ -= par * (tilesM * tilesN); } }
    const int tm = tid_ % tilesM, tn = tid_ / tilesM;
    const int bm0 = tm * BM, bn0 = tn * BN;

    // activation DMA: instruction j of a plane covers pixels [16j, 16j+16) x 4 octets; lane -> (pixel 16j + lane/4, slot lane%4)
    constexpr int XH = XI / 2;                                // hi instructions per wave (1 for BM=128, 2 for BM=256)
    int pimg[XH], poy[XH], pox[XH], pkg[XH];
    bool pok[XH];
#pragma unroll
    for (int i = 0; i < XH; ++i) {
        const int m_l = (wave + NW * i) * 16 + (lane >> 2);
        const int m = bm0 + m_l;
        pok[i] = m < M;
        const int hw = p.Hout * p.Wout;
        const int mm = pok[i] ? m : 0;
        pimg[i] = mm / hw;
        const int r = mm - pimg[i] * hw;
        if (p.pool) {            // window-major order: m = (pooled pixel) * 4 + (row in window) * 2 + (column in window)
            const int q = r >> 2, wp = p.Wout >> 1;
            const int yp = q / wp, xp = q - yp * wp;
            poy[i] = 2 * yp + ((r >> 1) & 1);
            pox[i] = 2 * xp + (r & 1);
        } else {
            poy[i] = r / p.Wout;
            pox[i] = r - poy[i] * p.Wout;
        }
        pkg[i] = (lane & 3) ^ ((m_l >> 2) & 3);               // the octet this lane fetches into slot lane%4
    }
    const KEntry* __restrict__ const ktab = p.ktab + par * (p.Kpad / 32);
    const int CoutPad = p.CoutPad, Nmax = p.Nmax;
    const u32x4* __restrict__ const wt_hi = reinterpret_cast<const u32x4*>(p.wt) + (size_t)par * (p.Kpad / 8) * CoutPad;
    const size_t wplane = (size_t)(p.Kpad / 8) * CoutPad * ((!X3 && p.fold) ? 4 : 1);     // units
    const u32x4* const zero = reinterpret_cast<const u32x4*>(p.zero16);
    const int ktiles = p.Kpad / 32;
    const unsigned ring_lds = (unsigned)(size_t)(__attribute__((address_space(3))) void*)ring;   // wave-uniform LDS address

    // SIMPLE state: pixel base pointers (tap (0,0) minus pad, channel 0 + this lane's octet), tap validity masks, and the
    // wave-uniform walk over (channel block, tap row, tap column) in k-tile order
    const uint16_t* sbase[XH];
    const uint16_t* sbaseB[XH];                               // SIMPLE == 2: second source
    unsigned long long smask[XH];
    size_t splane = 0, splaneB = 0;
    int sW = 0, sC = 0, s_ty = 0, s_tx = 0, s_cb = 0, nA = 0;
    if constexpr (SIMPLE == 2) {
        const KEntry eA = load_kentry(ktab);
        nA = eA.C / 32;
        const KEntry eB = load_kentry(ktab + nA);
        splane = (size_t)Nmax * eA.H * eA.W * eA.C;
        splaneB = (size_t)Nmax * eB.H * eB.W * eB.C;
        const int stA = (eA.flags >> 4) & 3, stB = (eB.flags >> 4) & 3;
#pragma unroll
        for (int i = 0; i < XH; ++i) {
            sbase[i] = reinterpret_cast<const uint16_t*>(eA.base) + ((size_t)(pimg[i] * eA.H + poy[i] * stA) * eA.W + pox[i] * stA) * eA.C + pkg[i] * 8;
            sbaseB[i] = reinterpret_cast<const uint16_t*>(eB.base) + ((size_t)(pimg[i] * eB.H + poy[i] * stB) * eB.W + pox[i] * stB) * eB.C + pkg[i] * 8;
            smask[i] = pok[i] ? 1ull : 0ull;
        }
    }
    if constexpr (SIMPLE == 1) {
        const KEntry e0 = load_kentry(ktab);                  // (channel block 0, tap 0): base, dims, dy = dx = -pad
        sW = e0.W; sC = e0.C;
        splane = (size_t)Nmax * e0.H * e0.W * e0.C;
#pragma unroll
        for (int i = 0; i < XH; ++i) {
            sbase[i] = reinterpret_cast<const uint16_t*>(e0.base) + ((long)(pimg[i] * e0.H + poy[i] - p.pad) * e0.W + pox[i] - p.pad) * e0.C + pkg[i] * 8;
            unsigned long long mk = 0;
            for (int ty = 0; ty < p.kh; ++ty)
                for (int tx = 0; tx < p.kw; ++tx) {
                    const int iy = poy[i] + ty - p.pad, ix = pox[i] + tx - p.pad;
                    if (pok[i] && iy >= 0 && ix >= 0 && iy < e0.H && ix < e0.W) mk |= 1ull << (ty * p.kw + tx);
                }
            smask[i] = mk;
        }
    }

    auto issue = [&](int kt, int stage) {
        const unsigned sbyte = ring_lds + (unsigned)(stage * STAGE_UNITS * 16);      // LDS byte address of the stage
        // ---- activations: hi plane instruction(s) then lo plane instruction(s) ----
        if constexpr (SIMPLE == 2) {
            const bool first = kt < nA;
            const int coff = (first ? kt : kt - nA) * 32;                    // elements, wave-uniform
            const size_t pln = first ? splane : splaneB;
#pragma unroll
            for (int i = 0; i < XH; ++i) {
                const bool ok = smask[i] != 0;
                const uint16_t* px = (first ? sbase[i] : sbaseB[i]) + coff;
#pragma unroll
                for (int pl = 0; pl < NPX; ++pl)
                    dma16(ok ? reinterpret_cast<const u32x4*>(px + pl * pln) : zero, sbyte + (unsigned)((pl * 4 * BM + (wave + NW * i) * 64) * 16));
            }
        } else if constexpr (SIMPLE == 1) {
            const int tap = s_ty * p.kw + s_tx;
            const long soff = (long)(s_ty * sW + s_tx) * sC + s_cb * 32;          // elements, wave-uniform
#pragma unroll
            for (int i = 0; i < XH; ++i) {
                    const bool ok = (smask[i] >> tap) & 1;
                const uint16_t* px = sbase[i] + soff;
#pragma unroll
                for (int pl = 0; pl < NPX; ++pl)
                    dma16(ok ? reinterpret_cast<const u32x4*>(px + pl * splane) : zero, sbyte + (unsigned)((pl * 4 * BM + (wave + NW * i) * 64) * 16));
            }
            if (++s_tx == p.kw) { s_tx = 0; if (++s_ty == p.kh) { s_ty = 0; ++s_cb; } }
        } else {
            const KEntry e = load_kentry(ktab + kt);
            const int st = (e.flags >> 4) & 3, up = e.flags & 1;
            const size_t plane = (size_t)Nmax * e.H * e.W * e.C;
#pragma unroll
            for (int i = 0; i < XH; ++i) {
                    int iy = poy[i] * st + e.dy, ix = pox[i] * st + e.dx;
                const bool ok = pok[i] && iy >= 0 && ix >= 0 && iy < (e.H << up) && ix < (e.W << up);
                iy >>= up; ix >>= up;
                const uint16_t* px = reinterpret_cast<const uint16_t*>(e.base) + ((size_t)(pimg[i] * e.H + iy) * e.W + ix) * e.C + pkg[i] * 8;
#pragma unroll
                for (int pl = 0; pl < NPX; ++pl)
                    dma16(ok ? reinterpret_cast<const u32x4*>(px + pl * plane) : zero, sbyte + (unsigned)((pl * 4 * BM + (wave + NW * i) * 64) * 16));
            }
        }
        // ---- weights: the stage image is the global image ----
#pragma unroll
        for (int i = 0; i < WPW; ++i) {
            const int jw = (wave + NW * i) % WI;              // weight instruction (wraps when WI < 8: duplicate fetch)
            const int wu = jw * 64 + lane;                    // unit inside the W region: [plane][kg][n]
            const int pl = wu / (4 * BN), r = wu % (4 * BN), kg = r / BN, n_l = r % BN;
            const u32x4* g = wt_hi + pl * wplane + (size_t)(kt * 4 + kg) * CoutPad + bn0 + n_l;
            dma16(g, sbyte + (unsigned)(X_UNITS * 16 + jw * 1024));
        }
    };

    // the same DMA instructions one at a time (n = 0 .. NDMA - 1: the X pieces (i, plane), then the weight pieces), for the main loop: it
    // issues them BEHIND its MFMA groups -- at the top of a k-tile, with the MFMA pipes drained by the barrier, each cost its full issue
    // latency (PMC of the bf16 x 3 form of this loop: pipes 56 % busy, a third of the wave cycles parked)
    KEntry pe;                                                 // (SIMPLE == 0: the gather entry of the tile being issued)
    auto piece = [&](int kt, int stage, int n) {
        const unsigned sbyte = ring_lds + (unsigned)(stage * STAGE_UNITS * 16);
        if (n < XH * NPX) {
            const int i = n / NPX, pl = n % NPX;
            if constexpr (SIMPLE == 2) {
                const bool first = kt < nA;
                const int coff = (first ? kt : kt - nA) * 32;
                const size_t pln = first ? splane : splaneB;
                const uint16_t* px = (first ? sbase[i] : sbaseB[i]) + coff;
                dma16(smask[i] != 0 ? reinterpret_cast<const u32x4*>(px + pl * pln) : zero, sbyte + (unsigned)((pl * 4 * BM + (wave + NW * i) * 64) * 16));
            } else if constexpr (SIMPLE == 1) {
                const int tap = s_ty * p.kw + s_tx;
                const long soff = (long)(s_ty * sW + s_tx) * sC + s_cb * 32;
                const uint16_t* px = sbase[i] + soff;
                dma16(((smask[i] >> tap) & 1) ? reinterpret_cast<const u32x4*>(px + pl * splane) : zero, sbyte + (unsigned)((pl * 4 * BM + (wave + NW * i) * 64) * 16));
                if (n == XH * NPX - 1) { if (++s_tx == p.kw) { s_tx = 0; if (++s_ty == p.kh) { s_ty = 0; ++s_cb; } } }
            } else {
                if (n == 0) pe = load_kentry(ktab + kt);
                const int st = (pe.flags >> 4) & 3, up = pe.flags & 1;
                const size_t plane = (size_t)Nmax * pe.H * pe.W * pe.C;
                int iy = poy[i] * st + pe.dy, ix = pox[i] * st + pe.dx;
                const bool ok = pok[i] && iy >= 0 && ix >= 0 && iy < (pe.H << up) && ix < (pe.W << up);
                iy >>= up; ix >>= up;
                const uint16_t* px = reinterpret_cast<const uint16_t*>(pe.base) + ((size_t)(pimg[i] * pe.H + iy) * pe.W + ix) * pe.C + pkg[i] * 8;
                dma16(ok ? reinterpret_cast<const u32x4*>(px + pl * plane) : zero, sbyte + (unsigned)((pl * 4 * BM + (wave + NW * i) * 64) * 16));
            }
        } else {
            const int i = n - XH * NPX;
            const int jw = (wave + NW * i) % WI;
            const int wu = jw * 64 + lane;
            const int pl = wu / (4 * BN), r = wu % (4 * BN), kg = r / BN, n_l = r % BN;
            dma16(wt_hi + pl * wplane + (size_t)(kt * 4 + kg) * CoutPad + bn0 + n_l, sbyte + (unsigned)(X_UNITS * 16 + jw * 1024));
        }
    };

    f32x16 acc[MT][NT];
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    constexpr int AHEAD = STAGES - 1;                         // k-tiles in flight
#pragma unroll
    for (int a = 0; a < AHEAD; ++a)
        if (a < ktiles) issue(a, a);
    const int frow = lane & 31, fk = lane >> 5;
    // With four stages (three k-tiles in flight) a k-tile waits until only AHEAD - 2 later tiles are outstanding: then tile kt + 1 is
    // complete -- and, behind the tile's barrier, visible -- as well, and the first k-step's fragments of tile kt + 1 are read in the TAIL of
    // tile kt (into the registers its own first k-step has released), so that a tile's first MFMAs follow its barrier directly instead of an
    // LDS round trip with the MFMA pipes drained (conv_dma3.hip).
    constexpr bool PRE = STAGES >= 4;
    u32x4 w[NPW][2][NT], x[NPX][2][MT];
    auto fragments_of = [&](int stage, int s) {
        const u32x4* Xs = ring + stage * STAGE_UNITS;
        const u32x4* Ws = Xs + X_UNITS;
        const int kg = 2 * s + fk;
#pragma unroll
        for (int pl = 0; pl < NPW; ++pl)
#pragma unroll
            for (int b = 0; b < NT; ++b) w[pl][s][b] = Ws[pl * 4 * BN + kg * BN + wn0 + b * 32 + frow];
#pragma unroll
        for (int a = 0; a < MT; ++a) {
            const int mrow = wm0 + a * 32 + frow;
            const int slot = mrow * 4 + (kg ^ ((mrow >> 2) & 3));        // [pixel][octet ^ swizzle]
#pragma unroll
            for (int pl = 0; pl < NPX; ++pl) x[pl][s][a] = Xs[pl * 4 * BM + slot];
        }
    };
    auto wait_later = [&](int later) {
        static_assert((AHEAD - 1) * NDMA < 64 && AHEAD <= 6, "vmcnt is a 6-bit counter");
        if (later <= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (later == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA) : "memory");
        else if (later == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NDMA) : "memory");
        else if (later == 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * NDMA) : "memory");
        else if (later == 4) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * NDMA) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(5 * NDMA) : "memory");
    };
    if constexpr (PRE) {
        wait_later(min(AHEAD - 1, ktiles - 1));       // tile 0 has landed
        __builtin_amdgcn_s_barrier();
        fragments_of(0, 0);
    }
    for (int kt = 0; kt < ktiles; ++kt) {
        if constexpr (PRE) {
            wait_later(min(AHEAD - 2, ktiles - 2 - kt));       // tiles kt and kt + 1 have landed
        } else {   // tile kt has landed once at most the DMAs of the tiles issued after it are outstanding
            const int later = min(AHEAD - 1, ktiles - 1 - kt);
            static_assert((AHEAD - 1) * NDMA < 64 && AHEAD <= 6, "vmcnt is a 6-bit counter");
            if (later == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else if (later == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA) : "memory");
            else if (later == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NDMA) : "memory");
            else if (later == 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * NDMA) : "memory");
            else if (later == 4) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * NDMA) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(5 * NDMA) : "memory");
        }
        __builtin_amdgcn_s_barrier();
        // schedule of one k-tile, pinned with sched_barriers (left alone, hipcc hoists the DMA issue to the top and sinks every LDS read to
        // just before its first use, which exposes the LDS latency four times per tile): the fragments of both k-steps (four stages: of
        // the second one only, the first came with the previous tile's tail) -> the MFMA groups, one or two DMA instructions of tile
        // kt + AHEAD behind each of the first ones -> (four stages) the next tile's first fragments before the last group.
        // Stage image: X planes [NPX][4][BM], then W planes [NPW][4][BN].
        // (bf16 x 3, nine DMA instructions and 24-48 MFMAs per k-tile: the round-2 order -- the whole issue between the two fragment reads --
        //  measured faster than the spread one on the strided 3x3 layers of the vgg-encoder monodepth: 53.7 vs 56.3 ms per 32 frames)
        constexpr bool SPREAD = !X3;
        if constexpr (!PRE) fragments_of(kt % STAGES, 0);
        if constexpr (!SPREAD) {
            __builtin_amdgcn_sched_barrier(0);
            if (kt + AHEAD < ktiles) issue(kt + AHEAD, (kt + AHEAD) % STAGES);
            __builtin_amdgcn_sched_barrier(0);
        }
        fragments_of(kt % STAGES, 1);
        __builtin_amdgcn_sched_barrier(0);
        const bool doissue = SPREAD && kt + AHEAD < ktiles;
        // (x plane, w plane) per product, small terms first.  bf16 x 3: hi*lo, lo*hi, mid*mid, hi*mid, mid*hi, hi*hi; bf16 x 2: hi*lo,
        // lo*hi, hi*hi; fp16 activations have no lo plane (hi*lo, hi*hi); W1: hi*hi only
        constexpr int NPR = X3 ? 6 : 3;
        constexpr int xp3[6] = {0, 2, 1, 0, 1, 0}, wp3[6] = {2, 0, 1, 1, 0, 0};
        constexpr int xp2[3] = {0, 1, 0}, wp2[3] = {1, 0, 0};
        constexpr int NPRA = X3 ? 6 : (3 - (F16 ? 1 : 0) - (W1 ? 1 : 0));      // products actually formed
        constexpr int NGROUPS = 2 * NPRA * NT;                                  // MFMA groups (MT MFMAs each) of a k-tile
        constexpr int PPG = (NDMA + NGROUPS - 1) / NGROUPS;                     // DMA pieces behind each of the first groups
        int gidx = 0;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
#pragma unroll
            for (int pr = 0; pr < NPR; ++pr) {
                if (!X3 && ((F16 && pr == 1) || (W1 && pr == 0))) continue;
                const int xi = X3 ? xp3[pr] : (F16 ? 0 : xp2[pr]), wi = X3 ? wp3[pr] : (W1 ? 0 : wp2[pr]);
#pragma unroll
                for (int b = 0; b < NT; ++b) {
                    if (PRE && gidx == NGROUPS - 1 && kt + 1 < ktiles) {      // (the first k-step's registers are free by now)
                        __builtin_amdgcn_sched_barrier(0);
                        fragments_of((kt + 1) % STAGES, 0);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    const u32x4 wv = (H2 && pr == 1) ? hs_wscaled(w[0][s][b]) : w[wi][s][b];      // (H2: x_lo multiplies w_hi * 2^-11)
#pragma unroll
                    for (int a = 0; a < MT; ++a)
                        acc[a][b] = mfma_frag<F16 || H2>(wv, x[xi][s][a], acc[a][b]);
                    if (gidx * PPG < NDMA) {
                        __builtin_amdgcn_sched_barrier(0);
                        if (doissue) {
#pragma unroll
                            for (int n = gidx * PPG; n < (gidx + 1) * PPG && n < NDMA; ++n) piece(kt + AHEAD, (kt + AHEAD) % STAGES, n);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    ++gidx;
                }
            }
        }
    }

    // ---- epilogue (as conv_split.hip): bias + activation, split once, LDS transpose, 16-byte runs per pixel ----
    __syncthreads();
    if constexpr (X3) {
        // bf16 x 3 output: the exact three-way split, one slab per plane and wave
        auto ep3 = [&](auto tag) {
            constexpr int ACT = decltype(tag)::value;
            constexpr int ROW = EPI_ROW;
            unsigned char* slab = reinterpret_cast<unsigned char*>(ring) + wave * (3 * 32 * ROW);
            constexpr int SEGS = NT * 4, PPP = 64 / SEGS;
            const int seg = lane % SEGS, prow = lane / SEGS;
            uint16_t* const out_hi = reinterpret_cast<uint16_t*>(p.out);
            const int m0 = bm0 + wm0, n0 = bn0 + wn0;
#pragma unroll
            for (int a = 0; a < MT; ++a) {
#pragma unroll
                for (int b = 0; b < NT; ++b)
#pragma unroll
                    for (int r4 = 0; r4 < 4; ++r4) {
                        const int nl = b * 32 + 8 * r4 + 4 * (lane >> 5);
                        f32x4 v = {acc[a][b][4 * r4], acc[a][b][4 * r4 + 1], acc[a][b][4 * r4 + 2], acc[a][b][4 * r4 + 3]};
                        v += *reinterpret_cast<const f32x4*>(p.bias + n0 + nl);
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
                    if (mo < M) {
                        uint16_t* o = out_hi + (size_t)mo * p.Cout + n0 + seg * 8;
#pragma unroll
                        for (int pl = 0; pl < 3; ++pl)
                            *reinterpret_cast<u32x4*>(o + pl * p.out_plane) = *reinterpret_cast<const u32x4*>(slab + pl * 32 * ROW + pix * ROW + seg * 16);
                    }
                }
                __builtin_amdgcn_wave_barrier();
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
        };
        if (p.act == ACT_RELU) ep3(ActTag<ACT_RELU>{});
        else if (p.act == ACT_ELU) ep3(ActTag<ACT_ELU>{});
        else ep3(ActTag<ACT_NONE>{});
        return;
    }
    auto epilogue = [&](auto tag, auto otag) {
        constexpr int ACT = decltype(tag)::value;
        constexpr int OF = decltype(otag)::value;              // output planes (the consumers' format): 0 bf16 hi + lo, 1 ONE fp16, 3 fp16 hi + scaled lo
        constexpr bool O16 = OF == 1;
        constexpr int ROW = EPI_ROW;
        unsigned char* sh = reinterpret_cast<unsigned char*>(ring) + wave * (2 * 32 * ROW);
        unsigned char* sl = sh + 32 * ROW;
        constexpr int SEGS = NT * 4, PPP = 64 / SEGS;
        const int seg = lane % SEGS, prow = lane / SEGS;
        uint16_t* const out_hi = reinterpret_cast<uint16_t*>(p.out);
        // 8 channels from channel ch of output pixel px: NHWC, or 16-channel sub-planes (TensorDesc::planar16)
        const size_t osub = p.out_plane / p.Cout * 16;
        auto oaddr = [&](size_t px, int ch) {
            return p.out_planar16 ? out_hi + (size_t)(ch >> 4) * osub + px * 16 + (ch & 8) : out_hi + px * p.Cout + ch;
        };
        const int m0 = bm0 + wm0, n0 = bn0 + wn0;
        // the bias of this lane's channels (32 b + 8 r4 + 4 (lane >> 5)), loaded once per tile instead of once per group of four values (conv_dma3.hip ep3)
        // (formed from an opaque copy of the lane id: the loads must not be hoisted above the k-loop, whose registers are all taken)
        int lane_b = lane;
        asm volatile("" : "+v"(lane_b));
        f32x4 bias_v[4 * NT];
#pragma unroll
        for (int i = 0; i < 4 * NT; ++i) bias_v[i] = *reinterpret_cast<const f32x4*>(p.bias + n0 + (i >> 2) * 32 + 8 * (i & 3) + 4 * (lane_b >> 5));
        if (p.pool) {
            // fused 2x2 max pool: the four pixels of a window are four consecutive accumulator columns = lanes 4j..4j+3;
            // max over the lane quad, THEN bias + activation (monotonic), one pooled pixel per quad
#pragma unroll
            for (int a = 0; a < MT; ++a) {
#pragma unroll
                for (int b = 0; b < NT; ++b)
#pragma unroll
                    for (int r4 = 0; r4 < 4; ++r4) {
                        const int nl = b * 32 + 8 * r4 + 4 * (lane >> 5);
                        f32x4 v;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            float mx = acc[a][b][4 * r4 + r];
                            mx = fmaxf(mx, __shfl_xor(mx, 1));
                            mx = fmaxf(mx, __shfl_xor(mx, 2));
                            v[r] = mx;
                        }
                        if constexpr (H2) v = v * p.alpha + bias_v[b * 4 + r4];
                        else v += bias_v[b * 4 + r4];
                        v = act_split4<ACT>(v);
                        uint2 h, l;
                        split4_fmt<OF>(v, h, l, p.sat);
                        if ((lane & 3) == 0) {
                            *reinterpret_cast<uint2*>(sh + ((lane & 31) >> 2) * ROW + nl * 2) = h;
                            if constexpr (!O16) *reinterpret_cast<uint2*>(sl + ((lane & 31) >> 2) * ROW + nl * 2) = l;
                        }
                    }
                __builtin_amdgcn_wave_barrier();
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int ps = 0; ps < (8 + PPP - 1) / PPP; ++ps) {
                    const int pix = ps * PPP + prow;                 // pooled pixel of this 32-row block (0..7)
                    const int mo = m0 + a * 32 + pix * 4;
                    if (pix < 8 && mo < M) {
                        const u32x4 h = *reinterpret_cast<const u32x4*>(sh + pix * ROW + seg * 16);
                        uint16_t* o = oaddr((size_t)(mo >> 2), n0 + seg * 8);
                        *reinterpret_cast<u32x4*>(o) = h;
                        if constexpr (!O16) *reinterpret_cast<u32x4*>(o + p.out_plane) = *reinterpret_cast<const u32x4*>(sl + pix * ROW + seg * 16);
                    }
                }
                __builtin_amdgcn_wave_barrier();
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
            return;
        }
#pragma unroll
        for (int a = 0; a < MT; ++a) {
#pragma unroll
            for (int b = 0; b < NT; ++b)
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    const int nl = b * 32 + 8 * r4 + 4 * (lane >> 5);
                    f32x4 v = {acc[a][b][4 * r4], acc[a][b][4 * r4 + 1], acc[a][b][4 * r4 + 2], acc[a][b][4 * r4 + 3]};
                    if constexpr (H2) v = v * p.alpha + bias_v[b * 4 + r4];
                    else v += bias_v[b * 4 + r4];
                    v = act_split4<ACT>(v);
                    uint2 h, l;
                    split4_fmt<OF>(v, h, l, p.sat);
                    *reinterpret_cast<uint2*>(sh + (lane & 31) * ROW + nl * 2) = h;
                    if constexpr (!O16) *reinterpret_cast<uint2*>(sl + (lane & 31) * ROW + nl * 2) = l;
                }
            __builtin_amdgcn_wave_barrier();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int ps = 0; ps < 32 / PPP; ++ps) {
                const int pix = ps * PPP + prow;
                const int mo = m0 + a * 32 + pix;
                const u32x4 h = *reinterpret_cast<const u32x4*>(sh + pix * ROW + seg * 16);
                if (mo < M) {
                    size_t opix = (size_t)mo;
                    if constexpr (!X3) {
                        if (p.fold) {            // source pixel (img, i, j) of parity (py, px) -> output pixel (2 i + py, 2 j + px)
                            const int hw = p.Hout * p.Wout, img = mo / hw, r = mo - img * hw, i = r / p.Wout, j = r - i * p.Wout;
                            opix = ((size_t)(img * 2 * p.Hout + 2 * i + (par >> 1))) * (2 * p.Wout) + 2 * j + (par & 1);
                        }
                    }
                    uint16_t* o = oaddr(opix, n0 + seg * 8);
                    *reinterpret_cast<u32x4*>(o) = h;
                    if constexpr (!O16) *reinterpret_cast<u32x4*>(o + p.out_plane) = *reinterpret_cast<const u32x4*>(sl + pix * ROW + seg * 16);
                }
            }
            __builtin_amdgcn_wave_barrier();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    };
    auto ep = [&](auto tag) {
        if constexpr (H2) epilogue(tag, IntTag<3>{});          // (an HS layer writes HS planes, nothing else)
        else { if (p.out_f16) epilogue(tag, IntTag<1>{}); else epilogue(tag, IntTag<0>{}); }
    };
    if (p.act == ACT_RELU) ep(ActTag<ACT_RELU>{});
    else if (p.act == ACT_ELU) ep(ActTag<ACT_ELU>{});
    else ep(ActTag<ACT_NONE>{});
}

// which layers take the DMA pipeline: vec layers with Cout a multiple of 64 and enough tiles to occupy the chip
// S3 / S2 / S1: ring depth of the three-product / two-product / one-product form (the stage shrinks with the planes it holds)
template <int WM, int WN, int MT, int NT, int S3, int S2, int S1>
void launch_dma_variant(const ConvParams& p, long M, hipStream_t s) {
    const int tilesM = (int)((M + WM * MT * 32 - 1) / (WM * MT * 32)), tilesN = p.Cout / (WN * NT * 32);
    const dim3 grid((unsigned)(tilesM * tilesN * ((p.fold && !p.x3) ? 4 : 1))), block(64 * WM * WN);
    if (p.fold && p.x3) return;                                // (folded GEMMs: the two-plane forms here, bf16 x 3 on conv_dma3.hip; the caller sees hipErrorInvalidValue)
    const int mode = (p.dbg & 16) ? 0 : p.simple;
    constexpr bool x3_fits = 2 * 12 * (WM * MT * 32 + WN * NT * 32) * 16 <= 160 * 1024;
    if constexpr (x3_fits) if (p.x3) {               // bf16 x 3: two stages
        if (mode == 2) hipLaunchKernelGGL((conv_dma_kernel<WM, WN, MT, NT, 2, 2, false, false, true>), grid, block, 0, s, p, (int)M, tilesM, tilesN);
        else if (mode == 1) hipLaunchKernelGGL((conv_dma_kernel<WM, WN, MT, NT, 1, 2, false, false, true>), grid, block, 0, s, p, (int)M, tilesM, tilesN);
        else hipLaunchKernelGGL((conv_dma_kernel<WM, WN, MT, NT, 0, 2, false, false, true>), grid, block, 0, s, p, (int)M, tilesM, tilesN);
        return;
    }
    if (p.f16 == 4) {                     // SD_PREC_F16X2: the three-product stage, fp16 HS planes
        if (mode == 2) hipLaunchKernelGGL((conv_dma_kernel<WM, WN, MT, NT, 2, S3, false, false, false, true>), grid, block, 0, s, p, (int)M, tilesM, tilesN);
        else if (mode == 1) hipLaunchKernelGGL((conv_dma_kernel<WM, WN, MT, NT, 1, S3, false, false, false, true>), grid, block, 0, s, p, (int)M, tilesM, tilesN);
        else hipLaunchKernelGGL((conv_dma_kernel<WM, WN, MT, NT, 0, S3, false, false, false, true>), grid, block, 0, s, p, (int)M, tilesM, tilesN);
        return;
    }
    if (p.f16 == 2) {
        if (mode == 2) hipLaunchKernelGGL((conv_dma_kernel<WM, WN, MT, NT, 2, S1, true, true>), grid, block, 0, s, p, (int)M, tilesM, tilesN);
        else if (mode == 1) hipLaunchKernelGGL((conv_dma_kernel<WM, WN, MT, NT, 1, S1, true, true>), grid, block, 0, s, p, (int)M, tilesM, tilesN);
        else hipLaunchKernelGGL((conv_dma_kernel<WM, WN, MT, NT, 0, S1, true, true>), grid, block, 0, s, p, (int)M, tilesM, tilesN);
    } else if (p.f16) {
        if (mode == 2) hipLaunchKernelGGL((conv_dma_kernel<WM, WN, MT, NT, 2, S2, true>), grid, block, 0, s, p, (int)M, tilesM, tilesN);
        else if (mode == 1) hipLaunchKernelGGL((conv_dma_kernel<WM, WN, MT, NT, 1, S2, true>), grid, block, 0, s, p, (int)M, tilesM, tilesN);
        else hipLaunchKernelGGL((conv_dma_kernel<WM, WN, MT, NT, 0, S2, true>), grid, block, 0, s, p, (int)M, tilesM, tilesN);
    } else if (mode == 2) hipLaunchKernelGGL((conv_dma_kernel<WM, WN, MT, NT, 2, S3>), grid, block, 0, s, p, (int)M, tilesM, tilesN);
    else if (mode == 1) hipLaunchKernelGGL((conv_dma_kernel<WM, WN, MT, NT, 1, S3>), grid, block, 0, s, p, (int)M, tilesM, tilesN);
    else hipLaunchKernelGGL((conv_dma_kernel<WM, WN, MT, NT, 0, S3>), grid, block, 0, s, p, (int)M, tilesM, tilesN);
}


}  // namespace sd
