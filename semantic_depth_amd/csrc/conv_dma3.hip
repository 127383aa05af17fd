// Implicit-GEMM convolution of the fp32-grade split engine (SD_PREC_BF16X3) for the layers with Cout % 256 == 0: fc6 / fc7, the ResNet
// block tails (conv3 + projection as one GEMM) and the wide 1x1 / strided layers.  Three bf16 planes per operand, six MFMA products per
// product (split_fmt.hpp), everything staged by LDS-DMA like conv_dma.hip.
//
// The 128 x 256 two-stage block of conv_dma.hip<X3> moves 72 KB per k-tile through the 16 B/clk L2 -> LDS path (4608 clk) for 48 MFMAs per
// wave (3072 clk): the DMA bounds it at 0.67 of the MFMA rate (measured 0.46-0.55).  Here the block is 256 x 256 -- 96 KB for 96 MFMAs per
// wave: 1 : 1 -- which fits LDS because the six products are grouped BY WEIGHT PLANE into three phases per k-tile, as in conv_direct3.hip:
//        phase lo : W_lo  x X_hi                16 MFMAs per wave
//        phase mid: W_mid x (X_mid, X_hi)       32
//        phase hi : W_hi  x (X_lo, X_mid, X_hi) 48
// The X fragments of the hi and mid planes stay in VGPRs across the phases of a k-tile (36 instead of 60 ds_read_b128 per 96 MFMAs), so
// every X plane is read from LDS in ONE phase only -- the phase that uses it first: lo reads X_hi, mid X_mid, hi X_lo -- exactly like the
// weight planes.  LDS is therefore a ring of four (X plane, weight plane) pairs of 16 + 16 KB (128 KB): the pair being read and the pairs
// of the next THREE phases in flight.  Phase q issues the pair of phase q + 3 (four DMA instructions per wave) and waits with a counted
// s_waitcnt vmcnt(8) for what was issued three phases ago.  (Round 3's first form double-buffered whole X k-tiles beside a three-slot
// weight ring -- 144 KB, two phases of look-ahead; PMC showed the MFMA pipes 56 % busy with a third of the wave cycles parked in waits.)
// 8 waves as 2 x 4, wave tile 128 pixels x 64 channels (acc 128 VGPRs); general gather through the KEntry table (any kernel size, stride,
// concatenated sources); epilogue = conv_dma.hip's X3 epilogue.
#include <cstdlib>
#include "kernels.hpp"
#include "split_fmt.hpp"

namespace sd {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void g3dma16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
// the same with a uniform base in SGPRs and a 32-bit per-lane byte offset (the weight panels: one VGPR per piece instead of a 64-bit
// pointer per piece and plane)
__device__ __forceinline__ void g3dma16s(const void* sbase, unsigned voff, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}
typedef int i32x8g __attribute__((ext_vector_type(8)));
__device__ __forceinline__ KEntry g3load_kentry(const KEntry* ptr) {
    i32x8g v;
    asm volatile("s_load_dwordx8 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(ptr) : "memory");
    KEntry e;
    e.base = reinterpret_cast<const float*>(((unsigned long long)(unsigned)v[1] << 32) | (unsigned)v[0]);
    e.H = v[2]; e.W = v[3]; e.C = v[4]; e.dy = v[5]; e.dx = v[6]; e.flags = v[7];
    return e;
}

constexpr int G3_BM = 256, G3_BN = 256, G3_NW = 8, G3_MT = 4, G3_NT = 2;
constexpr int G3_XPL = 4 * G3_BM;            // 16-B units of one X plane of a k-tile: [pixel][octet ^ swizzle]
constexpr int G3_WPL = 4 * G3_BN;            // units of one weight plane of a k-tile: [k-octet][channel]
constexpr int G3_RING = 4;                   // ring slots, one (X plane, weight plane) pair each: the phase being read + three in flight
constexpr int G3_PAIR = G3_XPL + G3_WPL;
constexpr int G3_LDS = G3_RING * G3_PAIR;    // units (128 KB)
constexpr int G3_ROW = G3_NT * 64 + 16;

// FLAT (ConvParams::flat): every k-tile reads its source at tap (0, 0) without an upsample -- the 1x1 layers: the ResNet block tails (conv3 +
// projection over the concatenated K axis, the shortcut source at the block's stride), conv1 of res4 / res5, fc7.  The pixel this lane
// fetches is then the same for every k-tile of a source: its byte offset is computed ONCE per source geometry (at most two) and a piece
// of the X DMA is a scalar base + that 32-bit offset, instead of ~20 VALU instructions with four quarter-rate multiplies per piece and
// phase (274 VALU per k-tile beside 96 MFMAs, and the two pipes of a SIMD do not overlap: profiles/r04_mfma_valu_overlap_probe.txt).
// Rows past M fetch pixel 0 (finite values; their outputs are never stored).
// MODE 2 (ConvParams::noup, not flat): any taps, no upsample, at most two source geometries -- the same precomputed pixel offset plus a
// SCALAR tap offset and the in-image test (fc6, the folded upconvs, the strided 3x3 layers): ~10 VALU per piece, no multiplies.
// HS (round 5; SD_PREC_F16X2, split_fmt.hpp "HS"): the same ring for TWO planes per operand and three fp16 products -- two phases per k-tile,
//        phase lo: W_lo x X_hi                       16 MFMAs per wave (the X_hi fragments stay in VGPRs for the k-tile)
//        phase hi: W_hi x (X_lo [scaled], X_hi)      32                (the X_lo product against w_hi * 2^-11, formed in registers)
// i.e. the lo and hi phases of the six-product form without its mid plane.  A pair is still 16 + 16 KB and phase q still issues the pair of phase
// q + 3, which is now one and a half k-tiles ahead: the lo phase of k-tile kt issues (X_lo, W_hi) of kt + 1, the hi phase (X_hi, W_lo) of kt + 2.
// 64 KB per k-tile against 48 MFMAs per wave: the L2 -> LDS path bounds it at 0.75 of the MFMA rate.  The accumulator times ConvParams::alpha,
// HS output planes.
// TIMED (SEMDEPTH_X3_DIAG=3; MODE 1 only, decomposition runs): s_memtime stamps around the counted wait, the barrier and the body of every phase, summed
// per wave; waves 0 and 4 of the middle workgroup print their sums (the instrumentation itself costs ~10 % of the wave cycles: read the SPLIT, not the total)
// S16 (round 5; the bf16 x 3 engine's production form): the same ring and the same LDS traffic, multiplied by v_mfma_f32_16x16x32 instead of 32x32x16 -- a k-tile is ONE
// k-step of 32, a wave's 128 x 64 tile 8 x 4 blocks of 16 x 16, the products grouped by X plane with the weight fragments kept.  In isolation the 16x16x32 form
// sustains 15-17 % more products per second on plane data under the power cap (half the accumulator traffic per MAC; profiles/r05_probe_mfma_shapes.txt); in this
// kernel the clock does rise by ~10 % but a phase takes 8-10 % more cycles (the DMA pieces / fragment prefetches between MFMA groups are covered by 16-clock instead of
// 32-clock MFMAs): -3 % on the layers of this kernel, +1 % end to end (profiles/r05_mfma16_ab.txt, r05_conv_dma3_hooks.txt).  Its sums differ in the last bits from the
// 32x32x16 form's (SEMDEPTH_MFMA32), as this block's always did from conv_dma.hip's: which of the two blocks a layer takes is decided per ENGINE, not per call
// (conv_dma3_eligible).  The three-product (HS) form takes it for its 1x1 layers (-1.9 %) and stays on 32x32x16 for fc6 (2 % slower there).
template <int MODE, bool HS = false, bool TIMED = false, bool S16 = false>
__global__ __launch_bounds__(512, 1) void conv_dma3_kernel(const ConvParams p, int M, int tilesM, int tilesN) {
    long long tm_wait = 0, tm_bar = 0, tm_body = 0, tm_t0 = 0, tm_pro = 0;
    if constexpr (TIMED) tm_t0 = __builtin_amdgcn_s_memtime();
    constexpr int NPL = HS ? 2 : 3;                          // planes per operand
    static_assert(G3_NW * NPL * 32 * G3_ROW <= G3_LDS * 16, "epilogue slabs fit in the ring");
    __shared__ __attribute__((aligned(16))) u32x4 lds[G3_LDS];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm0 = (wave & 1) * (G3_MT * 32), wn0 = (wave >> 1) * (G3_NT * 32);
    int tid_;
    {
        const int nwg = tilesM * tilesN * (p.fold ? 4 : 1), bid = blockIdx.x, xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
        tid_ = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    // upsample-folded conv (ConvParams::fold): the parity is the outermost tile index
    const int par = tid_ / (tilesM * tilesN);
    tid_ -= par * (tilesM * tilesN);
    int tm = tid_ % tilesM, tn = tid_ / tilesM;
    if (p.m_fastest == 0) { tn = tid_ % tilesN; tm = tid_ / tilesN; }      // (walk N first: the tiles that share an X panel run side by side)
    if (p.rowgrp && p.Hout >= 2 * p.pad) {
        // row-grouped tiles differ in length (border rows skip taps): dispatch the longest first -- rows by decreasing distance from the
        // border -- so that the CUs that finish early pick up the short ones (the workgroups start in blockIdx order)
        const int per_row = (tilesM / p.Hout) * tilesN, rank = (int)blockIdx.x / per_row, rest = (int)blockIdx.x - rank * per_row;
        const int full = p.Hout - 2 * p.pad;
        int oy = p.pad + rank;
        if (rank >= full) { const int j = rank - full, d = p.pad - 1 - (j >> 1); oy = (j & 1) ? p.Hout - 1 - d : d; }
        const int groups = tilesM / p.Hout;
        tn = rest / groups;
        tm = (rest - tn * groups) * p.Hout + oy;
    }
    const int bm0 = tm * G3_BM, bn0 = tn * G3_BN;

    // X DMA: instruction j (16 per plane) covers pixels [16 j, 16 j + 16) x 4 octets; this wave issues j = wave and wave + 8
    // per-lane geometry of its two X-DMA pixels, packed (registers): pyx = oy | ox << 16, pik = image | octet << 26 | inside << 30
    int pyx[2], pik[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int m_l = (wave + G3_NW * i) * 16 + (lane >> 2);
        const int m = bm0 + m_l;
        const bool ok = m < M;
        const int hw = p.Hout * p.Wout;
        const int mm = ok ? m : 0;
        int img = mm / hw;
        int r = mm - img * hw;
        int oy = r / p.Wout;
        if (p.rowgrp) {                  // pixel order (image group, row, image of the group, column): a tile is ONE row of rowgrp images
            const int q1 = mm / p.Wout, q2 = q1 / p.rowgrp;
            r = mm - q1 * p.Wout;        // (column)
            oy = q2 % p.Hout;
            img = (q2 / p.Hout) * p.rowgrp + (q1 - q2 * p.rowgrp);
            r += oy * p.Wout;
        }
        pyx[i] = oy | ((r - oy * p.Wout) << 16);
        pik[i] = img | (((lane & 3) ^ ((m_l >> 2) & 3)) << 26) | ((ok ? 1 : 0) << 30);      // (octet: the one this lane fetches into slot lane % 4)
    }
    const KEntry* __restrict__ const ktab = p.ktab + par * (p.Kpad / 32);
    const int CoutPad = p.CoutPad, Nmax = p.Nmax;
    const u32x4* __restrict__ const wt_hi = reinterpret_cast<const u32x4*>(p.wt) + (size_t)par * (p.Kpad / 8) * CoutPad;
    const size_t wplane = (size_t)(p.Kpad / 8) * CoutPad * (p.fold ? 4 : 1);     // units
    const u32x4* const zero = reinterpret_cast<const u32x4*>(p.zero16);
    // the k-tiles this tile runs: all of them, or (ConvParams::rowgrp: the tile is one output row) only the taps whose input row exists --
    // per 32-channel block the taps [t0, t0 + nt) of kh x kw; the skipped ones would multiply zero padding: the accumulators are bit for bit
    // what the full loop leaves
    int taps = p.Kpad / 32, t0 = 0, nt = taps;
    if (p.rowgrp) {
        const int oy_t = (bm0 / (p.rowgrp * p.Wout)) % p.Hout;
        const int dlo = oy_t < p.pad ? -oy_t : -p.pad, dhi = p.Hout - 1 - oy_t < p.pad ? p.Hout - 1 - oy_t : p.pad;
        taps = p.kh * p.kw; t0 = (dlo + p.pad) * p.kw; nt = (dhi - dlo + 1) * p.kw;
    }
    const int ktiles = (p.Kpad / 32 / taps) * nt;             // k-tiles this tile runs
    // cursor over them: actual k-tile index and position inside the window
    struct KCur { int idx, tt; };
    auto knext = [&](KCur c) { ++c.idx; if (++c.tt == nt) { c.tt = 0; c.idx += taps - nt; } return c; };
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) void*)lds;

    // FLAT: byte offsets of this lane's two pixels (+ its octet) in the source geometry of the first k-tile (A) and of the last one (B; the
    // same as A for a single source); a k-tile takes B when its geometry differs from A's
    unsigned offA0 = 0u, offA1 = 0u, offB0 = 0u, offB1 = 0u;
    int geoA_W = 0, geoA_C = 0, geoA_st = 0;
    constexpr bool FLAT = MODE == 1, PRE = MODE == 2;
    if constexpr (FLAT || PRE) {
        const KEntry ea = g3load_kentry(ktab), eb = g3load_kentry(ktab + p.Kpad / 32 - 1);
        geoA_W = ea.W; geoA_C = ea.C; geoA_st = (ea.flags >> 4) & 3;
        const int sa = geoA_st, sb = (eb.flags >> 4) & 3;
        auto off_of = [&](int yx, int ik, int H_, int W_, int C_, int st_) {
            const int oy = yx & 0xffff, ox = yx >> 16, img = ik & 0x3ffffff, oct = (ik >> 26) & 3;
            return ((ik >> 30) & 1) ? (unsigned)((((img * H_ + oy * st_) * W_ + ox * st_) * C_ + oct * 8) * 2) : (unsigned)(oct * 16);
        };
        offA0 = off_of(pyx[0], pik[0], ea.H, ea.W, ea.C, sa); offA1 = off_of(pyx[1], pik[1], ea.H, ea.W, ea.C, sa);
        if constexpr (FLAT) { offB0 = off_of(pyx[0], pik[0], eb.H, eb.W, eb.C, sb); offB1 = off_of(pyx[1], pik[1], eb.H, eb.W, eb.C, sb); }
    }
    int pyq[2] = {0, 0};                 // PRE: the only per-lane geometry the k-loop keeps beside the two offsets
    if constexpr (PRE) {
#pragma unroll
        for (int i = 0; i < 2; ++i) pyq[i] = (pyx[i] & 0x7fff) | (((pik[i] >> 30) & 1) << 15) | (pyx[i] & (int)0xffff0000);
    }
    // one DMA instruction of a pair: piece 0, 1 = the weight plane's two instructions of this wave, 2, 3 = the X plane's.
    // Everything of a piece's address that depends on the k-tile only -- the weight panel's row, the source's plane stride (a 64-bit product of four table
    // fields), the tap offset, the geometry test -- is formed ONCE per k-tile (KCtx, when the issue cursor moves on) instead of in every piece: a piece was
    // 12-35 scalar instructions between two MFMA groups, four times per phase, and the timed copy without its pieces runs a phase in 175 clocks less
    // (profiles/r05_conv_dma3_hooks.txt).
    struct KCtx { const u32x4* wrow; const uint16_t* xb; size_t plane; int isA; KEntry e; };
    auto make_ctx = [&](int kt, const KEntry& e) {
        KCtx c;
        c.e = e;
        c.wrow = wt_hi + (size_t)(kt * 4) * CoutPad + bn0;
        c.plane = (size_t)Nmax * e.H * e.W * e.C;
        c.isA = (e.W == geoA_W && e.C == geoA_C && ((e.flags >> 4) & 3) == geoA_st) ? 1 : 0;          // (wave-uniform)
        c.xb = reinterpret_cast<const uint16_t*>(e.base) + (PRE ? ((ptrdiff_t)e.dy * e.W + e.dx) * e.C : (ptrdiff_t)0);      // PRE: + the tap's offset
        return c;
    };
    auto issue_x1 = [&](const KCtx& c, int pl, int slot, int i) {     // X plane pl of the k-tile of context c -> ring slot
        const KEntry& e = c.e;
        const int st = (e.flags >> 4) & 3, up = e.flags & 1;
        const unsigned dst = lds0 + (unsigned)(slot * G3_PAIR * 16);
        if constexpr (FLAT) {
            const uint16_t* sbase = c.xb + (size_t)pl * c.plane;
            g3dma16s(sbase, c.isA ? (i == 0 ? offA0 : offA1) : (i == 0 ? offB0 : offB1), dst + (unsigned)((wave + G3_NW * i) * 1024));
            return;
        }
        if constexpr (PRE) {             // (ONE source geometry: ConvParams::noup is set for single-source layers only)
            // scalar: plane base + the tap's offset; per lane: the pixel offset of tap (0, 0) and the in-image test (pyq = oy | m < M << 15 | ox << 16)
            const uint16_t* sbase = c.xb + (ptrdiff_t)pl * (ptrdiff_t)c.plane;
            const int iy = ((pyq[i] & 0x7fff) << (st - 1)) + e.dy, ix = ((pyq[i] >> 16) << (st - 1)) + e.dx;      // (stride 1 or 2)
            const bool ok = ((pyq[i] >> 15) & 1) && (unsigned)iy < (unsigned)e.H && (unsigned)ix < (unsigned)e.W;
            const unsigned char* px = reinterpret_cast<const unsigned char*>(sbase) + (i == 0 ? offA0 : offA1);
            g3dma16(ok ? reinterpret_cast<const u32x4*>(px) : zero, dst + (unsigned)((wave + G3_NW * i) * 1024));
            return;
        }
        int iy = (pyx[i] & 0xffff) * st + e.dy, ix = (pyx[i] >> 16) * st + e.dx;
        const bool ok = ((pik[i] >> 30) & 1) && iy >= 0 && ix >= 0 && iy < (e.H << up) && ix < (e.W << up);
        iy >>= up; ix >>= up;
        const uint16_t* px = reinterpret_cast<const uint16_t*>(e.base) + ((size_t)((pik[i] & 0x3ffffff) * e.H + iy) * e.W + ix) * e.C + ((pik[i] >> 26) & 3) * 8 + (size_t)pl * c.plane;
        g3dma16(ok ? reinterpret_cast<const u32x4*>(px) : zero, dst + (unsigned)((wave + G3_NW * i) * 1024));
    };
    // weight DMA: piece i of a plane covers units [(wave + 8 i) 64, + 64) of [k-octet][BN channels]; per-lane byte offset into the panel
    // (piece 1 is piece 0 two k-octets on: the same per-lane offset from a scalar base 2 CoutPad units further)
    static_assert(G3_NW * 64 == 2 * G3_BN, "piece 1 of a weight plane starts two k-octets after piece 0");
    const int wu0 = wave * 64 + lane;                          // unit inside the plane: [kg][n]
    const unsigned woff0 = (unsigned)(((wu0 / G3_BN) * CoutPad + wu0 % G3_BN) * 16);
    auto issue_w1 = [&](const KCtx& c, int pl, int slot, int i) {       // weight plane pl of the k-tile of context c -> ring slot
        const u32x4* base = c.wrow + (size_t)pl * wplane + (size_t)(2 * i) * CoutPad;       // (wave-uniform: SGPRs)
        g3dma16s(base, woff0, lds0 + (unsigned)((slot * G3_PAIR + G3_XPL + (wave + G3_NW * i) * 64) * 16));
    };
    auto issue_pair = [&](int kt, int wpl, int xpl, int slot) {          // (prologue: all four instructions at once) weight plane wpl, X plane xpl of k-tile kt
        const KCtx c = make_ctx(kt, g3load_kentry(ktab + kt));
        issue_w1(c, wpl, slot, 0); issue_w1(c, wpl, slot, 1);
        issue_x1(c, xpl, slot, 0); issue_x1(c, xpl, slot, 1);
    };
    // plane indices in memory: X hi = 0 (bf16 x 3: mid = 1, lo = 2; HS: scaled lo = 1); W hi = 0 (bf16 x 3: mid = 1, lo = 2; HS: lo = 1).
    // Phase tag PH (0 lo, 1 mid, 2 hi; HS runs 0 and 2) -> the weight plane it multiplies and the X plane it reads for the first time
    // (S16 groups the products by X plane: phase lo reads X_lo and W_hi, mid X_mid and W_mid, hi X_hi and W_lo)
    auto wpl_of = [](int ph) { return S16 ? (HS ? (ph == 0 ? 0 : 1) : ph) : (HS ? (ph == 0 ? 1 : 0) : 2 - ph); };
    auto xpl_of = [](int ph) { return S16 ? (HS ? (ph == 0 ? 1 : 0) : 2 - ph) : (HS ? (ph == 0 ? 0 : 1) : ph); };
    constexpr int NPH = HS ? 2 : 3;                          // phases per k-tile

    f32x16 acc[S16 ? 1 : G3_MT][S16 ? 1 : G3_NT];
    f32x4 acc16[S16 ? 2 * G3_MT : 1][S16 ? 2 * G3_NT : 1];    // S16: [16-pixel block][16-channel block]; lane l holds pixel l & 15, channels 4 (l >> 4) .. + 3
    if constexpr (S16) {
#pragma unroll
        for (int a = 0; a < 2 * G3_MT; ++a)
#pragma unroll
            for (int b = 0; b < 2 * G3_NT; ++b) acc16[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    } else {
#pragma unroll
    for (int a = 0; a < G3_MT; ++a)
#pragma unroll
        for (int b = 0; b < G3_NT; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    }

    // phase q = 3 kt + ph reads weight plane (2 - ph) and, for the first and only time, X plane ph of k-tile kt from ring slot q % 4
    // (ph 0: lo, 1: mid, 2: hi)
    const int nphase = NPH * ktiles;
    // prologue = the issues of the (virtual) phases -3 .. -1: the pairs of phases 0, 1, 2 (HS: the third one is the lo pair of the SECOND k-tile)
    issue_pair(t0, wpl_of(0), xpl_of(0), 0);
    if constexpr (HS) {
        issue_pair(t0, wpl_of(2), xpl_of(2), 1);
        issue_pair(knext(KCur{t0, 0}).idx, wpl_of(0), xpl_of(0), 2);
    } else {
        issue_pair(t0, wpl_of(1), xpl_of(1), 1);
        issue_pair(t0, wpl_of(2), xpl_of(2), 2);
    }
    int prev1 = 4;                                            // DMA instructions this wave issued in the previous phase
    const int frow = lane & 31, fk = lane >> 5;
    // Fragment registers that live across phase boundaries: the LDS reads a phase starts with are issued in the TAIL of the phase before
    // it (pair q + 1 is complete and visible from barrier q on: every wave waits for it there), so a phase's first MFMAs follow its barrier
    // directly instead of waiting out an LDS round trip with the MFMA pipes drained.
    u32x4 xk[2][2][G3_MT];                                    // X fragments of the hi and mid planes, kept for the k-tile: [plane][k-step][a]
    u32x4 wn[G3_NT], xl[G3_MT];                               // the next phase's first-k-step weight fragments; X_lo fragments of a k-step
    u32x4 wk[2][2 * G3_NT], wl[2 * G3_NT], xa[G3_MT], xb[G3_MT];      // S16: kept W_hi / W_mid blocks, W_lo blocks, the two halves of the phase's X plane
    auto wfrag = [&](int slot, int s, int b) { return lds[slot * G3_PAIR + G3_XPL + (2 * s + fk) * G3_BN + wn0 + b * 32 + frow]; };
    auto xfrag = [&](int slot, int s, int a) {
        const int mrow = wm0 + a * 32 + frow;
        return lds[slot * G3_PAIR + mrow * 4 + ((2 * s + fk) ^ ((mrow >> 2) & 3))];
    };
    // S16 fragments: lane l supplies row / column l & 15 and the k-octet l >> 4 of the k-tile.  Weights: 16-channel block b (0 .. 3); X: pixel half h (0, 1), 16-pixel
    // block a4 (0 .. 3) of the half -- xk[plane][h][a4], xl[a4] and wn[0 .. 1] have the shapes of the 32x32x16 form's arrays
    const int l16 = lane & 15, ko = lane >> 4;
    auto wfrag16 = [&](int slot, int b) { return lds[slot * G3_PAIR + G3_XPL + ko * G3_BN + wn0 + b * 16 + l16]; };
    auto xfrag16 = [&](int slot, int h, int a4) {
        const int mrow = wm0 + (4 * h + a4) * 16 + l16;
        return lds[slot * G3_PAIR + mrow * 4 + (ko ^ ((mrow >> 2) & 3))];
    };
    // heads of phase 0: pairs 0 and 1 have landed once only the third prologue pair is outstanding
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int b = 0; b < G3_NT; ++b) { if constexpr (S16) { wk[0][b] = wfrag16(0, b); wk[0][b + G3_NT] = wfrag16(0, b + G3_NT); } else wn[b] = wfrag(0, 0, b); }
#pragma unroll
    for (int a = 0; a < G3_MT; ++a) { if constexpr (S16) xa[a] = xfrag16(0, 0, a); else xk[0][0][a] = xfrag(0, 0, a); }
    int q = 0;
    if constexpr (TIMED) tm_pro = __builtin_amdgcn_s_memtime() - tm_t0;
    KCur k1 = knext(KCur{t0, 0});                             // the k-tile whose pairs are being issued (the one after the k-tile being multiplied)
    KCtx c3 = make_ctx(k1.idx, g3load_kentry(ktab + k1.idx)); // (at least two k-tiles) its gather entry and address context
    // HS: the hi phase issues the lo pair of the k-tile after that one
    KCur k2 = knext(k1);
    KCtx c4 = c3;
    if constexpr (HS) { if (ktiles > 2) c4 = make_ctx(k2.idx, g3load_kentry(ktab + k2.idx)); }
    for (int kt = 0; kt < ktiles; ++kt) {
        auto phase = [&](auto ph_tag) {
            constexpr int PH = decltype(ph_tag)::value, NPX = PH + 1;
            long long ts0 = 0, ts1 = 0, ts2 = 0;
            if constexpr (TIMED) ts0 = __builtin_amdgcn_s_memtime();
            // pairs q and q + 1 have landed once at most the previous phase's DMAs are outstanding (pair q + 1 feeds the tail's prefetch)
            if (prev1 == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if constexpr (TIMED) ts1 = __builtin_amdgcn_s_memtime();
            __builtin_amdgcn_s_barrier();
            if constexpr (TIMED) ts2 = __builtin_amdgcn_s_memtime();
            // the pair of phase q + 3 goes into the slot phase q - 1 has just finished reading -- its four DMA instructions are spread
            // BEHIND the first four MFMA groups of this phase (at the top of the phase, with the MFMA pipe drained by the barrier, each
            // of them would cost its full issue latency)
            const bool doissue = q + 3 < nphase && !(TIMED && (p.sw & SW_X3_DIAG_TIMED));      // = 3 (kt + 1) + ph  (SEMDEPTH_X3_DIAG=7: timed copy without DMA pieces)
            auto piece = [&](int n) {
                if (!doissue) return;
                if constexpr (HS) {          // phase q + 3: from the lo phase the hi pair of the next k-tile, from the hi phase the lo pair of the one after it
                    if constexpr (PH == 0) { if (n < 2) issue_w1(c3, wpl_of(2), (q + 3) & 3, n); else issue_x1(c3, xpl_of(2), (q + 3) & 3, n - 2); }
                    else { if (n < 2) issue_w1(c4, wpl_of(0), (q + 3) & 3, n); else issue_x1(c4, xpl_of(0), (q + 3) & 3, n - 2); }
                } else {
                    if (n < 2) issue_w1(c3, 2 - PH, (q + 3) & 3, n);
                    else issue_x1(c3, PH, (q + 3) & 3, n - 2);
                }
            };
            prev1 = doissue ? 4 : 0;
            const int sq = q & 3, sn = (q + 1) & 3;            // this phase's ring slot, the next one's
            const bool next = q + 1 < nphase;
            // this phase's first k-step runs on fragments the phase before prefetched; its second k-step's are read now (lo / mid: behind
            // the barrier, under the first k-step's MFMAs; hi: between its k-steps, registers)
            u32x4 w[2][G3_NT];
#pragma unroll
            for (int b = 0; b < G3_NT; ++b) w[0][b] = wn[b];
            auto second_step = [&]() {                         // (issued behind the phase's first MFMA group)
                if constexpr (PH < 2) {
#pragma unroll
                    for (int b = 0; b < G3_NT; ++b) w[1][b] = wfrag(sq, 1, b);
#pragma unroll
                    for (int a = 0; a < G3_MT; ++a) xk[PH][1][a] = xfrag(sq, 1, a);
                }
            };
            // the tail's prefetch for phase q + 1: its first k-step's weight fragments and the first-k-step fragments of the X plane it
            // reads for the first time (lo -> mid: X_mid, mid -> hi: X_lo, hi -> lo of the next k-tile: X_hi, once the kept X_hi
            // fragments of this k-tile's first k-step have been used)
            auto tail = [&]() {
                if (!next) return;
#pragma unroll
                for (int b = 0; b < G3_NT; ++b) wn[b] = wfrag(sn, 0, b);
#pragma unroll
                for (int a = 0; a < G3_MT; ++a) {
                    if constexpr (PH == 0) { if constexpr (HS) xl[a] = xfrag(sn, 0, a); else xk[1][0][a] = xfrag(sn, 0, a); }      // (HS: the next phase is the hi phase)
                    else if constexpr (PH == 1) xl[a] = xfrag(sn, 0, a);
                    else xk[0][0][a] = xfrag(sn, 0, a);
                }
            };
#pragma unroll
            for (int s = 0; s < 2; ++s) {
#pragma unroll
                for (int pl = 0; pl < (PH < 2 ? NPX : (HS ? 1 : 2)); ++pl) {        // the kept planes
#pragma unroll
                    for (int b = 0; b < G3_NT; ++b) {
#pragma unroll
                        for (int a = 0; a < G3_MT; ++a)
                            acc[a][b] = mfma_frag<HS>(w[s][b], xk[pl][s][a], acc[a][b]);
                        const int grp = (s * (PH < 2 ? NPX : (HS ? 1 : 2)) + pl) * G3_NT + b;      // MFMA groups of four so far (lo phase: 4 in all)
                        if (PH == 0 ? true : (s == 0 && grp < 4)) {
                            __builtin_amdgcn_sched_barrier(0);
                            if (grp == 0) second_step();
                            piece(grp);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                    if (PH == 1 && s == 1 && pl == 0) {                  // mid: before its last eight MFMAs
                        __builtin_amdgcn_sched_barrier(0);
                        tail();
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                if (PH == 0 && s == 0) {                                 // lo: between its two k-steps
                    __builtin_amdgcn_sched_barrier(0);
                    tail();
                    __builtin_amdgcn_sched_barrier(0);
                }
                if constexpr (PH == 2) {
                    if (s == 1) {                                        // hi: before its last eight MFMAs (X_hi(0) of this k-tile is done with)
                        __builtin_amdgcn_sched_barrier(0);
                        tail();
                        __builtin_amdgcn_sched_barrier(0);
                    }
#pragma unroll
                    for (int b = 0; b < G3_NT; ++b) {
                        const u32x4 wx = HS ? hs_wscaled(w[s][b]) : w[s][b];      // (HS: the scaled-lo plane multiplies w_hi * 2^-11)
#pragma unroll
                        for (int a = 0; a < G3_MT; ++a)
                            acc[a][b] = mfma_frag<HS>(wx, xl[a], acc[a][b]);
                        if constexpr (HS) {
                            if (s == 0) {                                    // (HS: the hi phase has two kept-plane groups in its first k-step: pieces 2 and 3 go here)
                                __builtin_amdgcn_sched_barrier(0);
                                piece(2 + b);
                                __builtin_amdgcn_sched_barrier(0);
                            }
                        }
                    }
                    if (s == 0) {
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int b = 0; b < G3_NT; ++b) w[1][b] = wfrag(sq, 1, b);
#pragma unroll
                        for (int a = 0; a < G3_MT; ++a) xl[a] = xfrag(sq, 1, a);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
            ++q;
            if constexpr (TIMED) { const long long ts3 = __builtin_amdgcn_s_memtime(); tm_wait += ts1 - ts0; tm_bar += ts2 - ts1; tm_body += ts3 - ts2; }
        };
        // S16: the phases on 16x16x32 MFMAs, grouped BY X PLANE: a wave's tile is 128 pixels x 64 channels, so the weight fragments are the smaller set to keep
        // in registers (2 planes x 4 blocks = 32 VGPRs; the 32x32x16 form keeps 64 VGPRs of X fragments) -- with 32 accumulator tuples the kept-X form spills.
        //        phase lo : X_lo  x W_hi                  32 MFMAs per wave      (HS: X_lo x W_hi 2^-11)
        //        phase mid: X_mid x (W_hi, W_mid)         64
        //        phase hi : X_hi  x (W_hi, W_mid, W_lo)   96                     (HS: X_hi x (W_hi, W_lo))
        // A phase reads ONE new X plane (two halves of four 16-pixel blocks: xa prefetched by the tail of the phase before, xb behind the first MFMA group) and one
        // new weight plane (W_hi in lo -- its first two blocks prefetched by the hi phase's tail, straight into the kept registers once the hi phase's W_hi products
        // are done --, W_mid in mid, W_lo in hi: behind the first group, while the products of the kept planes run).  Groups of four MFMAs are half as long as the 32x32x16 form's: the DMA pieces go behind every other group.
        auto phase16 = [&](auto ph_tag) {
            constexpr int PH = decltype(ph_tag)::value;
            constexpr int KP = HS ? (PH == 0 ? 1 : 2) : PH + 1;       // weight planes multiplied in this phase
            // the six-product hi phase holds the most registers (accumulators + three weight planes + both X halves = 208) while its DMA pieces need their address
            // temporaries: the second half's X fragments are read behind the LAST piece there (five groups before their first use) instead of the first
            constexpr bool LATE_XB = KP == 3;
            long long ts0 = 0, ts1 = 0, ts2 = 0;
            if constexpr (TIMED) ts0 = __builtin_amdgcn_s_memtime();
            if (prev1 == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if constexpr (TIMED) ts1 = __builtin_amdgcn_s_memtime();
            __builtin_amdgcn_s_barrier();
            if constexpr (TIMED) ts2 = __builtin_amdgcn_s_memtime();
            const bool doissue = q + 3 < nphase && !(TIMED && (p.sw & SW_X3_DIAG_TIMED));      // (SEMDEPTH_X3_DIAG=7: the timed copy WITHOUT its DMA pieces -- what do the hooks cost?)
            auto piece = [&](int n) {
                if (!doissue) return;
                if constexpr (HS) {
                    if constexpr (PH == 0) { if (n < 2) issue_w1(c3, wpl_of(2), (q + 3) & 3, n); else issue_x1(c3, xpl_of(2), (q + 3) & 3, n - 2); }
                    else { if (n < 2) issue_w1(c4, wpl_of(0), (q + 3) & 3, n); else issue_x1(c4, xpl_of(0), (q + 3) & 3, n - 2); }
                } else {
                    if (n < 2) issue_w1(c3, wpl_of(PH), (q + 3) & 3, n);
                    else issue_x1(c3, xpl_of(PH), (q + 3) & 3, n - 2);
                }
            };
            prev1 = doissue ? 4 : 0;
            const int sq = q & 3, sn = (q + 1) & 3;
            const bool next = q + 1 < nphase;
            auto second_step = [&]() {                         // (issued behind the phase's first MFMA group)
                if constexpr (PH == 0) { }                     // (all four W_hi blocks came with the hi phase's tail: the lo phase's third group is only 128 clocks away)
                else {
#pragma unroll
                    for (int b = 0; b < 2 * G3_NT; ++b) { if constexpr (PH == 1) wk[1][b] = wfrag16(sq, b); else wl[b] = wfrag16(sq, b); }
                }
                if constexpr (!LATE_XB) {
#pragma unroll
                    for (int a = 0; a < G3_MT; ++a) xb[a] = xfrag16(sq, 1, a);
                }
            };
            auto tail = [&]() {                                // the next phase's first-half X fragments (+ from the hi phase: the first two blocks of the next W_hi)
                if (!next) return;
                if constexpr (PH == 2) {                       // (this k-tile's W_hi products are done: its four registers take the next k-tile's blocks)
#pragma unroll
                    for (int b = 0; b < 2 * G3_NT; ++b) wk[0][b] = wfrag16(sn, b);
                }
#pragma unroll
                for (int a = 0; a < G3_MT; ++a) xa[a] = xfrag16(sn, 0, a);
            };
            // weight plane of product kp: the kept planes first (in the first half the new plane's fragments are still arriving)
            auto wsel = [&](int h, int kp, int b) -> u32x4 {
                // 0 = W_hi, 1 = W_mid (HS: W_lo), 2 = W_lo -- the SAME order in both halves: a pixel's sum must not depend on where in a tile it lands (a frame
                // sits at different tile offsets in calls of different sizes); the new plane's fragments are still arriving during the first half's kept planes
                const int k = kp; (void)h;
                if constexpr (HS) {
                    if constexpr (PH == 0) return hs_wscaled(wk[0][b]);
                    else return k == 0 ? wk[0][b] : wl[b];
                } else return k == 0 ? wk[0][b] : k == 1 ? wk[1][b] : wl[b];
            };
#pragma unroll
            for (int h = 0; h < 2; ++h) {
#pragma unroll
                for (int kp = 0; kp < KP; ++kp) {
#pragma unroll
                    for (int b = 0; b < 2 * G3_NT; ++b) {
                        const u32x4 wv = wsel(h, kp, b);
#pragma unroll
                        for (int a = 0; a < G3_MT; ++a)
                            acc16[4 * h + a][b] = mfma_frag16<HS>(wv, h == 0 ? xa[a] : xb[a], acc16[4 * h + a][b]);
                        const int grp = (h * KP + kp) * (2 * G3_NT) + b;             // MFMA groups of four so far
                        if (!(grp & 1) && grp < 8) {
                            __builtin_amdgcn_sched_barrier(0);
                            if (grp == 0) second_step();
                            piece(grp >> 1);
                            if constexpr (LATE_XB) {
                                if (grp == 6) {
#pragma unroll
                                    for (int a = 0; a < G3_MT; ++a) xb[a] = xfrag16(sq, 1, a);
                                }
                            }
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                    if (KP > 1 && h == 1 && kp == 0) {                   // second half, after its first product (W_hi's: in the hi phase the tail refills its registers)
                        __builtin_amdgcn_sched_barrier(0);
                        tail();
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                if (KP == 1 && h == 0) {                                 // one product only: between the halves (xa is free from here on)
                    __builtin_amdgcn_sched_barrier(0);
                    tail();
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            ++q;
            if constexpr (TIMED) { const long long ts3 = __builtin_amdgcn_s_memtime(); tm_wait += ts1 - ts0; tm_bar += ts2 - ts1; tm_body += ts3 - ts2; }
        };
        if constexpr (S16) {
            phase16(IntTag<0>{});
            if constexpr (!HS) phase16(IntTag<1>{});
            phase16(IntTag<2>{});
        } else {
            phase(IntTag<0>{});
            if constexpr (!HS) phase(IntTag<1>{});
            phase(IntTag<2>{});
        }
        if constexpr (HS) {
            k1 = k2; c3 = c4;
            k2 = knext(k2);
            if (kt + 3 < ktiles) c4 = make_ctx(k2.idx, g3load_kentry(ktab + k2.idx));
        } else {
            k1 = knext(k1);
            if (kt + 2 < ktiles) c3 = make_ctx(k1.idx, g3load_kentry(ktab + k1.idx));       // (one scalar load + the context per k-tile, behind the hi phase's MFMAs)
        }
    }

    // ---- epilogue: bias + activation in f32, exact three-way split, LDS transpose (one slab per plane and wave), 16-byte runs ----
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // SEMDEPTH_X3_DIAG (decomposition runs, latched in the handle's switches; 0 in production): 1 = no output stores, 2 = no epilogue at all
    const int diag = TIMED ? 0 : SD_DIAG_BITS(p.sw);
    if (diag & 2) {
        if constexpr (S16) { if (acc16[0][0][0] == 12345.678f) p.out[0] = acc16[7][3][3]; }
        else { if (acc[0][0][0] == 12345.678f) p.out[0] = acc[1][1][3]; }
        return;
    }
    // (the epilogue's per-lane values are formed from an opaque copy of the lane id: nothing of it is hoisted above the k-loop, whose registers are all taken)
    int lane_e = lane;
    asm volatile("" : "+v"(lane_e));
    auto ep3 = [&](auto tag) {
        constexpr int ACT = decltype(tag)::value;
        constexpr int ROW = G3_ROW;
        unsigned char* slab = reinterpret_cast<unsigned char*>(lds) + wave * (NPL * 32 * ROW);
        constexpr int SEGS = G3_NT * 4, PPP = 64 / SEGS;
        const int seg = lane_e % SEGS, prow = lane_e / SEGS;
        uint16_t* const out_hi = reinterpret_cast<uint16_t*>(p.out);
        const int m0 = bm0 + wm0, n0 = bn0 + wn0;
        // the bias of this lane's channels, loaded ONCE (round 6: the loads sat inside the loop below, one L1 round trip in front of every group of four values and
        // not merged across the asm waits: 32 dependent loads per wave and tile): S16 four vectors (channels 16 r4 + 4 (lane >> 4)), 32x32x16 eight (32 b + 8 r4 + 4 (lane >> 5))
        f32x4 bias_v[S16 ? 4 : 4 * G3_NT];
#pragma unroll
        for (int i = 0; i < (S16 ? 4 : 4 * G3_NT); ++i)
            bias_v[i] = *reinterpret_cast<const f32x4*>(p.bias + n0 + (S16 ? i * 16 + 4 * (lane_e >> 4) : (i >> 2) * 32 + 8 * (i & 3) + 4 * (lane_e >> 5)));
#pragma unroll
        for (int a = 0; a < G3_MT; ++a) {
#pragma unroll
            for (int b = 0; b < G3_NT; ++b)
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    // four consecutive channels of one pixel per lane.  32x32x16: pixel lane & 31 of the 32, channels b 32 + 8 r4 + 4 (lane_e >> 5);
                    // S16: (b, r4) = (16-pixel block of the 32, 16-channel block): pixel 16 b + (lane_e & 15), channels 16 r4 + 4 (lane_e >> 4)
                    const int nl = S16 ? r4 * 16 + 4 * (lane_e >> 4) : b * 32 + 8 * r4 + 4 * (lane_e >> 5);
                    const int srow = S16 ? b * 16 + (lane_e & 15) : (lane_e & 31);
                    f32x4 v;
                    if constexpr (S16) v = acc16[2 * a + b][r4];
                    else v = f32x4{acc[a][b][4 * r4], acc[a][b][4 * r4 + 1], acc[a][b][4 * r4 + 2], acc[a][b][4 * r4 + 3]};
                    uint2 h, m, l;
                    if constexpr (HS) {
                        v = v * p.alpha + bias_v[S16 ? r4 : b * 4 + r4];
                        v = act_split4<ACT>(v);
                        split4_hs(v, h, m, p.sat);
                        *reinterpret_cast<uint2*>(slab + srow * ROW + nl * 2) = h;
                        *reinterpret_cast<uint2*>(slab + 32 * ROW + srow * ROW + nl * 2) = m;
                    } else {
                    v += bias_v[S16 ? r4 : b * 4 + r4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = act_x3<ACT>(v[r]);
                    split4_x3(v, h, m, l);
                    *reinterpret_cast<uint2*>(slab + srow * ROW + nl * 2) = h;
                    *reinterpret_cast<uint2*>(slab + 32 * ROW + srow * ROW + nl * 2) = m;
                    *reinterpret_cast<uint2*>(slab + 64 * ROW + srow * ROW + nl * 2) = l;
                    }
                }
            __builtin_amdgcn_wave_barrier();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int ps = 0; ps < 32 / PPP; ++ps) {
                const int pix = ps * PPP + prow;
                const int mo = m0 + a * 32 + pix;
                if (mo < M && !(diag & 1)) {
                    size_t opix = (size_t)mo;
                    if (p.rowgrp) {              // (image group, row, image of the group, column) -> (image, row, column)
                        const int q1 = mo / p.Wout, q2 = q1 / p.rowgrp, g = q2 / p.Hout;
                        opix = ((size_t)((g * p.rowgrp + (q1 - q2 * p.rowgrp)) * p.Hout + (q2 - g * p.Hout))) * p.Wout + (mo - q1 * p.Wout);
                    }
                    if (p.fold) {                // source pixel (img, i, j) of parity (py, px) -> output pixel (2 i + py, 2 j + px)
                        const int hw = p.Hout * p.Wout, img = mo / hw, r = mo - img * hw, i = r / p.Wout, j = r - i * p.Wout;
                        opix = ((size_t)(img * 2 * p.Hout + 2 * i + (par >> 1))) * (2 * p.Wout) + 2 * j + (par & 1);
                    }
                    uint16_t* o = out_hi + opix * p.Cout + n0 + seg * 8;
#pragma unroll
                    for (int pl = 0; pl < NPL; ++pl)
                        *reinterpret_cast<u32x4*>(o + pl * p.out_plane) = *reinterpret_cast<const u32x4*>(slab + pl * 32 * ROW + pix * ROW + seg * 16);
                }
            }
            __builtin_amdgcn_wave_barrier();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    };
    long long tm_e0 = 0;
    if constexpr (TIMED) tm_e0 = __builtin_amdgcn_s_memtime();
    if (p.act == ACT_RELU) ep3(ActTag<ACT_RELU>{});
    else if (p.act == ACT_ELU) ep3(ActTag<ACT_ELU>{});
    else ep3(ActTag<ACT_NONE>{});
    if constexpr (TIMED) {
        const long long te = __builtin_amdgcn_s_memtime();
        if ((int)blockIdx.x == (int)gridDim.x / 2 && lane == 0 && (wave == 0 || wave == 4))
            printf("[conv_dma3 timed] %s K=%d ktiles=%d wave %d: prologue %lld | per phase (%d phases): wait %.1f barrier %.1f body %.1f | loop %lld epilogue %lld total %lld ticks\n",
                   HS ? "HS" : "x3", p.Kpad, ktiles, wave, tm_pro, nphase, (double)tm_wait / nphase, (double)tm_bar / nphase, (double)tm_body / nphase,
                   tm_wait + tm_bar + tm_body, te - tm_e0, te - tm_t0);
    }
}

int conv_dma3_mode(const ConvParams& p);
// layers the 256 x 256 phased block takes: bf16 x 3, all-vec K axis, Cout a multiple of 256 and enough blocks to occupy the chip
bool conv_dma3_eligible(const ConvParams& p) {
    if (!(p.x3 || p.f16 == 4) || !p.vec || !p.zero16 || p.pool || p.out_planar16 || p.residual || p.Cout % G3_BN || p.Kpad < 64 || p.CoutPad != p.Cout) return false;
    // SD_PREC_F16X2: measured against the two-stage block of conv_dma.hip (profiles/r05_hs_phased_gemm_ab.txt), the two-phase ring wins on the 1x1 layers (fc7
    // 1.50 -> 1.29 ms, the res2 / res5 block tails 3-12 %) and -- since a piece's address arithmetic is formed once per k-tile -- on the row-grouped fc6
    // (6.93 -> 6.50 ms: it also skips the taps on padding rows), and loses on the small strided 3x3 of res4_6 (0.135 -> 0.178): the 1x1 layers and fc6 only
    // (SEMDEPTH_HS_PHASED_TAPS: every tap layer, the A/B switch of that measurement)
    // (the rule must not look at the call: the k x k stride-1 single-source layers -- fc6 --, row-grouped or not)
    if (p.f16 == 4 && conv_dma3_mode(p) != 1 && !(p.kh >= 3 && p.stride == 1 && !p.fold && p.noup) && !(p.sw & SW_HS_TAPS)) return false;
    if (p.fold) return true;                                   // (a folded layer always runs here when it can: its results must not depend on the batch)
    if (p.sw & SW_NO_DMA3) return false;                       // (A/B switch of the handle)
    // enough tiles to occupy the chip -- counted on a FULL pass of the engine (ConvParams::Nmax), not on the frames of this call: this block and conv_dma.hip's
    // two-stage block differ in the last bits of their sums, and a frame's result must not depend on how many frames the call carries (round 5: until then the
    // rule read the call's own pixel count, and at 512 x 1024 frame 0 of a call of one differed from frame 0 of a call of eight by 3e-7;
    // scripts/batch_independence_full_size.py)
    const long M = (long)(p.Nmax > 0 ? p.Nmax : p.N) * p.Hout * p.Wout;
    return ((M + G3_BM - 1) / G3_BM) * (p.Cout / G3_BN) >= 128;
}

// gather variant a layer runs: 1 = 1x1 layers (one offset per source geometry), 2 = tap layers without upsample (fc6, the folded upconvs, strided
// 3x3: precomputed pixel offset + scalar tap offset), 0 = the general gather
int conv_dma3_mode(const ConvParams& p) { return (p.flat && !p.fold && !p.rowgrp) ? 1 : p.noup ? 2 : 0; }

hipError_t launch_conv_dma3(const ConvParams& p, hipStream_t s) {
    if (!conv_dma3_eligible(p)) return hipErrorInvalidValue;
    const long M = (long)p.N * p.Hout * p.Wout;
    const int tilesM = (int)((M + G3_BM - 1) / G3_BM), tilesN = p.Cout / G3_BN;
    const dim3 grid((unsigned)(tilesM * tilesN * (p.fold ? 4 : 1)));
    const int mode = conv_dma3_mode(p);
    // the 16x16x32 form: every bf16 x 3 layer and the three-product engine's 1x1 layers (-1.9 % on them; its fc6 is 2 % faster on 32x32x16: r05_mfma16_ab.txt).
    // SEMDEPTH_MFMA32: the 32x32x16 form everywhere (same-box A/B)
    const bool s16 = !(p.sw & SW_MFMA32);
#define SD_G3(MODE_, HS_, TIMED_) do { if (s16) hipLaunchKernelGGL((conv_dma3_kernel<MODE_, HS_, TIMED_, !(HS_) || (MODE_) == 1>), grid, dim3(512), 0, s, p, (int)M, tilesM, tilesN); \
                                       else hipLaunchKernelGGL((conv_dma3_kernel<MODE_, HS_, TIMED_, false>), grid, dim3(512), 0, s, p, (int)M, tilesM, tilesN); } while (0)
#ifdef SD_DEV_VARIANTS
    if (mode == 1 && (p.sw & SW_X3_DIAG_NOSTORE) && (p.sw & SW_X3_DIAG_NOMFMA)) {          // SEMDEPTH_X3_DIAG=3: the timed copy of the 1x1 form
        if (p.f16 == 4) SD_G3(1, true, true); else SD_G3(1, false, true);
        return hipGetLastError();
    }
#endif
    if (p.f16 == 4) {           // SD_PREC_F16X2: the two-plane, two-phase form
        if (mode == 1) SD_G3(1, true, false); else if (mode == 2) SD_G3(2, true, false); else SD_G3(0, true, false);
        return hipGetLastError();
    }
    if (mode == 1) SD_G3(1, false, false); else if (mode == 2) SD_G3(2, false, false); else SD_G3(0, false, false);
#undef SD_G3
    return hipGetLastError();
}

}  // namespace sd
