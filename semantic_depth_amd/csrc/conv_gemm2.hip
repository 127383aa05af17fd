// The 1x1 GEMM layers of the three-product engine (SD_PREC_F16X2) as TWO workgroups per CU (round 6).
//
// conv_dma3.hip runs these layers -- the ResNet block tails (conv3 + projection over the concatenated K axis), conv1 of res4 / res5, fc7 -- on one 256 x 256 block per
// CU: eight waves, 256 registers each, 128 KB of ring.  Its MFMA pipes are 47-48 % busy at 2.0 GHz (profiles/r06_pmc_sq_f16x2.json: NOT at the power cap): a tile's
// prologue (index arithmetic + the cold round trip of its first pairs, 9-19 k clocks) and epilogue (bias + ELU + HS split + LDS transposition + stores, 19-47 k
// clocks with the two waves of a SIMD serialising on the VALU) run with the matrix pipe idle, and they are 30-50 % of a tile at K <= 640
// (profiles/r05_conv_dma3_timed_f16x2.txt).  A second accumulator set to slice one tile's epilogue into the next tile's k-loop does not exist (2 x 128 KB of
// accumulators in a 512-KB register file), and rounds 3-5 dismissed the half-size block because it moves 1.5 x the L2 -> LDS bytes per MFMA -- which bounds a block
// that runs its MFMAs back to back, not one that is half idle.
//
// Here the block is 256 x 128: FOUR waves (2 x 2, wave tile 128 pixels x 64 channels: the same 128 accumulator registers and the same fragment scheme as conv_dma3's
// 16x16x32 form), a ring of THREE (X plane, weight plane) pairs of 16 + 8 KB = 72 KB, <= 256 registers per wave -- so TWO workgroups are resident per CU, one wave of
// each per SIMD, and they run out of phase by themselves: one's prologue, epilogue and barrier waits are the other's k-loop.  The hardware interleaves what the
// 512-register single-wave form would have had to schedule by hand.
//
// Same products in the same order per accumulator as conv_dma3's S16 HS form (per k-tile: X_lo x W_hi 2^-11, X_hi x W_hi, X_hi x W_lo): bit-identical outputs.
//        phase lo: reads (X_lo, W_hi) of the k-tile     32 MFMAs per wave (v_mfma_f32_16x16x32_f16)
//        phase hi: reads (X_hi, W_lo), W_hi kept        64
// Phase q reads ring slot q % 3 and issues the pair of phase q + 2 into the slot phase q - 1 has just finished with (six DMA instructions per wave, spread behind
// the first MFMA groups); it waits with a counted vmcnt for what was issued TWO phases ago.  No cross-phase fragment prefetch: the other workgroup's wave on the
// same SIMD covers the LDS round trip at the head of a phase.
#include <cstdlib>
#include "kernels.hpp"
#include "split_fmt.hpp"

namespace sd {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// one LDS-DMA with a uniform base in SGPRs and a 32-bit per-lane byte offset (conv_dma3.hip g3dma16s)
__device__ __forceinline__ void g2dma16s(const void* sbase, unsigned voff, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}
typedef int i32x8h __attribute__((ext_vector_type(8)));
__device__ __forceinline__ KEntry g2load_kentry(const KEntry* ptr) {
    i32x8h v;
    asm volatile("s_load_dwordx8 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(ptr) : "memory");
    KEntry e;
    e.base = reinterpret_cast<const float*>(((unsigned long long)(unsigned)v[1] << 32) | (unsigned)v[0]);
    e.H = v[2]; e.W = v[3]; e.C = v[4]; e.dy = v[5]; e.dx = v[6]; e.flags = v[7];
    return e;
}

constexpr int H2_BM = 256, H2_BN = 128, H2_NW = 4, H2_MT = 4, H2_NT = 2;
constexpr int H2_XPL = 4 * H2_BM;            // 16-B units of one X plane of a k-tile: [pixel][octet ^ swizzle]   (16 KB)
constexpr int H2_WPL = 4 * H2_BN;            // units of one weight plane of a k-tile: [k-octet][channel]          (8 KB)
constexpr int H2_RING = 3;
constexpr int H2_PAIR = H2_XPL + H2_WPL;
constexpr int H2_LDS = H2_RING * H2_PAIR;    // 72 KB: two workgroups per CU
constexpr int H2_ROW = H2_NT * 64 + 16;
constexpr int H2_XI = 16 / H2_NW;            // X-DMA instructions per wave and plane
constexpr int H2_NP = H2_XI + 2;             // DMA instructions per wave and pair

__global__ __launch_bounds__(64 * H2_NW, 2) void conv_gemm2_hs_kernel(const ConvParams p, int M, int tilesM, int tilesN) {
    static_assert(H2_NW * 2 * 32 * H2_ROW <= H2_LDS * 16, "epilogue slabs fit in the ring");
    static_assert(H2_NW * 64 == 2 * H2_BN, "piece 1 of a weight plane starts two k-octets after piece 0");
    __shared__ __attribute__((aligned(16))) u32x4 lds[H2_LDS];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm0 = (wave & 1) * (H2_MT * 32), wn0 = (wave >> 1) * (H2_NT * 32);
    int tid_;
    {
        const int nwg = tilesM * tilesN, bid = blockIdx.x, xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
        tid_ = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    int tm = tid_ % tilesM, tn = tid_ / tilesM;
    if (p.m_fastest == 0) { tn = tid_ % tilesN; tm = tid_ / tilesN; }      // (walk N first: the tiles that share an X panel run side by side)
    const int bm0 = tm * H2_BM, bn0 = tn * H2_BN;

    // X DMA: instruction j (16 per plane) covers pixels [16 j, 16 j + 16) x 4 octets; this wave issues j = wave + 4 i.  The layers are 1x1 without upsample
    // (ConvParams::flat): the pixel a lane fetches is the same for every k-tile of a source, its byte offset is formed once per source geometry (at most two)
    const KEntry* __restrict__ const ktab = p.ktab;
    const int CoutPad = p.CoutPad, Nmax = p.Nmax;
    const u32x4* __restrict__ const wt_hi = reinterpret_cast<const u32x4*>(p.wt);
    const size_t wplane = (size_t)(p.Kpad / 8) * CoutPad;      // units
    const int ktiles = p.Kpad / 32;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) void*)lds;
    unsigned offA[H2_XI], offB[H2_XI];
    int geoA_W, geoA_C, geoA_st;
    {
        const KEntry ea = g2load_kentry(ktab), eb = g2load_kentry(ktab + __builtin_amdgcn_readfirstlane(ktiles - 1));
        geoA_W = ea.W; geoA_C = ea.C; geoA_st = (ea.flags >> 4) & 3;
        const int sb = (eb.flags >> 4) & 3;
#pragma unroll
        for (int i = 0; i < H2_XI; ++i) {
            const int m_l = (wave + H2_NW * i) * 16 + (lane >> 2);
            const int m = bm0 + m_l;
            const bool ok = m < M;
            const int hw = p.Hout * p.Wout;
            const int mm = ok ? m : 0;
            const int img = mm / hw;
            const int r = mm - img * hw;
            const int oy = r / p.Wout, ox = r - oy * p.Wout;
            const int oct = (lane & 3) ^ ((m_l >> 2) & 3);       // (the octet this lane fetches into slot lane % 4)
            offA[i] = ok ? (unsigned)((((img * ea.H + oy * geoA_st) * ea.W + ox * geoA_st) * ea.C + oct * 8) * 2) : (unsigned)(oct * 16);
            offB[i] = ok ? (unsigned)((((img * eb.H + oy * sb) * eb.W + ox * sb) * eb.C + oct * 8) * 2) : (unsigned)(oct * 16);
        }
    }
    // everything of a pair's addresses that depends on the k-tile only, formed once per k-tile
    struct KCtx { const u32x4* wrow; const uint16_t* xb; size_t plane; int isA; };
    auto make_ctx = [&](int kt) {
        const KEntry e = g2load_kentry(ktab + __builtin_amdgcn_readfirstlane(kt));      // (the index is wave-uniform: keep the address in SGPRs)
        KCtx c;
        c.wrow = wt_hi + (size_t)(kt * 4) * CoutPad + bn0;
        c.plane = (size_t)Nmax * e.H * e.W * e.C;
        c.isA = (e.W == geoA_W && e.C == geoA_C && ((e.flags >> 4) & 3) == geoA_st) ? 1 : 0;          // (wave-uniform)
        c.xb = reinterpret_cast<const uint16_t*>(e.base);
        return c;
    };
    const int wu0 = wave * 64 + lane;                          // weight unit inside the plane: [kg][n]
    const unsigned woff0 = (unsigned)(((wu0 / H2_BN) * CoutPad + wu0 % H2_BN) * 16);
    // DMA instruction n of a pair: 0, 1 = the weight plane's, 2 .. = the X plane's
    auto piece = [&](const KCtx& c, int wpl, int xpl, int slot, int n) {
        if (n < 2) {
            const u32x4* base = c.wrow + (size_t)wpl * wplane + (size_t)(2 * n) * CoutPad;       // (wave-uniform: SGPRs)
            g2dma16s(base, woff0, lds0 + (unsigned)((slot * H2_PAIR + H2_XPL + (wave + H2_NW * n) * 64) * 16));
        } else {
            const int i = n - 2;
            const uint16_t* sbase = c.xb + (size_t)xpl * c.plane;
            // (selects, not an indexed array: a run-time index would put the offsets into scratch memory)
            const unsigned oa = i == 0 ? offA[0] : i == 1 ? offA[1] : i == 2 ? offA[2] : offA[3];
            const unsigned ob = i == 0 ? offB[0] : i == 1 ? offB[1] : i == 2 ? offB[2] : offB[3];
            g2dma16s(sbase, c.isA ? oa : ob, lds0 + (unsigned)(slot * H2_PAIR * 16) + (unsigned)((wave + H2_NW * i) * 1024));
        }
    };
    // planes in memory: X hi = 0, X scaled lo = 1; W hi = 0, W lo = 1.  phase lo reads (X_lo, W_hi), phase hi (X_hi, W_lo)
    f32x4 acc16[2 * H2_MT][2 * H2_NT];      // [16-pixel block][16-channel block]; lane l holds pixel l & 15, channels 4 (l >> 4) .. + 3
#pragma unroll
    for (int a = 0; a < 2 * H2_MT; ++a)
#pragma unroll
        for (int b = 0; b < 2 * H2_NT; ++b) acc16[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nphase = 2 * ktiles;
    {   // prologue: the pairs of phases 0 and 1 (both of k-tile 0)
        const KCtx c0 = make_ctx(0);
#pragma unroll
        for (int n = 0; n < H2_NP; ++n) piece(c0, 0, 1, 0, n);
#pragma unroll
        for (int n = 0; n < H2_NP; ++n) piece(c0, 1, 0, 1, n);
    }
    const int l16 = lane & 15, ko = lane >> 4;
    auto wfrag16 = [&](int slot, int b) { return lds[slot * H2_PAIR + H2_XPL + ko * H2_BN + wn0 + b * 16 + l16]; };
    auto xfrag16 = [&](int slot, int h, int a4) {
        const int mrow = wm0 + (4 * h + a4) * 16 + l16;
        return lds[slot * H2_PAIR + mrow * 4 + (ko ^ ((mrow >> 2) & 3))];
    };
    u32x4 wk[2 * H2_NT];                      // W_hi blocks of the k-tile (read in its lo phase, kept for its hi phase)
    int q = 0, slot = 0;                      // phase counter and its ring slot (q % 3)
    int prev = H2_NP;                         // DMA instructions this wave issued in the previous phase
    KCtx cn = make_ctx(ktiles > 1 ? 1 : 0);   // the k-tile whose pairs are being issued (the one after the k-tile being multiplied)
    for (int kt = 0; kt < ktiles; ++kt) {
        const bool more = kt + 1 < ktiles;
        // ---------------- phase lo: X_lo x (W_hi 2^-11)
        {
            if (prev == H2_NP) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");       // (H2_NP: the pair issued in the previous phase may still be in flight)
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            const int sn = slot == 0 ? 2 : slot - 1;           // the slot phase q - 1 has just finished with = (q + 2) % 3
            u32x4 xa[H2_MT], xb[H2_MT];
#pragma unroll
            for (int b = 0; b < 2 * H2_NT; ++b) wk[b] = wfrag16(slot, b);
#pragma unroll
            for (int a = 0; a < H2_MT; ++a) xa[a] = xfrag16(slot, 0, a);
#pragma unroll
            for (int a = 0; a < H2_MT; ++a) xb[a] = xfrag16(slot, 1, a);
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int b = 0; b < 2 * H2_NT; ++b) {
                    const u32x4 wv = hs_wscaled(wk[b]);
#pragma unroll
                    for (int a = 0; a < H2_MT; ++a)
                        acc16[4 * h + a][b] = mfma_frag16<true>(wv, h == 0 ? xa[a] : xb[a], acc16[4 * h + a][b]);
                    const int grp = h * (2 * H2_NT) + b;       // MFMA groups of four so far: the pair of phase q + 2 (the lo pair of k-tile kt + 1) behind the first six
                    if (grp < H2_NP && more) {
                        __builtin_amdgcn_sched_barrier(0);
                        piece(cn, 0, 1, sn, grp);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            prev = more ? H2_NP : 0;
            ++q; slot = slot == 2 ? 0 : slot + 1;
        }
        // ---------------- phase hi: X_hi x (W_hi, W_lo)
        {
            if (prev == H2_NP) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            const int sn = slot == 0 ? 2 : slot - 1;
            u32x4 xa[H2_MT], xb[H2_MT], wl[2 * H2_NT];
#pragma unroll
            for (int a = 0; a < H2_MT; ++a) xa[a] = xfrag16(slot, 0, a);
#pragma unroll
            for (int b = 0; b < 2 * H2_NT; ++b) wl[b] = wfrag16(slot, b);
#pragma unroll
            for (int a = 0; a < H2_MT; ++a) xb[a] = xfrag16(slot, 1, a);
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int kp = 0; kp < 2; ++kp)
#pragma unroll
                    for (int b = 0; b < 2 * H2_NT; ++b) {
                        const u32x4 wv = kp == 0 ? wk[b] : wl[b];
#pragma unroll
                        for (int a = 0; a < H2_MT; ++a)
                            acc16[4 * h + a][b] = mfma_frag16<true>(wv, h == 0 ? xa[a] : xb[a], acc16[4 * h + a][b]);
                        const int grp = (h * 2 + kp) * (2 * H2_NT) + b;
                        if (grp < H2_NP && more) {             // the hi pair of k-tile kt + 1
                            __builtin_amdgcn_sched_barrier(0);
                            piece(cn, 1, 0, sn, grp);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
            prev = more ? H2_NP : 0;
            ++q; slot = slot == 2 ? 0 : slot + 1;
        }
        if (kt + 2 < ktiles) cn = make_ctx(kt + 2);            // (one scalar load + the context per k-tile)
    }
    (void)nphase; (void)q;

    // ---- epilogue (conv_dma3.hip ep3, HS / 16x16x32): bias + activation in f32, HS split, LDS transpose (one slab per plane and wave), 16-byte runs ----
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int lane_e = lane;
    asm volatile("" : "+v"(lane_e));         // (nothing of the epilogue's per-lane values is hoisted above the k-loop)
    auto ep = [&](auto tag) {
        constexpr int ACT = decltype(tag)::value;
        constexpr int ROW = H2_ROW;
        unsigned char* slab = reinterpret_cast<unsigned char*>(lds) + wave * (2 * 32 * ROW);
        constexpr int SEGS = H2_NT * 4, PPP = 64 / SEGS;
        const int seg = lane_e % SEGS, prow = lane_e / SEGS;
        uint16_t* const out_hi = reinterpret_cast<uint16_t*>(p.out);
        const int m0 = bm0 + wm0, n0 = bn0 + wn0;
        f32x4 bias_v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) bias_v[i] = *reinterpret_cast<const f32x4*>(p.bias + n0 + i * 16 + 4 * (lane_e >> 4));
#pragma unroll
        for (int a = 0; a < H2_MT; ++a) {
#pragma unroll
            for (int b = 0; b < H2_NT; ++b)
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    // (b, r4) = (16-pixel block of the 32, 16-channel block): pixel 16 b + (lane & 15), channels 16 r4 + 4 (lane >> 4)
                    const int nl = r4 * 16 + 4 * (lane_e >> 4);
                    const int srow = b * 16 + (lane_e & 15);
                    f32x4 v = acc16[2 * a + b][r4];
                    uint2 h, l;
                    v = v * p.alpha + bias_v[r4];
                    v = act_split4<ACT>(v);
                    split4_hs(v, h, l, p.sat);
                    *reinterpret_cast<uint2*>(slab + srow * ROW + nl * 2) = h;
                    *reinterpret_cast<uint2*>(slab + 32 * ROW + srow * ROW + nl * 2) = l;
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
                    for (int pl = 0; pl < 2; ++pl)
                        *reinterpret_cast<u32x4*>(o + pl * p.out_plane) = *reinterpret_cast<const u32x4*>(slab + pl * 32 * ROW + pix * ROW + seg * 16);
                }
            }
            __builtin_amdgcn_wave_barrier();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    };
    if (p.act == ACT_RELU) ep(ActTag<ACT_RELU>{});
    else if (p.act == ACT_ELU) ep(ActTag<ACT_ELU>{});
    else ep(ActTag<ACT_NONE>{});
}

// the layers it takes: what conv_dma3's flat 1x1 HS form takes (the rule must not look at the call: counted on a full pass of the engine)
bool conv_gemm2_eligible(const ConvParams& p) {
    if (p.f16 != 4 || p.out_f16 != 3 || !p.flat || p.fold || p.rowgrp || !conv_dma3_eligible(p) || conv_dma3_mode(p) != 1) return false;
    if (p.sw & (SW_NO_GEMM2 | SW_MFMA32)) return false;       // (SEMDEPTH_DISABLE=mfma16: the 32x32x16 form exists on conv_dma3 only)
    return p.Cout % H2_BN == 0 && p.Kpad >= 64;
}

hipError_t launch_conv_gemm2(const ConvParams& p, hipStream_t s) {
    if (!conv_gemm2_eligible(p)) return hipErrorInvalidValue;
    const long M = (long)p.N * p.Hout * p.Wout;
    const int tilesM = (int)((M + H2_BM - 1) / H2_BM), tilesN = p.Cout / H2_BN;
    hipLaunchKernelGGL(conv_gemm2_hs_kernel, dim3((unsigned)(tilesM * tilesN)), dim3(64 * H2_NW), 0, s, p, (int)M, tilesM, tilesN);
    return hipGetLastError();
}

}  // namespace sd
