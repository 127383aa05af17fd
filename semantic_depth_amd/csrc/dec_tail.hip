// The full-resolution tail of the monodepth decoder under SD_PREC_BF16X3 as ONE kernel (oracle/nets.py:138-160, level 1):
//        u    = elu(conv3x3(up2(iconv2)) + b)                 dec/upconv1   32 -> 16 channels, full resolution
//        i    = elu(conv3x3(concat(u, up2(disp2))) + b)       dec/iconv1    18 -> 16
//        disp = 0.3 sigmoid(conv3x3(i) + b)[..., 0]           dec/disp1     16 -> 1 (only disp_left_est[0] is fetched, semantic_depth.py:675)
// As three launches the two 16-channel full-resolution tensors make four 3.2-GB trips through HBM at 6 bytes per element (7.7 ms per 32
// frames, at neither roof); here they never leave the CU: a workgroup owns 8 x 28 pixel tiles of the output, keeps u on the tile + 2 and
// i on the tile + 1 in LDS and reads only the half-resolution iconv2 / disp2 tensors (1.7 GB) from memory.
//
// Stage 1 (upconv1) is the upsample-FOLDED form (plan.hpp OpDesc::fold): a 3x3 conv on a x2 nearest-neighbour upsampled source is, per output
// parity (y & 1, x & 1), a 2x2 conv on the source itself whose weights are sums of the 3x3 taps that read the same source pixel -- 4/9 of the
// multiplications.  A wave owns one parity: its 16 folded weight fragments stay in registers for the whole launch, the source fragments
// (16 pixels x 32 channels, one 2x2 tap = one k-step of v_mfma_f32_16x16x32_bf16) come from the 8 x 18 source tile in LDS and a source row
// is read once for the two output rows that use it.
// Stage 2 (iconv1): the K axis of an output row is three row blocks of two k-steps, [u(x-1) u(x) | u(x+1) e(x) 0]: 16-byte slots of 8 u
// channels, and e = the six values disp2(y, x-1 .. x+1) x 2 channels, kept beside the u tile.  A wave marches down a 16-pixel strip: the
// fragments of a u row are read once and serve the three output rows around it.  The result stays in LDS as f32.
// Stage 3 (disp1) is the f32 head the engine runs on reconstructed values anyway: 144 FMAs per pixel on the f32 tile.
//
// The weight fragments of BOTH MFMA stages (120 registers) plus the working set of either do not fit the 256 registers of a wave at two
// waves per SIMD, and with one wave per SIMD nothing overlaps the epilogues' VALU work (bias, the engine's ELU, the exact three-way split:
// as many cycles as the MFMAs) with the matrix pipe: 4.05 ms.  So the eight waves are two GROUPS, one of each per SIMD: waves 0-3 hold the
// stage-1 weights and produce the u tile of tile k + 1 while waves 4-7 hold the stage-2 weights and turn the u tile of tile k into i and
// disp; u / e are double-buffered (which is why the tile is 8 rows) and the groups meet at two barriers per tile.
// Six MFMA products per product as everywhere in this engine (split_fmt.hpp); outside the image u, i and up2(disp2) are ZERO (the zero
// padding of the layer that reads them), not convolved values.
#include "kernels.hpp"
#include "split_fmt.hpp"

namespace sd {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int DT_TH = 8, DT_TW = 28;
constexpr int DT_AR = DT_TH / 2 + 4, DT_AC = 18, DT_APL = DT_AR * DT_AC * 4;   // source tile: rows, columns, 16-byte units per plane ([pixel][octet ^ swizzle])
constexpr int DT_UR = DT_TH + 4, DT_UC = 34, DT_UPL = DT_UR * DT_UC * 2;       // u tile: [pixel][octet ^ swizzle] per plane
constexpr int DT_ER = DT_UR / 2, DT_EPL = DT_ER * DT_UC;                        // e tile: one unit per (half-resolution row, full-resolution column) and plane
constexpr int DT_IR = DT_TH + 2, DT_IC = 32, DT_IPIX = 80;                      // i tile (f32): bytes per pixel (64 + 16 of padding: conflict-free 16-byte reads)
constexpr int DT_DPL = DT_AR * DT_AC;                                           // disp2 tile: one 32-bit word (2 channels) per pixel and plane
constexpr int DT_NG = 256;                                                      // threads of a group
constexpr int DT_NPIX = DT_TH * DT_TW;

// HS = false: SD_PREC_BF16X3 (three bf16 planes per operand, six products per product).  HS = true: SD_PREC_F16X2 -- two planes per operand (fp16 hi +
// scaled lo activations, fp16 hi + lo of w * 2^12: split_fmt.hpp "HS"), THREE fp16 products per product, the x_lo one against w_hi * 2^-11, which is
// formed ONCE per launch here (the weight fragments are resident: it sits in the register slot of the bf16 form's third plane), the accumulators
// times DecTailParams::alpha (stage 1) / alpha2 (stage 2); the same tiles, stages and wave groups.
template <bool HS>
__global__ __launch_bounds__(512, 1) void dec_tail1_kernel(const DecTailParams p) {
    constexpr int NPL = HS ? 2 : 3;                                                 // planes per tensor
    constexpr int DT_UE = NPL * DT_UPL + NPL * DT_EPL;                              // units of one (u, e) buffer
    constexpr int DT_NA = (NPL * DT_APL + DT_NG - 1) / DT_NG, DT_ND = (NPL * DT_DPL + DT_NG - 1) / DT_NG, DT_NE = (NPL * DT_EPL + DT_NG - 1) / DT_NG;
    static_assert((2 * DT_UE + NPL * DT_APL) * 16 + DT_IR * DT_IC * DT_IPIX + NPL * DT_DPL * 4 <= 160 * 1024, "tiles fit the LDS of a CU");
    static_assert(DT_IR % 2 == 0 && DT_NPIX % 4 == 0 && DT_NPIX / 4 <= 64, "work split of stages 2 and 3");
    __shared__ __attribute__((aligned(16))) u32x4 UE[2 * DT_UE];
    __shared__ __attribute__((aligned(16))) u32x4 A[NPL * DT_APL];
    __shared__ __attribute__((aligned(16))) unsigned char I[DT_IR * DT_IC * DT_IPIX];
    __shared__ unsigned Dt[NPL * DT_DPL];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const bool s1 = wave < 4;                            // group 1: stage 1 (and the tile loads); group 2: stages 2 and 3
    const int wv = wave & 3, tg = t & (DT_NG - 1);       // wave / thread inside the group
    const int H = p.H, W = p.W, Hs = H >> 1, Ws = W >> 1;
    const int ntx = (W + DT_TW - 1) / DT_TW, nty = H / DT_TH;
    const int ntiles = ntx * nty * p.N;
    const int lp = lane & 15, lg = lane >> 4;            // MFMA fragment: pixel / output channel (lane & 15), k group (lane >> 4)
    // SEMDEPTH_X3_DIAG (decomposition runs; 0 in production): 1 / 2 / 3 = without stage 1 / 2 / 3
    const int diag = SD_DIAG_BITS(p.sw);

    struct Tile { int img, y0, x0; };
    auto tile_of = [&](int it) {
        Tile r;
        const int tx = it % ntx; it /= ntx;
        r.x0 = tx * DT_TW < W - DT_TW ? tx * DT_TW : W - DT_TW;        // (the last column of tiles is shifted inwards: it recomputes identical values)
        r.y0 = (it % nty) * DT_TH; r.img = it / nty;
        return r;
    };
    // six MFMA products of one (weight fragment, source fragment) pair, the smaller terms first
    // (HS: w = {hi, lo, hi * 2^-11}, x = {hi, scaled lo}: x_hi * w_lo, x_lo * w_hs, x_hi * w_hi; the xl argument is not read)
    auto mac6 = [](const u32x4 (&w)[3], const u32x4& xh, const u32x4& xm, const u32x4& xl, f32x4 acc) {
        if constexpr (HS) {
            acc = mfma_frag16<true>(w[1], xh, acc);
            acc = mfma_frag16<true>(w[2], xm, acc);
            acc = mfma_frag16<true>(w[0], xh, acc);
            return acc;
        }
        acc = mfma_frag16<false>(w[2], xh, acc);
        acc = mfma_frag16<false>(w[1], xm, acc);
        acc = mfma_frag16<false>(w[0], xl, acc);
        acc = mfma_frag16<false>(w[1], xh, acc);
        acc = mfma_frag16<false>(w[0], xm, acc);
        acc = mfma_frag16<false>(w[0], xh, acc);
        return acc;
    };
    // disp1 weights of channel 0 ([tap][16]) spread over the lanes of three registers: stage 3 broadcasts them with v_readlane
    const float wdr0 = p.wd[lane], wdr1 = p.wd[64 + lane], wdr2 = p.wd[lane < 16 ? 128 + lane : 0];
    auto wd_at = [&](int i) {           // i: compile-time constant after unrolling
        return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, i < 64 ? wdr0 : i < 128 ? wdr1 : wdr2), i & 63));
    };
    // stage 3 for pixel pi of the tile whose i tile is in LDS.  The reads of a tap of the 3 x 3 x 16 window go out two taps ahead of its FMAs (one
    // wave per SIMD runs this phase: nothing else hides an LDS round trip), the 16 weights of a tap are broadcast into 16 scalar registers before its FMAs (a
    // v_readlane directly in front of the FMA that uses it costs two wait states each) and two accumulators halve the dependent chain
    auto head = [&](const Tile& tl, int pi) {
        const int ry = pi / DT_TW, cx = pi - ry * DT_TW;
        f32x4 v[3][4];                                        // the 16 channels of a tap, two taps ahead of the FMAs
        auto tapload = [&](int slot, int tap) {
            const unsigned char* ip = I + ((ry + tap / 3) * DT_IC + cx + tap % 3) * DT_IPIX;
#pragma unroll
            for (int c4 = 0; c4 < 4; ++c4) v[slot][c4] = *reinterpret_cast<const f32x4*>(ip + 16 * c4);
        };
        tapload(0, 0); tapload(1, 1);
        float acc0 = p.bd[0], acc1 = 0.f;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            if (tap + 2 < 9) tapload((tap + 2) % 3, tap + 2);
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {                     // eight weights at a time (scalar registers are scarce too)
                float w[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) w[e] = wd_at(tap * 16 + 8 * hf + e);

#pragma unroll
                for (int e = 0; e < 8; e += 2) {
                    acc0 += v[tap % 3][2 * hf + (e >> 2)][e & 3] * w[e];
                    acc1 += v[tap % 3][2 * hf + ((e + 1) >> 2)][(e + 1) & 3] * w[e + 1];
                }

            }
        }
        const float acc = acc0 + acc1;
        p.out[((size_t)tl.img * H + tl.y0 + ry) * W + tl.x0 + cx] = 0.3f * (1.0f / (1.0f + expf(-acc)));
    };

    for (int i = t; i < 2 * DT_UE; i += 512) UE[i] = u32x4{0u, 0u, 0u, 0u};      // (columns 32, 33 of u are never computed: finite values)
    const int first = blockIdx.x, stride = gridDim.x;
    const int K = first < ntiles ? (ntiles - first + stride - 1) / stride : 0;      // tiles of this workgroup

    if (s1) {
        // =====================================================================================================================
        // group 1: tile loads, e, stage 1
        // =====================================================================================================================
        int ga[DT_NA];          // arow | acol << 8 | octet << 16 | plane << 20 | valid << 24
#pragma unroll
        for (int k = 0; k < DT_NA; ++k) {
            const int u = tg + DT_NG * k;
            const int pl = u / DT_APL, rem = u - pl * DT_APL, pix = rem >> 2, slot = rem & 3;
            const int arow = pix / DT_AC, acol = pix - arow * DT_AC;
            ga[k] = arow | (acol << 8) | ((slot ^ ((acol >> 1) & 3)) << 16) | (pl << 20) | ((u < NPL * DT_APL ? 1 : 0) << 24);
        }
        int gd[DT_ND];          // arow | acol << 8 | plane << 20 | valid << 24
#pragma unroll
        for (int k = 0; k < DT_ND; ++k) {
            const int d = tg + DT_NG * k;
            const int pl = d / DT_DPL, pix = d - pl * DT_DPL;
            const int arow = pix / DT_AC, acol = pix - arow * DT_AC;
            gd[k] = arow | (acol << 8) | (pl << 20) | ((d < NPL * DT_DPL ? 1 : 0) << 24);
        }
        int ge[DT_NE];          // er | uc << 8 | plane << 16 | valid << 24
        int gew[DT_NE];         // word of (drow, dcol(-1)) in Dt | step to dcol(0) << 16 | step from dcol(0) to dcol(+1) << 17
#pragma unroll
        for (int k = 0; k < DT_NE; ++k) {
            const int e = tg + DT_NG * k;
            const int pl = e / DT_EPL, pix = e - pl * DT_EPL;
            const int er = pix / DT_UC, uc = pix - er * DT_UC;
            auto dcol = [&](int dx) { int c = ((uc + dx) >> 1) + 1; return c < 0 ? 0 : c > DT_AC - 1 ? DT_AC - 1 : c; };
            ge[k] = er | (uc << 8) | (pl << 16) | ((e < NPL * DT_EPL ? 1 : 0) << 24);
            gew[k] = (pl * DT_DPL + (er + 1) * DT_AC + dcol(-1)) | ((dcol(0) - dcol(-1)) << 16) | ((dcol(1) - dcol(0)) << 17);
        }
        u32x4 pa[DT_NA];
        unsigned pd[DT_ND];
        auto prefetch = [&](const Tile& tl) {
            const int sy0 = (tl.y0 >> 1) - 2, sx0 = (tl.x0 >> 1) - 2;
            const uint16_t* const a = reinterpret_cast<const uint16_t*>(p.a);
            const uint16_t* const d2 = reinterpret_cast<const uint16_t*>(p.d2);
#pragma unroll
            for (int k = 0; k < DT_NA; ++k) {
                const int sy = sy0 + (ga[k] & 0xff), sx = sx0 + ((ga[k] >> 8) & 0xff);
                const bool ok = ((ga[k] >> 24) & 1) && (unsigned)sy < (unsigned)Hs && (unsigned)sx < (unsigned)Ws;
                const uint16_t* src = a + (size_t)((ga[k] >> 20) & 3) * p.a_plane + ((size_t)(tl.img * Hs + sy) * Ws + sx) * 32 + ((ga[k] >> 16) & 3) * 8;
                pa[k] = ok ? *reinterpret_cast<const u32x4*>(src) : u32x4{0u, 0u, 0u, 0u};
            }
#pragma unroll
            for (int k = 0; k < DT_ND; ++k) {
                const int sy = sy0 + (gd[k] & 0xff), sx = sx0 + ((gd[k] >> 8) & 0xff);
                const bool ok = ((gd[k] >> 24) & 1) && (unsigned)sy < (unsigned)Hs && (unsigned)sx < (unsigned)Ws;
                const uint16_t* src = d2 + (size_t)((gd[k] >> 20) & 3) * p.d_plane + ((size_t)(tl.img * Hs + sy) * Ws + sx) * 8;
                pd[k] = ok ? *reinterpret_cast<const unsigned*>(src) : 0u;
            }
        };
        auto stash = [&]() {           // the prefetched source pixels into LDS
#pragma unroll
            for (int k = 0; k < DT_NA; ++k)
                if ((ga[k] >> 24) & 1) A[tg + DT_NG * k] = pa[k];
#pragma unroll
            for (int k = 0; k < DT_ND; ++k)
                if ((gd[k] >> 24) & 1) Dt[tg + DT_NG * k] = pd[k];
        };
        const int q = wv, py = q >> 1, px = q & 1;                             // this wave's output parity
        u32x4 w1[4][3];                                                        // [2x2 tap][plane], resident for the whole launch
#pragma unroll
        for (int tp = 0; tp < 4; ++tp) {
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl) w1[tp][pl] = p.w1[((q * 4 + tp) * 3 + pl) * 64 + lane];
            if constexpr (HS) w1[tp][2] = hs_wscaled(w1[tp][0]);
        }
        const f32x4 bias1 = *reinterpret_cast<const f32x4*>(p.b1 + 4 * lg);

        Tile nxt = tile_of(K > 0 ? first : 0);                                // (one tile decode per iteration: the divisions are VALU work)
        if (K > 0) { prefetch(nxt); stash(); }
        __syncthreads();
        for (int k = 0; k <= K; ++k) {
            // ---------------- phase A: e and stage 1 of tile k (the other group: stage 2 of tile k - 1)
            if (k < K) {
                const Tile cur = nxt;
                const int y0 = cur.y0, x0 = cur.x0;
                u32x4* const Ub = UE + (k & 1) * DT_UE;
                if (k + 1 < K) { nxt = tile_of(first + (k + 1) * stride); prefetch(nxt); }
                // e = disp2(y, x - 1 .. x + 1) of every column of the u tile and every row pair (zero outside the image)
#pragma unroll
                for (int j = 0; j < DT_NE; ++j) {
                    const int er = ge[j] & 0xff, uc = (ge[j] >> 8) & 0xff, pl = (ge[j] >> 16) & 3;
                    const int ys = (y0 >> 1) - 1 + er, x = x0 - 2 + uc;
                    const int wi = gew[j] & 0xffff, i0 = wi + ((gew[j] >> 16) & 1), i1 = i0 + ((gew[j] >> 17) & 1);
                    const unsigned a = Dt[wi], b0 = Dt[i0], c = Dt[i1];            // (unconditional loads of in-tile words, then selects)
                    const bool oky = (unsigned)ys < (unsigned)Hs;
                    u32x4 ev;
                    ev[0] = oky && (unsigned)(x - 1) < (unsigned)W ? a : 0u;
                    ev[1] = oky && (unsigned)x < (unsigned)W ? b0 : 0u;
                    ev[2] = oky && (unsigned)(x + 1) < (unsigned)W ? c : 0u;
                    ev[3] = 0u;
                    if ((ge[j] >> 24) & 1) Ub[NPL * DT_UPL + pl * DT_EPL + er * DT_UC + uc] = ev;
                }
                // stage 1: u rows 2 n + py, n = 0 .. UR / 2 - 1, columns 2 m + px (m = lane & 15)
                if (diag != 1) {
                    // source fragment of tile row r, column shift b: pixel column m + px + b, octet lane >> 4
                    auto frag = [&](int r, int b, int pl) {
                        const int acol = lp + px + b;
                        return A[pl * DT_APL + (r * DT_AC + acol) * 4 + (lg ^ ((acol >> 1) & 3))];
                    };
                    u32x4 F[2][2][3];                                             // [row slot][b][plane]
#pragma unroll
                    for (int b = 0; b < 2; ++b)
#pragma unroll
                        for (int pl = 0; pl < NPL; ++pl) F[0][b][pl] = frag(py, b, pl);
#pragma unroll
                    for (int n = 0; n < DT_UR / 2; ++n) {
                        const int s0 = n & 1, s1_ = s0 ^ 1;                       // slots of source rows n + py (a = 0) and n + py + 1 (a = 1)
#pragma unroll
                        for (int b = 0; b < 2; ++b)
#pragma unroll
                            for (int pl = 0; pl < NPL; ++pl) F[s1_][b][pl] = frag(n + py + 1, b, pl);
                        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int b = 0; b < 2; ++b) {
                            acc0 = mac6(w1[b], F[s0][b][0], F[s0][b][1], F[s0][b][HS ? 1 : 2], acc0);
                            acc1 = mac6(w1[2 + b], F[s1_][b][0], F[s1_][b][1], F[s1_][b][HS ? 1 : 2], acc1);
                        }
                        f32x4 v = HS ? (acc0 + acc1) * p.alpha + bias1 : acc0 + acc1 + bias1;
                        v = act_split4<ACT_ELU>(v);      // (act_x3<ELU> and act_split<ELU> are the same function)
                        const int ru = 2 * n + py, uc = 2 * lp + px;
                        const bool in = (unsigned)(y0 - 2 + ru) < (unsigned)H && (unsigned)(x0 - 2 + uc) < (unsigned)W;
                        if (!in) v = f32x4{0.f, 0.f, 0.f, 0.f};
                        uint2 hh, mm, ll;
                        uint2* const up = reinterpret_cast<uint2*>(Ub) + ((ru * DT_UC + uc) * 2 + ((lg >> 1) ^ ((uc >> 2) & 1))) * 2 + (lg & 1);
                        if constexpr (HS) {
                            split4_hs(v, hh, mm, (sat_ptr_t) nullptr);
                            up[0] = hh; up[2 * DT_UPL] = mm;
                        } else {
                            split4_x3(v, hh, mm, ll);
                            up[0] = hh; up[2 * DT_UPL] = mm; up[4 * DT_UPL] = ll;
                        }
                    }
                }
            }
            __syncthreads();
            // ---------------- phase B: the source pixels of tile k + 1 into LDS (the other group: stage 3 of tile k - 1)
            if (k + 1 < K) stash();
            __syncthreads();
        }
    } else {
        // =====================================================================================================================
        // group 2: stages 2 and 3
        // =====================================================================================================================
        u32x4 w2[6][3];                                                        // [row block dy, half][plane], resident for the whole launch
#pragma unroll
        for (int ks = 0; ks < 6; ++ks) {
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl) w2[ks][pl] = p.w2[(ks * 3 + pl) * 64 + lane];
            if constexpr (HS) w2[ks][2] = hs_wscaled(w2[ks][0]);
        }
        const f32x4 bias2 = *reinterpret_cast<const f32x4*>(p.b2 + 4 * lg);
        const int strip = wv & 1, ri0 = (DT_IR / 2) * (wv >> 1);
        const int c = 16 * strip + lp;
        // 16-byte slot of this lane in the two k-steps of a row block: A = [u(x-1) | u(x)], B = [u(x+1) | e(x), (zero weights)]
        const int uA = c + 1 + ((lg >> 1) - 1), uB = c + 2;
        const int offA = uA * 2 + ((lg & 1) ^ ((uA >> 2) & 1)), offB = uB * 2 + ((lg & 1) ^ ((uB >> 2) & 1));
        __syncthreads();
        for (int k = 0; k <= K; ++k) {
            const Tile cur = tile_of(first + (k >= 1 ? k - 1 : 0) * stride);
            // ---------------- phase A: stage 2 of tile k - 1: i rows ri0 .. of the 16-pixel strip (the other group: stage 1 of tile k)
            if (k >= 1 && diag != 2) {
                const u32x4* const Ub = UE + ((k - 1) & 1) * DT_UE;
                u32x4 G[3][2][3];                                             // [u row slot][half][plane]
                auto gload = [&](int slot, int ru) {
#pragma unroll
                    for (int pl = 0; pl < NPL; ++pl) {
                        G[slot][0][pl] = Ub[pl * DT_UPL + ru * (DT_UC * 2) + offA];
                        G[slot][1][pl] = Ub[lg < 2 ? pl * DT_UPL + ru * (DT_UC * 2) + offB : NPL * DT_UPL + pl * DT_EPL + (ru >> 1) * DT_UC + c + 1];
                    }
                };
                gload(0, ri0); gload(1, ri0 + 1);
#pragma unroll
                for (int it = 0; it < DT_IR / 2; ++it) {
                    const int ri = ri0 + it;
                    gload((it + 2) % 3, ri + 2);
                    f32x4 acc[3];
#pragma unroll
                    for (int dy = 0; dy < 3; ++dy) {
                        acc[dy] = f32x4{0.f, 0.f, 0.f, 0.f};
                        const int sl = (it + dy) % 3;
                        acc[dy] = mac6(w2[2 * dy], G[sl][0][0], G[sl][0][1], G[sl][0][HS ? 1 : 2], acc[dy]);
                        acc[dy] = mac6(w2[2 * dy + 1], G[sl][1][0], G[sl][1][1], G[sl][1][HS ? 1 : 2], acc[dy]);
                    }
                    f32x4 v = HS ? ((acc[0] + acc[1]) + acc[2]) * p.alpha2 + bias2 : (acc[0] + acc[1]) + acc[2] + bias2;
                    v = act_split4<ACT_ELU>(v);      // (act_x3<ELU> and act_split<ELU> are the same function)
                    const bool in = (unsigned)(cur.y0 - 1 + ri) < (unsigned)H && (unsigned)(cur.x0 - 1 + c) < (unsigned)W;
                    if (!in) v = f32x4{0.f, 0.f, 0.f, 0.f};
                    *reinterpret_cast<f32x4*>(I + (ri * DT_IC + c) * DT_IPIX + 16 * lg) = v;
                }
            }
            __syncthreads();
            // ---------------- phase B: stage 3 of tile k - 1, a quarter of the tile's pixels per wave: VALU time is per wave instruction, so
            //                  one pass on each SIMD (56 of 64 lanes) instead of two passes on two of them
            if (k >= 1 && diag != 3 && lane < DT_NPIX / 4) head(cur, wv * (DT_NPIX / 4) + lane);
            __syncthreads();
        }
    }
}

bool dec_tail1_eligible(int H, int W) { return H % DT_TH == 0 && W % 2 == 0 && W >= DT_TW + 2; }

hipError_t launch_dec_tail1(const DecTailParams& p, hipStream_t s) {
    if (!dec_tail1_eligible(p.H, p.W)) return hipErrorInvalidValue;
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return hipErrorInvalidDevice;
        cus = prop.multiProcessorCount;
    }
    const int tiles = ((p.W + DT_TW - 1) / DT_TW) * (p.H / DT_TH) * p.N;
    const int wgs = cus - p.reserve_cus > 0 ? cus - p.reserve_cus : 1;
    if (p.hs) hipLaunchKernelGGL(dec_tail1_kernel<true>, dim3((unsigned)(tiles < wgs ? tiles : wgs)), dim3(512), 0, s, p);
    else hipLaunchKernelGGL(dec_tail1_kernel<false>, dim3((unsigned)(tiles < wgs ? tiles : wgs)), dim3(512), 0, s, p);
    return hipGetLastError();
}

}  // namespace sd
