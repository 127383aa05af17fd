// The full-resolution tail of the monodepth decoder under SD_PREC_BF16X3 as ONE kernel (oracle/nets.py:138-160, level 1):
//        u    = elu(conv3x3(up2(iconv2)) + b)                 dec/upconv1   32 -> 16 channels, full resolution
//        i    = elu(conv3x3(concat(u, up2(disp2))) + b)       dec/iconv1    18 -> 16
//        disp = 0.3 sigmoid(conv3x3(i) + b)[..., 0]           dec/disp1     16 -> 1 (only disp_left_est[0] is fetched, semantic_depth.py:675)
// As three launches the two 16-channel full-resolution tensors make four 3.2-GB trips through HBM at 6 bytes per element (7.7 ms per 32
// frames, at neither roof); here they never leave the CU: a workgroup owns a 16 x 28 pixel tile of the output, keeps u on the tile + 2 and
// i on the tile + 1 in LDS and reads only the half-resolution iconv2 / disp2 tensors (1.7 GB) from memory.
//
// Stage 1 (upconv1) is the upsample-FOLDED form (plan.hpp OpDesc::fold): a 3x3 conv on a x2 nearest-neighbour upsampled source is, per output
// parity (y & 1, x & 1), a 2x2 conv on the source itself whose weights are sums of the 3x3 taps that read the same source pixel -- 4/9 of the
// multiplications.  A wave owns one parity: its 16 folded weight fragments stay in registers for the whole launch, the source fragments
// (16 pixels x 32 channels, one 2x2 tap = one k-step of v_mfma_f32_16x16x32_bf16) come from the 12 x 18 source tile in LDS and a source row
// is read once for the two output rows that use it.
// Stage 2 (iconv1): the K axis of an output row is three row blocks of two k-steps, [u(x-1) u(x) | u(x+1) e(x) 0]: 16-byte slots of 8 u
// channels, and e = the six values disp2(y, x-1 .. x+1) x 2 channels, written beside u as a third "octet" of every pixel.  A wave marches
// down a 16-pixel strip: the fragments of a u row are read once and serve the three output rows around it.  The result stays in LDS as f32.
// Stage 3 (disp1) is the f32 head the engine runs on reconstructed values anyway: 144 FMAs per pixel on the f32 tile.
// Six MFMA products per product as everywhere in this engine (split_fmt.hpp); bias + the engine's ELU + the exact three-way split in the
// epilogues; outside the image u, i and up2(disp2) are ZERO (the zero padding of the layer that reads them), not convolved values.
#include "kernels.hpp"
#include "split_fmt.hpp"

namespace sd {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int DT_TH = 16, DT_TW = 28;
constexpr int DT_AR = 12, DT_AC = 18, DT_APL = DT_AR * DT_AC * 4;        // source tile: rows, columns, 16-byte units per plane ([pixel][octet ^ swizzle])
constexpr int DT_UR = 20, DT_UC = 34, DT_UPL = DT_UR * DT_UC * 3;        // u tile: [pixel][u octet 0 | u octet 1 | e] per plane
constexpr int DT_IR = 18, DT_IC = 32, DT_IPIX = 80;                      // i tile (f32): bytes per pixel (64 + 16 of padding: conflict-free 16-byte reads)
constexpr int DT_R0 = (DT_IR * DT_IC * DT_IPIX > 3 * DT_APL * 16 ? DT_IR * DT_IC * DT_IPIX : 3 * DT_APL * 16) / 16;     // units: the source tile, later the i tile
constexpr int DT_DPL = DT_AR * DT_AC;                                    // disp2 tile: one 32-bit word (2 channels) per pixel and plane
constexpr int DT_NT = 256;                                               // threads: ONE wave per SIMD, so that a wave owns 512 registers (both weight sets stay resident)
constexpr int DT_NA = (3 * DT_APL + DT_NT - 1) / DT_NT, DT_ND = (3 * DT_DPL + DT_NT - 1) / DT_NT, DT_NE = (3 * DT_UR * DT_UC + DT_NT - 1) / DT_NT;

__global__ __launch_bounds__(DT_NT, 1) void dec_tail1_x3_kernel(const DecTailParams p) {
    static_assert((DT_R0 + 3 * DT_UPL) * 16 + 3 * DT_DPL * 4 <= 160 * 1024, "tiles fit the LDS of a CU");
    __shared__ __attribute__((aligned(16))) u32x4 R0[DT_R0];
    __shared__ __attribute__((aligned(16))) u32x4 U[3 * DT_UPL];
    __shared__ unsigned Dt[3 * DT_DPL];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int H = p.H, W = p.W, Hs = H >> 1, Ws = W >> 1;
    const int ntx = (W + DT_TW - 1) / DT_TW, nty = H / DT_TH;
    const int ntiles = ntx * nty * p.N;
    const int lp = lane & 15, lg = lane >> 4;           // MFMA fragment: pixel / output channel (lane & 15), k group (lane >> 4)

    // ---- per-thread geometry of the tile loads (the same for every tile)
    int ga[DT_NA];          // arow | acol << 8 | octet << 16 | plane << 20 | valid << 24
#pragma unroll
    for (int k = 0; k < DT_NA; ++k) {
        const int u = t + DT_NT * k;
        const int pl = u / DT_APL, rem = u - pl * DT_APL, pix = rem >> 2, slot = rem & 3;
        const int arow = pix / DT_AC, acol = pix - arow * DT_AC;
        ga[k] = arow | (acol << 8) | ((slot ^ ((acol >> 1) & 3)) << 16) | (pl << 20) | ((u < 3 * DT_APL ? 1 : 0) << 24);
    }
    int gd[DT_ND];          // arow | acol << 8 | plane << 20 | valid << 24
#pragma unroll
    for (int k = 0; k < DT_ND; ++k) {
        const int d = t + DT_NT * k;
        const int pl = d / DT_DPL, pix = d - pl * DT_DPL;
        const int arow = pix / DT_AC, acol = pix - arow * DT_AC;
        gd[k] = arow | (acol << 8) | (pl << 20) | ((d < 3 * DT_DPL ? 1 : 0) << 24);
    }
    // e items of this thread: destination unit in U | disp2-tile word of dx = -1 .. the words of dx = 0, +1 are at + dc0, + dc1 (0 or 1)
    int ge[DT_NE];          // ru | uc << 8 | valid << 24
    int gew[DT_NE];         // word index of (drow, dcol(-1)) in Dt | step to dcol(0) << 16 | step from dcol(0) to dcol(+1) << 17
#pragma unroll
    for (int k = 0; k < DT_NE; ++k) {
        const int e = t + DT_NT * k;
        const int pl = e / (DT_UR * DT_UC), pix = e - pl * (DT_UR * DT_UC);
        const int ru = pix / DT_UC, uc = pix - ru * DT_UC;
        auto dcol = [&](int dx) { int c = ((uc + dx) >> 1) + 1; return c < 0 ? 0 : c > DT_AC - 1 ? DT_AC - 1 : c; };
        ge[k] = ru | (uc << 8) | (pl << 16) | ((e < 3 * DT_UR * DT_UC ? 1 : 0) << 24);
        gew[k] = (pl * DT_DPL + ((ru >> 1) + 1) * DT_AC + dcol(-1)) | ((dcol(0) - dcol(-1)) << 16) | ((dcol(1) - dcol(0)) << 17);
    }
    struct Tile { int img, y0, x0; };
    auto tile_of = [&](int it) {
        Tile r;
        const int tx = it % ntx; it /= ntx;
        r.x0 = tx * DT_TW < W - DT_TW ? tx * DT_TW : W - DT_TW;        // (the last column of tiles is shifted inwards: it recomputes identical values)
        r.y0 = (it % nty) * DT_TH; r.img = it / nty;
        return r;
    };
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

    // ---- weights: resident in registers for the whole launch
    const int q = wave, py = q >> 1, px = q & 1;                         // stage 1: this wave's output parity
    u32x4 w1[4][3];                                                        // [2x2 tap][plane]
#pragma unroll
    for (int tp = 0; tp < 4; ++tp)
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) w1[tp][pl] = p.w1[((q * 4 + tp) * 3 + pl) * 64 + lane];
    u32x4 w2[6][3];                                                        // [row block dy, half][plane]
#pragma unroll
    for (int ks = 0; ks < 6; ++ks)
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) w2[ks][pl] = p.w2[(ks * 3 + pl) * 64 + lane];
    const f32x4 bias1 = *reinterpret_cast<const f32x4*>(p.b1 + 4 * lg), bias2 = *reinterpret_cast<const f32x4*>(p.b2 + 4 * lg);

    // six MFMA products of one (weight fragment, source fragment) pair, the smaller terms first
    auto mac6 = [](const u32x4 (&w)[3], const u32x4& xh, const u32x4& xm, const u32x4& xl, f32x4 acc) {
        acc = mfma_frag16<false>(w[2], xh, acc);
        acc = mfma_frag16<false>(w[1], xm, acc);
        acc = mfma_frag16<false>(w[0], xl, acc);
        acc = mfma_frag16<false>(w[1], xh, acc);
        acc = mfma_frag16<false>(w[0], xm, acc);
        acc = mfma_frag16<false>(w[0], xh, acc);
        return acc;
    };

    for (int i = t; i < 3 * DT_UPL; i += DT_NT) U[i] = u32x4{0u, 0u, 0u, 0u};
    // disp1 weights of channel 0 ([tap][16]) spread over the lanes of three registers: stage 3 broadcasts them with v_readlane (a read per
    // weight from LDS instead costs more LDS time than the whole i tile)
    const float wdr0 = p.wd[lane], wdr1 = p.wd[64 + lane], wdr2 = p.wd[lane < 16 ? 128 + lane : 0];
    auto wd_at = [&](int i) {           // i: compile-time constant after unrolling
        return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, i < 64 ? wdr0 : i < 128 ? wdr1 : wdr2), i & 63));
    };      // (columns 32, 33 of u are never computed: finite values)
    // SEMDEPTH_X3_DIAG (decomposition runs; 0 in production): 1 / 2 / 3 = without stage 1 / 2 / 3
    const int diag = ((p.sw & SW_X3_DIAG_NOSTORE) ? 1 : 0) | ((p.sw & SW_X3_DIAG_NOMFMA) ? 2 : 0);
    int tile = blockIdx.x;
    if (tile >= ntiles) return;
    Tile cur = tile_of(tile);
    prefetch(cur);
    __syncthreads();
    for (; tile < ntiles; tile += gridDim.x) {
        const int y0 = cur.y0, x0 = cur.x0;
        // ---- the tile's source pixels into LDS
#pragma unroll
        for (int k = 0; k < DT_NA; ++k)
            if ((ga[k] >> 24) & 1) R0[t + DT_NT * k] = pa[k];
#pragma unroll
        for (int k = 0; k < DT_ND; ++k)
            if ((gd[k] >> 24) & 1) Dt[t + DT_NT * k] = pd[k];
        __syncthreads();
        // ---- e = disp2(y, x - 1 .. x + 1) of every pixel of the u tile (zero outside the image)
#pragma unroll
        for (int k = 0; k < DT_NE; ++k) {
            const int ru = ge[k] & 0xff, uc = (ge[k] >> 8) & 0xff, pl = (ge[k] >> 16) & 3;
            const int y = y0 - 2 + ru, x = x0 - 2 + uc;
            const int wi = gew[k] & 0xffff, i0 = wi + ((gew[k] >> 16) & 1), i1 = i0 + ((gew[k] >> 17) & 1);
            const unsigned a = Dt[wi], b0 = Dt[i0], c = Dt[i1];            // (unconditional loads of in-tile words, then selects)
            const bool oky = (unsigned)y < (unsigned)H;
            u32x4 ev;
            ev[0] = oky && (unsigned)(x - 1) < (unsigned)W ? a : 0u;
            ev[1] = oky && (unsigned)x < (unsigned)W ? b0 : 0u;
            ev[2] = oky && (unsigned)(x + 1) < (unsigned)W ? c : 0u;
            ev[3] = 0u;
            if ((ge[k] >> 24) & 1) U[pl * DT_UPL + (ru * DT_UC + uc) * 3 + 2] = ev;
        }
        // ---- stage 1: u rows 2 n + py, n = 0 .. 9, columns 2 m + px (m = lane & 15) of the u tile
        if (diag != 1) {
            constexpr int n0 = 0;
            // source fragment of tile row r, column shift b: pixel column m + px + b, octet lane >> 4
            auto frag = [&](int r, int b, int pl) {
                const int acol = lp + px + b;
                return R0[pl * DT_APL + (r * DT_AC + acol) * 4 + (lg ^ ((acol >> 1) & 3))];
            };
            u32x4 F[2][2][3];                                             // [row slot][b][plane]
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) F[0][b][pl] = frag(n0 + py, b, pl);
#pragma unroll
            for (int it = 0; it < 10; ++it) {
                const int n = n0 + it;
                const int s0 = it & 1, s1 = s0 ^ 1;                       // slots of source rows n + py (a = 0) and n + py + 1 (a = 1)
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) F[s1][b][pl] = frag(n + py + 1, b, pl);
                f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    acc0 = mac6(w1[b], F[s0][b][0], F[s0][b][1], F[s0][b][2], acc0);
                    acc1 = mac6(w1[2 + b], F[s1][b][0], F[s1][b][1], F[s1][b][2], acc1);
                }
                f32x4 v = acc0 + acc1 + bias1;
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = act_f32<ACT_ELU>(v[r]);
                const int ru = 2 * n + py, uc = 2 * lp + px;
                const bool in = (unsigned)(y0 - 2 + ru) < (unsigned)H && (unsigned)(x0 - 2 + uc) < (unsigned)W;
                if (!in) v = f32x4{0.f, 0.f, 0.f, 0.f};
                uint2 hh, mm, ll;
                split4_x3(v, hh, mm, ll);
                uint2* const up = reinterpret_cast<uint2*>(U) + ((ru * DT_UC + uc) * 3 + (lg >> 1)) * 2 + (lg & 1);
                up[0] = hh; up[2 * DT_UPL] = mm; up[4 * DT_UPL] = ll;
            }
        }
        __syncthreads();
        // ---- the next tile's source pixels travel while stages 2 and 3 run
        const int nxt = tile + (int)gridDim.x;
        const Tile ntl = nxt < ntiles ? tile_of(nxt) : cur;
        if (nxt < ntiles) prefetch(ntl);
        // ---- stage 2: i rows ri0 .. ri0 + 8 of the 16-pixel strip `strip`
        if (diag != 2) {
            const int strip = wave & 1;
            const int ri0 = 9 * (wave >> 1);
            const int c = 16 * strip + lp;
            // 16-byte slot of this lane in the two k-steps of a row block: A = [u(x-1) | u(x)], B = [u(x+1) | e(x), (zero weights)]
            const int offA = (c + 1 + ((lg >> 1) - 1)) * 3 + (lg & 1);
            const int offB = lg < 2 ? (c + 2) * 3 + lg : (c + 1) * 3 + 2;
            u32x4 G[3][2][3];                                             // [u row slot][half][plane]
            auto gload = [&](int slot, int ru) {
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) {
                    G[slot][0][pl] = U[pl * DT_UPL + ru * (DT_UC * 3) + offA];
                    G[slot][1][pl] = U[pl * DT_UPL + ru * (DT_UC * 3) + offB];
                }
            };
            gload(0, ri0); gload(1, ri0 + 1);
#pragma unroll
            for (int it = 0; it < 9; ++it) {
                {
                    const int ri = ri0 + it;
                    gload((it + 2) % 3, ri + 2);
                    f32x4 acc[3];
#pragma unroll
                    for (int dy = 0; dy < 3; ++dy) {
                        acc[dy] = f32x4{0.f, 0.f, 0.f, 0.f};
                        const int sl = (it + dy) % 3;
                        acc[dy] = mac6(w2[2 * dy], G[sl][0][0], G[sl][0][1], G[sl][0][2], acc[dy]);
                        acc[dy] = mac6(w2[2 * dy + 1], G[sl][1][0], G[sl][1][1], G[sl][1][2], acc[dy]);
                    }
                    f32x4 v = (acc[0] + acc[1]) + acc[2] + bias2;
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = act_f32<ACT_ELU>(v[r]);
                    const bool in = (unsigned)(y0 - 1 + ri) < (unsigned)H && (unsigned)(x0 - 1 + c) < (unsigned)W;
                    if (!in) v = f32x4{0.f, 0.f, 0.f, 0.f};
                    *reinterpret_cast<f32x4*>(reinterpret_cast<unsigned char*>(R0) + (ri * DT_IC + c) * DT_IPIX + 16 * lg) = v;
                }
            }
        }
        __syncthreads();
        // ---- stage 3: disp1 channel 0 on the f32 tile.  A thread owns the vertical pixel pair (2 rp, 2 rp + 1) x cx: the four i rows around
        //      it are read once for both (consecutive lanes = consecutive pixels of a row: conflict-free 16-byte reads)
        if (t < (DT_TH / 2) * DT_TW && diag != 3) {
            const int rp = t / DT_TW, cx = t - rp * DT_TW;
            float acc0 = p.bd[0], acc1 = acc0;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    const unsigned char* ip = reinterpret_cast<const unsigned char*>(R0) + ((2 * rp + r) * DT_IC + cx + dx) * DT_IPIX;
#pragma unroll
                    for (int c4 = 0; c4 < 4; ++c4) {
                        const f32x4 v = *reinterpret_cast<const f32x4*>(ip + 16 * c4);
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            if (r < 3) acc0 += v[e] * wd_at((r * 3 + dx) * 16 + 4 * c4 + e);
                            if (r > 0) acc1 += v[e] * wd_at(((r - 1) * 3 + dx) * 16 + 4 * c4 + e);
                        }
                    }
                }
            }
            float* o = p.out + ((size_t)cur.img * H + y0 + 2 * rp) * W + x0 + cx;
            o[0] = 0.3f * (1.0f / (1.0f + expf(-acc0)));
            o[W] = 0.3f * (1.0f / (1.0f + expf(-acc1)));
        }
        __syncthreads();
        cur = ntl;
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
    hipLaunchKernelGGL(dec_tail1_x3_kernel, dim3((unsigned)(tiles < cus ? tiles : cus)), dim3(DT_NT), 0, s, p);
    return hipGetLastError();
}

}  // namespace sd
