// Direct 3x3 stride-1 convolution of the fp32-grade split engine (SD_PREC_BF16X3): every layer conv_direct.hip takes under the other
// split precisions (VGG conv1_2..conv5_3 with their pools, the ResNet 3x3 layers, the monodepth decoder), on THREE bf16 planes per
// operand -- v = hi + mid + lo exactly (split_fmt.hpp) -- and SIX MFMA products per product:
//        x*w  =  x_hi*w_hi + x_mid*w_hi + x_lo*w_hi  +  x_hi*w_mid + x_mid*w_mid  +  x_hi*w_lo     (+ three terms below 2^-23 |x w|)
// Same tiling as conv_direct.hip (a workgroup of 8 waves owns a 16 x 32 pixel tile, passes of 64 output channels, per 16-channel chunk
// the 18 x 34 halo tile is DMAed into LDS once and serves all nine taps), but three planes of X (60 KB) plus three of W (54 KB) per chunk
// do not fit twice into 160 KB.  The products are therefore grouped BY WEIGHT PLANE into three phases per chunk,
//        phase lo : W_lo  x (X_hi)                     36 MFMAs per wave (NB = 2)
//        phase mid: W_mid x (X_mid, X_hi)              72
//        phase hi : W_hi  x (X_lo, X_mid, X_hi)       108
// X is double-buffered per chunk (2 x 60 KB) and the weight planes stream through a two-slot ring (2 x 18 KB): 156 KB + 2 KB of bias.
// While a phase multiplies, the LDS-DMA brings the next weight plane and, spread over the hi / mid phases, the next chunk's X planes:
//        phase lo issues  W_mid(c)                               18 KB under  36 MFMAs
//        phase mid issues W_hi(c)   + X_hi(c+1)                  38 KB under  72
//        phase hi issues  W_lo(c+1) + X_mid(c+1) + X_lo(c+1)     58 KB under 108
// so every phase moves at most ~0.55 of the bytes the 16 B/clk L2 -> LDS path could move in its MFMA time.  One barrier per phase.
// The (tile, pass, chunk, phase) sequence is one software pipeline across tiles (persistent workgroups, XCD-aware tile order).
// Epilogue: bias + activation in f32 (the f32 engine's own ELU), the exact three-way split, LDS transpose in the consumed X buffer,
// 16-byte runs of 8 channels per pixel and plane; optional fused 2x2 max pool.
#include <cstdlib>
#include "kernels.hpp"
#include "split_fmt.hpp"

namespace sd {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// one LDS-DMA (64 lanes x 16 B, lane-linear destination); inline asm so hipcc neither counts nor serialises them (conv_dma.hip)
__device__ __forceinline__ void d3dma16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
typedef int i32x8t __attribute__((ext_vector_type(8)));
__device__ __forceinline__ DirectChunk d3load_chunk(const DirectChunk* ptr) {
    i32x8t v;
    asm volatile("s_load_dwordx8 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(ptr) : "memory");
    DirectChunk e;
    e.base = reinterpret_cast<const void*>(((unsigned long long)(unsigned)v[1] << 32) | (unsigned)v[0]);
    e.H = v[2]; e.W = v[3]; e.C = v[4]; e.up = v[5]; e.nvalid = v[6]; e.pad = v[7];
    return e;
}

constexpr int T3_TW = 32, T3_HW = T3_TW + 2, T3_WAVES = 8, T3_TH = 16, T3_HH = T3_TH + 2, T3_MT = 2;

// NB = 32-channel blocks of output channels per pass (Cout <= 32 NB).  UP: every source is read through a x2 nearest-neighbour
// upsample (the upconv layers): the LDS tile holds the 10 x 18 SOURCE pixels under the halo (conv_direct.hip).
// KEEP: X planes (hi first) whose MFMA fragments stay in VGPRs across the phases of a chunk instead of being re-read from LDS: the
// kernel issues 0.58 ds_read_b128 per MFMA without it (x: 72, w: 54 per 216 MFMAs and wave) and the LDS read path (128 B/clk/CU),
// not the MFMA pipe, bounds it; KEEP = 2 reads every X fragment once (x: 36).
// WSLOTS: slots of the weight-plane ring.  2 (NB = 2: all the LDS there is): every phase waits for everything issued in the phase before it.
// 3 (NB = 1, the 16 / 32-channel full-resolution layers): phases of 18 / 36 / 54 MFMAs per wave are shorter than the latency of the DMAs
// they would have to cover, so a weight plane is issued TWO phases before its phase, each X plane of the next chunk two phases before its
// first read, and a phase waits with a counted vmcnt only for what was issued two phases ago.
// FOLD (UP layers with ONE source; ConvDirectParams::fold): the upsample-folded form -- per output parity (y & 1, x & 1) a 2x2 conv on the
// source with the 3x3 taps that read the same source pixel added up (plan.hpp OpDesc::fold): 16 tap matrices per chunk instead of 9, but
// 4 instead of 9 MFMAs per output pixel.  Wave w owns parity w & 3 and the source rows 4 (w >> 2) .. + 3 of the tile: its two 32-pixel MFMA
// column groups are 2 source rows x 16 source columns each.
// TIMED (SEMDEPTH_X3_DIAG=4; <2, false, 2, 2> only, decomposition runs): s_memtime stamps around every phase's wait + barrier, the barrier in front of the
// epilogue and the epilogue itself, summed per wave over the items of a workgroup; waves 0 and 4 of the middle workgroup print their sums.
template <int NB, bool UP, int KEEP, int WSLOTS, bool FOLD = false, bool TIMED = false>
__global__ __launch_bounds__(512, 1) void conv_direct3_kernel(const ConvDirectParams p) {
    int tm_loop = 0, tm_wb = 0, tm_wb0 = 0, tm_bar = 0, tm_ep = 0, tm_start = 0, tm_pro = 0, tm_epv = 0, tm_epw = 0, tm_eps = 0;      // (low 32 bits of s_memtime)
    auto now = []() { return (int)__builtin_amdgcn_s_memtime(); };
    int tm_items = 0;
    if constexpr (TIMED) tm_start = now();
    static_assert(!FOLD || (UP && WSLOTS == 2 && KEEP == 2), "the folded form: upconv layers, two-slot ring, kept X fragments");
    constexpr int NTAP = FOLD ? 16 : 9;                        // tap matrices per chunk
    constexpr int S_HH = UP ? T3_HH / 2 + 1 : T3_HH, S_HW = UP ? T3_HW / 2 + 1 : T3_HW;      // stored tile
    constexpr int XI = (S_HH * S_HW * 2 + 63) / 64;            // DMA instructions per halo plane (2 octet slots per pixel)
    constexpr int XUNITS = XI * 64;
    constexpr int XS = (XI + T3_WAVES - 1) / T3_WAVES;         // X-DMA slots per wave and plane
    constexpr int WI = NTAP * NB;                              // DMA instructions per weight plane
    constexpr int WUNITS = NTAP * 2 * 32 * NB;                 // taps x octets x output channels
    constexpr int WS = (WI + T3_WAVES - 1) / T3_WAVES;         // weight-DMA slots per wave and plane
    constexpr int ROW = 64 * NB + 16;                          // epilogue slab row (bytes per pixel and plane, + pad)
    // planes per epilogue round: with 32 output channels a wave's three planes of 32 pixels fit the consumed X buffer side by side, so a
    // row of outputs costs ONE LDS write -> read round trip instead of three (the 16 / 32-channel layers have two or three chunks per
    // tile: their epilogue is a third of the tile's time and is a chain of LDS round trips with the MFMA pipes idle)
    constexpr int EPL = NB == 1 ? 3 : 1;
    constexpr int SLAB = T3_WAVES * EPL * 32 * ROW / 16;       // units: EPL planes of 32 pixels per wave
    constexpr int XBUF = 3 * XUNITS < SLAB ? SLAB : 3 * XUNITS;
    static_assert((2 * XBUF + WSLOTS * WUNITS) * 16 + 2048 <= 160 * 1024, "two X buffers + the weight ring + bias fit the LDS of a CU");
    static_assert(WS + 2 * XS <= (FOLD ? 4 : 9) * NB, "one DMA slot per MFMA group of a phase");
    __shared__ __attribute__((aligned(16))) u32x4 lds[2 * XBUF + WSLOTS * WUNITS];
    __shared__ __attribute__((aligned(16))) float sbias[512];
    auto hpix = [](int hy, int hx) { return UP ? ((hy + 1) >> 1) * S_HW + ((hx + 1) >> 1) : hy * S_HW + hx; };

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int tiles_x = p.W / T3_TW, tiles_y = (p.H + T3_TH - 1) / T3_TH;
    const int total = tiles_x * tiles_y * p.N;
    const u32x4* const zero = reinterpret_cast<const u32x4*>(p.zero16);
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) void*)lds;      // LDS byte address
    const int frow = lane & 31, fk = lane >> 5;
    const int fpar = wave & 3, fpy = fpar >> 1, fpx = fpar & 1, fhh = wave >> 2;     // FOLD: this wave's parity and half of the tile's source rows
    // SEMDEPTH_X3_DIAG (decomposition runs, latched in the handle's switches): 1 no output stores, 2 no MFMAs; 0 in production
    const int diag = SD_DIAG_BITS(p.sw);

    // work item = (tile, pass of <= 64 output channels); the passes of a tile are neighbouring items
    const int items = total * p.nsplit;
    struct Tile { int img, ty0, tx0, half; };
    auto tile_of = [&](int it) {
        if ((items & 7) == 0) it = (it & 7) * (items >> 3) + (it >> 3);     // neighbouring items (shared halos) on one XCD
        Tile r;
        int tid = it / p.nsplit;
        r.half = it - tid * p.nsplit;
        const int bx = tid % tiles_x; tid /= tiles_x;
        r.tx0 = bx * T3_TW; r.ty0 = (tid % tiles_y) * T3_TH; r.img = tid / tiles_y;
        return r;
    };

    // halo geometry of this lane's X-DMA slots (instruction j = wave + 8 i of a plane): constant over tiles and chunks.  The element offset
    // of slot i inside a source is ((gy0 + ry) W + gx0 + rx) C + octet: the lane's pixel share ry W + rx is the same for every tile and every
    // source that is not read through an upsample of its own (they all have the layer's width) -- kept in a register per slot; the tile's
    // share is scalar arithmetic and the product with C one full-rate 24-bit multiply-add.  (Recomputing row and column per slot cost two
    // quarter-rate multiplies + ten VALU per DMA instruction, and the MFMA and VALU pipes of a SIMD do not overlap.)
    auto slot_geo = [&](int i, int& ry, int& rx, int& oct) {    // halo row / column / channel octet of slot i; false: beyond the halo
        const int u = (wave + T3_WAVES * i) * 64 + lane;
        const int pix = u >> 1;
        oct = (u & 1) ^ ((pix >> 3) & 1);
        ry = pix / S_HW; rx = pix - ry * S_HW;
        return pix < S_HH * S_HW;
    };
    const int Wsrc = UP ? p.W >> 1 : p.W;                      // width of the sources read at the layer's own resolution
    unsigned pshare[XS];                                       // ry W + rx
    unsigned octb = 0;                                         // bit i: octet of slot i
#pragma unroll
    for (int i = 0; i < XS; ++i) {
        int ry, rx, oct;
        slot_geo(i, ry, rx, oct);
        pshare[i] = (unsigned)(ry * Wsrc + rx);
        octb |= (unsigned)oct << i;
    }
    unsigned okA = 0, okB = 0;            // bit i: slot i reads an existing pixel (okB: ... and the first channel octet of a chunk)
    int tgy = 0, tgx = 0;                 // source coordinates of halo pixel (0, 0) of the cursor tile (wave-uniform)
    auto set_tile = [&](const Tile& tl) {
        okA = 0; okB = 0;
        tgy = (UP ? (tl.ty0 >> 1) : tl.ty0) - 1; tgx = (UP ? (tl.tx0 >> 1) : tl.tx0) - 1;
#pragma unroll
        for (int i = 0; i < XS; ++i) {
            int ry, rx, oct;
            const bool halo = slot_geo(i, ry, rx, oct);
            const int gy = tgy + ry, gx = tgx + rx;
            const bool in = halo && (unsigned)gy < (unsigned)(UP ? p.H >> 1 : p.H) && (unsigned)gx < (unsigned)(UP ? p.W >> 1 : p.W);
            okA |= (in ? 1u : 0u) << i;
            okB |= ((in && !oct) ? 1u : 0u) << i;
        }
    };
    // a chunk in flight: source image + plane stride, its X buffer, its weight block (hi plane; plane stride = nchunks * WUNITS)
    // (tile0: the source element under halo pixel (0, 0) of the tile -- the 64-bit products of a slot's address are formed ONCE per chunk, not per DMA instruction)
    struct ChunkCtx { const uint16_t* tile0; size_t plane; unsigned rowel, C, okm, xbyte; int up, gy0, gx0; const u32x4* w; };
    auto begin_chunk = [&](const DirectChunk& ch, const Tile& tl, int c, int buf) {
        ChunkCtx k;
        k.plane = (size_t)p.Nmax * ch.H * ch.W * (ch.pad ? ch.pad : ch.C);      // elements (pad: the channel count of a tensor stored as 16-channel sub-planes)
        k.rowel = (unsigned)(ch.W * ch.C); k.C = (unsigned)ch.C; k.up = UP ? 0 : ch.up;
        k.okm = ch.nvalid >= 2 ? okA : okB;
        k.gy0 = tgy; k.gx0 = tgx;                           // (the tile of the cursor at this moment: a chunk's X planes may be issued after the cursor moved on)
        k.tile0 = reinterpret_cast<const uint16_t*>(ch.base) + (size_t)tl.img * ch.H * ch.W * ch.C + ((ptrdiff_t)k.gy0 * (ptrdiff_t)k.rowel + (ptrdiff_t)k.gx0 * (ptrdiff_t)k.C);
        k.xbyte = lds0 + (unsigned)(buf * XBUF * 16);
        k.w = p.wt + ((size_t)(3 * tl.half) * p.nchunks + c) * WUNITS;
        return k;
    };
    auto xslot = [&](const ChunkCtx& k, int pl, int i) {       // X-DMA instruction wave + 8 i of plane pl
        const int j = wave + T3_WAVES * i;
        if (j >= XI) return;
        const uint16_t* src;
        if (k.up == 0) {        // (wave-uniform) scalar tile share + the lane's pixel share x C
            const unsigned off = __umul24(pshare[i], k.C) + (((octb >> i) & 1u) << 3);
            src = k.tile0 + (ptrdiff_t)pl * (ptrdiff_t)k.plane + off;
        } else {                // a x2-upsampled source of a mixed layer (up2(disp) of the iconv layers): halo coordinates halved per pixel
            int ry, rx, oct;
            slot_geo(i, ry, rx, oct);
            const int gy = k.gy0 + ry, gx = k.gx0 + rx;
            const unsigned off = (unsigned)(gy >> 1) * k.rowel + ((unsigned)(gx >> 1) * k.C + (unsigned)(oct << 3));
            const uint16_t* img = k.tile0 - ((ptrdiff_t)k.gy0 * (ptrdiff_t)k.rowel + (ptrdiff_t)k.gx0 * (ptrdiff_t)k.C);
            src = img + (size_t)pl * k.plane + off;
        }
        d3dma16(((k.okm >> i) & 1u) ? reinterpret_cast<const u32x4*>(src) : zero, k.xbyte + (unsigned)((pl * XUNITS + j * 64) * 16));
    };
    const size_t wplane_u = (size_t)p.nchunks * WUNITS;      // units between the planes of a weight block
    auto wslot = [&](const u32x4* wbase, int pl, int i, int slot) {     // weight-DMA instruction wave + 8 i of plane pl into ring slot
        const int jw = wave + T3_WAVES * i;
        if (jw >= WI) return;
        d3dma16(wbase + (size_t)pl * wplane_u + jw * 64 + lane, lds0 + (unsigned)((2 * XBUF + slot * WUNITS + jw * 64) * 16));     // slot < WSLOTS
    };

    sbias[threadIdx.x] = (int)threadIdx.x < p.nsplit * p.Cout ? p.bias[threadIdx.x] : 0.f;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    int tid = blockIdx.x;
    if (tid >= items) return;
    Tile cur = tile_of(tid);
    // the issue cursor: the (tile, chunk) item whose X planes and W_hi go into the ring next
    int itid = tid, ic = 0;
    Tile icur = cur;
    set_tile(icur);
    auto advance = [&]() {
        if (++ic == p.nchunks) { ic = 0; itid += gridDim.x; if (itid < items) { icur = tile_of(itid); set_tile(icur); } }
    };
    const u32x4* wcur;                   // weight block of the chunk being multiplied
    ChunkCtx kc;                         // ... and its context (WSLOTS == 3: its X_lo plane is issued during its own first phase)
    // DMA instructions this wave issues per weight plane / per X plane (the counted waits of the three-slot pipeline)
    int nwv = 0, nxv = 0;
#pragma unroll
    for (int i = 0; i < WS; ++i) nwv += (wave + T3_WAVES * i < WI) ? 1 : 0;
#pragma unroll
    for (int i = 0; i < XS; ++i) nxv += (wave + T3_WAVES * i < XI) ? 1 : 0;
    int prev = 0;                        // DMA instructions this wave issued in the previous phase
    auto wait_prev = [&]() {             // everything issued BEFORE the previous phase has landed
        switch (prev) {
            case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
            case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
            case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
            case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
            case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
            case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
            default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        }
    };
    {
        kc = begin_chunk(d3load_chunk(p.chunks), icur, 0, 0);
        if constexpr (WSLOTS == 3) {
            // the issues of the virtual phases -2 and -1: (W_lo, X_hi) and (W_mid, X_mid) of chunk 0; X_lo and W_hi follow in phase 0
#pragma unroll
            for (int i = 0; i < WS; ++i) wslot(kc.w, 2, i, 0);
#pragma unroll
            for (int i = 0; i < XS; ++i) xslot(kc, 0, i);
#pragma unroll
            for (int i = 0; i < WS; ++i) wslot(kc.w, 1, i, 1);
#pragma unroll
            for (int i = 0; i < XS; ++i) xslot(kc, 1, i);
            prev = nwv + nxv;
        } else {
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
#pragma unroll
                for (int i = 0; i < XS; ++i) xslot(kc, pl, i);
#pragma unroll
            for (int i = 0; i < WS; ++i) wslot(kc.w, 2, i, 0);         // the first phase of a chunk multiplies W_lo
        }
        wcur = kc.w;
        advance();
    }
    int g = 0, q = 0;                    // chunks / phases consumed so far
    if constexpr (TIMED) tm_pro = now() - tm_start;
    for (; tid < items; tid += gridDim.x) {
        const int half = cur.half;
        int tm_i0 = 0;
        if constexpr (TIMED) { tm_i0 = now(); ++tm_items; }
        f32x16 acc[T3_MT][NB];
#pragma unroll
        for (int a = 0; a < T3_MT; ++a)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][nb][r] = 0.f;
        for (int c = 0; c < p.nchunks; ++c, ++g) {
            const bool more = itid < items;        // the cursor item exists: its X goes into buffer (g + 1) & 1 during this chunk
            ChunkCtx kn;
            const u32x4* const Xb = lds + (g & 1) * XBUF;
            auto xload = [&](int pl, int dx, int r) {
                const int lp = hpix(T3_MT * wave + r, frow + dx);
                return Xb[pl * XUNITS + lp * 2 + (fk ^ ((lp >> 3) & 1))];
            };
            // FOLD: fragment (column shift tb, row offset ro = 2 a + ta) of plane pl: lane = (source row r2 = frow >> 4, source column frow & 15)
            // -> halo pixel (4 hh + ro + r2 + py, column + tb + px) of the stored source tile
            auto xloadf = [&](int pl, int tb, int ro) {
                const int lp = (4 * fhh + ro + (frow >> 4) + fpy) * S_HW + (frow & 15) + tb + fpx;
                return Xb[pl * XUNITS + lp * 2 + (fk ^ ((lp >> 3) & 1))];
            };
            // the X fragments of the first KEEP planes stay in registers from the phase that first reads them to the end of the chunk
            u32x4 xk[KEEP > 0 ? KEEP : 1][3][T3_MT + 2];
            // one phase: the weight plane in ring slot q & 1 times X planes PH .. 0 (phase 0: W_lo x X_hi; 1: W_mid x (X_mid, X_hi);
            // 2: W_hi x (X_lo, X_mid, X_hi)); `issue(grp)` is called behind MFMA group grp (one DMA instruction per call).
            // The LDS reads run AHEAD of the MFMAs that consume them (an explicit software pipeline: left to itself the compiler puts every
            // weight fragment's ds_read + s_waitcnt directly in front of its 2 NPX MFMAs, and in the lo / mid phases the two waves of a
            // SIMD do not have enough MFMAs per group to cover an LDS round trip): the weight fragment of group g + 2 and the X rows of the
            // plane this phase reads for the first time (row dy + 2 / the first two rows of the next dx) are issued before group g's MFMAs.
            auto phase = [&](auto ph_tag, auto nomfma_tag, auto&& issue) {
                constexpr int PH = decltype(ph_tag)::value, NPX = PH + 1, G = 9 * NB;
                constexpr bool NOMFMA = decltype(nomfma_tag)::value != 0;
                constexpr bool NEWKEPT = PH < KEEP;              // the plane first read in this phase stays in xk
                const u32x4* const Wq = lds + 2 * XBUF + (q % WSLOTS) * WUNITS;
                auto wfrag = [&](int grp) {
                    const int dx = grp / (3 * NB), dy = (grp / NB) % 3, nb = grp % NB;
                    return Wq[((dy * 3 + dx) * 2 + fk) * (32 * NB) + nb * 32 + frow];
                };
                u32x4 xn[T3_MT + 2];                             // rows of the new plane when it is not kept
                u32x4 xo[KEEP < PH ? PH - KEEP : 1][T3_MT + 2];  // older planes that are not kept: re-read per dx
                auto xnew = [&](int dx, int r) {
                    if constexpr (NEWKEPT) xk[PH][dx][r] = xload(PH, dx, r);
                    else xn[r] = xload(PH, dx, r);
                };
                u32x4 wq[3];
                wq[0] = wfrag(0);
                wq[1] = wfrag(1);
                xnew(0, 0); xnew(0, 1);
#pragma unroll
                for (int grp = 0; grp < G; ++grp) {
                    const int dx = grp / (3 * NB), dy = (grp / NB) % 3, nb = grp % NB;
                    if (dy == 0 && nb == 0) {
#pragma unroll
                        for (int pl = KEEP; pl < PH; ++pl)
#pragma unroll
                            for (int r = 0; r < T3_MT + 2; ++r) xo[pl - KEEP][r] = xload(pl, dx, r);
                    }
                    if (grp + 2 < G) wq[(grp + 2) % 3] = wfrag(grp + 2);
                    if (nb == 0) {
                        if (dy < 2) xnew(dx, dy + 2);
                        else if (dx < 2) { xnew(dx + 1, 0); xnew(dx + 1, 1); }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr (!NOMFMA) {
                        const u32x4 wv = wq[grp % 3];
#pragma unroll
                        for (int pl = NPX - 1; pl >= 0; --pl)          // the smaller planes first
#pragma unroll
                            for (int a = 0; a < T3_MT; ++a) {
                                const u32x4 xv = pl < KEEP ? xk[pl < KEEP ? pl : 0][dx][a + dy] : (pl == PH ? xn[a + dy] : xo[pl >= KEEP && pl < PH ? pl - KEEP : 0][a + dy]);
                                acc[a][nb] = mfma_frag<false>(wv, xv, acc[a][nb]);
                            }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    issue(grp);
                    __builtin_amdgcn_sched_barrier(0);
                }
            };
            // the folded form of a phase: 4 NB groups (tap t = 2 ta + tb, block nb) of 2 NPX MFMAs; the eight fragments of the plane the phase
            // reads for the first time go out before its first MFMA, the weight fragment of group g + 2 before group g's MFMAs
            auto phase_fold = [&](auto ph_tag, auto&& issue) {
                constexpr int PH = decltype(ph_tag)::value, NPX = PH + 1, G = 4 * NB;
                const u32x4* const Wq = lds + 2 * XBUF + (q % WSLOTS) * WUNITS;
                auto wfrag = [&](int grp) {
                    const int tp = grp / NB, nb = grp % NB;
                    return Wq[((fpar * 4 + tp) * 2 + fk) * (32 * NB) + nb * 32 + frow];
                };
                u32x4 xn[2][4];                                  // the lo plane (read in the hi phase only)
                u32x4 wq[3];
                wq[0] = wfrag(0);
                wq[1] = wfrag(1);
#pragma unroll
                for (int tb = 0; tb < 2; ++tb)
#pragma unroll
                    for (int ro = 0; ro < 4; ++ro) {
                        if constexpr (PH < KEEP) xk[PH][tb][ro] = xloadf(PH, tb, ro);
                        else xn[tb][ro] = xloadf(PH, tb, ro);
                    }
#pragma unroll
                for (int grp = 0; grp < G; ++grp) {
                    const int tp = grp / NB, ta = tp >> 1, tb = tp & 1, nb = grp % NB;
                    if (grp + 2 < G) wq[(grp + 2) % 3] = wfrag(grp + 2);
                    __builtin_amdgcn_sched_barrier(0);
                    const u32x4 wv = wq[grp % 3];
#pragma unroll
                    for (int pl = NPX - 1; pl >= 0; --pl)          // the smaller planes first
#pragma unroll
                        for (int a = 0; a < T3_MT; ++a) {
                            const u32x4 xv = pl < KEEP ? xk[pl < KEEP ? pl : 0][tb][2 * a + ta] : xn[tb][2 * a + ta];
                            acc[a][nb] = mfma_frag<false>(wv, xv, acc[a][nb]);
                        }
                    __builtin_amdgcn_sched_barrier(0);
                    issue(grp);
                    __builtin_amdgcn_sched_barrier(0);
                }
            };
            // (SEMDEPTH_X3_DIAG & 2 selects the copy of a phase without MFMAs; production runs the branch-free one)
#define SD_PHASE(PH_, ...) do { if constexpr (FOLD) phase_fold(IntTag<PH_>{}, __VA_ARGS__); \
                                else if (diag & 2) phase(IntTag<PH_>{}, IntTag<1>{}, __VA_ARGS__); else phase(IntTag<PH_>{}, IntTag<0>{}, __VA_ARGS__); } while (0)
            if constexpr (WSLOTS == 3) {
                // ---- three-slot pipeline: phase q issues the weight plane of phase q + 2 and the X plane first read in phase q + 2
                // lo(c): W_hi(c) + X_lo(c)
                wait_prev();
                __builtin_amdgcn_s_barrier();
                SD_PHASE(0, [&](int grp) {
                    if (grp < WS) wslot(wcur, 0, grp, (q + 2) % 3);
                    else if (grp < WS + XS) xslot(kc, 2, grp - WS);
                });
                prev = nwv + nxv;
                ++q;
                // mid(c): W_lo(c+1) + X_hi(c+1)
                wait_prev();
                __builtin_amdgcn_s_barrier();
                if (more) kn = begin_chunk(d3load_chunk(p.chunks + ic), icur, ic, (g + 1) & 1);
                SD_PHASE(1, [&](int grp) {
                    if (!more) return;
                    if (grp < WS) wslot(kn.w, 2, grp, (q + 2) % 3);
                    else if (grp < WS + XS) xslot(kn, 0, grp - WS);
                });
                prev = more ? nwv + nxv : 0;
                ++q;
                // hi(c): W_mid(c+1) + X_mid(c+1)
                wait_prev();
                __builtin_amdgcn_s_barrier();
                SD_PHASE(2, [&](int grp) {
                    if (!more) return;
                    if (grp < WS) wslot(kn.w, 1, grp, (q + 2) % 3);
                    else if (grp < WS + XS) xslot(kn, 1, grp - WS);
                });
                prev = more ? nwv + nxv : 0;
                ++q;
                if (more) { kc = kn; }
            } else {
            // ---- phase lo: W_lo x X_hi (36 MFMAs per wave at NB = 2); brings W_mid of this chunk
            int tm_a = 0;
            if constexpr (TIMED) tm_a = now();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if constexpr (TIMED) { const int d = now() - tm_a; tm_wb += d; if (c == 0) tm_wb0 += d; }
            SD_PHASE(0, [&](int grp) {
                if (grp < WS) wslot(wcur, 1, grp, (q + 1) & 1);
            });
            ++q;
            // ---- phase mid: W_mid x (X_mid, X_hi) (72); brings W_hi of this chunk and X_hi of the cursor chunk
            if constexpr (TIMED) tm_a = now();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if constexpr (TIMED) tm_wb += now() - tm_a;
            if (more) kn = begin_chunk(d3load_chunk(p.chunks + ic), icur, ic, (g + 1) & 1);
            SD_PHASE(1, [&](int grp) {
                if (grp < WS) wslot(wcur, 0, grp, (q + 1) & 1);
                else if (more && grp < WS + XS) xslot(kn, 0, grp - WS);
            });
            ++q;
            // ---- phase hi: W_hi x (X_lo, X_mid, X_hi) (108); brings W_lo and X_mid, X_lo of the cursor chunk
            if constexpr (TIMED) tm_a = now();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if constexpr (TIMED) tm_wb += now() - tm_a;
            SD_PHASE(2, [&](int grp) {
                if (!more) return;
                if (grp < WS) wslot(kn.w, 2, grp, (q + 1) & 1);
                else if (grp < WS + 2 * XS) { const int s_ = grp - WS; xslot(kn, 1 + s_ / XS, s_ % XS); }
            });
            ++q;
            }
            if (more) { wcur = kn.w; advance(); }
        }

        // ---- epilogue: bias + activation, the exact three-way split, LDS transpose one plane at a time in the X buffer just consumed
        //      (the other one is being filled for the next item), 16-byte runs of 8 channels per pixel and plane ----
        int tm_e0 = 0, tm_e1 = 0;
        if constexpr (TIMED) tm_e0 = now();
        __builtin_amdgcn_s_barrier();
        if constexpr (TIMED) { tm_e1 = now(); tm_loop += tm_e0 - tm_i0; tm_bar += tm_e1 - tm_e0; }
        auto epilogue = [&](auto tag) {
            constexpr int ACT = decltype(tag)::value;
            constexpr int SEGS = 4 * NB, PPP = 64 / SEGS;        // 16-byte segments per pixel, pixels per store pass
            unsigned char* sh = reinterpret_cast<unsigned char*>(lds + ((g - 1) & 1) * XBUF) + wave * (EPL * 32 * ROW);
            const int seg = lane % SEGS, prow = lane / SEGS;
            uint16_t* const out_hi = reinterpret_cast<uint16_t*>(p.out);
            const int n0 = half * p.Cout;                 // first output channel of this pass
            f32x4 bias[4 * NB];
#pragma unroll
            for (int r4 = 0; r4 < 4 * NB; ++r4) bias[r4] = *reinterpret_cast<const f32x4*>(sbias + n0 + 8 * r4 + 4 * fk);
            if (p.pool) {
                // fused 2x2 max pool: vertical max across the wave's two rows (same lane), horizontal across lane pairs
                // (pixel = lane & 31), THEN bias + activation (monotonic) on a quarter of the values
                uint2 pp[3][4 * NB];
#pragma unroll
                for (int r4 = 0; r4 < 4 * NB; ++r4) {
                    if (8 * r4 >= p.Cout) continue;
                    const int nb = r4 >> 2, q4 = r4 & 3;
                    f32x4 v;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float m = fmaxf(acc[0][nb][4 * q4 + r], acc[1][nb][4 * q4 + r]);
                        v[r] = fmaxf(m, __shfl_xor(m, 1));
                    }
                    v += bias[r4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = act_x3<ACT>(v[r]);
                    split4_x3(v, pp[0][r4], pp[1][r4], pp[2][r4]);
                }
                const int pq = frow >> 1;                    // pooled pixel of this lane pair
                const int yp = (cur.ty0 >> 1) + wave, Hp = p.H >> 1, Wp = p.W >> 1;
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) {
                    if (!(lane & 1)) {
#pragma unroll
                        for (int r4 = 0; r4 < 4 * NB; ++r4) {
                            if (8 * r4 >= p.Cout) continue;
                            *reinterpret_cast<uint2*>(sh + pq * ROW + (8 * r4 + 4 * fk) * 2) = pp[pl][r4];
                        }
                    }
                    __builtin_amdgcn_wave_barrier();
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                    for (int ps = 0; ps < (16 + PPP - 1) / PPP; ++ps) {
                        const int pix = ps * PPP + prow;
                        if (pix < 16 && yp < Hp && seg * 8 < p.Cout && !(diag & 1)) {
                            uint16_t* o = out_hi + ((size_t)(cur.img * Hp + yp) * Wp + (cur.tx0 >> 1) + pix) * p.Cstride + n0 + seg * 8;
                            *reinterpret_cast<u32x4*>(o + pl * p.out_plane) = *reinterpret_cast<const u32x4*>(sh + pix * ROW + seg * 16);
                        }
                    }
                    __builtin_amdgcn_wave_barrier();
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                }
                return;
            }
#pragma unroll
            for (int a = 0; a < T3_MT; ++a) {
                const int y = cur.ty0 + T3_MT * wave + a;
                uint2 pp[3][4 * NB];
                int tm_x = 0;
                if constexpr (TIMED) tm_x = now();
#pragma unroll
                for (int r4 = 0; r4 < 4 * NB; ++r4) {
                    if (8 * r4 >= p.Cout) continue;         // rows past Cout are padding, never stored
                    const int nb = r4 >> 2, q4 = r4 & 3;
                    f32x4 v = {acc[a][nb][4 * q4], acc[a][nb][4 * q4 + 1], acc[a][nb][4 * q4 + 2], acc[a][nb][4 * q4 + 3]};
                    v += bias[r4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = act_x3<ACT>(v[r]);
                    split4_x3(v, pp[0][r4], pp[1][r4], pp[2][r4]);
                }
                if constexpr (TIMED) { __builtin_amdgcn_sched_barrier(0); const int n = now(); tm_epv += n - tm_x; tm_x = n; }
#pragma unroll
                for (int pl0 = 0; pl0 < 3; pl0 += EPL) {
#pragma unroll
                    for (int e = 0; e < EPL; ++e)
#pragma unroll
                        for (int r4 = 0; r4 < 4 * NB; ++r4) {
                            if (8 * r4 >= p.Cout) continue;
                            *reinterpret_cast<uint2*>(sh + e * (32 * ROW) + frow * ROW + (8 * r4 + 4 * fk) * 2) = pp[pl0 + e][r4];
                        }
                    __builtin_amdgcn_wave_barrier();
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    if constexpr (TIMED) { const int n = now(); tm_epw += n - tm_x; tm_x = n; }
#pragma unroll
                    for (int ps = 0; ps < 32 / PPP; ++ps) {
                        const int pix = ps * PPP + prow;
                        // FOLD: pixel pix of column group a = source (row 4 hh + 2 a + (pix >> 4), column pix & 15) of this wave's parity
                        const int yo = FOLD ? cur.ty0 + 2 * (4 * fhh + 2 * a + (pix >> 4)) + fpy : y;
                        const int xo = FOLD ? cur.tx0 + 2 * (pix & 15) + fpx : cur.tx0 + pix;
                        if (yo < p.H && seg * 8 < p.Cout && !(diag & 1)) {
                            uint16_t* o = out_hi + ((size_t)(cur.img * p.H + yo) * p.W + xo) * p.Cstride + n0 + seg * 8;
#pragma unroll
                            for (int e = 0; e < EPL; ++e)
                                *reinterpret_cast<u32x4*>(o + (pl0 + e) * p.out_plane) = *reinterpret_cast<const u32x4*>(sh + e * (32 * ROW) + pix * ROW + seg * 16);
                        }
                    }
                    __builtin_amdgcn_wave_barrier();
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    if constexpr (TIMED) { const int n = now(); tm_eps += n - tm_x; tm_x = n; }
                }
            }
        };
        if (p.act == ACT_RELU) epilogue(ActTag<ACT_RELU>{});
        else if (p.act == ACT_ELU) epilogue(ActTag<ACT_ELU>{});
        else epilogue(ActTag<ACT_NONE>{});
        if constexpr (TIMED) tm_ep += now() - tm_e1;
        if constexpr (WSLOTS == 3) {
            // gfx9 counts stores in vmcnt too and loads / stores may complete out of order with respect to each other: a COUNTED wait is
            // only sound over the DMA loads alone.  Drain the epilogue's stores (and the DMAs of the last phase with them) once per tile.
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            prev = 0;
        }
        if (tid + (int)gridDim.x < items) cur = tile_of(tid + gridDim.x);
    }
#undef SD_PHASE
    if constexpr (TIMED) {
        const int total = now() - tm_start;
        if ((int)blockIdx.x == (int)gridDim.x / 2 && lane == 0 && (wave == 0 || wave == 4))
            printf("[conv_direct3 timed] H=%d W=%d Cout=%d chunks=%d items=%d wave %d: prologue %d | per item: loop %.0f (wait+barrier of its %d phases %.0f, of the first one %.0f) "
                   "barrier before the epilogue %.0f epilogue %.0f (bias + act + split %.0f, LDS writes + wait %.0f, LDS reads + stores + wait %.0f) | total %d ticks\n",
                   p.H, p.W, p.Cout * p.nsplit, p.nchunks, tm_items, wave, tm_pro, (double)tm_loop / tm_items, 3 * p.nchunks, (double)tm_wb / tm_items, (double)tm_wb0 / tm_items,
                   (double)tm_bar / tm_items, (double)tm_ep / tm_items, (double)tm_epv / tm_items, (double)tm_epw / tm_items, (double)tm_eps / tm_items, total);
    }
}

hipError_t launch_conv_direct3(const ConvDirectParams& p, hipStream_t s) {
    if (p.act == ACT_SIGMOID03 || p.nreal || p.out_planar16 || p.f16 || p.out_f16) return hipErrorInvalidValue;   // (heads run as small-N kernels)
    if (p.W % T3_TW || p.Cout > 64 || p.Cout % 8 || p.nsplit < 1 || p.nsplit > 8 || (p.nsplit > 1 && p.Cout != 64)) return hipErrorInvalidValue;
    if (p.pool && ((p.H & 1) || (p.W & 1))) return hipErrorInvalidValue;
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return hipErrorInvalidDevice;
        cus = prop.multiProcessorCount;
    }
    const bool up = p.all_up && !(p.H & 1) && !(p.W & 1) && !(p.sw & SW_NO_UPTILE);
    const int tiles = (p.W / T3_TW) * ((p.H + T3_TH - 1) / T3_TH) * p.N * p.nsplit;
    const int wgs = cus - p.reserve_cus > 0 ? cus - p.reserve_cus : 1;
    const dim3 grid((unsigned)(tiles < wgs ? tiles : wgs));        // persistent: one workgroup per CU (158 KB of LDS)
    if (p.fold) {             // upsample-folded upconv layers: source-resolution tiles, two-slot ring, kept fragments
        if (!up || p.pool) return hipErrorInvalidValue;
        if (p.Cout <= 32) hipLaunchKernelGGL((conv_direct3_kernel<1, true, 2, 2, true>), grid, dim3(512), 0, s, p);
        else hipLaunchKernelGGL((conv_direct3_kernel<2, true, 2, 2, true>), grid, dim3(512), 0, s, p);
        return hipGetLastError();
    }
    // Production: X fragments of the hi / mid planes kept in registers (KEEP = 2), two-slot weight ring.  The variants measured and not
    // adopted -- KEEP = 0 (re-read fragments: -1 %, profiles/r03d_conv_direct3_keep_ab.txt), the three-slot ring of the NB = 1 layers (equal
    // or 3-6 % slower, profiles/r03g_conv_direct3_ring3_ab.txt) -- are compiled only with -DSD_DEV_VARIANTS (SEMDEPTH_X3_KEEP=0,
    // SEMDEPTH_X3_RING3 then select them); the default build carries six instantiations of this kernel instead of sixteen.
    const ConvDirectParams& pd = p;
#ifdef SD_DEV_VARIANTS
    if ((p.sw & SW_X3_DIAG_TIMED) && p.Cout > 32 && !up) {          // SEMDEPTH_X3_DIAG=4: the timed copy of the dominant form
        hipLaunchKernelGGL((conv_direct3_kernel<2, false, 2, 2, false, true>), grid, dim3(512), 0, s, pd);
        return hipGetLastError();
    }
    const int keep = (p.sw & SW_X3_NOKEEP) ? 0 : 2;
    const bool ring3 = (p.sw & SW_X3_RING3) != 0;
#define SD_D3(NB_, UP_, WS_) do { if (keep >= 2) hipLaunchKernelGGL((conv_direct3_kernel<NB_, UP_, 2, WS_>), grid, dim3(512), 0, s, pd); \
                                  else hipLaunchKernelGGL((conv_direct3_kernel<NB_, UP_, 0, WS_>), grid, dim3(512), 0, s, pd); } while (0)
    if (p.Cout <= 32) {
        if (ring3) { if (up) SD_D3(1, true, 3); else SD_D3(1, false, 3); }
        else { if (up) SD_D3(1, true, 2); else SD_D3(1, false, 2); }
    } else {
        if (up) SD_D3(2, true, 2); else SD_D3(2, false, 2);
    }
#else
#define SD_D3(NB_, UP_, WS_) hipLaunchKernelGGL((conv_direct3_kernel<NB_, UP_, 2, WS_>), grid, dim3(512), 0, s, pd)
    if (p.Cout <= 32) { if (up) SD_D3(1, true, 2); else SD_D3(1, false, 2); }
    else { if (up) SD_D3(2, true, 2); else SD_D3(2, false, 2); }
#endif
#undef SD_D3
    return hipGetLastError();
}

const char* conv_direct3_kernel_name(const ConvDirectParams& p) {
    if (p.fold) return p.Cout <= 32 ? "conv_direct_x3_fold_kernel<1>" : "conv_direct_x3_fold_kernel<2>";
    return p.Cout <= 32 ? "conv_direct_x3_kernel<1,2>" : "conv_direct_x3_kernel<2,2>";
}

}  // namespace sd
