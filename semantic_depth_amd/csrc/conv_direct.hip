// Direct 3x3 stride-1 convolution for the full-resolution, few-channel layers of the split-bf16 engine
// (monodepth decoder levels 1-2: Cout <= 32, sources of 8..64 channels, optional x2 nearest-neighbour upsample, concat).
//
// For these layers an im2col GEMM is bound by the gather, not by the MFMA: every activation would be fetched nine times
// (once per tap) for only 32 output channels.  Here a workgroup (8 waves) owns a 16 x 32 pixel output tile and, per
// 16-channel chunk of the input, DMAs the 18 x 34 pixel HALO TILE once into LDS (32 B per pixel and plane, octet slot
// XOR-swizzled by (pixel >> 3) & 1 on the source side so the shifted ds_read_b128 fragment reads stay conflict free)
// together with the chunk's 9 x 16 x 32 weights; the nine taps read their shifted MFMA fragments straight out of the halo
// tile, each input row fragment serving three taps.  Two 58 KiB stages: the DMA of chunk c+1 runs under the MFMAs of chunk c,
// one barrier per chunk.  Each wave: 2 rows x 32 pixels x 32 channels, 3 x v_mfma_f32_32x32x16_bf16 per product.
#include <cstdlib>
#include "kernels.hpp"
#include "split_fmt.hpp"

namespace sd {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// one LDS-DMA (64 lanes x 16 B, lane-linear destination); inline asm so hipcc does not serialise them (see conv_dma.hip)
__device__ __forceinline__ void ddma16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

constexpr int D_TH = 16, D_TW = 32;                      // output tile
constexpr int D_HW = D_TW + 2, D_HH = D_TH + 2;          // halo tile 18 x 34
constexpr int D_XI = 20;                                 // DMA instructions per halo plane: 612 pixels x 2 slots = 1224 units
constexpr int D_XUNITS = D_XI * 64;
constexpr int D_WI = 9;                                  // DMA instructions per weight plane
constexpr int D_WUNITS = 9 * 2 * 32;                     // taps x octets x 32 output channels
constexpr int D_STAGE = 2 * D_XUNITS + 2 * D_WUNITS;     // 3712 units = 58 KiB
constexpr int D_NDMA = 2 * D_XI + 2 * D_WI;              // 58 DMA instructions per stage, 8 waves
constexpr int D_WAVES = 8;


// the chunk descriptor through the scalar cache (a compiler-visible vector load would bring a vmcnt(0) that drains the DMAs)
typedef int i32x8d __attribute__((ext_vector_type(8)));
__device__ __forceinline__ DirectChunk load_chunk(const DirectChunk* ptr) {
    i32x8d v;
    asm volatile("s_load_dwordx8 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(ptr) : "memory");
    DirectChunk e;
    e.base = reinterpret_cast<const void*>(((unsigned long long)(unsigned)v[1] << 32) | (unsigned)v[0]);
    e.H = v[2]; e.W = v[3]; e.C = v[4]; e.up = v[5]; e.nvalid = v[6]; e.pad = v[7];
    return e;
}

// persistent: workgroup b walks tiles b, b + grid, ...; the (tile, chunk) sequence is one software pipeline, so the first
// chunk of the next tile lands while the current tile's epilogue runs.
__global__ __launch_bounds__(512, 1) void conv_direct_kernel(const ConvDirectParams p) {
    __shared__ __attribute__((aligned(16))) u32x4 lds[2 * D_STAGE];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int tiles_x = p.W / D_TW, tiles_y = (p.H + D_TH - 1) / D_TH;
    const int total = tiles_x * tiles_y * p.N;
    const u32x4* const zero = reinterpret_cast<const u32x4*>(p.zero16);
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) void*)lds;      // LDS byte address
    const int frow = lane & 31, fk = lane >> 5;

    struct Tile { int img, ty0, tx0; };
    auto tile_of = [&](int tid) {
        if ((total & 7) == 0) tid = (tid & 7) * (total >> 3) + (tid >> 3);     // neighbouring tiles (shared halos) on one XCD
        Tile r;
        const int bx = tid % tiles_x; tid /= tiles_x;
        r.tx0 = bx * D_TW; r.ty0 = (tid % tiles_y) * D_TH; r.img = tid / tiles_y;
        return r;
    };

    // halo geometry of this lane's five X-DMA slots (instruction j = wave + 8 i, i < 5; i = 5..7 are weight DMAs): constant
    // over tiles and chunks.  packed: ry | rx << 8 | octet << 16 | inside-halo << 17
    int geo[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const int j = wave + D_WAVES * i;
        const int u = (j - (j >= D_XI ? D_XI : 0)) * 64 + lane;
        const int pix = u >> 1;
        const int oct = (u & 1) ^ ((pix >> 3) & 1);
        const int ry = pix / D_HW, rx = pix - ry * D_HW;
        geo[i] = ry | (rx << 8) | (oct << 16) | ((pix < D_HH * D_HW ? 1 : 0) << 17);
    }

    // stage image: Xh[1280] Xl[1280] Wh[576] Wl[576]; X unit = [halo pixel][octet ^ ((pixel >> 3) & 1)]
    auto issue = [&](const Tile& tl, int c, int stage) {
        const DirectChunk ch = load_chunk(p.chunks + c);
        const size_t plane = (size_t)p.Nmax * ch.H * ch.W * ch.C;      // elements
        const unsigned sbyte = lds0 + (unsigned)(stage * D_STAGE * 16);
        const uint16_t* const img_hi = reinterpret_cast<const uint16_t*>(ch.base) + (size_t)tl.img * ch.H * ch.W * ch.C;
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const int j = wave + D_WAVES * i;
            if (p.dbg & 1) continue;
            const int ry = geo[i] & 0xff, rx = (geo[i] >> 8) & 0xff, oct = (geo[i] >> 16) & 1;
            const int gy = tl.ty0 - 1 + ry, gx = tl.tx0 - 1 + rx;
            const bool ok = (geo[i] >> 17) && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W && oct < ch.nvalid;
            const unsigned off = (unsigned)(((gy >> ch.up) * ch.W + (gx >> ch.up)) * ch.C + oct * 8);
            const uint16_t* src = (j >= D_XI ? img_hi + plane : img_hi) + off;
            ddma16(ok ? reinterpret_cast<const u32x4*>(src) : zero, sbyte + (unsigned)(j * 1024));
        }
#pragma unroll
        for (int i = 5; i < (D_NDMA + D_WAVES - 1) / D_WAVES; ++i) {
            const int j = wave + D_WAVES * i;
            if (j < D_NDMA && !(p.dbg & 2)) {
                const int jw = j - 2 * D_XI;
                const int pl = jw >= D_WI ? 1 : 0;
                const u32x4* gw = p.wt + ((size_t)pl * p.nchunks + c) * D_WUNITS + (jw - pl * D_WI) * 64 + lane;
                ddma16(gw, sbyte + (unsigned)((2 * D_XUNITS + jw * 64) * 16));
            }
        }
    };

    // bias of this lane's accumulator rows, once
    f32x4 bias[4];
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4) {
        const int nl = 8 * r4 + 4 * fk;
        bias[r4] = nl < p.Cout ? *reinterpret_cast<const f32x4*>(p.bias + nl) : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    int tid = blockIdx.x;
    if (tid >= total) return;
    Tile cur = tile_of(tid);
    issue(cur, 0, 0);
    int g = 0;                                     // stages consumed so far
    for (; tid < total; tid += gridDim.x) {
        f32x16 acc[2];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
        Tile nxt = cur;
        for (int c = 0; c < p.nchunks; ++c, ++g) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();          // stage g has landed for every wave; everyone is done with stage g-1
            if (c + 1 < p.nchunks) issue(cur, c + 1, (g + 1) & 1);
            else if (tid + (int)gridDim.x < total) { nxt = tile_of(tid + gridDim.x); issue(nxt, 0, (g + 1) & 1); }
            const u32x4* Xh = lds + (g & 1) * D_STAGE;
            const u32x4* Xl = Xh + D_XUNITS;
            const u32x4* Wh = Xl + D_XUNITS;
            const u32x4* Wl = Wh + D_WUNITS;
            if (!(p.dbg & 4))
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                bf16x8 xh[4], xl[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int lp = (2 * wave + r) * D_HW + frow + dx;
                    const int idx = lp * 2 + (fk ^ ((lp >> 3) & 1));
                    xh[r] = __builtin_bit_cast(bf16x8, Xh[idx]);
                    xl[r] = __builtin_bit_cast(bf16x8, Xl[idx]);
                }
#pragma unroll
                for (int dy = 0; dy < 3; ++dy) {
                    const int wi = ((dy * 3 + dx) * 2 + fk) * 32 + frow;
                    const bf16x8 wh = __builtin_bit_cast(bf16x8, Wh[wi]);
                    const bf16x8 wl = __builtin_bit_cast(bf16x8, Wl[wi]);
#pragma unroll
                    for (int pr = 0; pr < 3; ++pr)
#pragma unroll
                        for (int a = 0; a < 2; ++a)
                            acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pr == 0 ? wl : wh, pr == 1 ? xl[a + dy] : xh[a + dy], acc[a], 0, 0, 0);
                }
            }
        }

        // ---- epilogue: bias + activation, split once, LDS transpose (in the stage just consumed; the other one is being
        //      filled for the next tile), 16-byte runs of 8 channels per pixel and plane ----
        __builtin_amdgcn_s_barrier();
        auto epilogue = [&](auto tag) {
            constexpr int ACT = decltype(tag)::value;
            constexpr int ROW = 64 + 16;
            unsigned char* sh = reinterpret_cast<unsigned char*>(lds + ((g - 1) & 1) * D_STAGE) + wave * (2 * 32 * ROW);
            unsigned char* sl = sh + 32 * ROW;
            const int seg = lane & 3, prow = lane >> 2;          // 4 segments of 8 channels, 16 pixels per pass
            uint16_t* const out_hi = reinterpret_cast<uint16_t*>(p.out);
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                const int y = cur.ty0 + 2 * wave + a;
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    if (8 * r4 >= p.Cout) continue;         // rows past Cout are padding, never stored
                    const int nl = 8 * r4 + 4 * fk;
                    f32x4 v = {acc[a][4 * r4], acc[a][4 * r4 + 1], acc[a][4 * r4 + 2], acc[a][4 * r4 + 3]};
                    v += bias[r4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = act_split<ACT>(v[r]);
                    uint2 h, l;
                    split4(v, h, l);
                    *reinterpret_cast<uint2*>(sh + frow * ROW + nl * 2) = h;
                    *reinterpret_cast<uint2*>(sl + frow * ROW + nl * 2) = l;
                }
                __builtin_amdgcn_wave_barrier();
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int ps = 0; ps < 2; ++ps) {
                    const int pix = ps * 16 + prow;
                    const u32x4 h = *reinterpret_cast<const u32x4*>(sh + pix * ROW + seg * 16);
                    const u32x4 l = *reinterpret_cast<const u32x4*>(sl + pix * ROW + seg * 16);
                    if (y < p.H && seg * 8 < p.Cout && !(p.dbg & 8)) {
                        uint16_t* o = out_hi + ((size_t)(cur.img * p.H + y) * p.W + cur.tx0 + pix) * p.Cout + seg * 8;
                        *reinterpret_cast<u32x4*>(o) = h;
                        *reinterpret_cast<u32x4*>(o + p.out_plane) = l;
                    }
                }
                __builtin_amdgcn_wave_barrier();
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
        };
        if (p.act == ACT_RELU) epilogue(ActTag<ACT_RELU>{});
        else if (p.act == ACT_ELU) epilogue(ActTag<ACT_ELU>{});
        else epilogue(ActTag<ACT_NONE>{});
        cur = nxt;
    }
}

hipError_t launch_conv_direct(const ConvDirectParams& p, hipStream_t s) {
    if (p.W % D_TW || p.Cout > 32 || p.Cout % 8) return hipErrorInvalidValue;
    const int tiles = (p.W / D_TW) * ((p.H + D_TH - 1) / D_TH) * p.N;
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return hipErrorInvalidDevice;
        cus = prop.multiProcessorCount;
    }
    ConvDirectParams q = p;
    static const char* dbg = std::getenv("SEMDEPTH_DIRECT_DBG");
    q.dbg = dbg ? atoi(dbg) : 0;
    hipLaunchKernelGGL(conv_direct_kernel, dim3((unsigned)(tiles < cus ? tiles : cus)), dim3(512), 0, s, q);
    return hipGetLastError();
}

}  // namespace sd
