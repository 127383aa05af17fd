// extern "C" surface of libsemdepth.so (include/semdepth.h).  Owns the layer plans and the launch sequences;
// owns no device memory (the caller binds arenas it allocated, e.g. torch tensors).
#include "../../include/semdepth.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "kernels.hpp"
#include "plan.hpp"

using namespace sd;

static_assert(sizeof(RwResultDev) == sizeof(sd_rw_result), "sd_rw_result layout");
static_assert(sizeof(F2fResultDev) == sizeof(sd_f2f_result), "sd_f2f_result layout");

struct sd_handle {
    int device = 0, H = 0, W = 0, max_batch = 0, enc = 0, chunk = 0, cap = 0, prec = 0;
    unsigned sw = 0;      // SEMDEPTH_* switches, latched in sd_create
    unsigned fuse_epoch = 0;   // epoch of the look-back words in the one-pass fuse scratch (0: scratch must be zeroed first)
    int last_mono_frames = 0;  // frames of the last sd_monodepth_forward whose raw pair is still in the activation arena (0: none / chunked)
    NetPlan fcn, mono;
    bool bound = false;
    char* wf = nullptr;   // FCN weight arena
    char* wm = nullptr;   // monodepth weight arena
    char* ws = nullptr;   // workspace arena
    // workspace carve (byte offsets)
    size_t o_fcn = 0, o_mono = 0, o_fuse = 0, o_fuse1 = 0, o_cams = 0, o_bufA = 0, o_bufB = 0, o_rgbA = 0, o_rgbB = 0, o_cnt = 0, o_plane = 0, o_o3d = 0, o_misc = 0, o_rsz = 0, o_cmp = 0;
    std::vector<int> rsz_host;      // tap tables of the last sd_resize_cubic_u8 geometry (kept alive for the async upload)
    int rsz_key[4] = {0, 0, 0, 0};
    size_t ws_bytes = 0;
    int last_fcn_images = 0, last_mono_images = 0;
    int reserve_cus = 0;            // sd_set_reserved_cus: CUs the persistent conv launches leave free
    std::vector<CamDev> cams_stage;
    std::map<std::pair<int, int>, std::vector<float>> bias_host;   // (net, slot) -> host copy, for bias slots that are summed
    std::map<std::pair<int, int>, std::vector<float>> w_host;      // (net, slot) -> f32 tensor of the SD_PREC_F16X2 slots that share a weight scale (sd_load_weight)
    std::string err;
    // profiling (sd_profile): event pairs around conv launches
    bool prof = false;
    struct ProfRec { const char* kernel; double flops; hipEvent_t a, b; const char* op; int M, N, K; double bytes; };
    std::vector<ProfRec> prof_recs;
    // one event per conv launch: the end event of a conv is the start event of the conv launched right behind it (two events per
    // launch cost 1.8 % of a step: every record is a barrier packet on the stream)
    std::vector<hipEvent_t> prof_pool;
    size_t prof_used = 0;
    hipEvent_t prof_last = nullptr;      // end event of the previous conv launch, if nothing else was launched since
};

namespace {

size_t al(size_t v) { return (v + 255) / 256 * 256; }
constexpr int RSZ_MAX = 16384;     // largest destination extent of sd_resize_cubic_u8
constexpr size_t SAT_OFF = 512;     // the fp16 saturation counter inside the o_misc scratch (zero16 lives at +256)

sd_status fail(sd_handle* h, sd_status code, const std::string& msg) {
    if (h) h->err = msg;
    return code;
}
#define HIPCHK(h, expr)                                                                              \
    do {                                                                                             \
        hipError_t e_ = (expr);                                                                      \
        if (e_ != hipSuccess) return fail(h, SD_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

NetPlan& plan_of(sd_handle* h, sd_net net) { return net == SD_NET_FCN8S ? h->fcn : h->mono; }
const NetPlan& plan_of(const sd_handle* h, sd_net net) { return net == SD_NET_FCN8S ? h->fcn : h->mono; }
char* warena(sd_handle* h, sd_net net) { return net == SD_NET_FCN8S ? h->wf : h->wm; }

void carve_workspace(sd_handle* h) {
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t r = off; off += al(bytes); return r; };
    const size_t B = (size_t)h->max_batch, cap = (size_t)h->cap;
    h->o_fcn = take(h->fcn.act_bytes);
    h->o_mono = take(h->mono.act_bytes);
    h->o_fuse = take(fuse_scratch_bytes(h->max_batch, h->H, h->W));
    h->o_fuse1 = take(fuse_onepass_scratch_bytes(h->max_batch, h->H, h->W));
    h->o_cams = take(sizeof(CamDev) * B);
    h->o_bufA = take(B * cap * 3 * sizeof(float));
    h->o_bufB = take(B * cap * 3 * sizeof(float));
    h->o_rgbA = take(B * cap * 3);            // colours travel with the points through every filter (semantic_depth.py:206-245)
    h->o_rgbB = take(B * cap * 3);
    h->o_cnt = take(B * sizeof(int32_t) * 24);
    h->o_plane = take(B * sizeof(double) * 4);
    h->o_o3d = take(o3d_scratch_bytes(h->max_batch, h->cap));
    h->o_misc = take(4096 + al(B * 7 * sizeof(int32_t)) + al(B * 12 * sizeof(double)));   // scalars | f2f counts | f2f planes
    h->o_cmp = take(cmp_scratch_bytes(h->max_batch));
    h->o_rsz = take((size_t)RSZ_MAX * 16 * sizeof(int));          // resize tap tables: [RSZ_MAX][4] x idx | x weight | y idx | y weight
    h->ws_bytes = off;
}

sd_status upload_tables(sd_handle* h, sd_net net) {
    NetPlan& p = plan_of(h, net);
    char* wbase = warena(h, net);
    const char* abase = h->ws + (net == SD_NET_FCN8S ? h->o_fcn : h->o_mono);
    for (const OpDesc& op : p.ops) {
        if (op.kind == OP_CONV_DIRECT) {
            std::vector<DirectChunk> chunks;
            build_direct_chunks(p, op, abase, chunks);
            HIPCHK(h, hipMemcpy(wbase + op.tab_offset, chunks.data(), chunks.size() * sizeof(DirectChunk), hipMemcpyHostToDevice));
            continue;
        }
        if (op.kind != OP_CONV) continue;
        std::vector<KEntry> ktab;
        build_conv_tables(p, op, abase, wbase + op.tab_offset, ktab);
        HIPCHK(h, hipMemcpy(wbase + op.tab_offset, ktab.data(), ktab.size() * sizeof(KEntry), hipMemcpyHostToDevice));
    }
    return SD_OK;
}

struct HeadOut { float* logits; uint8_t* road; uint8_t* fence; uint8_t* argmax; };

// next event of the handle's profiling pool (created on demand), nullptr on failure
static hipEvent_t prof_event(sd_handle* h) {
    if (h->prof_used == h->prof_pool.size()) {
        hipEvent_t a;
        if (hipEventCreate(&a) != hipSuccess) return nullptr;
        h->prof_pool.push_back(a);
    }
    return h->prof_pool[h->prof_used++];
}

// one chunk through a plan.  frames: u8 [nframes,H,W,3] (device)
sd_status run_plan(sd_handle* h, sd_net net, const uint8_t* frames, int nframes, const HeadOut* head, hipStream_t s) {
    NetPlan& p = plan_of(h, net);
    char* wbase = warena(h, net);
    char* abase = h->ws + (net == SD_NET_FCN8S ? h->o_fcn : h->o_mono);
    const int N = nframes * (p.images / p.frames);
    auto T = [&](int t) -> float* { return reinterpret_cast<float*>(abase + p.tensors[t].offset); };
    auto Wp = [&](int w) -> const float* { return reinterpret_cast<const float*>(wbase + p.weights[w].offset); };
    auto PL = [&](int t) -> size_t { const TensorDesc& d = p.tensors[t]; return (size_t)p.images * d.H * d.W * d.C; };   // lo-plane offset
    // algorithmic HBM bytes of a conv op for N of the plan's images: sources + output once, + the weights (sd_profile_bucket::bytes)
    auto op_bytes = [&](const OpDesc& op) -> double {
        double b = 0;
        for (int j = 0; j < op.nsrc; ++j) b += (double)p.tensors[op.src[j]].bytes;
        if (op.dst >= 0) b += (double)p.tensors[op.dst].bytes;
        b = b * N / p.images;
        if (op.w >= 0) b += (double)p.weights[op.w].bytes;
        return b;
    };
    // 0 f32, 1 split bf16, 2 ONE fp16 plane, 4 bf16 x 3, 5 fp16 hi + scaled lo (fp16 hi + lo tensors never reach the ops that take this)
    auto FMT = [&](int t) -> int { return p.tensors[t].fmt ? (p.tensors[t].x3 ? 4 : p.tensors[t].f16 == 3 ? 5 : p.tensors[t].f16 ? 2 : 1) : 0; };
    for (const WeightSlot& wsl : p.weights)
        if (!wsl.loaded) return fail(h, SD_ERR_STATE, "weight not loaded: " + wsl.name);
    h->prof_last = nullptr;                 // (other work may have been put on the stream since the previous call)
    for (const OpDesc& op : p.ops) {
        hipError_t e = hipSuccess;
        bool conv_op = false;
        switch (op.kind) {
            case OP_PRE_VGG:
                e = launch_pre_vgg(frames, T(op.dst), (long)nframes * h->H * h->W, FMT(op.dst), PL(op.dst), s);
                break;
            case OP_PRE_MONO:
                e = launch_pre_mono(frames, T(op.dst), nframes, h->H, h->W, FMT(op.dst), PL(op.dst), p.input_scale == 1.f ? 1 : 0, s);
                break;
            case OP_CONV: {
                const TensorDesc& d = p.tensors[op.dst];
                const TensorDesc& s0 = p.tensors[op.src[0]];
                ConvParams c{};
                c.nsrc = op.nsrc; c.Ctot = op.Ctot;
                c.N = N; c.Hin = s0.H * (op.up[0] ? 2 : 1); c.Win = s0.W * (op.up[0] ? 2 : 1);
                c.pool = op.fuse_pool;
                c.Hout = d.H << c.pool; c.Wout = d.W << c.pool; c.Cout = d.C; c.CoutPad = p.weights[op.w].CoutPad;
                c.kh = c.kw = op.k; c.stride = op.sstride[0]; c.pad = op.pad;
                c.K = op.K; c.Kpad = op.Kpad;
                c.wt = Wp(op.w); c.bias = Wp(op.b);
                c.ktab = reinterpret_cast<const KEntry*>(wbase + op.tab_offset);
                c.vec = op.vec; c.vtiles = op.Kvec / 32;
                c.simple = op.nsrc == 1 && op.vec && op.Kvec == op.Kpad && op.sstride[0] == 1 && !op.up[0] && op.k * op.k <= 64 &&
                           op.Kpad == op.k * op.k * s0.C;
                if (!c.simple && op.nsrc == 2 && op.k == 1 && op.vec && op.Kvec == op.Kpad && !op.up[0] && !op.up[1] && op.pad == 0 &&
                    s0.C % 32 == 0 && p.tensors[op.src[1]].C % 32 == 0 && op.Kpad == s0.C + p.tensors[op.src[1]].C)
                    c.simple = 2;
                c.residual = op.residual >= 0 ? T(op.residual) : nullptr;
                c.out = T(op.dst);
                c.act = op.act; c.m_fastest = op.m_fastest;
                c.out_plane = PL(op.dst); c.Nmax = p.images;
                c.zero16 = h->ws + h->o_misc + 256;        // the arena is zero-filled and nothing writes here
                const bool split = p.prec != 0;
                c.f16 = op.f16; c.out_f16 = d.f16; c.alpha = op.f16 == 4 ? 1.f / p.weights[op.w].wscale : 1.f;
                c.src0 = T(op.src[0]); c.src0_plane = PL(op.src[0]);
                c.out_planar16 = d.planar16;
                c.sw = h->sw; c.reserve_cus = h->reserve_cus;
                c.sat = reinterpret_cast<unsigned long long*>(h->ws + h->o_misc + SAT_OFF);
                c.x3 = p.x3;
                // conv_dma3.hip, k x k stride-1 layers on one source (fc6): one output row of 256 / Wout images per tile, taps on padding rows skipped
                const bool ph3 = p.x3 || p.h2;          // the phased 256 x 256 GEMM block of conv_dma3.hip exists for these two engines
                if (ph3 && !op.fold && op.nsrc == 1 && op.vec && op.Kvec == op.Kpad && op.k >= 3 && op.sstride[0] == 1 && !op.up[0] && !c.pool &&
                    op.Kpad == op.k * op.k * s0.C && c.Wout > 0 && 256 % c.Wout == 0 && N % (256 / c.Wout) == 0 && !(h->sw & SW_NO_ROWSKIP))
                    c.rowgrp = 256 / c.Wout;
                c.flat = (ph3 && op.k == 1 && !op.fold && !op.up[0] && !(op.nsrc > 1 && op.up[1]) && op.nsrc <= 2 && !(h->sw & SW_NO_FLAT)) ? 1 : 0;
                c.noup = (ph3 && op.nsrc == 1 && !(h->sw & SW_NO_FLAT)) ? 1 : 0;       // (conv_dma3's precomputed gather: ONE source geometry)
                for (int j = 0; j < op.nsrc; ++j) {      // (32-bit byte offsets inside a plane of every source; strides 1 or 2)
                    if (PL(op.src[j]) * 2 >= ((size_t)1 << 32)) c.flat = c.noup = 0;
                    // (a folded op keeps up[0] = 1 from the plan, but its table reads the source at its own resolution: tap layers without upsample)
                    if ((op.up[j] && !op.fold) || op.sstride[j] < 1 || op.sstride[j] > 2) c.noup = 0;
                }
                if (op.fold) {          // (conv_dma3.hip: the GEMM's pixel space is the source itself)
                    c.fold = 1; c.simple = 0; c.Hin = s0.H; c.Win = s0.W; c.Hout = s0.H; c.Wout = s0.W; c.kh = c.kw = 2;
                }
                hipEvent_t ea = nullptr, eb = nullptr;
                if (h->prof) {
                    ea = h->prof_last;
                    if (!ea) { if (!(ea = prof_event(h))) return fail(h, SD_ERR_HIP, "hipEventCreate"); hipEventRecord(ea, s); }
                    if (!(eb = prof_event(h))) return fail(h, SD_ERR_HIP, "hipEventCreate");
                }
                const bool dma3 = split && conv_dma3_eligible(c) && !(h->sw & SW_NO_DMA);
                const bool dma = !dma3 && split && conv_dma_variant(c) != 0 && !(h->sw & SW_NO_DMA);
                const bool stem = split && !dma && !dma3 && conv_stem_eligible(c);
                if (c.out_planar16 && !stem && !dma) return fail(h, SD_ERR_STATE, "sub-planar output needs the LDS-DMA or the stem conv kernel");
                if (c.pool && !dma) return fail(h, SD_ERR_STATE, "fused pool needs the LDS-DMA conv kernel");
                if (c.fold && !dma3 && !(dma && !c.x3)) return fail(h, SD_ERR_STATE, "an upsample-folded conv needs the conv_dma3 kernel (bf16 x 3) or a two-plane form of conv_dma");
                e = dma3 ? launch_conv_dma3(c, s) : dma ? launch_conv_dma(c, s) : stem ? launch_conv_stem(c, s) : split ? launch_conv_split(c, s) : launch_conv_igemm(c, s);
                if (h->prof) {
                    hipEventRecord(eb, s);
                    h->prof_last = eb; conv_op = true;
                    static const char* const dma3_names[3] = {"conv_dma3_kernel<0>", "conv_dma3_kernel<1>", "conv_dma3_kernel<2>"};
                    static const char* const dmah_names[3] = {"conv_dma_hs_phased_kernel<0>", "conv_dma_hs_phased_kernel<1>", "conv_dma_hs_phased_kernel<2>"};
                    // (one bucket for the bench line; the per-layer listing of SEMDEPTH_PROFILE_VERBOSE names the gather variant)
                    const char* const dma3_name = (h->sw & SW_PROFILE_VERBOSE) ? (c.f16 == 4 ? dmah_names : dma3_names)[conv_dma3_mode(c)]
                                                                               : (c.f16 == 4 ? "conv_dma_hs_phased_kernel" : "conv_dma3_kernel");
                    h->prof_recs.push_back({dma3 ? dma3_name : dma ? conv_dma_kernel_name(c) : stem ? (c.x3 ? "conv_stem_x3_kernel" : c.f16 == 4 ? "conv_stem_hs_kernel" : c.f16 ? "conv_stem_f16w_kernel" : "conv_stem_kernel") : split ? conv_split_kernel_name(c) : conv_igemm_kernel_name(c), op.flops * N / p.images, ea, eb,
                                            op.name.c_str(), N * c.Hout * c.Wout * (c.fold ? 4 : 1), d.C, op.K, op_bytes(op)});
                }
                break;
            }
            case OP_CONV_DIRECT: {
                const TensorDesc& d = p.tensors[op.dst];
                ConvDirectParams c{};
                c.chunks = reinterpret_cast<const DirectChunk*>(wbase + op.tab_offset);
                c.nchunks = op.nchunks;
                c.pool = op.fuse_pool;
                c.N = N; c.H = d.H << c.pool; c.W = d.W << c.pool;
                c.nsplit = op.nsplit; c.Cout = d.C / op.nsplit; c.Cstride = d.C; c.out_planar16 = d.planar16;
                c.nreal = (d.Ctf > 0 && d.Ctf < d.C) ? d.Ctf : 0;
                c.all_up = 1;
                for (int i = 0; i < op.nsrc; ++i) c.all_up = c.all_up && op.up[i] == 1;
                c.fold = op.fold;
                c.wt = reinterpret_cast<const u32x4_t*>(Wp(op.w)); c.bias = Wp(op.b);
                c.out = T(op.dst); c.out_plane = PL(op.dst); c.act = op.act; c.Nmax = p.images;
                c.zero16 = h->ws + h->o_misc + 256;
                c.rows_per_wave = 2;
                c.f16 = op.f16; c.out_f16 = d.f16; c.alpha = op.f16 == 4 ? 1.f / p.weights[op.w].wscale : 1.f;
                c.sw = h->sw; c.reserve_cus = h->reserve_cus;
                c.sat = reinterpret_cast<unsigned long long*>(h->ws + h->o_misc + SAT_OFF);
                hipEvent_t ea = nullptr, eb = nullptr;
                if (h->prof) {
                    ea = h->prof_last;
                    if (!ea) { if (!(ea = prof_event(h))) return fail(h, SD_ERR_HIP, "hipEventCreate"); hipEventRecord(ea, s); }
                    if (!(eb = prof_event(h))) return fail(h, SD_ERR_HIP, "hipEventCreate");
                }
                e = p.x3 ? launch_conv_direct3(c, s) : launch_conv_direct(c, s);
                if (h->prof) {
                    hipEventRecord(eb, s);
                    h->prof_last = eb; conv_op = true;
                    h->prof_recs.push_back({p.x3 ? conv_direct3_kernel_name(c) : conv_direct_kernel_name(c), op.flops * N / p.images, ea, eb, op.name.c_str(), N * c.H * c.W, d.C, op.K, op_bytes(op)});
                }
                break;
            }
            case OP_DEC_TAIL1: {
                const TensorDesc& d = p.tensors[op.dst];
                DecTailParams c{};
                c.a = T(op.src[0]); c.a_plane = PL(op.src[0]); c.d2 = T(op.src[1]); c.d_plane = PL(op.src[1]);
                c.N = N; c.H = d.H; c.W = d.W;
                c.w1 = reinterpret_cast<const u32x4_t*>(Wp(op.w)); c.b1 = Wp(op.b);
                c.w2 = reinterpret_cast<const u32x4_t*>(Wp(op.w2)); c.b2 = Wp(op.b2);
                c.wd = Wp(op.w3); c.bd = Wp(op.b3);
                c.out = T(op.dst); c.sw = h->sw; c.reserve_cus = h->reserve_cus;
                c.hs = p.h2; c.alpha = p.h2 ? 1.f / p.weights[op.w].wscale : 1.f; c.alpha2 = p.h2 ? 1.f / p.weights[op.w2].wscale : 1.f;
                hipEvent_t ea = nullptr, eb = nullptr;
                if (h->prof) {
                    ea = h->prof_last;
                    if (!ea) { if (!(ea = prof_event(h))) return fail(h, SD_ERR_HIP, "hipEventCreate"); hipEventRecord(ea, s); }
                    if (!(eb = prof_event(h))) return fail(h, SD_ERR_HIP, "hipEventCreate");
                }
                e = launch_dec_tail1(c, s);
                if (h->prof) {
                    hipEventRecord(eb, s);
                    h->prof_last = eb; conv_op = true;
                    h->prof_recs.push_back({p.h2 ? "dec_tail1_hs_kernel" : "dec_tail1_x3_kernel", op.flops * N / p.images, ea, eb, op.name.c_str(), N * d.H * d.W, 16, op.K, op_bytes(op)});
                }
                break;
            }
            case OP_SMALLN: {
                const TensorDesc& s0 = p.tensors[op.src[0]];
                SmallNParams c{};
                c.x = T(op.src[0]); c.N = N; c.H = s0.H; c.W = s0.W; c.C = s0.C; c.k = op.k; c.nout = op.nout;
                c.wt = Wp(op.w); c.bias = Wp(op.b); c.out = T(op.dst); c.act = op.act;
                c.in_split = FMT(op.src[0]) != 0; c.out_split = FMT(op.dst) != 0; c.in_plane = PL(op.src[0]); c.out_plane = PL(op.dst);
                c.f16 = s0.f16; c.out_f16 = p.tensors[op.dst].f16; c.x3 = p.x3;
                c.in_sub = s0.planar16 ? (size_t)p.images * s0.H * s0.W * 16 : 0;
                c.out_c = p.tensors[op.dst].C;
                c.zero16 = h->ws + h->o_misc + 256;
                c.sw = h->sw;
                e = launch_conv_smalln(c, s);
                break;
            }
            case OP_POOL2: {
                const TensorDesc& s0 = p.tensors[op.src[0]];
                e = launch_maxpool2(T(op.src[0]), T(op.dst), N, s0.H, s0.W, s0.C, FMT(op.src[0]), PL(op.src[0]), PL(op.dst), s);
                break;
            }
            case OP_POOL3Z: {
                const TensorDesc& s0 = p.tensors[op.src[0]];
                e = launch_maxpool3z(T(op.src[0]), T(op.dst), N, s0.H, s0.W, s0.C, FMT(op.src[0]), PL(op.src[0]), PL(op.dst), s0.planar16 ? p.images : 0, s);
                break;
            }
            case OP_DECONV4_ADD: {
                const TensorDesc& s0 = p.tensors[op.src[0]];
                e = launch_deconv4s2_add(T(op.src[0]), Wp(op.w), Wp(op.b), T(op.residual), T(op.dst), N, s0.H, s0.W, s);
                break;
            }
            case OP_HEAD16: {
                const TensorDesc& s0 = p.tensors[op.src[0]];
                e = launch_deconv16s8_head(T(op.src[0]), Wp(op.w), Wp(op.b), N, s0.H, s0.W, head->logits, head->road, head->fence,
                                           head->argmax, s);
                break;
            }
        }
        if (!conv_op) h->prof_last = nullptr;       // something else went onto the stream: the next conv records its own start
        if (e != hipSuccess) return fail(h, SD_ERR_HIP, "launch " + op.name + ": " + hipGetErrorString(e));
    }
    (net == SD_NET_FCN8S ? h->last_fcn_images : h->last_mono_images) = N;
    return SD_OK;
}

CamDev make_cam(const sd_camera& c) {
    // np.float32([[1,0,0,-cx],[0,-1,0,cy],[0,0,0,-f],[0,0,1/b,0]]) then promoted to double by OpenCV
    CamDev d;
    const float q[16] = {1.f, 0.f, 0.f, (float)(-c.cx), 0.f, -1.f, 0.f, (float)c.cy, 0.f, 0.f, 0.f, (float)(-c.f), 0.f, 0.f, (float)(1.0 / c.b), 0.f};
    for (int i = 0; i < 16; ++i) d.q[i] = (double)q[i];
    d.mult = (float)c.disp_mult;
    return d;
}

int32_t* cnt_slot(sd_handle* h, int i) { return reinterpret_cast<int32_t*>(h->ws + h->o_cnt) + (size_t)i * h->max_batch; }

}  // namespace

extern "C" {

#ifndef SD_DEFAULT_PLAN_FCN
#define SD_DEFAULT_PLAN_FCN "conv1_2,conv3_3:x,conv4_1:x,conv4_2:1,conv4_3:1,conv5_1:1,conv5_2:1,conv5_3:1,fc6:1,fc7:1"
#endif
#ifndef SD_DEFAULT_PLAN_MONO
#define SD_DEFAULT_PLAN_MONO "enc/conv1,enc/res*:1,dec/upconv*:1,dec/iconv*:1,dec/disp4,dec/disp3,dec/disp1"
#endif
#ifndef SD_SOURCE_HASH
#define SD_SOURCE_HASH "unhashed"
#endif
// "... src=<hash>": sha256 prefix of the sources this binary was built from (semantic_depth_amd/build.py source_hash);
// the Python loader refuses a library whose hash differs from the tree's
const char* sd_version(void) { return "semdepth 0.4 (gfx950; f32 MFMA, split-bf16 x2 / x3 and fp16 MFMA) src=" SD_SOURCE_HASH; }

const char* sd_status_string(sd_status s) {
    switch (s) {
        case SD_OK: return "ok";
        case SD_ERR_INVALID: return "invalid argument";
        case SD_ERR_HIP: return "HIP error";
        case SD_ERR_STATE: return "invalid state";
        case SD_ERR_NOTFOUND: return "not found";
        default: return "unknown";
    }
}

// the default precision plan (SD_PREC_PLAN): the output of scripts/calibrate_precision.py on the MI355X (per-group error of the
// 2-product form against the exact-f32 engine, greedy by error per saved product under 3e-4 per network on 4 frames of 512 x 1024;
// profiles/r02_precision_calibration.json; DESIGN.md §3 "Precision plan" has the table).  FCN-8s: 69 % of its FLOPs on two products
// at 2.9e-4 (left on three: conv1_1, conv2_1, conv3_x, conv4_1 -- the layers whose INPUT tensor also feeds a score head, or whose
// own error is ~2e-4), monodepth-resnet50: 96 % at 2.7e-4 (left on three: the stem and decoder levels 3 and 2).
static const char* const kDefaultPlanFcn = SD_DEFAULT_PLAN_FCN;
static const char* const kDefaultPlanMono = SD_DEFAULT_PLAN_MONO;

static sd_status create_impl(sd_handle** out, int device, int H, int W, int max_batch, sd_encoder enc, sd_precision prec,
                             const char* fcn_f16, const char* mono_f16);

sd_status sd_create(sd_handle** out, int device, int H, int W, int max_batch, sd_encoder enc, sd_precision prec) {
    if (prec != SD_PREC_F32 && prec != SD_PREC_BF16X2 && prec != SD_PREC_MIXED && prec != SD_PREC_PLAN && prec != SD_PREC_BF16X3 && prec != SD_PREC_F16X2)
        return SD_ERR_INVALID;
    const char* fcn = prec == SD_PREC_PLAN ? kDefaultPlanFcn : "";
    // (the calibrated monodepth plan names ResNet-50 layers; the vgg encoder -- the ill-conditioned one of the two in the tests --
    //  stays on three products under SD_PREC_PLAN)
    const char* mono = prec == SD_PREC_PLAN ? (enc == SD_ENC_RESNET50 ? kDefaultPlanMono : "") : (prec == SD_PREC_MIXED ? "*" : "");
    return create_impl(out, device, H, W, max_batch, enc, prec, fcn, mono);
}

sd_status sd_create_with_plan(sd_handle** out, int device, int H, int W, int max_batch, sd_encoder enc, const char* fcn_f16_layers,
                              const char* mono_f16_layers) {
    return create_impl(out, device, H, W, max_batch, enc, SD_PREC_PLAN, fcn_f16_layers ? fcn_f16_layers : "", mono_f16_layers ? mono_f16_layers : "");
}

const char* sd_default_plan(sd_net net) { return net == SD_NET_FCN8S ? kDefaultPlanFcn : kDefaultPlanMono; }

sd_status sd_precision_plan(const sd_handle* h, sd_net net, char* layers_out, size_t cap, double* flop_share_out) {
    if (!h) return SD_ERR_INVALID;
    const NetPlan& p = plan_of(h, net);
    if (layers_out && cap) { std::strncpy(layers_out, p.f16_ops.c_str(), cap - 1); layers_out[cap - 1] = 0; }
    if (flop_share_out) *flop_share_out = p.flops_per_image > 0 ? p.flops_f16 / p.flops_per_image : 0.0;
    return p.f16_ops.size() + 1 > cap && layers_out ? SD_ERR_INVALID : SD_OK;
}

static sd_status create_impl(sd_handle** out, int device, int H, int W, int max_batch, sd_encoder enc, sd_precision prec,
                             const char* fcn_f16, const char* mono_f16) {
    if (!out || H <= 0 || W <= 0 || max_batch <= 0) return SD_ERR_INVALID;
    {
        const std::string bad = sd_disable_unknown();
        if (!bad.empty()) {
            std::fprintf(stderr, "sd_create: SEMDEPTH_DISABLE names no switch '%s' (dma dma3 direct stem fold tail1 pool_fuse planar n16 fuse1 fuse4 flat rowskip dma_big mfma16)\n", bad.c_str());
            return SD_ERR_INVALID;
        }
    }
    sd_handle* h = new sd_handle();
    h->sw = latch_switches();
    if (const char* e = std::getenv("SEMDEPTH_RESERVE_CUS")) h->reserve_cus = std::min(128, std::max(0, atoi(e)));
    h->device = device; h->H = H; h->W = W; h->max_batch = max_batch; h->enc = (int)enc; h->cap = H * W; h->prec = (int)prec;
    int chunk = 32;      // frames per network pass: the deep layers (M = 512 px per frame) need ~32 frames to fill 256 CUs;
                         // activations of a 32-frame chunk are ~17 GB at 512x1024, nothing on a 288 GB part
    if (const char* e = std::getenv("SEMDEPTH_CHUNK")) chunk = std::max(1, atoi(e));
    h->chunk = std::min(max_batch, chunk);
    try {
        // exact f32 MFMA | split, three bf16 planes | split, fp16 hi + scaled lo planes | split (bf16 x 2 + the fp16 forms of the plan)
        const int eng = h->prec == SD_PREC_F32 ? 0 : h->prec == SD_PREC_BF16X3 ? 2 : h->prec == SD_PREC_F16X2 ? 3 : 1;
        h->fcn = build_fcn8s(h->chunk, H, W, eng, fcn_f16);
        h->mono = build_monodepth(enc == SD_ENC_VGG ? 0 : 1, h->chunk, H, W, eng, mono_f16);
    } catch (const std::exception& ex) {
        std::fprintf(stderr, "sd_create: %s\n", ex.what());
        delete h;
        return SD_ERR_INVALID;
    }
    carve_workspace(h);
    *out = h;
    return SD_OK;
}

sd_status sd_destroy(sd_handle* h) {
    if (h)
        for (hipEvent_t e : h->prof_pool) hipEventDestroy(e);      // (the library owns no device memory; the profiling events are its only HIP objects)
    delete h;
    return SD_OK;
}

const char* sd_last_error(const sd_handle* h) { return h ? h->err.c_str() : "null handle"; }

sd_status sd_query_memory(const sd_handle* h, size_t* fcn_w, size_t* mono_w, size_t* ws) {
    if (!h) return SD_ERR_INVALID;
    if (fcn_w) *fcn_w = h->fcn.weight_bytes;
    if (mono_w) *mono_w = h->mono.weight_bytes;
    if (ws) *ws = h->ws_bytes;
    return SD_OK;
}

sd_status sd_bind_memory(sd_handle* h, void* wf, void* wm, void* ws) {
    if (!h || !wf || !wm || !ws) return SD_ERR_INVALID;
    HIPCHK(h, hipSetDevice(h->device));
    h->wf = (char*)wf; h->wm = (char*)wm; h->ws = (char*)ws;
    h->bound = true;
    HIPCHK(h, hipMemset(h->ws + h->o_misc, 0, 4096));      // zero page (+256), saturation counter (+512), the scalar slot (+0)
    sd_status st = upload_tables(h, SD_NET_FCN8S);
    if (st != SD_OK) return st;
    return upload_tables(h, SD_NET_MONODEPTH);
}

int sd_weight_count(const sd_handle* h, sd_net net) { return h ? (int)plan_of(h, net).weights.size() : 0; }

sd_status sd_weight_info(const sd_handle* h, sd_net net, int index, char* name_out, int64_t* shape_out, int* rank_out) {
    if (!h) return SD_ERR_INVALID;
    const NetPlan& p = plan_of(h, net);
    if (index < 0 || index >= (int)p.weights.size()) return SD_ERR_INVALID;
    const WeightSlot& s = p.weights[index];
    if (name_out) { std::strncpy(name_out, s.name.c_str(), 63); name_out[63] = 0; }
    if (shape_out) for (int i = 0; i < 4; ++i) shape_out[i] = s.shape[i];
    if (rank_out) *rank_out = s.rank;
    return SD_OK;
}

sd_status sd_load_weight(sd_handle* h, sd_net net, const char* name, const float* data, const int64_t* shape, int rank) {
    if (!h || !name || !data || !shape) return SD_ERR_INVALID;
    if (!h->bound) return fail(h, SD_ERR_STATE, "sd_bind_memory first");
    NetPlan& p = plan_of(h, net);
    auto it = p.weight_by_name.find(name);
    if (it == p.weight_by_name.end()) return fail(h, SD_ERR_NOTFOUND, std::string("unknown weight ") + name);
    WeightSlot& s = p.weights[it->second];
    if (rank != s.rank) return fail(h, SD_ERR_INVALID, std::string("rank mismatch for ") + name);
    for (int i = 0; i < rank; ++i)
        if (shape[i] != s.shape[i]) return fail(h, SD_ERR_INVALID, std::string("shape mismatch for ") + name);
    std::vector<float> scaled;
    size_t nel = 1;
    for (int i = 0; i < rank; ++i) nel *= (size_t)shape[i];
    if (s.scale != 1.f) {               // (monodepth stem with integer input: the weights carry the 1/255)
        scaled.resize(nel);
        for (size_t i = 0; i < nel; ++i) scaled[i] = data[i] * s.scale;
        data = scaled.data();
    }
    const int idx = it->second;
    // re-layout + upload of one slot from its TensorFlow-layout tensor
    auto upload = [&](int j, const float* w) -> sd_status {
        WeightSlot& t = p.weights[j];
        std::vector<float> buf;
        relayout_weight(t, w, buf);
        char* base = warena(h, net) + t.offset;
        if (t.layout == WL_IGEMM) {                     // rows [k_off, k_off+Kpad) of a [Ktotal/4][CoutPad][4] matrix
            HIPCHK(h, hipMemcpy(base + (size_t)t.k_off * t.CoutPad * 4, buf.data(), t.bytes, hipMemcpyHostToDevice));
        } else if (t.layout == WL_IGEMM_SPLIT) {        // the same rows in the hi plane and in the lo plane
            const size_t plane = (size_t)t.Ktotal * t.CoutPad * 2, rows = (size_t)t.Kpad * t.CoutPad * 2, ro = (size_t)t.k_off * t.CoutPad * 2;
            HIPCHK(h, hipMemcpy(base + ro, buf.data(), rows, hipMemcpyHostToDevice));
            HIPCHK(h, hipMemcpy(base + plane + ro, reinterpret_cast<char*>(buf.data()) + rows, rows, hipMemcpyHostToDevice));
            if (t.x3) HIPCHK(h, hipMemcpy(base + 2 * plane + ro, reinterpret_cast<char*>(buf.data()) + 2 * rows, rows, hipMemcpyHostToDevice));
        } else {
            const int root = t.owner >= 0 ? t.owner : j;
            bool grouped = t.owner >= 0;
            for (const WeightSlot& o : p.weights) grouped = grouped || (o.owner == root && o.layout == WL_RAW && &o != &t);
            if (grouped && t.layout == WL_RAW && t.rank == 1) {      // summed bias group (ResNet conv3 + projection)
                h->bias_host[{(int)net, j}] = buf;
                std::vector<float> sum(buf.size(), 0.f);
                for (int q = 0; q < (int)p.weights.size(); ++q)
                    if (q == root || p.weights[q].owner == root) {
                        auto f = h->bias_host.find({(int)net, q});
                        if (f != h->bias_host.end()) for (size_t i = 0; i < sum.size(); ++i) sum[i] += f->second[i];
                    }
                HIPCHK(h, hipMemcpy(base, sum.data(), t.bytes, hipMemcpyHostToDevice));
            } else {
                HIPCHK(h, hipMemcpy(base, buf.data(), t.bytes, hipMemcpyHostToDevice));
            }
        }
        return SD_OK;
    };
    if (s.hs) {
        // SD_PREC_F16X2: the planes hold w' = w * 2^k in fp16 (hi + lo), k chosen HERE per layer so that the largest stored |w'| lies in
        // [2^12, 2^13) -- any finite f32 weight tensor loads (round 5 used a fixed 2^12 and refused |w| >= 16), and a layer of tiny weights
        // keeps the low plane's bits.  A stored value of an upsample-folded layer is a sum of up to four taps: bounded by 4 max |w|.  Slots
        // that feed ONE accumulator (ResNet conv3 + projection shortcut) share the scale: the members' tensors are kept on the host and the
        // ones already loaded are laid out again when a later member moves the group's exponent.
        const int root = s.owner >= 0 ? s.owner : idx;
        std::vector<int> members;
        for (int j = 0; j < (int)p.weights.size(); ++j)
            if ((j == root || p.weights[j].owner == root) && p.weights[j].hs) members.push_back(j);
        const bool grouped = members.size() > 1;
        for (size_t i = 0; i < nel; ++i)
            if (!std::isfinite(data[i])) return fail(h, SD_ERR_INVALID, std::string("weight ") + name + ": non-finite value");
        if (grouped) h->w_host[{(int)net, idx}].assign(data, data + nel);
        float mx = 0.f;
        auto amax = [&](const float* w, size_t n, const WeightSlot& t) {
            float m = 0.f;
            for (size_t i = 0; i < n; ++i) m = std::max(m, std::fabs(w[i]));
            return m * ((t.fold || t.layout == WL_TAIL_UP) ? 4.f : 1.f);
        };
        mx = amax(data, nel, s);
        if (grouped)
            for (int j : members) {
                auto f = h->w_host.find({(int)net, j});
                if (j != idx && f != h->w_host.end()) mx = std::max(mx, amax(f->second.data(), f->second.size(), p.weights[j]));
            }
        int k = 12;
        if (mx > 0.f) k = std::min(100, std::max(-100, 12 - std::ilogb(mx)));
        const float wscale = std::ldexp(1.f, k);
        const bool moved = wscale != p.weights[root].wscale;
        for (int j : members) p.weights[j].wscale = wscale;
        if (grouped && moved)
            for (int j : members) {
                auto f = h->w_host.find({(int)net, j});
                if (j != idx && p.weights[j].loaded && f != h->w_host.end()) {
                    sd_status st = upload(j, f->second.data());
                    if (st != SD_OK) return st;
                }
            }
    }
    {
        sd_status st = upload(idx, data);
        if (st != SD_OK) return st;
    }
    s.loaded = true;
    return SD_OK;
}

sd_status sd_fcn8s_forward(sd_handle* h, const uint8_t* frames, int B, float* logits, uint8_t* road, uint8_t* fence,
                           uint8_t* argmax, void* stream) {
    if (!h || !frames || B <= 0 || B > h->max_batch) return fail(h, SD_ERR_INVALID, "sd_fcn8s_forward: bad arguments");
    if (!h->bound) return fail(h, SD_ERR_STATE, "sd_bind_memory first");
    const size_t npix = (size_t)h->H * h->W;
    for (int b0 = 0; b0 < B; b0 += h->chunk) {
        const int nb = std::min(h->chunk, B - b0);
        HeadOut ho{logits ? logits + (size_t)b0 * npix * 3 : nullptr, road ? road + (size_t)b0 * npix : nullptr,
                   fence ? fence + (size_t)b0 * npix : nullptr, argmax ? argmax + (size_t)b0 * npix : nullptr};
        sd_status st = run_plan(h, SD_NET_FCN8S, frames + (size_t)b0 * npix * 3, nb, &ho, (hipStream_t)stream);
        if (st != SD_OK) return st;
    }
    return SD_OK;
}

// OpenCV's scalar INTER_CUBIC tables for one axis (imgproc/resize.cpp: resize() + interpolateCubic, A = -0.75, weights as
// cvRound(w * 2048) int16, tap indices replicated at the borders); mirrored by oracle/resize.py
static void resize_tables(int dst, int src, int* idx, int* wgt) {
    const double inv = (double)dst / (double)src, scale = 1.0 / inv;
    for (int d = 0; d < dst; ++d) {
        float f = (float)((d + 0.5) * scale - 0.5);
        const int s = (int)std::floor(f);
        const float x = f - (float)s;
        const float A = -0.75f;
        float c[4];
        c[0] = ((A * (x + 1.f) - 5.f * A) * (x + 1.f) + 8.f * A) * (x + 1.f) - 4.f * A;
        c[1] = ((A + 2.f) * x - (A + 3.f)) * x * x + 1.f;
        const float xm = 1.f - x;
        c[2] = ((A + 2.f) * xm - (A + 3.f)) * xm * xm + 1.f;
        c[3] = 1.f - c[0] - c[1] - c[2];
        for (int k = 0; k < 4; ++k) {
            long v = std::lrint((double)(c[k] * 2048.f));          // cvRound: round half to even
            v = v < -32768 ? -32768 : (v > 32767 ? 32767 : v);
            wgt[d * 4 + k] = (int)v;
            int i = s - 1 + k;
            idx[d * 4 + k] = i < 0 ? 0 : (i > src - 1 ? src - 1 : i);
        }
    }
}

sd_status sd_resize_cubic_u8(sd_handle* h, const uint8_t* src, int B, int src_h, int src_w, int channels, uint8_t* dst, int dst_h,
                             int dst_w, void* stream) {
    if (!h || !src || !dst || B <= 0 || src_h <= 0 || src_w <= 0 || channels <= 0 || channels > 4 || dst_h <= 0 || dst_w <= 0 ||
        dst_h > RSZ_MAX || dst_w > RSZ_MAX)
        return fail(h, SD_ERR_INVALID, "sd_resize_cubic_u8: bad arguments (destination extent at most 16384)");
    if (!h->bound) return fail(h, SD_ERR_STATE, "sd_bind_memory first");
    hipStream_t s = (hipStream_t)stream;
    if (src_h == dst_h && src_w == dst_w) {                        // cv2.resize returns a copy
        HIPCHK(h, hipMemcpyAsync(dst, src, (size_t)B * src_h * src_w * channels, hipMemcpyDeviceToDevice, s));
        return SD_OK;
    }
    int* dev = reinterpret_cast<int*>(h->ws + h->o_rsz);
    int *xi = dev, *xa = dev + 4 * (size_t)RSZ_MAX, *yi = dev + 8 * (size_t)RSZ_MAX, *ya = dev + 12 * (size_t)RSZ_MAX;
    const int key[4] = {src_h, src_w, dst_h, dst_w};
    if (std::memcmp(key, h->rsz_key, sizeof(key)) != 0) {
        HIPCHK(h, hipStreamSynchronize(s));                        // the previous tables may still be in use
        h->rsz_host.assign((size_t)RSZ_MAX * 16, 0);
        int* hx = h->rsz_host.data();
        resize_tables(dst_w, src_w, hx, hx + 4 * (size_t)RSZ_MAX);
        resize_tables(dst_h, src_h, hx + 8 * (size_t)RSZ_MAX, hx + 12 * (size_t)RSZ_MAX);
        HIPCHK(h, hipMemcpy(dev, hx, h->rsz_host.size() * sizeof(int), hipMemcpyHostToDevice));
        std::memcpy(h->rsz_key, key, sizeof(key));
    }
    HIPCHK(h, launch_resize_cubic_u8(src, dst, B, src_h, src_w, dst_h, dst_w, channels, xi, xa, yi, ya, s));
    return SD_OK;
}

sd_status sd_post_process(sd_handle* h, const float* disp_raw, int B, float* disp_pp, void* stream) {
    if (!h || !disp_raw || !disp_pp || B <= 0) return fail(h, SD_ERR_INVALID, "sd_post_process: bad arguments");
    HIPCHK(h, launch_post_process(disp_raw, disp_pp, B, h->H, h->W, (hipStream_t)stream));
    return SD_OK;
}

sd_status sd_monodepth_forward(sd_handle* h, const uint8_t* frames, int B, float* disp_pp, float* disp_raw, void* stream) {
    if (!h || !frames || B <= 0 || B > h->max_batch) return fail(h, SD_ERR_INVALID, "sd_monodepth_forward: bad arguments");
    if (!disp_pp && B > h->chunk && !disp_raw)
        return fail(h, SD_ERR_INVALID, "sd_monodepth_forward: disp_pp may be NULL only when the raw pair survives (B <= chunk) or disp_raw is given");
    if (!h->bound) return fail(h, SD_ERR_STATE, "sd_bind_memory first");
    const size_t npix = (size_t)h->H * h->W;
    hipStream_t s = (hipStream_t)stream;
    const NetPlan& p = h->mono;
    const float* raw = reinterpret_cast<const float*>(h->ws + h->o_mono + p.tensors[p.t_output].offset);
    for (int b0 = 0; b0 < B; b0 += h->chunk) {
        const int nb = std::min(h->chunk, B - b0);
        sd_status st = run_plan(h, SD_NET_MONODEPTH, frames + (size_t)b0 * npix * 3, nb, nullptr, s);
        if (st != SD_OK) return st;
        if (disp_pp) HIPCHK(h, launch_post_process(raw, disp_pp + (size_t)b0 * npix, nb, h->H, h->W, s));
        if (disp_raw)
            HIPCHK(h, hipMemcpyAsync(disp_raw + (size_t)b0 * 2 * npix, raw, (size_t)nb * 2 * npix * sizeof(float), hipMemcpyDeviceToDevice, s));
    }
    h->last_mono_frames = B <= h->chunk ? B : 0;
    return SD_OK;
}

static sd_status fuse_impl(sd_handle* h, const float* disp_pp, const float* disp_raw, float* pp_out, const uint8_t* road, const uint8_t* fence,
                           const uint8_t* frames, const sd_camera* cams, int B, int cap, float* dense, float* road_xyz, uint8_t* road_rgb,
                           int32_t* n_road, float* fence_xyz, uint8_t* fence_rgb, int32_t* n_fence, void* stream) {
    if (!h || (!disp_pp && !disp_raw) || !cams || B <= 0 || B > h->max_batch || cap <= 0) return fail(h, SD_ERR_INVALID, "sd_fuse_backproject: bad arguments");
    if (!h->bound) return fail(h, SD_ERR_STATE, "sd_bind_memory first");
    if ((road_xyz && (!road || !n_road)) || (fence_xyz && (!fence || !n_fence)))
        return fail(h, SD_ERR_INVALID, "sd_fuse_backproject: a cloud output needs its mask and its counter");
    hipStream_t s = (hipStream_t)stream;
    std::vector<CamDev> cd(B);
    for (int b = 0; b < B; ++b) cd[b] = make_cam(cams[b]);
    CamDev* dcams = reinterpret_cast<CamDev*>(h->ws + h->o_cams);
    if (h->cams_stage.size() != (size_t)B || std::memcmp(h->cams_stage.data(), cd.data(), sizeof(CamDev) * B) != 0) {
        h->cams_stage = cd;
        HIPCHK(h, hipMemcpyAsync(dcams, h->cams_stage.data(), sizeof(CamDev) * B, hipMemcpyHostToDevice, s));
    }
    FuseParams p{};
    p.disp_pp = disp_pp; p.road = road_xyz ? road : nullptr; p.fence = fence_xyz ? fence : nullptr; p.frames = frames; p.cams = dcams;
    p.B = B; p.H = h->H; p.W = h->W; p.cap = cap; p.sw = h->sw;
    p.dense = dense; p.road_xyz = road_xyz; p.road_rgb = frames ? road_rgb : nullptr; p.n_road = n_road;
    p.fence_xyz = fence_xyz; p.fence_rgb = frames ? fence_rgb : nullptr; p.n_fence = n_fence;
    const size_t nblk = ((size_t)h->H * h->W + 255) / 256;
    p.blk_counts = reinterpret_cast<int32_t*>(h->ws + h->o_fuse);
    p.blk_offsets = p.blk_counts + (size_t)h->max_batch * nblk * 2;
    // one-pass form: look-back words + tickets; the words carry an epoch, so the scratch is zeroed only every 1023 launches
    p.disp_raw = disp_raw; p.pp_out = pp_out;
    // (layout of fuse_onepass_scratch_bytes: the look-back words of max_batch frames, then the tickets)
    p.lb_state = reinterpret_cast<unsigned long long*>(h->ws + h->o_fuse1);
    p.lb_ticket = reinterpret_cast<int32_t*>(h->ws + h->o_fuse1 + (fuse_onepass_scratch_bytes(h->max_batch, h->H, h->W) - 256 - (size_t)h->max_batch * 4));
    if (h->fuse_epoch == 0 || h->fuse_epoch >= 1023) {
        HIPCHK(h, hipMemsetAsync(h->ws + h->o_fuse1, 0, fuse_onepass_scratch_bytes(h->max_batch, h->H, h->W), s));
        h->fuse_epoch = 0;
    }
    p.epoch = ++h->fuse_epoch;
    HIPCHK(h, launch_fuse(p, s));
    return SD_OK;
}

sd_status sd_fuse_backproject(sd_handle* h, const float* disp_pp, const uint8_t* road, const uint8_t* fence, const uint8_t* frames,
                              const sd_camera* cams, int B, int cap, float* dense, float* road_xyz, uint8_t* road_rgb,
                              int32_t* n_road, float* fence_xyz, uint8_t* fence_rgb, int32_t* n_fence, void* stream) {
    if (!disp_pp) return fail(h, SD_ERR_INVALID, "sd_fuse_backproject: bad arguments");
    return fuse_impl(h, disp_pp, nullptr, nullptr, road, fence, frames, cams, B, cap, dense, road_xyz, road_rgb, n_road, fence_xyz, fence_rgb,
                     n_fence, stream);
}

sd_status sd_postprocess_fuse_backproject(sd_handle* h, const float* disp_raw, float* disp_pp_out, const uint8_t* road, const uint8_t* fence,
                                          const uint8_t* frames, const sd_camera* cams, int B, int cap, float* road_xyz, uint8_t* road_rgb,
                                          int32_t* n_road, float* fence_xyz, uint8_t* fence_rgb, int32_t* n_fence, void* stream) {
    if (!h || !disp_pp_out) return fail(h, SD_ERR_INVALID, "sd_postprocess_fuse_backproject: bad arguments");
    if (!disp_raw) {            // the raw pair of the handle's last sd_monodepth_forward, still in the activation arena
        if (h->last_mono_frames < B || B > h->chunk)
            return fail(h, SD_ERR_STATE, "sd_postprocess_fuse_backproject: no raw disparities of a monodepth pass of >= B frames in the arena");
        const NetPlan& p = h->mono;
        disp_raw = reinterpret_cast<const float*>(h->ws + h->o_mono + p.tensors[p.t_output].offset);
    }
    return fuse_impl(h, nullptr, disp_raw, disp_pp_out, road, fence, frames, cams, B, cap, nullptr, road_xyz, road_rgb, n_road, fence_xyz,
                     fence_rgb, n_fence, stream);
}

sd_status sd_road_width(sd_handle* h, const float* road_xyz, const uint8_t* road_rgb, const int32_t* n_road, int B, int cap,
                        const sd_rw_params* prm, sd_rw_result* results, float* final_xyz, uint8_t* final_rgb, int32_t* n_final,
                        void* stream) {
    if (!h || !road_xyz || !n_road || !prm || !results || B <= 0 || B > h->max_batch || cap <= 0 || cap > h->cap)
        return fail(h, SD_ERR_INVALID, "sd_road_width: bad arguments");
    if (!h->bound) return fail(h, SD_ERR_STATE, "sd_bind_memory first");
    hipStream_t s = (hipStream_t)stream;
    float* A = reinterpret_cast<float*>(h->ws + h->o_bufA);
    int32_t *n1 = cnt_slot(h, 1), *n2 = cnt_slot(h, 2), *n3 = cnt_slot(h, 3), *n4 = cnt_slot(h, 4), *n5 = cnt_slot(h, 5),
            *n6 = cnt_slot(h, 6);
    double* plane = reinterpret_cast<double*>(h->ws + h->o_plane);
    void* o3d = h->ws + h->o_o3d;
    RwResultDev* res = reinterpret_cast<RwResultDev*>(results);
    // the stages ping-pong between arenas A and B2: with distinct input and output the ordered compaction of a frame runs on
    // 64 workgroups instead of one (pcl.hip: multi-block compaction)
    float* B2 = reinterpret_cast<float*>(h->ws + h->o_bufB);
    // colours (nullable): the reference carries road_colors through every filter (semantic_depth.py:206-245)
    uint8_t* cA = road_rgb ? reinterpret_cast<uint8_t*>(h->ws + h->o_rgbA) : nullptr;
    uint8_t* cB = road_rgb ? reinterpret_cast<uint8_t*>(h->ws + h->o_rgbB) : nullptr;
    void* cmp = h->ws + h->o_cmp;
    HIPCHK(h, launch_filter_coord({road_xyz, road_rgb, n_road}, {A, cA, n1}, B, cap, F_LT_NEG, 2, prm->z_cut, cmp, s));
    HIPCHK(h, launch_mad_filter({A, cA, n1}, {B2, cB, n2}, B, cap, 1, prm->mad_y, nullptr, cmp, s));
    HIPCHK(h, launch_mad_filter({B2, cB, n2}, {A, cA, n3}, B, cap, 0, prm->mad_x, nullptr, cmp, s));
    HIPCHK(h, launch_plane_filter({A, cA, n3}, {B2, cB, n4}, B, cap, 1, prm->plane_thr, plane, cmp, s));
    const int32_t* nlast = n4;
    const float* fin = B2;
    if (prm->use_o3d) {
        HIPCHK(h, launch_sor({B2, cB, n4}, {A, cA, n5}, B, cap, prm->sor_k, prm->sor_ratio, o3d, nullptr, cmp, s));
        HIPCHK(h, launch_ror({A, cA, n5}, {B2, cB, n6}, B, cap, prm->ror_n, prm->ror_r, o3d, cmp, s));
        nlast = n6;
    } else {
        n5 = n4; n6 = n4;
    }
    HIPCHK(h, launch_end_points({fin, nullptr, nlast}, B, cap, prm->depth - prm->depth_offset, prm->window, res, s));
    HIPCHK(h, launch_record_counts(res, B, n_road, n1, n2, n3, n4, n5, n6, plane, s));
    if (final_xyz) HIPCHK(h, hipMemcpyAsync(final_xyz, fin, (size_t)B * cap * 3 * sizeof(float), hipMemcpyDeviceToDevice, s));
    if (final_rgb) {
        if (!road_rgb) return fail(h, SD_ERR_INVALID, "sd_road_width: final_rgb needs road_rgb");
        HIPCHK(h, hipMemcpyAsync(final_rgb, cB, (size_t)B * cap * 3, hipMemcpyDeviceToDevice, s));
    }
    if (n_final) HIPCHK(h, hipMemcpyAsync(n_final, nlast, (size_t)B * sizeof(int32_t), hipMemcpyDeviceToDevice, s));
    return SD_OK;
}

sd_status sd_fence_to_fence(sd_handle* h, const float* fence_xyz, const uint8_t* fence_rgb, const int32_t* n_fence, int B, int cap,
                            const sd_rw_result* road, const sd_f2f_params* prm, sd_f2f_result* results, float* left_xyz,
                            uint8_t* left_rgb, float* right_xyz, uint8_t* right_rgb, void* stream) {
    if (!h || !fence_xyz || !n_fence || !road || !prm || !results || B <= 0 || B > h->max_batch || cap <= 0 || cap > h->cap)
        return fail(h, SD_ERR_INVALID, "sd_fence_to_fence: bad arguments");
    if (!h->bound) return fail(h, SD_ERR_STATE, "sd_bind_memory first");
    hipStream_t s = (hipStream_t)stream;
    float* A = reinterpret_cast<float*>(h->ws + h->o_bufA);     // fence, then left fence
    float* Rb = reinterpret_cast<float*>(h->ws + h->o_bufB);    // right fence
    float* Lb = reinterpret_cast<float*>(h->ws + h->o_o3d);     // left fence (the Open3D scratch is idle here)
    const size_t pts = (size_t)h->max_batch * h->cap;
    uint8_t* cA = fence_rgb ? reinterpret_cast<uint8_t*>(h->ws + h->o_rgbA) : nullptr;
    uint8_t* cR = fence_rgb ? reinterpret_cast<uint8_t*>(h->ws + h->o_rgbB) : nullptr;
    uint8_t* cL = fence_rgb ? reinterpret_cast<uint8_t*>(h->ws + h->o_o3d) + al(pts * 3 * sizeof(float)) : nullptr;
    if ((left_rgb || right_rgb) && !fence_rgb) return fail(h, SD_ERR_INVALID, "sd_fence_to_fence: colour outputs need fence_rgb");
    int32_t* cnt = cnt_slot(h, 8);                              // [7][max_batch]: slots 8..14
    auto C = [&](int j) { return cnt + (size_t)j * h->max_batch; };
    int32_t* packed = reinterpret_cast<int32_t*>(h->ws + h->o_misc + 4096);
    double* planes = reinterpret_cast<double*>(h->ws + h->o_misc + 4096 + al((size_t)h->max_batch * 7 * sizeof(int32_t)));   // road | left | right, [B][4] each
    double *p_road = planes, *p_left = planes + (size_t)B * 4, *p_right = planes + (size_t)B * 8;
    HIPCHK(h, hipMemcpyAsync(C(0), n_fence, (size_t)B * sizeof(int32_t), hipMemcpyDeviceToDevice, s));
    HIPCHK(h, launch_gather_planes(reinterpret_cast<const RwResultDev*>(road), B, p_road, s));
    HIPCHK(h, launch_mad_filter({fence_xyz, fence_rgb, n_fence}, {A, cA, C(1)}, B, cap, 1, prm->mad_y, nullptr, nullptr, s));
    HIPCHK(h, launch_filter_coord({A, cA, C(1)}, {A, cA, C(2)}, B, cap, F_ABS_LT, 2, prm->z_max, nullptr, s));
    HIPCHK(h, launch_extract_pcls({A, cA, C(2)}, {Lb, cL, C(3)}, {Rb, cR, C(4)}, B, cap, 0, nullptr, s));
    HIPCHK(h, launch_mad_filter({Lb, cL, C(3)}, {Lb, cL, C(5)}, B, cap, 0, prm->mad_left, nullptr, nullptr, s));
    HIPCHK(h, launch_plane_filter({Lb, cL, C(5)}, {Lb, cL, C(5)}, B, cap, 0, prm->plane_thr, p_left, nullptr, s));
    HIPCHK(h, launch_mad_filter({Rb, cR, C(4)}, {Rb, cR, C(6)}, B, cap, 0, prm->mad_right, nullptr, nullptr, s));
    HIPCHK(h, launch_plane_filter({Rb, cR, C(6)}, {Rb, cR, C(6)}, B, cap, 0, prm->plane_thr, p_right, nullptr, s));
    // denoised left / right fence clouds (the reference writes them to *_FENCE.ply, semantic_depth.py:412-415); sizes = counts[5], counts[6]
    if (left_xyz) HIPCHK(h, hipMemcpyAsync(left_xyz, Lb, (size_t)B * cap * 3 * sizeof(float), hipMemcpyDeviceToDevice, s));
    if (right_xyz) HIPCHK(h, hipMemcpyAsync(right_xyz, Rb, (size_t)B * cap * 3 * sizeof(float), hipMemcpyDeviceToDevice, s));
    if (left_rgb) HIPCHK(h, hipMemcpyAsync(left_rgb, cL, (size_t)B * cap * 3, hipMemcpyDeviceToDevice, s));
    if (right_rgb) HIPCHK(h, hipMemcpyAsync(right_rgb, cR, (size_t)B * cap * 3, hipMemcpyDeviceToDevice, s));
    // counts array for the kernel is [7][B] contiguous: compact the strided slots
    for (int j = 0; j < 7; ++j)
        HIPCHK(h, hipMemcpyAsync(packed + (size_t)j * B, C(j), (size_t)B * sizeof(int32_t), hipMemcpyDeviceToDevice, s));
    HIPCHK(h, launch_f2f(p_road, p_left, p_right, B, prm->depth, packed, reinterpret_cast<F2fResultDev*>(results), s));
    return SD_OK;
}

// ---------------------------------------------------------------- pcl.py function by function (single cloud)
static sd_status set_n(sd_handle* h, int n, int32_t** dn, hipStream_t s) {
    *dn = reinterpret_cast<int32_t*>(h->ws + h->o_misc);
    HIPCHK(h, hipMemsetD32Async((hipDeviceptr_t)*dn, n, 1, s));
    return SD_OK;
}
#define PCL_PROLOGUE()                                                                                         \
    if (!h || !xyz || !xyz_out || !n_out || n < 0 || (size_t)n > (size_t)h->max_batch * h->cap)                \
        return fail(h, SD_ERR_INVALID, "pcl: bad arguments");                                                  \
    if (!h->bound) return fail(h, SD_ERR_STATE, "sd_bind_memory first");                                       \
    hipStream_t s = (hipStream_t)stream;                                                                       \
    int32_t* dn = nullptr;                                                                                     \
    { sd_status st_ = set_n(h, n, &dn, s); if (st_ != SD_OK) return st_; }                                     \
    const int cap1 = std::max(n, 1);

sd_status sd_pcl_remove_from_to(sd_handle* h, const float* xyz, const uint8_t* rgb, int n, int axis, double to_meter,
                                float* xyz_out, uint8_t* rgb_out, int32_t* n_out, void* stream) {
    PCL_PROLOGUE();
    HIPCHK(h, launch_filter_coord({xyz, rgb, dn}, {xyz_out, rgb_out, n_out}, 1, cap1, F_LT_NEG, axis, to_meter, h->ws + h->o_cmp, s));
    return SD_OK;
}
sd_status sd_pcl_threshold_complete(sd_handle* h, const float* xyz, const uint8_t* rgb, int n, int axis, double threshold,
                                    float* xyz_out, uint8_t* rgb_out, int32_t* n_out, void* stream) {
    PCL_PROLOGUE();
    HIPCHK(h, launch_filter_coord({xyz, rgb, dn}, {xyz_out, rgb_out, n_out}, 1, cap1, F_ABS_LT, axis, threshold, h->ws + h->o_cmp, s));
    return SD_OK;
}
sd_status sd_pcl_remove_noise_by_mad(sd_handle* h, const float* xyz, const uint8_t* rgb, int n, int axis, double threshold,
                                     float* xyz_out, uint8_t* rgb_out, int32_t* n_out, float* stats_out, void* stream) {
    PCL_PROLOGUE();
    HIPCHK(h, launch_mad_filter({xyz, rgb, dn}, {xyz_out, rgb_out, n_out}, 1, cap1, axis, threshold, stats_out, h->ws + h->o_cmp, s));
    return SD_OK;
}
sd_status sd_pcl_remove_noise_by_fitting_plane(sd_handle* h, const float* xyz, const uint8_t* rgb, int n, int axis, double threshold,
                                               float* xyz_out, uint8_t* rgb_out, int32_t* n_out, double* coeff_out, void* stream) {
    PCL_PROLOGUE();
    if (axis < 0 || axis > 2) return fail(h, SD_ERR_INVALID, "axis");
    HIPCHK(h, launch_plane_filter({xyz, rgb, dn}, {xyz_out, rgb_out, n_out}, 1, cap1, axis, threshold, coeff_out, h->ws + h->o_cmp, s));
    return SD_OK;
}
sd_status sd_pcl_extract_pcls(sd_handle* h, const float* xyz, const uint8_t* rgb, int n, int axis, float* left_xyz, uint8_t* left_rgb,
                              int32_t* n_left, float* right_xyz, uint8_t* right_rgb, int32_t* n_right, float* mean_out, void* stream) {
    float* xyz_out = left_xyz;
    int32_t* n_out = n_left;
    PCL_PROLOGUE();
    if (!right_xyz || !n_right || axis < 0 || axis > 2) return fail(h, SD_ERR_INVALID, "pcl: bad arguments");
    HIPCHK(h, launch_extract_pcls({xyz, rgb, dn}, {left_xyz, left_rgb, n_left}, {right_xyz, right_rgb, n_right}, 1, cap1, axis, mean_out, s));
    return SD_OK;
}
sd_status sd_pcl_get_end_points_of_road(sd_handle* h, const float* xyz, int n, double depth, double window, sd_rw_result* out,
                                        void* stream) {
    if (!h || !xyz || !out || n < 0) return fail(h, SD_ERR_INVALID, "pcl: bad arguments");
    if (!h->bound) return fail(h, SD_ERR_STATE, "sd_bind_memory first");
    hipStream_t s = (hipStream_t)stream;
    int32_t* dn = nullptr;
    sd_status st = set_n(h, n, &dn, s);
    if (st != SD_OK) return st;
    HIPCHK(h, launch_end_points({xyz, nullptr, dn}, 1, std::max(n, 1), depth, window, reinterpret_cast<RwResultDev*>(out), s));
    return SD_OK;
}
sd_status sd_o3d_statistical_outlier_removal(sd_handle* h, const float* xyz, const uint8_t* rgb, int n, int nb_neighbors,
                                             double std_ratio, float* xyz_out, uint8_t* rgb_out, int32_t* n_out,
                                             double* mean_dist_out, void* stream) {
    PCL_PROLOGUE();
    if (nb_neighbors < 1 || nb_neighbors > 16) return fail(h, SD_ERR_INVALID, "nb_neighbors must be in 1..16");
    if (o3d_scratch_bytes(1, cap1) > o3d_scratch_bytes(h->max_batch, h->cap)) return fail(h, SD_ERR_INVALID, "cloud too large");
    HIPCHK(h, launch_sor({xyz, rgb, dn}, {xyz_out, rgb_out, n_out}, 1, cap1, nb_neighbors, std_ratio, h->ws + h->o_o3d, mean_dist_out, h->ws + h->o_cmp, s));
    return SD_OK;
}
sd_status sd_o3d_radius_outlier_removal(sd_handle* h, const float* xyz, const uint8_t* rgb, int n, int nb_points, double radius,
                                        float* xyz_out, uint8_t* rgb_out, int32_t* n_out, void* stream) {
    PCL_PROLOGUE();
    if (o3d_scratch_bytes(1, cap1) > o3d_scratch_bytes(h->max_batch, h->cap)) return fail(h, SD_ERR_INVALID, "cloud too large");
    HIPCHK(h, launch_ror({xyz, rgb, dn}, {xyz_out, rgb_out, n_out}, 1, cap1, nb_points, radius, h->ws + h->o_o3d, h->ws + h->o_cmp, s));
    return SD_OK;
}

// ---------------------------------------------------------------- introspection
sd_status sd_net_tensor(sd_handle* h, sd_net net, const char* name, float* out, size_t cap_floats, int64_t* shape_out, void* stream) {
    if (!h || !name) return SD_ERR_INVALID;
    NetPlan& p = plan_of(h, net);
    auto it = p.tensor_by_name.find(name);
    if (it == p.tensor_by_name.end()) return fail(h, SD_ERR_NOTFOUND, std::string("unknown tensor ") + name);
    const TensorDesc& t = p.tensors[it->second];
    const int N = net == SD_NET_FCN8S ? h->last_fcn_images : h->last_mono_images;
    if (shape_out) { shape_out[0] = N; shape_out[1] = t.H; shape_out[2] = t.W; shape_out[3] = t.fmt ? t.Ctf : t.C; }
    if (!out) return SD_OK;
    if (!h->bound) return fail(h, SD_ERR_STATE, "sd_bind_memory first");
    const size_t numel = (size_t)N * t.H * t.W * (t.fmt ? t.Ctf : t.C);
    if (numel > cap_floats) return fail(h, SD_ERR_INVALID, "output buffer too small");
    const char* abase = h->ws + (net == SD_NET_FCN8S ? h->o_fcn : h->o_mono);
    if (t.fmt)      // split-bf16 planes -> f32
        HIPCHK(h, launch_unsplit(reinterpret_cast<const float*>(abase + t.offset), out, (long)N * t.H * t.W, t.C, t.Ctf, (size_t)p.images * t.H * t.W * t.C,
                                 t.planar16 ? (size_t)p.images * t.H * t.W * 16 : 0, t.x3 ? -1 : t.f16, (hipStream_t)stream));
    else
        HIPCHK(h, hipMemcpyAsync(out, abase + t.offset, numel * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return SD_OK;
}

sd_status sd_profile(sd_handle* h, int enable) {
    if (!h) return SD_ERR_INVALID;
    h->prof = enable != 0;
    if (!h->prof) { h->prof_recs.clear(); h->prof_used = 0; h->prof_last = nullptr; }
    return SD_OK;
}

sd_status sd_profile_read(sd_handle* h, sd_profile_bucket* out, int cap_buckets, int* n_out) {
    if (!h || !out || !n_out || cap_buckets < 1) return SD_ERR_INVALID;
    HIPCHK(h, hipDeviceSynchronize());
    const bool verbose = (h->sw & SW_PROFILE_VERBOSE) != 0;
    int n = 0;
    for (auto& r : h->prof_recs) {
        float ms = 0.f;
        HIPCHK(h, hipEventElapsedTime(&ms, r.a, r.b));
        if (verbose)
            std::fprintf(stderr, "[sd_profile] %-28s M=%-8d N=%-5d K=%-6d %8.3f ms %7.2f TF/s  %s\n", r.op, r.M, r.N, r.K, ms,
                         r.flops / (ms * 1e-3) / 1e12, r.kernel);
        int b = 0;
        while (b < n && std::strcmp(out[b].kernel, r.kernel) != 0) ++b;
        if (b == n) {
            if (n == cap_buckets) continue;
            std::memset(&out[n], 0, sizeof(out[n]));
            std::strncpy(out[n].kernel, r.kernel, 63);
            ++n;
        }
        out[b].launches += 1;
        out[b].ms += ms;
        out[b].flops += r.flops;
        out[b].bytes += r.bytes;
    }
    *n_out = n;
    h->prof_recs.clear();
    h->prof_used = 0;
    h->prof_last = nullptr;
    return SD_OK;
}

double sd_net_flops_per_image(const sd_handle* h, sd_net net) { return h ? plan_of(h, net).flops_per_image : 0.0; }

int sd_pass_frames(const sd_handle* h) { return h ? h->chunk : 0; }

sd_status sd_saturation_count(sd_handle* h, uint64_t* count_out, int reset) {
    if (!h) return SD_ERR_INVALID;
    if (!h->bound) return fail(h, SD_ERR_STATE, "sd_bind_memory first");
    HIPCHK(h, hipDeviceSynchronize());
    unsigned long long v = 0;
    if (count_out) {
        HIPCHK(h, hipMemcpy(&v, h->ws + h->o_misc + SAT_OFF, sizeof(v), hipMemcpyDeviceToHost));
        *count_out = (uint64_t)v;
    }
    if (reset) HIPCHK(h, hipMemset(h->ws + h->o_misc + SAT_OFF, 0, sizeof(v)));
    return SD_OK;
}

sd_status sd_set_reserved_cus(sd_handle* h, int n) {
    if (!h || n < 0 || n > 128) return SD_ERR_INVALID;
    h->reserve_cus = n;
    return SD_OK;
}

// the same counter without a device synchronisation: an 8-byte device-to-host copy enqueued on `stream` (host_dst: pinned memory of the caller; it
// holds the count once the work enqueued on the stream before this call has finished)
sd_status sd_saturation_count_async(sd_handle* h, uint64_t* host_dst, void* stream) {
    if (!h || !host_dst) return SD_ERR_INVALID;
    if (!h->bound) return fail(h, SD_ERR_STATE, "sd_bind_memory first");
    HIPCHK(h, hipMemcpyAsync(host_dst, h->ws + h->o_misc + SAT_OFF, sizeof(uint64_t), hipMemcpyDeviceToHost, (hipStream_t)stream));
    return SD_OK;
}

}  // extern "C"
