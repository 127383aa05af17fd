// Layer plans: FCN-8s (VGG16 encoder + fcn8s/fcn.py:159-215 decoder) and monodepth (vgg / resnet50).
// Architectures per SURVEY.md Appendix A/B.  Host-only code.
#include "plan.hpp"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <stdexcept>

namespace sd {

// SEMDEPTH_DISABLE=name[,name...]: the ONE run-time variable that switches specialised kernels off in favour of the generic ones behind them (parity tests, A/B
// runs).  Names: dma dma3 direct stem fold tail1 pool_fuse planar n16 fuse1 fuse4 flat rowskip dma_big mfma16.  Read when a handle is created (plan + switches).
// Closed A/Bs of earlier rounds (SEMDEPTH_NO_UPTILE, _NO_N16_MT1, _NO_DMA32, _NO_SMALLN_TILE, _X3_RING3, _X3_KEEP, _HS_PHASED_TAPS, _DMA_DBG, _NO_PLANAR_WIDE,
// _NO_DIRECT128, _DIRECT_MINPIX, _NO_MFMA_HEADS, _X3_NO_POOL_FUSE) and the decomposition runs (SEMDEPTH_X3_DIAG) exist in -DSD_DEV_VARIANTS builds only
// (SEMDEPTH_DEV_BUILD=1 python -m semantic_depth_amd.build --force).
bool sd_disabled(const char* what) {
    const char* e = std::getenv("SEMDEPTH_DISABLE");
    if (!e) return false;
    const size_t n = std::strlen(what);
    for (const char* p = e; *p;) {
        const char* q = std::strchr(p, ',');
        const size_t len = q ? (size_t)(q - p) : std::strlen(p);
        if (len == n && !std::strncmp(p, what, n)) return true;
        p += len + (q ? 1 : 0);
    }
    return false;
}
// the first token of SEMDEPTH_DISABLE that names nothing (a typo must not silently leave the specialised kernel on): sd_create refuses the handle
std::string sd_disable_unknown() {
    static const char* const known[] = {"dma", "dma3", "direct", "stem", "fold", "tail1", "pool_fuse", "planar", "n16", "fuse1", "fuse4", "flat", "rowskip", "dma_big", "mfma16"};
    const char* e = std::getenv("SEMDEPTH_DISABLE");
    if (!e) return "";
    for (const char* p = e; *p;) {
        const char* q = std::strchr(p, ',');
        const size_t len = q ? (size_t)(q - p) : std::strlen(p);
        bool ok = len == 0;
        for (const char* k : known) ok = ok || (std::strlen(k) == len && !std::strncmp(p, k, len));
        if (!ok) return std::string(p, len);
        p += len + (q ? 1 : 0);
    }
    return "";
}
static const char* dev_env(const char* name) {
#ifdef SD_DEV_VARIANTS
    return std::getenv(name);
#else
    (void)name;
    return nullptr;
#endif
}

unsigned latch_switches() {
    static const struct { const char* name; unsigned bit; } tab[] = {
        {"n16", SW_NO_N16}, {"dma_big", SW_NO_DMA_BIG}, {"stem", SW_NO_STEM}, {"fuse4", SW_NO_FUSE4}, {"dma", SW_NO_DMA}, {"fuse1", SW_NO_FUSE1},
        {"dma3", SW_NO_DMA3}, {"fold", SW_NO_FOLD}, {"tail1", SW_NO_TAIL1}, {"rowskip", SW_NO_ROWSKIP}, {"flat", SW_NO_FLAT}, {"mfma16", SW_MFMA32}};
    static const struct { const char* name; unsigned bit; } dev[] = {
        {"SEMDEPTH_NO_UPTILE", SW_NO_UPTILE}, {"SEMDEPTH_NO_N16_MT1", SW_NO_N16_MT1}, {"SEMDEPTH_NO_DMA32", SW_NO_DMA32}, {"SEMDEPTH_NO_SMALLN_TILE", SW_NO_SMALLN_TILE},
        {"SEMDEPTH_X3_RING3", SW_X3_RING3}, {"SEMDEPTH_HS_PHASED_TAPS", SW_HS_TAPS}};
    unsigned sw = 0;
    for (const auto& e : tab)
        if (sd_disabled(e.name)) sw |= e.bit;
    for (const auto& e : dev)
        if (dev_env(e.name)) sw |= e.bit;
    if (const char* d = dev_env("SEMDEPTH_DMA_DBG")) if (atoi(d) & 16) sw |= SW_DMA_DBG16;
    if (const char* v = std::getenv("SEMDEPTH_PROFILE_VERBOSE")) if (v[0] == '1') sw |= SW_PROFILE_VERBOSE;
    if (const char* v = dev_env("SEMDEPTH_X3_KEEP")) if (atoi(v) == 0) sw |= SW_X3_NOKEEP;
    if (const char* v = dev_env("SEMDEPTH_X3_DIAG")) sw |= ((atoi(v) & 1) ? SW_X3_DIAG_NOSTORE : 0u) | ((atoi(v) & 2) ? SW_X3_DIAG_NOMFMA : 0u) | ((atoi(v) & 4) ? SW_X3_DIAG_TIMED : 0u);
    return sw;
}

namespace {

constexpr size_t ALIGN = 256;
size_t align_up(size_t v) { return (v + ALIGN - 1) / ALIGN * ALIGN; }

struct Src { int t; int up; int stride = 0; /* 0: the op's stride */ };

struct Builder {
    NetPlan p;

    int tensor(const std::string& name, int N, int H, int W, int C) {
        TensorDesc t;
        t.name = name; t.N = N; t.H = H; t.W = W; t.C = C; t.Ctf = C;
        t.fmt = p.prec ? 1 : 0;          // the split engine keeps its activations as split-bf16 planes
        t.x3 = p.x3;
        if (p.h2) t.f16 = 3;             // fp16 hi + scaled lo planes (SD_PREC_F16X2)
        t.bytes = (size_t)N * H * W * C * (p.x3 ? 6 : sizeof(float));        // bf16 x 3: three 16-bit planes
        p.tensors.push_back(t);
        p.tensor_by_name[name] = (int)p.tensors.size() - 1;
        return (int)p.tensors.size() - 1;
    }
    void alias(const std::string& name, int t) { p.tensor_by_name[name] = t; }

    int wslot(const std::string& name, std::initializer_list<int64_t> shape, int layout, int Kpad = 0, int CoutPad = 0, int nout = 0, int nsplit = 1) {
        WeightSlot s;
        s.name = name; s.rank = (int)shape.size(); s.layout = layout; s.Kpad = Kpad; s.CoutPad = CoutPad; s.nout = nout; s.nsplit = nsplit;
        int i = 0;
        for (auto v : shape) s.shape[i++] = v;
        size_t n = 0;
        switch (layout) {
            case WL_IGEMM: case WL_IGEMM_SPLIT: n = (size_t)Kpad * CoutPad; break;
            case WL_DIRECT_SPLIT: n = (size_t)Kpad * CoutPad * nsplit; break;     // Kpad = nchunks * 9 * 16, CoutPad = 32 or 64
            case WL_SMALLN: n = (size_t)s.shape[0] * s.shape[1] * s.shape[2] * 4; break;
            case WL_BIAS4: n = 4; break;
            case WL_TAIL_UP: n = 4 * 4 * 3 * 64 * 4; break;         // [parity][tap][plane][lane] x 16 bytes
            case WL_TAIL_ICONV: n = 6 * 3 * 64 * 4; break;          // [row block, half][plane][lane] x 16 bytes
            default: n = 1; for (int j = 0; j < s.rank; ++j) n *= (size_t)s.shape[j];
        }
        s.bytes = n * sizeof(float);
        if (p.x3 && (layout == WL_IGEMM_SPLIT || layout == WL_DIRECT_SPLIT)) { s.x3 = 1; s.bytes = n * 6; }      // three bf16 planes
        if (p.h2 && (layout == WL_IGEMM_SPLIT || layout == WL_DIRECT_SPLIT)) { s.f16 = 1; s.hs = 1; }            // two fp16 planes of w * 2^k (WeightSlot::wscale)
        if (p.h2 && (layout == WL_TAIL_UP || layout == WL_TAIL_ICONV)) s.hs = 1;                                 // the same in planes 0 and 1 of the fragments
        p.weights.push_back(s);
        p.weight_by_name[name] = (int)p.weights.size() - 1;
        return (int)p.weights.size() - 1;
    }

    void push(OpDesc& op) { p.ops.push_back(op); }

    // dense conv on the MFMA engine
    int conv(const std::string& name, std::vector<Src> srcs, int Cout, int k, int stride, int act, const std::string& wname,
             const std::string& bname, int residual = -1) {
        OpDesc op;
        op.kind = OP_CONV; op.name = name; op.nsrc = (int)srcs.size();
        int Ctot = 0, Ctf = 0, Hin = 0, Win = 0, N = 0;
        bool vec = true;
        int srcCtf[3] = {0, 0, 0}, srcCpad[3] = {0, 0, 0}, srcVec[3] = {0, 0, 0};
        int Cvec = 0, CqPad = 0;
        for (int i = 0; i < op.nsrc; ++i) {
            const TensorDesc& t = p.tensors[srcs[i].t];
            op.src[i] = srcs[i].t; op.up[i] = srcs[i].up;
            const int h = t.H * (srcs[i].up ? 2 : 1), w = t.W * (srcs[i].up ? 2 : 1);
            op.sstride[i] = srcs[i].stride ? srcs[i].stride : stride;
            const int pad_ = (k - 1) / 2;
            const int ho = (h + 2 * pad_ - k) / op.sstride[i] + 1, wo = (w + 2 * pad_ - k) / op.sstride[i] + 1;
            if (i == 0) { Hin = h; Win = w; N = t.N; }
            else {
                const int ho0 = (Hin + 2 * pad_ - k) / op.sstride[0] + 1, wo0 = (Win + 2 * pad_ - k) / op.sstride[0] + 1;
                if (ho != ho0 || wo != wo0 || t.N != N) throw std::runtime_error("conv " + name + ": source dims disagree");
            }
            srcCtf[i] = t.Ctf; srcCpad[i] = (t.C + 3) / 4 * 4;     // the K axis holds whole channel quads per source
            Ctot += srcCpad[i]; Ctf += t.Ctf;
            srcVec[i] = (t.C % 32 == 0) ? 1 : 0;
            if (srcVec[i]) Cvec += t.C; else { CqPad += srcCpad[i]; vec = false; }
        }
        op.k = k; op.stride = stride; op.pad = (k - 1) / 2; op.act = act; op.residual = residual;
        const int Hout = (Hin + 2 * op.pad - k) / op.sstride[0] + 1, Wout = (Win + 2 * op.pad - k) / op.sstride[0] + 1;
        op.Ctot = Ctot; op.K = k * k * Ctot; op.vec = vec ? 1 : 0;
        op.Kvec = k * k * Cvec; op.CqPad = CqPad;
        op.Kpad = op.Kvec + (k * k * CqPad + 31) / 32 * 32;
        // full-resolution few-channel 3x3 layers of the split engine go to the direct (halo-tile) kernel
        // 128 .. 512 output channels: 2 .. 8 passes of 64 per tile (256 and more only where an image has enough tiles: the
        // choice must not depend on the batch)
        const char* mp = dev_env("SEMDEPTH_DIRECT_MINPIX");
        const int64_t minpix = mp ? std::atoll(mp) : 512;
        const bool split128 = (Cout == 128 && !dev_env("SEMDEPTH_NO_DIRECT128")) ||
                              ((Cout == 256 || Cout == 512) && (int64_t)Hin * Win >= minpix);
        bool direct = p.prec && k == 3 && stride == 1 && (Cout <= 32 || Cout == 64 || split128) && Cout % 8 == 0 && Win % 32 == 0 && residual < 0 &&
                      !sd_disabled("direct");
        for (int i = 0; i < op.nsrc; ++i)
            if (p.tensors[op.src[i]].C % 8 || op.sstride[i] != 1) direct = false;
        // bf16 x 3: an upconv layer (3x3 on a x2 nearest-neighbour upsampled source) with a wide output runs as four 2x2 convs on the source
        // itself, one per output parity, on the 256 x 256 GEMM block (OpDesc::fold).  The choice depends on the layer alone, never on the batch.
        // The two-plane split engines (SD_PREC_F16X2, and bf16 x 2 / the precision plan: VERDICT r4 item 7): the same on conv_dma.hip, for outputs of 128
        // channels and more (upconv6 / 5 / 4; measured at 64 on f16x2: upconv3 0.74 ms as a direct 3x3 conv on the upsampled source, 0.89 ms folded on the
        // 256 x 64 GEMM block)
        if (((p.x3 && Cout % 256 == 0) || (p.prec && !p.x3 && Cout % 128 == 0)) && k == 3 && stride == 1 && op.nsrc == 1 && op.up[0] && p.tensors[op.src[0]].C % 32 == 0 &&
            residual < 0 && !(latch_switches() & (SW_NO_FOLD | SW_NO_DMA))) {        // (the folded GEMM form exists on conv_dma3 only: SEMDEPTH_NO_DMA implies no fold)
            const TensorDesc& t = p.tensors[op.src[0]];
            op.fold = 1; op.vec = 1;
            op.Ctot = t.C; op.K = 4 * t.C; op.Kpad = op.Kvec = 4 * t.C; op.CqPad = 0;
            op.w = wslot(wname, {k, k, Ctf, Cout}, WL_IGEMM_SPLIT, 4 * op.Kpad, Cout);
            WeightSlot& ws = p.weights[op.w];
            ws.fold = 1; ws.nsrc = 1; ws.vec = 1; ws.Ktotal = 4 * op.Kpad; ws.srcCtf[0] = ws.srcCpad[0] = t.C; ws.srcVec[0] = 1;
            op.b = wslot(bname, {Cout}, WL_RAW);
            op.dst = tensor(name, N, Hout, Wout, Cout);
            op.tab_bytes = (size_t)4 * (op.Kpad / 32) * sizeof(KEntry);
            op.flops = 2.0 * (double)N * Hout * Wout * Cout * (double)(4 * Ctf);      // the multiplications that run (4/9 of the 3x3 form's)
            op.m_fastest = 1;
            push(op);
            return op.dst;
        }
        if (direct) {
            op.kind = OP_CONV_DIRECT;
            int nch = 0;
            for (int i = 0; i < op.nsrc; ++i) nch += (p.tensors[op.src[i]].C + 15) / 16;
            op.nchunks = nch;
            op.nsplit = Cout > 64 ? Cout / 64 : 1;
            // bf16 x 3 and (round 6) SD_PREC_F16X2, an upconv layer (one x2-upsampled source): the upsample-folded form of the direct kernel (OpDesc::fold;
            // f16x2: the 32 / 64-channel forms -- a 16-channel upconv1 outside the fused decoder tail runs the 16-wide MFMA kernel unfolded)
            op.fold = ((p.x3 || (p.h2 && Cout >= 32)) && op.nsrc == 1 && op.up[0] && p.tensors[op.src[0]].C % 16 == 0 && Hout % 2 == 0 && !(latch_switches() & (SW_NO_FOLD | SW_NO_UPTILE))) ? 1 : 0;
            op.w = wslot(wname, {k, k, Ctf, Cout}, WL_DIRECT_SPLIT, nch * (op.fold ? 16 : 9) * 16, Cout <= 32 ? 32 : 64, 0, op.nsplit);
            WeightSlot& ws = p.weights[op.w];
            ws.nsrc = op.nsrc; ws.fold = op.fold;
            for (int i = 0; i < 3; ++i) { ws.srcCtf[i] = srcCtf[i]; ws.srcCpad[i] = i < op.nsrc ? p.tensors[op.src[i]].C : 0; }
            op.b = wslot(bname, {Cout}, WL_RAW);
            op.dst = tensor(name, N, Hout, Wout, Cout);
            op.tab_bytes = (size_t)nch * sizeof(DirectChunk);
            op.flops = 2.0 * (double)N * Hout * Wout * Cout * (double)((op.fold ? 4 : k * k) * Ctf);      // (folded: the multiplications that run)
            if (op.fold) op.K = 4 * Ctot;
            push(op);
            return op.dst;
        }
        const int bn = p.prec ? conv_split_tile_n(Cout) : conv_tile_n(Cout);
        const int CoutPad = (Cout + bn - 1) / bn * bn;
        op.w = wslot(wname, {k, k, Ctf, Cout}, p.prec ? WL_IGEMM_SPLIT : WL_IGEMM, op.Kpad, CoutPad);
        {
            WeightSlot& ws = p.weights[op.w];
            ws.nsrc = op.nsrc; ws.vec = op.vec; ws.Ktotal = op.Kpad;
            for (int i = 0; i < 3; ++i) { ws.srcCtf[i] = srcCtf[i]; ws.srcCpad[i] = srcCpad[i]; ws.srcVec[i] = srcVec[i]; }
        }
        op.b = wslot(bname, {Cout}, WL_RAW);
        op.dst = tensor(name, N, Hout, Wout, Cout);
        // one descriptor per k-tile, plus 8 quad descriptors for every k-tile of the quad tail
        op.tab_bytes = ((size_t)(op.Kpad / 32) + (size_t)((op.Kpad - op.Kvec) / 32) * 8) * sizeof(KEntry);
        const double M = (double)N * Hout * Wout;
        op.flops = 2.0 * M * Cout * (double)(k * k * Ctf);      // algorithmic (TF) K, not the padded one
        // block order: walk M first when the weight matrix is the larger operand (it then stays L2-resident per N panel)
        // concurrent blocks walk M for one weight panel: the panel streams through L2 once per XCD and is shared by all of them
        op.m_fastest = 1;
        (void)Ctf;
        push(op);
        return op.dst;
    }

    // ResNet bottleneck tail: act( conv1x1(a; wA) + conv1x1_stride_s(b; wB) + biasA + biasB ) as ONE GEMM over K = Ca + Cb
    int conv_pair_1x1(const std::string& name, int a, int b, int stride_b, int Cout, int act, const std::string& wA,
                      const std::string& bA, const std::string& wB, const std::string& bB) {
        OpDesc op;
        op.kind = OP_CONV; op.name = name; op.nsrc = 2;
        const TensorDesc& ta = p.tensors[a];
        const TensorDesc& tb = p.tensors[b];
        if (ta.C % 32 || tb.C % 32) throw std::runtime_error("conv_pair_1x1 " + name + ": channels must be multiples of 32");
        if ((tb.H - 1) / stride_b + 1 != ta.H || (tb.W - 1) / stride_b + 1 != ta.W || ta.N != tb.N)
            throw std::runtime_error("conv_pair_1x1 " + name + ": source dims disagree");
        op.src[0] = a; op.src[1] = b; op.sstride[0] = 1; op.sstride[1] = stride_b;
        op.k = 1; op.stride = 1; op.pad = 0; op.act = act;
        op.Ctot = ta.C + tb.C; op.K = op.Ctot; op.Kpad = op.K; op.Kvec = op.K; op.vec = 1;
        const int bn = p.prec ? conv_split_tile_n(Cout) : conv_tile_n(Cout);
        const int CoutPad = (Cout + bn - 1) / bn * bn;
        const int lay = p.prec ? WL_IGEMM_SPLIT : WL_IGEMM;
        op.w = wslot(wA, {1, 1, ta.C, Cout}, lay, ta.C, CoutPad);
        const int w2 = wslot(wB, {1, 1, tb.C, Cout}, lay, tb.C, CoutPad);
        {
            WeightSlot& s0 = p.weights[op.w];
            s0.nsrc = 1; s0.srcCtf[0] = s0.srcCpad[0] = ta.C; s0.srcVec[0] = 1; s0.vec = 1; s0.Ktotal = op.Kpad; s0.k_off = 0;
            WeightSlot& s1 = p.weights[w2];
            s1.nsrc = 1; s1.srcCtf[0] = s1.srcCpad[0] = tb.C; s1.srcVec[0] = 1; s1.vec = 1; s1.Ktotal = op.Kpad; s1.k_off = ta.C; s1.owner = op.w;
        }
        op.b = wslot(bA, {Cout}, WL_RAW);
        const int b2 = wslot(bB, {Cout}, WL_RAW);
        p.weights[b2].owner = op.b;
        op.dst = tensor(name, ta.N, ta.H, ta.W, Cout);
        op.tab_bytes = (size_t)(op.Kpad / 32) * sizeof(KEntry);
        op.flops = 2.0 * ta.N * ta.H * ta.W * Cout * (double)op.K;
        op.m_fastest = 1;
        push(op);
        return op.dst;
    }

    int smalln(const std::string& name, int src, int nout, int k, int act, const std::string& wname, const std::string& bname, int cout_tf,
               bool feeds_conv = false) {
        OpDesc op;
        op.kind = OP_SMALLN; op.name = name; op.nsrc = 1; op.src[0] = src; op.k = k; op.pad = (k - 1) / 2; op.act = act; op.nout = nout;
        const TensorDesc& t = p.tensors[src];
        // a 3x3 head whose output feeds the next iconv (disp4..disp2, written as one zero-padded octet per pixel) is a direct
        // conv on the 16-wide MFMA: its two real output channels ride in a 16-column weight image
        if (feeds_conv && p.prec && !p.x3 && k == 3 && t.W % 32 == 0 && t.C % 8 == 0 && nout <= 8 && nout == cout_tf &&
            !sd_disabled("direct") && !sd_disabled("n16") && !dev_env("SEMDEPTH_NO_MFMA_HEADS")) {
            op.kind = OP_CONV_DIRECT; op.stride = 1;
            op.nchunks = (t.C + 15) / 16;
            op.nsplit = 1;
            op.w = wslot(wname, {k, k, t.C, cout_tf}, WL_DIRECT_SPLIT, op.nchunks * 9 * 16, 32, 0, 1);
            WeightSlot& ws = p.weights[op.w];
            ws.nsrc = 1; ws.srcCtf[0] = t.Ctf ? t.Ctf : t.C; ws.srcCpad[0] = t.C;
            op.b = wslot(bname, {cout_tf}, WL_BIAS4, 0, 0, nout);
            op.dst = tensor(name, t.N, t.H, t.W, 8);
            p.tensors[op.dst].Ctf = nout;
            op.tab_bytes = (size_t)op.nchunks * sizeof(DirectChunk);
            op.K = k * k * t.C;
            op.flops = 2.0 * t.N * t.H * t.W * nout * k * k * t.C;
            push(op);
            return op.dst;
        }
        op.w = wslot(wname, {k, k, t.C, cout_tf}, WL_SMALLN, 0, 0, nout);
        op.b = wslot(bname, {cout_tf}, WL_BIAS4, 0, 0, nout);
        // a disparity map that feeds the next iconv is stored as ONE zero-padded channel octet per pixel in the split
        // engine (8 stored channels, 2 real), so the direct 3x3 kernel can DMA it like any other source
        const bool octet = feeds_conv && p.prec;
        op.dst = tensor(name, t.N, t.H, t.W, octet ? 8 : nout);
        p.tensors[op.dst].Ctf = nout;
        if (!feeds_conv) p.tensors[op.dst].fmt = 0;      // consumed by f32 kernels (deconv ladder, post-processing)
        op.flops = 2.0 * p.tensors[src].N * p.tensors[src].H * p.tensors[src].W * nout * k * k * p.tensors[src].C;
        push(op);
        return op.dst;
    }

    // bf16 x 3: upconv1 -> iconv1 -> disp1 (channel 0) in one launch (dec_tail.hip).  a = dec/iconv2 (32 channels), d2 = dec/disp2 (octet)
    int dec_tail1(int a, int d2) {
        OpDesc op;
        op.kind = OP_DEC_TAIL1; op.name = "dec/tail1"; op.nsrc = 2; op.src[0] = a; op.src[1] = d2; op.k = 3; op.pad = 1; op.act = ACT_SIGMOID03;
        const TensorDesc ta = p.tensors[a];
        if (ta.C != 32 || p.tensors[d2].C != 8 || p.tensors[d2].H != ta.H || p.tensors[d2].W != ta.W)
            throw std::runtime_error("dec_tail1: needs a 32-channel source and a disparity octet of the same size");
        op.w = wslot("dec/upconv1/weights", {3, 3, 32, 16}, WL_TAIL_UP);
        op.b = wslot("dec/upconv1/biases", {16}, WL_RAW);
        op.w2 = wslot("dec/iconv1/weights", {3, 3, 18, 16}, WL_TAIL_ICONV);
        op.b2 = wslot("dec/iconv1/biases", {16}, WL_RAW);
        op.w3 = wslot("dec/disp1/weights", {3, 3, 16, 2}, WL_SMALLN, 0, 0, 1);
        op.b3 = wslot("dec/disp1/biases", {2}, WL_BIAS4, 0, 0, 1);
        op.dst = tensor("dec/disp1", ta.N, 2 * ta.H, 2 * ta.W, 1);
        p.tensors[op.dst].Ctf = 1; p.tensors[op.dst].fmt = 0; p.tensors[op.dst].x3 = 0;
        p.tensors[op.dst].bytes = (size_t)ta.N * 2 * ta.H * 2 * ta.W * sizeof(float);
        op.K = 4 * 32 + 9 * 18 + 9;        // multiplications per output pixel and channel that run: folded upconv1 + iconv1 (16 channels each) + the head
        op.flops = 2.0 * ta.N * 4.0 * ta.H * ta.W * (16.0 * (4 * 32) + 16.0 * (9 * 18) + 9.0 * 16);
        push(op);
        return op.dst;
    }

    int pool(const std::string& name, int src, bool zero3) {
        OpDesc op;
        op.kind = zero3 ? OP_POOL3Z : OP_POOL2; op.name = name; op.nsrc = 1; op.src[0] = src;
        const TensorDesc t = p.tensors[src];
        const int Ho = zero3 ? (t.H - 1) / 2 + 1 : t.H / 2, Wo = zero3 ? (t.W - 1) / 2 + 1 : t.W / 2;
        // a direct conv whose only consumer is this pool applies it in its epilogue (max commutes with bias + ReLU/ELU);
        // the full-resolution tensor is then never written
        if (!zero3 && !(p.x3 && dev_env("SEMDEPTH_X3_NO_POOL_FUSE")) && !p.ops.empty() && p.ops.back().kind == OP_CONV_DIRECT && p.ops.back().dst == src && t.H % 2 == 0 && t.W % 2 == 0 &&
            (p.ops.back().act == ACT_RELU || p.ops.back().act == ACT_ELU || p.ops.back().act == ACT_NONE) && !sd_disabled("pool_fuse")) {
            OpDesc& prod = p.ops.back();
            prod.fuse_pool = 1;
            for (auto it = p.tensor_by_name.begin(); it != p.tensor_by_name.end();)
                it = it->second == src ? p.tensor_by_name.erase(it) : std::next(it);
            prod.dst = tensor(name, t.N, Ho, Wo, t.C);
            return prod.dst;
        }
        // the LDS-DMA conv kernel does the same by walking its output pixels in 2x2-window-major order
        if (!zero3 && p.prec && !p.x3 && !p.ops.empty() && p.ops.back().kind == OP_CONV && p.ops.back().dst == src && p.ops.back().vec &&
            p.ops.back().nsrc == 1 && p.ops.back().Kvec == p.ops.back().Kpad && p.ops.back().Kpad >= 64 && t.C % 64 == 0 &&
            p.ops.back().residual < 0 && t.H % 2 == 0 && t.W % 2 == 0 &&
            (p.ops.back().act == ACT_RELU || p.ops.back().act == ACT_ELU || p.ops.back().act == ACT_NONE) && !sd_disabled("pool_fuse") &&
            !sd_disabled("dma")) {
            OpDesc& prod = p.ops.back();
            prod.fuse_pool = 1;
            for (auto it = p.tensor_by_name.begin(); it != p.tensor_by_name.end();)
                it = it->second == src ? p.tensor_by_name.erase(it) : std::next(it);
            prod.dst = tensor(name, t.N, Ho, Wo, t.C);
            return prod.dst;
        }
        op.dst = tensor(name, t.N, Ho, Wo, t.C);
        push(op);
        return op.dst;
    }

    // a tensor written by a direct conv (or by a conv that always takes the LDS-DMA kernel) and read only by direct convs and
    // LDS-tiled few-channel heads is handed over as 16-channel sub-planes (TensorDesc::planar16): the reader's 16-channel
    // chunk of a pixel row is then one contiguous run instead of 32 bytes out of every pixel's line
    void mark_planar() {
        if (!p.prec || sd_disabled("planar")) return;
        if (p.x3) {
            // bf16 x 3 (round 5): the one hand-off whose writer and reader both know the layout -- a stem conv (conv_stem.hip) whose output is read by direct 3x3
            // convs only (FCN-8s conv1_1 -> conv1_2): a 16-channel chunk of conv1_2's halo is then a contiguous run of the sub-plane instead of 32 bytes out of
            // every pixel's 128-byte line, which four chunk passes fetched four times (6.4 GB of input read as ~20)
            for (size_t ti = 0; ti < p.tensors.size(); ++ti) {
                TensorDesc& t = p.tensors[ti];
                if (t.C % 16 || (t.C != 32 && t.C != 64) || t.W % 32 || (int)ti == p.t_output || (int)ti == p.t_input) continue;
                bool made = false, ok = true;
                int readers = 0;
                for (const OpDesc& op : p.ops) {
                    if (op.dst == (int)ti)
                        made = op.kind == OP_CONV && op.nsrc == 1 && op.src[0] == p.t_input && (op.k & 1) && op.k <= 7 && !op.fold && !op.fuse_pool && op.residual < 0;
                    bool reads = op.residual == (int)ti;
                    for (int j = 0; j < op.nsrc; ++j) reads = reads || op.src[j] == (int)ti;
                    if (!reads) continue;
                    ++readers;
                    // (and, round 5, the 3x3 stride-2 pool: monodepth's enc/conv1 -> pool1 + the skip of iconv2, whose four chunk passes fetched 3.2 GB as 12.9)
                    ok = ok && (op.kind == OP_CONV_DIRECT || (op.kind == OP_POOL3Z && t.C % 8 == 0)) && op.residual != (int)ti;
                }
                if (made && ok && readers > 0 && !(latch_switches() & SW_NO_STEM)) t.planar16 = 1;
            }
            return;
        }
        const bool wide = !dev_env("SEMDEPTH_NO_PLANAR_WIDE");
        for (size_t ti = 0; ti < p.tensors.size(); ++ti) {
            TensorDesc& t = p.tensors[ti];
            if (t.C % 16 || (int)ti == p.t_output || (int)ti == p.t_input) continue;
            bool made = false, ok = true;
            int readers = 0;
            for (const OpDesc& op : p.ops) {
                if (op.dst == (int)ti)
                    made = op.kind == OP_CONV_DIRECT ||
                           (wide && op.kind == OP_CONV && op.vec && op.Kvec == op.Kpad && t.C % 64 == 0 && op.Kpad >= 64 &&
                            !sd_disabled("dma")) ||
                           // (three-product engine, round 5: a stem conv's output, as on bf16 x 3 above)
                           (p.h2 && op.kind == OP_CONV && op.nsrc == 1 && op.src[0] == p.t_input && (op.k & 1) && op.k <= 7 && !op.fold && !op.fuse_pool &&
                            op.residual < 0 && (t.C == 32 || t.C == 64) && t.W % 32 == 0 && !(latch_switches() & SW_NO_STEM));
                bool reads = op.residual == (int)ti;
                for (int j = 0; j < op.nsrc; ++j) reads = reads || op.src[j] == (int)ti;
                if (!reads) continue;
                ++readers;
                const bool head = wide && op.kind == OP_SMALLN && conv_smalln_tiled(1, op.k, t.W, t.C, op.nout, latch_switches());
                ok = ok && (op.kind == OP_CONV_DIRECT || head || (p.h2 && op.kind == OP_POOL3Z && t.C % 8 == 0)) && op.residual != (int)ti;
            }
            if (made && ok && readers > 0) t.planar16 = 1;
        }
    }

    // precision plan: mark the conv layers named in p.f16_spec, then close the choice under the one-format-per-tensor rule
    void apply_precision_plan() {
        if (p.h2) {                      // SD_PREC_F16X2: every conv in the three-product HS form (the tensors and weight slots were marked when they were made)
            for (OpDesc& op : p.ops)
                if (op.kind == OP_CONV || op.kind == OP_CONV_DIRECT) op.f16 = 4;
            return;
        }
        if (!p.prec || p.x3 || p.f16_spec.empty()) return;
        std::vector<std::string> toks;
        {
            size_t a = 0;
            while (a <= p.f16_spec.size()) {
                size_t b = p.f16_spec.find(',', a);
                if (b == std::string::npos) b = p.f16_spec.size();
                std::string t = p.f16_spec.substr(a, b - a);
                while (!t.empty() && t.front() == ' ') t.erase(t.begin());
                while (!t.empty() && t.back() == ' ') t.pop_back();
                if (!t.empty()) toks.push_back(t);
                a = b + 1;
            }
        }
        auto is_conv = [](const OpDesc& op) { return op.kind == OP_CONV || op.kind == OP_CONV_DIRECT; };
        // token = layer name | prefix* | *, optionally followed by ":1" = ONE MFMA product (x * w_hi: plain fp16) instead of two, or by
        // ":x" = two products on fp16 hi + lo ACTIVATION planes and w_hi (the error is the weight's rounding, the input tensor keeps 22
        // bits: for layers whose input also feeds a score head; direct 3x3 layers only)
        auto match = [](const std::string& t, const std::string& n) {
            return t == "*" || t == n || (t.size() > 1 && t.back() == '*' && n.compare(0, t.size() - 1, t, 0, t.size() - 1) == 0);
        };
        for (std::string t : toks) {
            int mode = 1;
            if (t.size() > 2 && t.compare(t.size() - 2, 2, ":1") == 0) { mode = 2; t.erase(t.size() - 2); }
            else if (t.size() > 2 && t.compare(t.size() - 2, 2, ":x") == 0) { mode = 3; t.erase(t.size() - 2); }
            bool any = false;
            for (OpDesc& op : p.ops) {
                if (!match(t, op.name)) continue;
                any = true;          // (a layer that is not an MFMA conv at this geometry, e.g. a small-N head, follows its input's format)
                // a per-thread head named EXACTLY (no wildcard): its input tensor becomes ONE fp16 plane -- the producer writes half the
                // bytes and the head reads half (dec/disp1 reads the full-resolution iconv1 output: 2.1 -> 1.07 GB each way)
                if (op.kind == OP_SMALLN && t == op.name && mode == 1 && p.tensors[op.src[0]].fmt && p.tensors[op.src[0]].f16 < 1)
                    p.tensors[op.src[0]].f16 = 1;
                if (!is_conv(op)) continue;
                if (mode == 3) {
                    if (op.k != 3 || op.stride != 1 || p.tensors[op.dst].C % 64)
                        throw std::runtime_error("precision plan of " + p.net + ": ':x' needs a 3x3 stride-1 layer with a multiple of 64 output channels (" + op.name + ")");
                    // the form exists in the direct kernel only, on planes a direct kernel's epilogue wrote: where this geometry routes the
                    // layer or its producer elsewhere (narrow images, a stand-alone pool, the stem kernel) the layer keeps three products
                    bool ok = op.kind == OP_CONV_DIRECT && op.nsplit * 64 == p.tensors[op.dst].C && !op.up[0];
                    for (int j = 0; j < op.nsrc && ok; ++j)
                        for (const OpDesc& q : p.ops)
                            if (q.dst == op.src[j] && q.kind != OP_CONV_DIRECT) ok = false;
                    if (ok) op.f16 = 3;
                } else if (op.f16 != 3) op.f16 = std::max(op.f16, mode);
            }
            if (!any) throw std::runtime_error("precision plan of " + p.net + ": no layer matches '" + t + "'");
        }
        // closure.  tensor.f16: 0 bf16 hi+lo, 1 ONE fp16 plane, 2 fp16 hi+lo (its hi plane IS format 1: every fp16 layer can read it)
        for (bool changed = true; changed;) {
            changed = false;
            for (OpDesc& op : p.ops) {
                if (is_conv(op)) {
                    bool reads16 = false;
                    for (int j = 0; j < op.nsrc; ++j) reads16 = reads16 || p.tensors[op.src[j]].f16;
                    if (reads16 && !op.f16) { op.f16 = 1; changed = true; }
                    if (op.f16)
                        for (int j = 0; j < op.nsrc; ++j) {
                            TensorDesc& t = p.tensors[op.src[j]];
                            const int want = op.f16 == 3 ? 2 : 1;
                            if (t.fmt && t.f16 < want) { t.f16 = want; changed = true; }
                        }
                } else if (op.kind == OP_POOL2 || op.kind == OP_POOL3Z) {      // a stand-alone pool keeps the format of its source
                    TensorDesc &a = p.tensors[op.src[0]], &b = p.tensors[op.dst];
                    if (a.fmt && b.fmt && a.f16 != b.f16) { a.f16 = b.f16 = std::max(a.f16, b.f16); changed = true; }
                }
            }
        }
        // an fp16 hi+lo tensor is written by a direct conv's epilogue and read by convs / per-thread heads only
        for (const OpDesc& op : p.ops) {
            if (op.dst >= 0 && p.tensors[op.dst].fmt && p.tensors[op.dst].f16 == 2 && op.kind != OP_CONV_DIRECT)
                throw std::runtime_error("precision plan of " + p.net + ": the input of a ':x' layer must come from a direct 3x3 layer (" + op.name + ")");
            if ((op.kind == OP_POOL2 || op.kind == OP_POOL3Z) && p.tensors[op.src[0]].f16 == 2)
                throw std::runtime_error("precision plan of " + p.net + ": a stand-alone pool cannot read fp16 hi+lo planes (" + op.name + ")");
        }
        if (p.net.compare(0, 9, "monodepth") == 0)
            for (OpDesc& op : p.ops)
                if (is_conv(op) && op.f16 && op.nsrc == 1 && op.src[0] == p.t_input) {
                    p.input_scale = 1.f;
                    p.weights[op.w].scale = 1.f / 255.f;
                }
        double fl = 0;
        for (OpDesc& op : p.ops) {
            if (!is_conv(op) || !op.f16) continue;
            p.f16_ops += (p.f16_ops.empty() ? "" : ",") + op.name + (op.f16 == 2 ? ":1" : op.f16 == 3 ? ":x" : "");
            fl += op.flops;
            if (op.f16 == 2) p.flops_f16x1 += op.flops / std::max(1, p.images);
            p.weights[op.w].f16 = 1;
            for (WeightSlot& s : p.weights)
                if (s.owner == op.w && (s.layout == WL_IGEMM_SPLIT || s.layout == WL_DIRECT_SPLIT)) s.f16 = 1;
        }
        p.flops_f16 = fl / std::max(1, p.images);
    }

    void finish() {
        apply_precision_plan();
        mark_planar();
        // liveness
        for (size_t i = 0; i < p.ops.size(); ++i) {
            const OpDesc& op = p.ops[i];
            auto touch = [&](int t) {
                if (t < 0) return;
                if (p.tensors[t].first < 0) p.tensors[t].first = (int)i;
                p.tensors[t].last = (int)i;
            };
            touch(op.dst);
            for (int j = 0; j < op.nsrc; ++j) touch(op.src[j]);
            touch(op.residual);
        }
        // weight arena: slots, then per-op tables
        size_t off = 0;
        for (auto& s : p.weights) {
            if (s.owner >= 0) continue;
            s.offset = off;
            const bool ig = s.layout == WL_IGEMM || s.layout == WL_IGEMM_SPLIT;
            off += align_up(ig ? (size_t)s.Ktotal * s.CoutPad * (s.x3 ? 6 : sizeof(float)) : s.bytes);
        }
        for (auto& s : p.weights) if (s.owner >= 0) s.offset = p.weights[s.owner].offset;
        for (auto& op : p.ops)
            if (op.kind == OP_CONV || op.kind == OP_CONV_DIRECT) {
                op.tab_offset = off; off += align_up(op.tab_bytes);
            }
        p.weight_bytes = off;
        // activation arena: first-fit over lifetimes (SEMDEPTH_KEEP_ACTIVATIONS=1: no reuse, for layer-by-layer tests)
        const char* keep = std::getenv("SEMDEPTH_KEEP_ACTIVATIONS");
        const bool no_reuse = keep && keep[0] == '1';
        std::vector<int> order;
        for (size_t i = 0; i < p.tensors.size(); ++i) if (p.tensors[i].first >= 0) order.push_back((int)i);
        std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return p.tensors[a].first < p.tensors[b].first; });
        std::vector<int> placed;
        size_t top = 0;
        for (int ti : order) {
            TensorDesc& t = p.tensors[ti];
            const size_t sz = align_up(t.bytes);
            std::vector<std::pair<size_t, size_t>> busy;
            for (int pj : placed) {
                const TensorDesc& o = p.tensors[pj];
                if (no_reuse || o.last >= t.first) busy.push_back({o.offset, o.offset + align_up(o.bytes)});
            }
            std::sort(busy.begin(), busy.end());
            size_t pos = 0;
            for (auto& b : busy) {
                if (pos + sz <= b.first) break;
                pos = std::max(pos, b.second);
            }
            t.offset = pos;
            top = std::max(top, pos + sz);
            placed.push_back(ti);
        }
        p.act_bytes = top;
        double fl = 0;
        for (auto& op : p.ops) fl += op.flops;
        p.flops_per_image = fl / std::max(1, p.images);
    }
};

}  // namespace

// ---------------------------------------------------------------------------------------------
NetPlan build_fcn8s(int frames, int H, int W, int prec, const char* f16_layers) {
    if (H % 32 || W % 32) throw std::runtime_error("FCN-8s needs H, W multiples of 32");
    Builder b;
    b.p.prec = prec ? 1 : 0; b.p.x3 = prec == 2; b.p.h2 = prec == 3; b.p.f16_spec = (prec == 1 && f16_layers) ? f16_layers : "";
    b.p.net = "fcn8s"; b.p.frames = frames; b.p.images = frames; b.p.H = H; b.p.W = W;
    int x = b.tensor("input_pre", frames, H, W, 4);      // 4th channel is zero: float4 gathers in conv1_1
    b.p.tensors[x].Ctf = 3;
    { OpDesc op; op.kind = OP_PRE_VGG; op.name = "pre"; op.dst = x; b.push(op); }
    b.p.t_input = x;
    const char* blocks[5][3] = {{"conv1_1", "conv1_2", nullptr}, {"conv2_1", "conv2_2", nullptr}, {"conv3_1", "conv3_2", "conv3_3"},
                                {"conv4_1", "conv4_2", "conv4_3"}, {"conv5_1", "conv5_2", "conv5_3"}};
    const int ch[5] = {64, 128, 256, 512, 512};
    int pools[5];
    for (int s = 0; s < 5; ++s) {
        for (int j = 0; j < 3 && blocks[s][j]; ++j) {
            std::string n = blocks[s][j];
            const int xin = x;
            x = b.conv(n, {{x, 0}}, ch[s], 3, 1, ACT_RELU, "vgg/" + n + "/filter", "vgg/" + n + "/biases");
            // conv1_1 (stem kernel) -> conv1_2 (direct kernel): hand the 64 channels over as four 16-channel sub-planes
            if (s == 0 && j == 1 && prec == 1 && b.p.ops.back().kind == OP_CONV_DIRECT && b.p.tensors[xin].C == 64 && W % 32 == 0 &&
                !sd_disabled("planar") && !sd_disabled("stem")) {
                b.p.tensors[xin].planar16 = 1;
            }
        }
        x = b.pool("pool" + std::to_string(s + 1), x, false);
        pools[s] = x;
    }
    b.alias("layer3_out", pools[2]);
    b.alias("layer4_out", pools[3]);
    x = b.conv("fc6", {{pools[4], 0}}, 4096, 7, 1, ACT_RELU, "vgg/fc6/filter", "vgg/fc6/biases");
    x = b.conv("fc7", {{x, 0}}, 4096, 1, 1, ACT_RELU, "vgg/fc7/filter", "vgg/fc7/biases");
    b.alias("layer7_out", x);
    const int s7 = b.smalln("score7", x, 3, 1, ACT_NONE, "dec/score7/kernel", "dec/score7/bias", 3);
    const int s4 = b.smalln("score4", pools[3], 3, 1, ACT_NONE, "dec/score4/kernel", "dec/score4/bias", 3);
    const int s3 = b.smalln("score3", pools[2], 3, 1, ACT_NONE, "dec/score3/kernel", "dec/score3/bias", 3);
    auto deconv4 = [&](const std::string& name, int src, int skip, const std::string& wn) {
        OpDesc op;
        op.kind = OP_DECONV4_ADD; op.name = name; op.nsrc = 1; op.src[0] = src; op.residual = skip;
        op.w = b.wslot("dec/" + wn + "/kernel", {4, 4, 3, 3}, WL_RAW);
        op.b = b.wslot("dec/" + wn + "/bias", {3}, WL_RAW);
        const TensorDesc& t = b.p.tensors[src];
        op.dst = b.tensor(name, t.N, t.H * 2, t.W * 2, 3);
        b.p.tensors[op.dst].fmt = 0;          // the deconv ladder is f32
        b.push(op);
        return op.dst;
    };
    const int first_skip = deconv4("first_skip", s7, s4, "deconv1");
    const int second_skip = deconv4("second_skip", first_skip, s3, "deconv2");
    {
        OpDesc op;
        op.kind = OP_HEAD16; op.name = "head"; op.nsrc = 1; op.src[0] = second_skip;
        op.w = b.wslot("dec/deconv3/kernel", {16, 16, 3, 3}, WL_RAW);
        op.b = b.wslot("dec/deconv3/bias", {3}, WL_RAW);
        b.push(op);
    }
    b.p.t_output = second_skip;
    b.finish();
    return b.p;
}

// ---------------------------------------------------------------------------------------------
NetPlan build_monodepth(int encoder, int frames, int H, int W, int prec, const char* f16_layers) {
    const int mult = encoder == 0 ? 128 : 64;
    if (H % mult || W % mult) throw std::runtime_error("monodepth needs H, W multiples of 128 (vgg) / 64 (resnet50)");
    Builder b;
    b.p.prec = prec ? 1 : 0; b.p.x3 = prec == 2; b.p.h2 = prec == 3; b.p.f16_spec = (prec == 1 && f16_layers) ? f16_layers : "";
    b.p.net = encoder == 0 ? "monodepth-vgg" : "monodepth-resnet50";
    b.p.frames = frames; b.p.images = 2 * frames; b.p.H = H; b.p.W = W;
    const int N = 2 * frames;
    int x = b.tensor("input_pre", N, H, W, 4);
    b.p.tensors[x].Ctf = 3;
    { OpDesc op; op.kind = OP_PRE_MONO; op.name = "pre"; op.dst = x; b.push(op); }
    b.p.t_input = x;
    auto cv = [&](const std::string& name, std::vector<Src> srcs, int C, int k, int s, int act = ACT_ELU, int res = -1) {
        return b.conv(name, srcs, C, k, s, act, name + "/weights", name + "/biases", res);
    };
    std::map<int, int> skips;   // decoder level -> skip tensor
    int top, enc_out;
    if (encoder == 0) {
        const int spec[7][2] = {{32, 7}, {64, 5}, {128, 3}, {256, 3}, {512, 3}, {512, 3}, {512, 3}};
        int feats[7];
        for (int i = 0; i < 7; ++i) {
            const std::string n = "enc/conv" + std::to_string(i + 1);
            x = cv(n + "a", {{x, 0}}, spec[i][0], spec[i][1], 1);
            x = cv(n + "b", {{x, 0}}, spec[i][0], spec[i][1], 2);
            feats[i] = x;
        }
        for (int lvl = 2; lvl <= 7; ++lvl) skips[lvl] = feats[lvl - 2];
        top = 7; enc_out = feats[6];
    } else {
        const int conv1 = cv("enc/conv1", {{x, 0}}, 64, 7, 2);
        const int pool1 = b.pool("enc/pool1", conv1, true);
        x = pool1;
        const int stage_n[4] = {64, 128, 256, 512}, stage_blocks[4] = {3, 4, 6, 3};
        int stage_out[4];
        for (int s = 0; s < 4; ++s) {
            for (int blk = 1; blk <= stage_blocks[s]; ++blk) {
                const std::string pfx = "enc/res" + std::to_string(s + 2) + "_" + std::to_string(blk);
                const int stride = blk == stage_blocks[s] ? 2 : 1;     // the LAST block of a stage strides
                const int n = stage_n[s];
                const int c1 = cv(pfx + "/conv1", {{x, 0}}, n, 1, 1);
                const int c2 = cv(pfx + "/conv2", {{c1, 0}}, n, 3, stride);
                // elu(conv3(c2) + proj(x)): the shortcut is always a 1x1 projection (upstream quirk), so both 1x1 convs
                // are one GEMM over the concatenated K axis; the shortcut source is read at the block's stride
                x = b.conv_pair_1x1(pfx + "/conv3", c2, x, stride, 4 * n, ACT_ELU, pfx + "/conv3/weights", pfx + "/conv3/biases",
                                    pfx + "/proj/weights", pfx + "/proj/biases");
            }
            stage_out[s] = x;
        }
        b.alias("enc/conv2", stage_out[0]); b.alias("enc/conv3", stage_out[1]);
        b.alias("enc/conv4", stage_out[2]); b.alias("enc/conv5", stage_out[3]);
        skips[6] = stage_out[2]; skips[5] = stage_out[1]; skips[4] = stage_out[0]; skips[3] = pool1; skips[2] = conv1;
        top = 6; enc_out = stage_out[3];
    }
    const int dec_ch[8] = {0, 16, 32, 64, 128, 256, 512, 512};
    x = enc_out;
    int disp_prev = -1;
    for (int lvl = top; lvl >= 1; --lvl) {
        const std::string L = std::to_string(lvl);
        if (lvl == 1 && (b.p.x3 || b.p.h2) && dec_tail1_eligible(H, W) && !(latch_switches() & SW_NO_TAIL1)) {
            disp_prev = b.dec_tail1(x, disp_prev);
            break;
        }
        const int u = cv("dec/upconv" + L, {{x, 1}}, dec_ch[lvl], 3, 1);
        std::vector<Src> cat = {{u, 0}};
        if (skips.count(lvl)) cat.push_back({skips[lvl], 0});
        if (lvl <= 3) cat.push_back({disp_prev, 1});
        x = cv("dec/iconv" + L, cat, dec_ch[lvl], 3, 1);
        if (lvl <= 4) {
            // scales 4..2 feed udisp (both channels); at scale 1 only channel 0 = disp_left_est[0] is fetched (semantic_depth.py:675)
            const int nout = lvl == 1 ? 1 : 2;
            disp_prev = b.smalln("dec/disp" + L, x, nout, 3, ACT_SIGMOID03, "dec/disp" + L + "/weights", "dec/disp" + L + "/biases", 2,
                                 /*feeds_conv=*/lvl > 1);
        }
    }
    b.p.t_output = disp_prev;
    b.finish();
    return b.p;
}

// f32 -> IEEE fp16 bits, round to nearest even, subnormals kept, overflow -> inf (the weights of the 2-product scheme)
static uint16_t f32_to_f16_rne(float v) {
    uint32_t u; std::memcpy(&u, &v, 4);
    const uint32_t sign = (u >> 16) & 0x8000u;
    u &= 0x7fffffffu;
    if (u >= 0x7f800000u) return (uint16_t)(sign | (u > 0x7f800000u ? 0x7e00u : 0x7c00u));    // NaN / inf
    if (u >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u);                                   // rounds to >= 65520 -> inf
    if (u < 0x38800000u) {                                                                     // below 2^-14: subnormal
        if (u < 0x33000000u) return (uint16_t)sign;                                            // < 2^-25 -> 0
        const int e = (int)(u >> 23);                                                          // 102..112
        uint32_t m = (u & 0x7fffffu) | 0x800000u;                                              // 24-bit significand
        const int shift = 126 - e;                                                             // 14..24: value = m * 2^(e-150), unit 2^-24
        const uint32_t q = m >> shift, rem = m & ((1u << shift) - 1u), half = 1u << (shift - 1);
        uint32_t r = q + ((rem > half || (rem == half && (q & 1u))) ? 1u : 0u);
        return (uint16_t)(sign | r);
    }
    uint32_t r = u - 0x38000000u;                                                              // rebias 127 -> 15
    const uint32_t rem = r & 0x1fffu;
    r >>= 13;
    if (rem > 0x1000u || (rem == 0x1000u && (r & 1u))) ++r;
    return (uint16_t)(sign | r);
}

static float f16_bits_to_f32(uint16_t h) {
    const uint32_t sign = (uint32_t)(h & 0x8000u) << 16, e = (h >> 10) & 0x1fu, m = h & 0x3ffu;
    uint32_t u;
    if (e == 0) {
        if (m == 0) u = sign;
        else {                                   // subnormal: m * 2^-24
            float f = (float)m * 5.9604644775390625e-08f;
            std::memcpy(&u, &f, 4);
            u |= sign;
        }
    } else if (e == 31) u = sign | 0x7f800000u | (m << 13);
    else u = sign | ((e + 112u) << 23) | (m << 13);
    float f; std::memcpy(&f, &u, 4);
    return f;
}
// the two fp16 weight planes of the 2-product scheme (split_fmt.hpp): w_hi = RNE(w), w_lo = RNE(w - w_hi) (22 bits together; w_lo of
// an ordinary weight is an fp16 subnormal, which the MFMA honours)
static inline void f16_split(float w, uint16_t& hi, uint16_t& lo) {
    hi = f32_to_f16_rne(w);
    lo = f32_to_f16_rne(w - f16_bits_to_f32(hi));
}

// ---------------------------------------------------------------------------------------------
void relayout_weight(const WeightSlot& s, const float* w, std::vector<float>& out) {
    out.assign((s.bytes + sizeof(float) - 1) / sizeof(float), 0.f);
    if (s.layout == WL_IGEMM || s.layout == WL_IGEMM_SPLIT) {
        const bool split = s.layout == WL_IGEMM_SPLIT;
        uint16_t* hi = reinterpret_cast<uint16_t*>(out.data());
        uint16_t* lo = hi + (size_t)s.Kpad * s.CoutPad;             // (bf16 x 3: the mid plane; the lo plane follows it)
        uint16_t* lo3 = lo + (size_t)s.Kpad * s.CoutPad;
        auto bf16 = [](float v) -> uint16_t {          // round to nearest even, like v_cvt_pk_bf16_f32
            uint32_t u; std::memcpy(&u, &v, 4);
            u += 0x7FFFu + ((u >> 16) & 1u);
            return (uint16_t)(u >> 16);
        };
        const bool f16 = s.f16 != 0;                   // two fp16 planes: hi = fp16(w) (RNE, subnormals kept), lo = fp16(w - hi)
        auto bf16_to_f = [](uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; std::memcpy(&f, &u, 4); return f; };
        if (s.fold) {
            // upsample-folded 3x3 (OpDesc::fold): parity q = 2 py + px, tap t = 2 a + b reads source pixel (i + a - 1 + py, j + b - 1 + px); the
            // 3x3 taps dy that land on source row a of parity py are  py = 0: a = 0 <- {0}, a = 1 <- {1, 2};  py = 1: a = 0 <- {0, 1}, a = 1 <- {2}
            // (the same for columns).  The taps are added in double and rounded once to f32 -- the grade of every other f32 operation of
            // the layer -- and that f32 value is split exactly into its three bf16 planes like any other weight.
            // K order inside a parity panel: (32-channel block, tap, channel), as the vec region of an ordinary layer.
            const int64_t C = s.shape[2], Cout = s.shape[3];
            const int64_t Kp = s.Kpad / 4;              // rows of one parity panel (4 C)
            auto rows_of = [](int par, int a, int* d) { if (par == 0) { if (a == 0) { d[0] = 0; return 1; } d[0] = 1; d[1] = 2; return 2; }
                                                         if (a == 0) { d[0] = 0; d[1] = 1; return 2; } d[0] = 2; return 1; };
            for (int q = 0; q < 4; ++q)
                for (int tp = 0; tp < 4; ++tp) {
                    int dys[2], dxs[2];
                    const int ny = rows_of(q >> 1, tp >> 1, dys), nx = rows_of(q & 1, tp & 1, dxs);
                    for (int64_t c = 0; c < C; ++c) {
                        const int64_t k = q * Kp + ((c / 32) * 4 + tp) * 32 + c % 32;
                        const size_t base = (size_t)(k / 8) * s.CoutPad * 8 + (k % 8);
                        for (int64_t n = 0; n < Cout; ++n) {
                            double acc = 0.0;
                            for (int iy = 0; iy < ny; ++iy)
                                for (int ix = 0; ix < nx; ++ix) acc += (double)w[((int64_t)(dys[iy] * 3 + dxs[ix]) * C + c) * Cout + n];
                            const float wf = (float)acc;
                            if (f16) { f16_split(s.hs ? wf * s.wscale : wf, hi[base + n * 8], lo[base + n * 8]); continue; }      // (SD_PREC_F16X2: planes of w * 2^k)
                            const uint16_t h = bf16(wf);
                            hi[base + n * 8] = h;
                            const float r1 = wf - bf16_to_f(h);
                            lo[base + n * 8] = bf16(r1);
                            if (s.x3) lo3[base + n * 8] = bf16(r1 - bf16_to_f(lo[base + n * 8]));
                        }
                    }
                }
            return;
        }
        const int64_t taps = s.shape[0] * s.shape[1], Ctf = s.shape[2], Cout = s.shape[3];
        int CtotPad = 0;
        for (int i = 0; i < s.nsrc; ++i) CtotPad += s.srcCpad[i];
        int Cvec = 0, CqPad = 0;
        for (int i = 0; i < s.nsrc; ++i) { if (s.srcVec[i]) Cvec += s.srcCpad[i]; else CqPad += s.srcCpad[i]; }
        const int64_t Kvec = taps * Cvec;
        (void)CtotPad;
        for (int64_t tap = 0; tap < taps; ++tap) {
            int cb_tf = 0, cv = 0, cq = 0;
            for (int i = 0; i < s.nsrc; ++i) {
                for (int c = 0; c < s.srcCtf[i]; ++c) {
                    // vec sources walk K as (32-channel block, tap, channel): the taps of one channel block are adjacent
                    // k-tiles, so a block re-reads the same input rows from L1/L2; the other sources form a (tap, quad) tail
                    int64_t k;
                    if (s.srcVec[i]) { const int64_t cp = cv + c; k = ((cp / 32) * taps + tap) * 32 + cp % 32; }
                    else k = Kvec + tap * CqPad + cq + c;
                    const float* src = w + (tap * Ctf + cb_tf + c) * Cout;
                    if (!split) {
                        float* dst = out.data() + (size_t)(k / 4) * s.CoutPad * 4 + (k % 4);
                        for (int64_t n = 0; n < Cout; ++n) dst[n * 4] = src[n];
                    } else {                          // two bf16 planes [k/8][n][8]
                        const size_t base = (size_t)(k / 8) * s.CoutPad * 8 + (k % 8);
                        for (int64_t n = 0; n < Cout; ++n) {
                            if (f16) { f16_split(s.hs ? src[n] * s.wscale : src[n], hi[base + n * 8], lo[base + n * 8]); continue; }      // (hs: planes of w * 2^k)
                            const uint16_t h = bf16(src[n]);
                            hi[base + n * 8] = h;
                            const float r1 = src[n] - bf16_to_f(h);
                            lo[base + n * 8] = bf16(r1);
                            if (s.x3) lo3[base + n * 8] = bf16(r1 - bf16_to_f(lo[base + n * 8]));        // exact: w = hi + mid + lo
                        }
                    }
                }
                cb_tf += s.srcCtf[i];
                if (s.srcVec[i]) cv += s.srcCpad[i]; else cq += s.srcCpad[i];
            }
        }
    } else if (s.layout == WL_DIRECT_SPLIT) {
        // [split][plane][chunk][tap 9][octet 2][CoutPad n][8]: chunk = 16 stored channels of one source
        const int64_t Ctf = s.shape[2], Cout = s.shape[3];
        const size_t plane = (size_t)s.Kpad * s.CoutPad;
        uint16_t* const hi0 = reinterpret_cast<uint16_t*>(out.data());
        auto bf16 = [](float v) -> uint16_t { uint32_t u; std::memcpy(&u, &v, 4); u += 0x7FFFu + ((u >> 16) & 1u); return (uint16_t)(u >> 16); };
        auto bf16_to_f = [](uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; std::memcpy(&f, &u, 4); return f; };
        if (s.fold) {
            // upsample-folded (OpDesc::fold): [split][plane][chunk][parity q][2x2 tap t][octet][CoutPad n][8]; the 3x3 taps that land on source
            // row a of parity py are  py = 0: a = 0 <- {0}, a = 1 <- {1, 2};  py = 1: a = 0 <- {0, 1}, a = 1 <- {2}  (columns alike), added in
            // double and rounded once to f32, then split exactly like any other weight
            auto rows_of = [](int par, int a, int* d) { if (par == 0) { if (a == 0) { d[0] = 0; return 1; } d[0] = 1; d[1] = 2; return 2; }
                                                         if (a == 0) { d[0] = 0; d[1] = 1; return 2; } d[0] = 2; return 1; };
            for (int c0 = 0, chunk = 0; c0 < s.srcCpad[0]; c0 += 16, ++chunk)
                for (int q = 0; q < 4; ++q)
                    for (int tp = 0; tp < 4; ++tp) {
                        int dys[2], dxs[2];
                        const int ny = rows_of(q >> 1, tp >> 1, dys), nx = rows_of(q & 1, tp & 1, dxs);
                        for (int c = c0; c < std::min(c0 + 16, s.srcCtf[0]); ++c) {
                            const int oct = (c - c0) / 8, e = (c - c0) % 8;
                            const size_t base = ((((size_t)chunk * 4 + q) * 4 + tp) * 2 + oct) * s.CoutPad * 8 + e;
                            for (int64_t n = 0; n < Cout; ++n) {
                                double acc = 0.0;
                                for (int iy = 0; iy < ny; ++iy)
                                    for (int ix = 0; ix < nx; ++ix) acc += (double)w[((int64_t)(dys[iy] * 3 + dxs[ix]) * Ctf + c) * Cout + n];
                                const float wf = (float)acc;
                                uint16_t* hi = hi0 + (size_t)(n / s.CoutPad) * (s.x3 ? 3 : 2) * plane + base + (n % s.CoutPad) * 8;
                                if (s.f16) { f16_split(s.hs ? wf * s.wscale : wf, *hi, hi[plane]); continue; }       // (SD_PREC_F16X2: two fp16 planes of w * 2^k)
                                const uint16_t h = bf16(wf);
                                *hi = h;
                                const float r1 = wf - bf16_to_f(h);
                                hi[plane] = bf16(r1);
                                hi[2 * plane] = bf16(r1 - bf16_to_f(hi[plane]));
                            }
                        }
                    }
            return;
        }
        int chunk = 0, cb_tf = 0;
        for (int i = 0; i < s.nsrc; ++i) {
            for (int c0 = 0; c0 < s.srcCpad[i]; c0 += 16, ++chunk)
                for (int tap = 0; tap < 9; ++tap)
                    for (int c = c0; c < std::min(c0 + 16, s.srcCtf[i]); ++c) {
                        const int oct = (c - c0) / 8, e = (c - c0) % 8;
                        const float* src = w + ((int64_t)tap * Ctf + cb_tf + c) * Cout;
                        const size_t base = (((size_t)chunk * 9 + tap) * 2 + oct) * s.CoutPad * 8 + e;
                        for (int64_t n = 0; n < Cout; ++n) {
                            uint16_t* hi = hi0 + (size_t)(n / s.CoutPad) * (s.x3 ? 3 : 2) * plane + base + (n % s.CoutPad) * 8;
                            if (s.f16) { f16_split(s.hs ? src[n] * s.wscale : src[n], *hi, hi[plane]); continue; }
                            const uint16_t h = bf16(src[n]);
                            *hi = h;
                            const float r1 = src[n] - bf16_to_f(h);
                            hi[plane] = bf16(r1);
                            if (s.x3) hi[2 * plane] = bf16(r1 - bf16_to_f(hi[plane]));                    // exact: w = hi + mid + lo
                        }
                    }
            cb_tf += s.srcCtf[i];
        }
    } else if (s.layout == WL_TAIL_UP || s.layout == WL_TAIL_ICONV) {
        // MFMA A fragments of dec_tail.hip (v_mfma_f32_16x16x32_bf16: lane l = output channel l & 15, k group l >> 4 = eight consecutive k),
        // three bf16 planes each: out = u32x4 [fragment][plane][lane]
        auto bf16 = [](float v) -> uint16_t { uint32_t u; std::memcpy(&u, &v, 4); u += 0x7FFFu + ((u >> 16) & 1u); return (uint16_t)(u >> 16); };
        auto bf16_to_f = [](uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; std::memcpy(&f, &u, 4); return f; };
        uint16_t* const o16 = reinterpret_cast<uint16_t*>(out.data());
        auto put = [&](int frag, int lane, int e, float v) {             // element e of the lane's eight, all three planes
            if (s.hs) {              // SD_PREC_F16X2: fp16 hi + lo of w * 2^k in planes 0 and 1 (the kernel forms w_hi * 2^-11 in the third slot)
                uint16_t h16, l16;
                f16_split(v * s.wscale, h16, l16);
                o16[(((size_t)frag * 3 + 0) * 64 + lane) * 8 + e] = h16;
                o16[(((size_t)frag * 3 + 1) * 64 + lane) * 8 + e] = l16;
                return;
            }
            const uint16_t h = bf16(v);
            const float r1 = v - bf16_to_f(h);
            const uint16_t m = bf16(r1);
            const uint16_t l = bf16(r1 - bf16_to_f(m));
            o16[(((size_t)frag * 3 + 0) * 64 + lane) * 8 + e] = h;
            o16[(((size_t)frag * 3 + 1) * 64 + lane) * 8 + e] = m;
            o16[(((size_t)frag * 3 + 2) * 64 + lane) * 8 + e] = l;
        };
        const int64_t C = s.shape[2], Cout = s.shape[3];
        if (s.layout == WL_TAIL_UP) {
            // upsample-folded upconv1 (see the fold branch of WL_IGEMM_SPLIT above): fragment (parity q, tap a b) holds Wf[q][a][b][c = 8 kg + e][n]
            auto rows_of = [](int par, int a, int* d) { if (par == 0) { if (a == 0) { d[0] = 0; return 1; } d[0] = 1; d[1] = 2; return 2; }
                                                         if (a == 0) { d[0] = 0; d[1] = 1; return 2; } d[0] = 2; return 1; };
            for (int q = 0; q < 4; ++q)
                for (int tp = 0; tp < 4; ++tp) {
                    int dys[2], dxs[2];
                    const int ny = rows_of(q >> 1, tp >> 1, dys), nx = rows_of(q & 1, tp & 1, dxs);
                    for (int lane = 0; lane < 64; ++lane)
                        for (int e = 0; e < 8; ++e) {
                            const int n = lane & 15, c = 8 * (lane >> 4) + e;
                            double acc = 0.0;
                            for (int iy = 0; iy < ny; ++iy)
                                for (int ix = 0; ix < nx; ++ix) acc += (double)w[((int64_t)(dys[iy] * 3 + dxs[ix]) * C + c) * Cout + n];
                            put(q * 4 + tp, lane, e, (float)acc);
                        }
                }
        } else {
            // iconv1 on concat(u 16, up2(disp2) 2): fragment (row block dy, half): half 0 = [u(x-1) octets 0 1 | u(x) octets 0 1],
            // half 1 = [u(x+1) octets 0 1 | disp2 (dx -1, 0, +1) x 2 channels + two zeros | zeros]
            for (int dy = 0; dy < 3; ++dy)
                for (int hf = 0; hf < 2; ++hf)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int e = 0; e < 8; ++e) {
                            const int n = lane & 15, sl = lane >> 4;
                            float v = 0.f;
                            if (hf == 0) v = w[((int64_t)(dy * 3 + (sl >> 1)) * C + 8 * (sl & 1) + e) * Cout + n];
                            else if (sl < 2) v = w[((int64_t)(dy * 3 + 2) * C + 8 * sl + e) * Cout + n];
                            else if (sl == 2 && e < 6) v = w[((int64_t)(dy * 3 + (e >> 1)) * C + 16 + (e & 1)) * Cout + n];
                            put(dy * 2 + hf, lane, e, v);
                        }
        }
    } else if (s.layout == WL_SMALLN) {
        const int64_t K = s.shape[0] * s.shape[1] * s.shape[2], Cout = s.shape[3];
        for (int64_t k = 0; k < K; ++k)
            for (int j = 0; j < s.nout; ++j) out[(size_t)j * K + k] = w[k * Cout + j];     // [nout][K]
    } else if (s.layout == WL_BIAS4) {
        for (int j = 0; j < s.nout; ++j) out[j] = w[j];
    } else {
        std::memcpy(out.data(), w, s.bytes);
    }
}

void build_conv_tables(const NetPlan& p, const OpDesc& op, const char* act_base, const char* tab_dev, std::vector<KEntry>& ktab) {
    if (op.fold) {      // four parity tables of Kpad / 32 entries: k-tile = (32-channel block, 2x2 tap) on the source at its own resolution
        const TensorDesc& t = p.tensors[op.src[0]];
        const int kt_n = op.Kpad / 32;
        ktab.assign((size_t)4 * kt_n, KEntry{nullptr, 1, 1, 1, 0, 0, 0});
        for (int q = 0; q < 4; ++q)
            for (int kt = 0; kt < kt_n; ++kt) {
                const int cb = kt / 4, tp = kt % 4;
                KEntry& e = ktab[(size_t)q * kt_n + kt];
                e.base = reinterpret_cast<const float*>(act_base + t.offset + (size_t)cb * 32 * 2);
                e.H = t.H; e.W = t.W; e.C = t.C;
                e.dy = (tp >> 1) - 1 + (q >> 1); e.dx = (tp & 1) - 1 + (q & 1);
                e.flags = (1 << 4) | (4 << 8) | 0x10000;
            }
        return;
    }
    const int taps = op.k * op.k;
    // channel runs: vec sources concatenated (cv), the other sources concatenated with quad padding (cq)
    struct Run { int src, c0; };
    auto describe = [&](int s, int cl, int tap) {
        const TensorDesc& t = p.tensors[op.src[s]];
        KEntry e{nullptr, 1, 1, 1, 0, 0, 0};
        e.base = reinterpret_cast<const float*>(act_base + t.offset + (size_t)cl * (t.fmt ? 2 : 4));   // hi plane when split
        e.H = t.H; e.W = t.W; e.C = t.C;
        e.dy = tap / op.k - op.pad; e.dx = tap % op.k - op.pad;
        const int nv = std::max(0, std::min(4, t.C - cl));
        e.flags = (op.up[s] ? 1 : 0) | (op.sstride[s] << 4) | (nv << 8) | (nv > 0 ? 0x10000 : 0);
        return e;
    };
    auto locate = [&](int c, bool vecpart, int& s, int& cl) {     // c indexes the vec (or quad) concatenation
        int base = 0;
        for (s = 0; s < op.nsrc; ++s) {
            const TensorDesc& t = p.tensors[op.src[s]];
            const bool v = t.C % 32 == 0;
            if (v != vecpart) continue;
            const int cpad = (t.C + 3) / 4 * 4;
            if (c < base + cpad) { cl = c - base; return true; }
            base += cpad;
        }
        return false;
    };
    const int ktiles = op.Kpad / 32, vtiles = op.Kvec / 32, qtiles = ktiles - vtiles;
    ktab.assign((size_t)ktiles + (size_t)qtiles * 8, KEntry{nullptr, 1, 1, 1, 0, 0, 0});
    for (int kt = 0; kt < vtiles; ++kt) {            // vec region: k-tile = (channel block cb, tap)
        const int cb = kt / taps, tap = kt % taps;
        int s, cl;
        if (locate(cb * 32, true, s, cl)) ktab[kt] = describe(s, cl, tap);
    }
    for (int qt = 0; qt < qtiles; ++qt) {            // quad tail: (tap, channel quad) order, 8 quads per k-tile
        KEntry& head = ktab[vtiles + qt];
        head.flags = 0x20000;                        // "quad tile": base points at its 8 quad descriptors
        head.base = reinterpret_cast<const float*>(tab_dev + ((size_t)ktiles + (size_t)qt * 8) * sizeof(KEntry));
        for (int j = 0; j < 8; ++j) {
            const int kq = (qt * 8 + j) * 4;         // offset inside the quad tail
            if (op.CqPad == 0 || kq >= taps * op.CqPad) continue;
            const int tap = kq / op.CqPad, c = kq % op.CqPad;
            int s, cl;
            if (locate(c, false, s, cl)) ktab[(size_t)ktiles + (size_t)qt * 8 + j] = describe(s, cl, tap);
        }
    }
}

void build_direct_chunks(const NetPlan& p, const OpDesc& op, const char* act_base, std::vector<DirectChunk>& chunks) {
    chunks.clear();
    for (int i = 0; i < op.nsrc; ++i) {
        const TensorDesc& t = p.tensors[op.src[i]];
        for (int c0 = 0; c0 < t.C; c0 += 16) {
            DirectChunk ch;
            ch.base = act_base + t.offset + (size_t)c0 * 2;      // hi plane, bf16 elements
            ch.H = t.H; ch.W = t.W; ch.C = t.C; ch.up = op.up[i];
            ch.nvalid = std::min(2, (t.C - c0) / 8);
            ch.pad = 0;
            if (t.planar16) {     // sub-plane c0/16 of [C/16][Nmax][H][W][16]: pixel stride 16, `pad` carries the tensor's channel count
                ch.base = act_base + t.offset + (size_t)(c0 / 16) * p.images * t.H * t.W * 16 * 2;
                ch.C = 16; ch.pad = t.C;
            }
            chunks.push_back(ch);
        }
    }
}

}  // namespace sd
