// Road-width tail on gfx950 (HBM / latency bound; no MFMA): the reference's "hand-made point cloud library"
// (semantic_depth_lib/pcl.py) and the two Open3D outlier filters of semantic_depth.py:227-245, for B frames at once.
// SURVEY.md §2.2 rows K20-K24.  Compiled with -ffp-contract=off: every comparison that selects points must
// round exactly like numpy / Open3D do on the CPU.
//
// Layout: a cloud is xyz f32 [cap][3] (+ optional rgb u8 [cap][3]) with its size in a DEVICE int32; frame b
// lives at offset b*cap.  The reductions (medians, moments, arg-min/max) run one 1024-thread workgroup per
// frame with wave ballots / shuffles inside; the Open3D filters run one thread per point over a uniform grid.
// All filters keep the input row order (ordered compaction), which the reference's "first min / first max"
// end-point pick depends on (pcl.py:307-311, semantic_depth.py:259).
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include "kernels.hpp"

namespace sd {

constexpr int TB = 1024;          // threads of a per-frame workgroup
constexpr int NW = TB / 64;       // waves per workgroup

// ------------------------------------------------------------------------------------------ helpers
__device__ __forceinline__ unsigned f2key(float v) {       // monotone float -> uint
    unsigned u = __float_as_uint(v);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key2f(unsigned k) {
    unsigned u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
    return __uint_as_float(u);
}

struct Lds {
    unsigned hist[256];
    unsigned sh[8];
    int wsum[4][NW];
    int base;
    double dred[NW];
    float fred[NW];
    int ired[NW];
    unsigned ured[NW];
};

// rank-th smallest (0-based) of val(0..n-1); 8-bit MSB-first radix select, wave-aggregated LDS histogram
template <class F>
__device__ float block_select(F val, int n, unsigned rank, Lds& L) {
    const int tid = threadIdx.x, lane = tid & 63;
    unsigned prefix = 0, mask = 0;
    for (int pass = 3; pass >= 0; --pass) {
        for (int i = tid; i < 256; i += TB) L.hist[i] = 0;
        __syncthreads();
        const int shift = pass * 8;
        constexpr int U = 4;       // independent loads in flight per thread (the passes are latency bound otherwise)
        for (int i0 = 0; i0 < n; i0 += TB * U) {
            unsigned keys[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int i = i0 + u * TB + tid;
                keys[u] = i < n ? f2key(val(i)) : 0u;
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int i = i0 + u * TB + tid;
                const unsigned k = keys[u];
                bool valid = i < n && (k & mask) == prefix;
                const unsigned bin = (k >> shift) & 255u;
                // a few rounds of wave aggregation (the top bytes of a coordinate column take only a handful of values)
                unsigned long long act = __ballot(valid);
                for (int it = 0; it < 4 && act; ++it) {
                    const int leader = __ffsll((long long)act) - 1;
                    const unsigned b0 = __shfl(bin, leader);
                    const unsigned long long m = __ballot(valid && bin == b0);
                    if (lane == leader) atomicAdd(&L.hist[b0], (unsigned)__popcll(m));
                    if (bin == b0) valid = false;
                    act = __ballot(valid);
                }
                if (valid) atomicAdd(&L.hist[bin], 1u);
            }
        }
        __syncthreads();
        if (tid < 64) {            // first bin whose cumulative count exceeds rank: 4 bins per lane, shuffle scan over the wave
            const unsigned h0 = L.hist[4 * tid], h1 = L.hist[4 * tid + 1], h2 = L.hist[4 * tid + 2], h3 = L.hist[4 * tid + 3];
            const unsigned own = h0 + h1 + h2 + h3;
            unsigned incl = own;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const unsigned t = __shfl_up(incl, o); if (tid >= o) incl += t; }
            const unsigned excl = incl - own;
            const unsigned long long hit = __ballot(incl > rank);
            const int first = hit ? __ffsll((long long)hit) - 1 : 63;          // (rank < n always: some lane hits)
            if (tid == first) {
                unsigned cum = excl;
                int b = 4 * tid;
                if (cum + h0 > rank) { }
                else if (cum + h0 + h1 > rank) { cum += h0; b += 1; }
                else if (cum + h0 + h1 + h2 > rank) { cum += h0 + h1; b += 2; }
                else { cum += h0 + h1 + h2; b += 3; }
                L.sh[0] = (unsigned)b;
                L.sh[1] = rank - cum;
            }
        }
        __syncthreads();
        prefix |= L.sh[0] << shift;
        mask |= 0xFFu << shift;
        rank = L.sh[1];
        __syncthreads();
    }
    return key2f(prefix);
}

__device__ __forceinline__ int block_sum_int(int v, Lds& L) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) L.ired[threadIdx.x >> 6] = v;
    __syncthreads();
    int s = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) s += L.ired[w];
    return s;
}
__device__ __forceinline__ double block_sum_f64(double v, Lds& L) {   // fixed order -> deterministic
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) L.dred[threadIdx.x >> 6] = v;
    __syncthreads();
    double s = 0.0;
#pragma unroll
    for (int w = 0; w < NW; ++w) s += L.dred[w];
    return s;
}
__device__ __forceinline__ unsigned block_min_u32(unsigned v, Lds& L) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { unsigned t = __shfl_xor(v, o); v = t < v ? t : v; }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) L.ured[threadIdx.x >> 6] = v;
    __syncthreads();
    unsigned s = 0xFFFFFFFFu;
#pragma unroll
    for (int w = 0; w < NW; ++w) s = L.ured[w] < s ? L.ured[w] : s;
    return s;
}

// np.median of val(0..n-1) (float32 semantics: odd -> middle, even -> (a+b)/2 in f32; any NaN -> NaN)
template <class F>
__device__ float block_median(F val, int n, Lds& L) {
    if (n <= 0) return __uint_as_float(0x7fc00000u);
    int nan_local = 0;
    for (int i0 = threadIdx.x; i0 < n; i0 += TB * 4) {
        float v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = i0 + u * TB < n ? val(i0 + u * TB) : 0.f;
#pragma unroll
        for (int u = 0; u < 4; ++u) nan_local |= (v[u] != v[u]);
    }
    if (block_sum_int(nan_local, L) > 0) return __uint_as_float(0x7fc00000u);
    if (n & 1) return block_select(val, n, (unsigned)(n / 2), L);
    const float a = block_select(val, n, (unsigned)(n / 2 - 1), L);
    // b = element of rank n/2: a again if enough copies <= a, else the smallest element > a
    int le = 0;
    unsigned nxt = 0xFFFFFFFFu;
    const unsigned ka = f2key(a);
    for (int i0 = threadIdx.x; i0 < n; i0 += TB * 4) {
        unsigned k[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) k[u] = i0 + u * TB < n ? f2key(val(i0 + u * TB)) : 0xFFFFFFFFu;
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (i0 + u * TB < n) { if (k[u] <= ka) ++le; else nxt = k[u] < nxt ? k[u] : nxt; }
    }
    const int cnt_le = block_sum_int(le, L);
    const unsigned kmin = block_min_u32(nxt, L);
    const float b = (cnt_le >= n / 2 + 1) ? a : key2f(kmin);
    return (a + b) / 2.0f;
}

// np.median again, in ~1.3 passes over the column instead of ~6: a systematic sample of 4096 values brackets the
// median ranks (select on the sample in LDS), ONE pass over the column counts what lies below the bracket and collects
// the keys inside it into LDS, and the exact order statistics are selected from that buffer.  Exact by construction;
// if the bracket misses (adversarial order) or overflows, the full radix select above runs instead.
constexpr int MED_S = 4096;         // sample size
constexpr int MED_D = 160;          // bracket half-width in sample ranks (5 sigma of the sample-rank error at the median)
constexpr int MED_CAP = 16384;      // keys collected inside the bracket (expected ~8 % of n <= 14 k at n = 175 k)
struct MedLds {
    unsigned samp[MED_S];
    unsigned buf[MED_CAP];
    unsigned cnt, below, nan;
};
template <class F>
__device__ float block_median_fast(F val, int n, Lds& L, MedLds& M) {
    if (n <= 0) return __uint_as_float(0x7fc00000u);
    if (n < 4 * MED_S) return block_median(val, n, L);
    const int tid = threadIdx.x;
    for (int j = tid; j < MED_S; j += TB) M.samp[j] = f2key(val((int)(((long)j * n) / MED_S)));
    if (tid == 0) { M.cnt = 0; M.below = 0; M.nan = 0; }
    __syncthreads();
    const unsigned r1 = (unsigned)((n - 1) / 2), r2 = (unsigned)(n / 2);       // the two middle ranks (equal when n is odd)
    const int ts = (int)(((long)r1 * MED_S) / n);
    const int slo = max(ts - MED_D, 0), shi = min(ts + MED_D, MED_S - 1);
    auto sval = [&](int j) { return key2f(M.samp[j]); };
    const unsigned klo = f2key(block_select(sval, MED_S, (unsigned)slo, L));
    const unsigned khi = f2key(block_select(sval, MED_S, (unsigned)shi, L));
    // one pass: count keys below the bracket, collect the keys inside it
    unsigned below = 0, nanf = 0;
    for (int i0 = tid; i0 < n; i0 += TB * 4) {
        float v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = i0 + u * TB < n ? val(i0 + u * TB) : 0.f;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (i0 + u * TB >= n) continue;
            nanf |= (v[u] != v[u]);
            const unsigned k = f2key(v[u]);
            if (k < klo) ++below;
            else if (k <= khi) {
                const unsigned pos = atomicAdd(&M.cnt, 1u);
                if (pos < MED_CAP) M.buf[pos] = k;
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { below += __shfl_xor(below, o); nanf |= __shfl_xor(nanf, o); }
    if ((tid & 63) == 0) { atomicAdd(&M.below, below); if (nanf) atomicOr(&M.nan, 1u); }
    __syncthreads();
    const unsigned cnt = M.cnt, bel = M.below;
    if (M.nan) return __uint_as_float(0x7fc00000u);
    if (cnt > MED_CAP || r1 < bel || r2 >= bel + cnt) return block_median(val, n, L);       // bracket missed: exact fallback
    auto bval = [&](int j) { return key2f(M.buf[j]); };
    const float a = block_select(bval, (int)cnt, r1 - bel, L);
    if (r1 == r2) return a;
    const float b = block_select(bval, (int)cnt, r2 - bel, L);
    return (a + b) / 2.0f;
}

// ordered compaction of one frame: keeps rows with pred(i, x, y, z); in and out may alias
template <class P>
__device__ void block_compact(const float* __restrict__ xyz, const uint8_t* __restrict__ rgb, int n, float* oxyz,
                              uint8_t* orgb, int32_t* n_out, int cap, P pred, Lds& L) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) L.base = 0;
    __syncthreads();
    constexpr int U = 4;           // row chunks per iteration: 4 independent loads per thread, one barrier pair per 4 chunks
    for (int i0 = 0; i0 < n; i0 += TB * U) {
        float x[U], y[U], z[U];
        uint8_t c0[U], c1[U], c2[U];
        bool keep[U];
        unsigned long long m[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = i0 + u * TB + tid;
            x[u] = y[u] = z[u] = 0.f; c0[u] = c1[u] = c2[u] = 0;
            if (i < n) {
                x[u] = xyz[(size_t)i * 3]; y[u] = xyz[(size_t)i * 3 + 1]; z[u] = xyz[(size_t)i * 3 + 2];
                if (rgb) { c0[u] = rgb[(size_t)i * 3]; c1[u] = rgb[(size_t)i * 3 + 1]; c2[u] = rgb[(size_t)i * 3 + 2]; }
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = i0 + u * TB + tid;
            keep[u] = i < n && pred(i, x[u], y[u], z[u]);
            m[u] = __ballot(keep[u]);
            if (lane == 0) L.wsum[u][wave] = __popcll(m[u]);
        }
        __syncthreads();           // all rows of these chunks are in registers; wave totals visible
        int run = L.base, tot_all = 0;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            int pos = run + __popcll(m[u] & ((1ull << lane) - 1ull));
            int tot = 0;
#pragma unroll
            for (int w = 0; w < NW; ++w) { const int c = L.wsum[u][w]; if (w < wave) pos += c; tot += c; }
            if (keep[u] && pos < cap) {
                oxyz[(size_t)pos * 3] = x[u]; oxyz[(size_t)pos * 3 + 1] = y[u]; oxyz[(size_t)pos * 3 + 2] = z[u];
                if (orgb) { orgb[(size_t)pos * 3] = c0[u]; orgb[(size_t)pos * 3 + 1] = c1[u]; orgb[(size_t)pos * 3 + 2] = c2[u]; }
            }
            run += tot; tot_all += tot;
        }
        __syncthreads();
        if (tid == 0) L.base += tot_all;
        __syncthreads();
    }
    if (tid == 0) *n_out = L.base;
}

#define FRAME_VIEW()                                                                          \
    const int b = blockIdx.x;                                                                 \
    const float* xyz = in.xyz + (size_t)b * cap * 3;                                          \
    const uint8_t* rgb = in.rgb ? in.rgb + (size_t)b * cap * 3 : nullptr;                     \
    const int n = min(in.n[b], cap);                                                          \
    float* oxyz = out.xyz + (size_t)b * cap * 3;                                              \
    uint8_t* orgb = (out.rgb && in.rgb) ? out.rgb + (size_t)b * cap * 3 : nullptr;            \
    int32_t* on = out.n + b;                                                                  \
    __shared__ Lds L;

// ------------------------------------------------------------------------------------------ multi-block compaction
// A frame's ordered compaction by ONE workgroup (block_compact above) moves 2 MB at ~12 GB/s.  When input and output do not
// alias, the same result comes from CMP_G workgroups per frame in two launches: every block counts the rows it keeps in
// its contiguous slice, then writes them at (kept rows of the slices before it) + (its own ballot prefix).  The filters
// below compute their per-frame statistics with one workgroup as before (COMPACT = false: statistics only, written to the
// frame's parameter record) and leave the data movement to these two kernels.
constexpr int CMP_G = 64;            // slices (workgroups) per frame
constexpr int CMP_PARAMS = 8;        // doubles per frame: MAD {median, mad}; plane {C0, C1, C2}; statistical filter {threshold}
struct MedG { unsigned klo, khi, cnt, below, nan, done; float med, madv; };      // per-frame state of a multi-block median
struct CmpScratch { int* blk_cnt; double* params; MedG* med; unsigned* medbuf; };
size_t cmp_scratch_bytes(int B) {
    return (size_t)B * (CMP_G * sizeof(int) + CMP_PARAMS * sizeof(double) + sizeof(MedG) + (size_t)MED_CAP * sizeof(unsigned)) + 1024;
}
static CmpScratch cmp_carve(void* base, int B) {
    CmpScratch c;
    c.params = reinterpret_cast<double*>(base);
    char* q = reinterpret_cast<char*>(base) + ((size_t)B * CMP_PARAMS * sizeof(double) + 255) / 256 * 256;
    c.blk_cnt = reinterpret_cast<int*>(q);
    q += ((size_t)B * CMP_G * sizeof(int) + 255) / 256 * 256;
    c.med = reinterpret_cast<MedG*>(q);
    q += ((size_t)B * sizeof(MedG) + 255) / 256 * 256;
    c.medbuf = reinterpret_cast<unsigned*>(q);
    return c;
}
__device__ __forceinline__ void cmp_slice(int n, int g, int& lo, int& hi) {
    const int S = ((n + CMP_G - 1) / CMP_G + 255) / 256 * 256;
    lo = min(n, g * S); hi = min(n, lo + S);
}
template <class P>
__global__ __launch_bounds__(256) void cmp_count_kernel(CloudView in, int cap, P pred, int* blk_cnt) {
    const int b = blockIdx.y, g = blockIdx.x;
    const int n = min(in.n[b], cap);
    const float* xyz = in.xyz + (size_t)b * cap * 3;
    int lo, hi;
    cmp_slice(n, g, lo, hi);
    int c = 0;
    for (int i = lo + threadIdx.x; i < hi; i += 256) c += pred(b, i, xyz[(size_t)i * 3], xyz[(size_t)i * 3 + 1], xyz[(size_t)i * 3 + 2]) ? 1 : 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
    __shared__ int ws[4];
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) blk_cnt[b * CMP_G + g] = ws[0] + ws[1] + ws[2] + ws[3];
}
template <class P>
__global__ __launch_bounds__(256) void cmp_scatter_kernel(CloudView in, CloudOut out, int cap, P pred, const int* blk_cnt) {
    const int b = blockIdx.y, g = blockIdx.x;
    const int n = min(in.n[b], cap);
    const float* xyz = in.xyz + (size_t)b * cap * 3;
    const uint8_t* rgb = in.rgb ? in.rgb + (size_t)b * cap * 3 : nullptr;
    float* oxyz = out.xyz + (size_t)b * cap * 3;
    uint8_t* orgb = (out.rgb && in.rgb) ? out.rgb + (size_t)b * cap * 3 : nullptr;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // kept rows before this slice (and of the whole frame): 64 counts, one per lane
    const int mine = blk_cnt[b * CMP_G + lane];
    int before = lane < g ? mine : 0, total = mine;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { before += __shfl_xor(before, o); total += __shfl_xor(total, o); }
    if (g == 0 && threadIdx.x == 0) out.n[b] = total;
    int lo, hi;
    cmp_slice(n, g, lo, hi);
    __shared__ int wsum[4];
    int base = before;
    for (int i0 = lo; i0 < hi; i0 += 256) {
        const int i = i0 + threadIdx.x;
        float x = 0.f, y = 0.f, z = 0.f;
        uint8_t c0 = 0, c1 = 0, c2 = 0;
        bool keep = false;
        if (i < hi) {
            x = xyz[(size_t)i * 3]; y = xyz[(size_t)i * 3 + 1]; z = xyz[(size_t)i * 3 + 2];
            if (rgb) { c0 = rgb[(size_t)i * 3]; c1 = rgb[(size_t)i * 3 + 1]; c2 = rgb[(size_t)i * 3 + 2]; }
            keep = pred(b, i, x, y, z);
        }
        const unsigned long long m = __ballot(keep);
        if (lane == 0) wsum[wave] = __popcll(m);
        __syncthreads();
        int pos = base + __popcll(m & ((1ull << lane) - 1ull));
        int tot = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) { const int c = wsum[w]; if (w < wave) pos += c; tot += c; }
        if (keep && pos < cap) {
            oxyz[(size_t)pos * 3] = x; oxyz[(size_t)pos * 3 + 1] = y; oxyz[(size_t)pos * 3 + 2] = z;
            if (orgb) { orgb[(size_t)pos * 3] = c0; orgb[(size_t)pos * 3 + 1] = c1; orgb[(size_t)pos * 3 + 2] = c2; }
        }
        base += tot;
        __syncthreads();
    }
}
template <class P>
static void cmp_run(CloudView in, CloudOut out, int B, int cap, P pred, const CmpScratch& c, hipStream_t s) {
    hipLaunchKernelGGL(cmp_count_kernel<P>, dim3(CMP_G, B), dim3(256), 0, s, in, cap, pred, c.blk_cnt);
    hipLaunchKernelGGL(cmp_scatter_kernel<P>, dim3(CMP_G, B), dim3(256), 0, s, in, out, cap, pred, c.blk_cnt);
}
// the predicates of the filters below, with the same expressions as their single-workgroup forms
struct CoordPred {
    int kind, axis; float t;
    __device__ bool operator()(int, int, float x, float y, float z) const {
        const float v = axis == 0 ? x : (axis == 1 ? y : z);
        return kind == F_LT_NEG ? (v < -t) : (fabsf(v) < t);
    }
};
struct MadPred {
    int axis; float thr; const double* params;
    __device__ bool operator()(int b, int, float x, float y, float z) const {
        const float med = (float)params[b * CMP_PARAMS], madv = (float)params[b * CMP_PARAMS + 1];
        const float v = axis == 0 ? x : (axis == 1 ? y : z);
        const float pen = 0.6745f * fabsf(v - med) / madv;
        return pen < thr;
    }
};
struct PlanePred {
    int iu, iv, id; double thr; const double* params;
    __device__ bool operator()(int b, int, float x, float y, float z) const {
        const double C0 = params[b * CMP_PARAMS], C1 = params[b * CMP_PARAMS + 1], C2 = params[b * CMP_PARAMS + 2];
        const double p[3] = {(double)x, (double)y, (double)z};
        const double a = ((C0 * p[iu] + C1 * p[iv]) - p[id]) + C2;
        return fabs(a) < thr;
    }
};
struct SorPred {
    const double* mean_d; int cap; const double* params;
    __device__ bool operator()(int b, int i, float, float, float) const {
        const double v = mean_d[(size_t)b * cap + i];
        return v > 0.0 && v < params[b * CMP_PARAMS];
    }
};
struct KeepPred {
    const uint8_t* keep; int cap;
    __device__ bool operator()(int b, int i, float, float, float) const { return keep[(size_t)b * cap + i] != 0; }
};

// ------------------------------------------------------------------------------------------ K20 / threshold
// pcl.remove_from_to (pcl.py:30-43): keep coord < -t.   pcl.threshold_complete (pcl.py:240-250): keep |coord| < t.
// The comparison is float32 vs the float32-rounded literal, as numpy does for a float32 column.
__global__ __launch_bounds__(TB) void filter_coord_kernel(CloudView in, CloudOut out, int cap, int kind, int axis, float t) {
    FRAME_VIEW();
    block_compact(xyz, rgb, n, oxyz, orgb, on, cap, [=](int, float x, float y, float z) {
        const float v = axis == 0 ? x : (axis == 1 ? y : z);
        return kind == F_LT_NEG ? (v < -t) : (fabsf(v) < t);
    }, L);
}
hipError_t launch_filter_coord(CloudView in, CloudOut out, int B, int cap, int kind, int axis, double t, void* cscratch, hipStream_t s) {
    if (cscratch && in.xyz != out.xyz) {
        cmp_run(in, out, B, cap, CoordPred{kind, axis, (float)t}, cmp_carve(cscratch, B), s);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(filter_coord_kernel, dim3(B), dim3(TB), 0, s, in, out, cap, kind, axis, (float)t);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------ multi-block median
// block_median_fast spread over the chip: (1) one workgroup per frame samples the column and brackets the middle ranks,
// (2) CMP_G workgroups per frame make the one pass over the column (count below the bracket, append the keys inside it to
// a global buffer), (3) one workgroup per frame selects the order statistics from the buffer -- or, if the bracket missed,
// runs the full single-workgroup select.  MODE 0: median of the column; MODE 1: median of |column - median| (the MAD).
template <int MODE>
__device__ __forceinline__ float med_value(const float* xyz, int i, int axis, float med) {
    const float v = xyz[(size_t)i * 3 + axis];
    return MODE == 0 ? v : fabsf(v - med);
}
template <int MODE>
__global__ __launch_bounds__(TB) void med_sample_kernel(CloudView in, int cap, int axis, MedG* G) {
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* xyz = in.xyz + (size_t)b * cap * 3;
    const int n = min(in.n[b], cap);
    __shared__ Lds L;
    __shared__ unsigned samp[MED_S];
    const float med = MODE == 1 ? G[b].med : 0.f;
    auto val = [=](int i) { return med_value<MODE>(xyz, i, axis, med); };
    if (n < 4 * MED_S) {                                        // small cloud: the plain select, done here
        const float r = block_median(val, n, L);
        if (tid == 0) { G[b].done = 1; if (MODE == 0) G[b].med = r; else G[b].madv = r; }
        return;
    }
    for (int j = tid; j < MED_S; j += TB) samp[j] = f2key(val((int)(((long)j * n) / MED_S)));
    __syncthreads();
    const unsigned r1 = (unsigned)((n - 1) / 2);
    const int ts = (int)(((long)r1 * MED_S) / n);
    const int slo = max(ts - MED_D, 0), shi = min(ts + MED_D, MED_S - 1);
    auto sval = [&](int j) { return key2f(samp[j]); };
    const unsigned klo = f2key(block_select(sval, MED_S, (unsigned)slo, L));
    const unsigned khi = f2key(block_select(sval, MED_S, (unsigned)shi, L));
    if (tid == 0) { G[b].klo = klo; G[b].khi = khi; G[b].cnt = 0; G[b].below = 0; G[b].nan = 0; G[b].done = 0; }
}
template <int MODE>
__global__ __launch_bounds__(256) void med_collect_kernel(CloudView in, int cap, int axis, MedG* G, unsigned* medbuf) {
    const int b = blockIdx.y, g = blockIdx.x;
    if (G[b].done) return;
    const float* xyz = in.xyz + (size_t)b * cap * 3;
    const int n = min(in.n[b], cap);
    const unsigned klo = G[b].klo, khi = G[b].khi;
    const float med = MODE == 1 ? G[b].med : 0.f;
    unsigned* buf = medbuf + (size_t)b * MED_CAP;
    int lo, hi;
    cmp_slice(n, g, lo, hi);
    unsigned below = 0, nanf = 0;
    for (int i = lo + threadIdx.x; i < hi; i += 256) {
        const float v = med_value<MODE>(xyz, i, axis, med);
        nanf |= (v != v);
        const unsigned k = f2key(v);
        if (k < klo) ++below;
        else if (k <= khi) {
            const unsigned pos = atomicAdd(&G[b].cnt, 1u);
            if (pos < MED_CAP) buf[pos] = k;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { below += __shfl_xor(below, o); nanf |= __shfl_xor(nanf, o); }
    if ((threadIdx.x & 63) == 0) { if (below) atomicAdd(&G[b].below, below); if (nanf) atomicOr(&G[b].nan, 1u); }
}
template <int MODE>
__global__ __launch_bounds__(TB) void med_pick_kernel(CloudView in, int cap, int axis, MedG* G, const unsigned* medbuf, float* stats, double* params) {
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* xyz = in.xyz + (size_t)b * cap * 3;
    const int n = min(in.n[b], cap);
    __shared__ Lds L;
    float r;
    if (G[b].done) {
        r = MODE == 0 ? G[b].med : G[b].madv;
    } else {
        const unsigned cnt = G[b].cnt, bel = G[b].below;
        const unsigned r1 = (unsigned)((n - 1) / 2), r2 = (unsigned)(n / 2);
        const float med = MODE == 1 ? G[b].med : 0.f;
        if (G[b].nan) {
            r = __uint_as_float(0x7fc00000u);
        } else if (cnt > MED_CAP || r1 < bel || r2 >= bel + cnt) {            // bracket missed: exact fallback
            r = block_median([=](int i) { return med_value<MODE>(xyz, i, axis, med); }, n, L);
        } else {
            const unsigned* buf = medbuf + (size_t)b * MED_CAP;
            auto bval = [=](int j) { return key2f(buf[j]); };
            const float a = block_select(bval, (int)cnt, r1 - bel, L);
            r = a;
            if (r1 != r2) r = (a + block_select(bval, (int)cnt, r2 - bel, L)) / 2.0f;
        }
    }
    if (tid == 0) {
        if (MODE == 0) {
            G[b].med = r;
        } else {
            G[b].madv = r;
            const float med = G[b].med;
            if (stats) { stats[b * 2] = med; stats[b * 2 + 1] = r; }
            params[b * CMP_PARAMS] = (double)med; params[b * CMP_PARAMS + 1] = (double)r;
        }
    }
}
template <int MODE>
static void med_run(CloudView in, int B, int cap, int axis, const CmpScratch& c, float* stats, hipStream_t s) {
    hipLaunchKernelGGL(med_sample_kernel<MODE>, dim3(B), dim3(TB), 0, s, in, cap, axis, c.med);
    hipLaunchKernelGGL(med_collect_kernel<MODE>, dim3(CMP_G, B), dim3(256), 0, s, in, cap, axis, c.med, c.medbuf);
    hipLaunchKernelGGL(med_pick_kernel<MODE>, dim3(B), dim3(TB), 0, s, in, cap, axis, c.med, c.medbuf, stats, c.params);
}

// ------------------------------------------------------------------------------------------ K21 MAD
// pcl.remove_noise_by_mad + mad (pcl.py:46-81), all float32 like numpy on a float32 column:
//   med = median(v); dev = |v - med|; MAD = median(dev); keep 0.6745f*dev/MAD < thr
template <bool COMPACT>
__global__ __launch_bounds__(TB) void mad_filter_kernel(CloudView in, CloudOut out, int cap, int axis, float thr, float* stats, double* params) {
    FRAME_VIEW();
    __shared__ MedLds M;
    auto col = [=](int i) { return xyz[(size_t)i * 3 + axis]; };
    const float med = block_median_fast(col, n, L, M);
    auto dev = [=](int i) { return fabsf(xyz[(size_t)i * 3 + axis] - med); };
    const float madv = block_median_fast(dev, n, L, M);
    if (stats && threadIdx.x == 0) { stats[b * 2] = med; stats[b * 2 + 1] = madv; }
    if (!COMPACT) {
        if (threadIdx.x == 0) { params[b * CMP_PARAMS] = (double)med; params[b * CMP_PARAMS + 1] = (double)madv; }
        return;
    }
    block_compact(xyz, rgb, n, oxyz, orgb, on, cap, [=](int, float x, float y, float z) {
        const float v = axis == 0 ? x : (axis == 1 ? y : z);
        const float pen = 0.6745f * fabsf(v - med) / madv;
        return pen < thr;
    }, L);
}
hipError_t launch_mad_filter(CloudView in, CloudOut out, int B, int cap, int axis, double thr, float* stats, void* cscratch, hipStream_t s) {
    if (cscratch && in.xyz != out.xyz) {
        const CmpScratch c = cmp_carve(cscratch, B);
        med_run<0>(in, B, cap, axis, c, nullptr, s);              // median, then the median of the deviations from it
        med_run<1>(in, B, cap, axis, c, stats, s);
        cmp_run(in, out, B, cap, MadPred{axis, (float)thr, c.params}, c, s);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(mad_filter_kernel<true>, dim3(B), dim3(TB), 0, s, in, out, cap, axis, (float)thr, stats, (double*)nullptr);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------ K22 plane fit
// pcl.remove_noise_by_fitting_plane (pcl.py:84-209): least squares dep = C0*u + C1*v + C2 in float64
// (the reference calls scipy.linalg.lstsq on float64 columns; here: centred normal equations, float64,
// deterministic reduction order), then keep |C0*u + C1*v - dep + C2| < thr (float64).
template <bool COMPACT>
__global__ __launch_bounds__(TB) void plane_filter_kernel(CloudView in, CloudOut out, int cap, int axis, double thr, double* coeff, double* params) {
    FRAME_VIEW();
    const int iu = axis == 0 ? 1 : 0, iv = axis == 2 ? 1 : 2, id = axis;
    double su = 0, sv = 0, sd_ = 0;
#pragma unroll 4
    for (int i = threadIdx.x; i < n; i += TB) {
        su += (double)xyz[(size_t)i * 3 + iu]; sv += (double)xyz[(size_t)i * 3 + iv]; sd_ += (double)xyz[(size_t)i * 3 + id];
    }
    const double inv_n = 1.0 / (double)n;
    const double mu = block_sum_f64(su, L) * inv_n, mv = block_sum_f64(sv, L) * inv_n, md = block_sum_f64(sd_, L) * inv_n;
    double suu = 0, suv = 0, svv = 0, sud = 0, svd = 0;
#pragma unroll 4
    for (int i = threadIdx.x; i < n; i += TB) {
        const double u = (double)xyz[(size_t)i * 3 + iu] - mu, v = (double)xyz[(size_t)i * 3 + iv] - mv;
        const double d = (double)xyz[(size_t)i * 3 + id] - md;
        suu += u * u; suv += u * v; svv += v * v; sud += u * d; svd += v * d;
    }
    suu = block_sum_f64(suu, L); suv = block_sum_f64(suv, L); svv = block_sum_f64(svv, L);
    sud = block_sum_f64(sud, L); svd = block_sum_f64(svd, L);
    const double det = suu * svv - suv * suv;
    const double C0 = (sud * svv - svd * suv) / det;
    const double C1 = (svd * suu - sud * suv) / det;
    const double C2 = md - C0 * mu - C1 * mv;
    if (coeff && threadIdx.x == 0) {
        double* c = coeff + (size_t)b * 4;     // Cx, Cy, Cz, C  (pcl.py:130, :168, :204)
        if (axis == 0)      { c[0] = -1.0; c[1] = C0; c[2] = C1; }
        else if (axis == 1) { c[0] = C0; c[1] = -1.0; c[2] = C1; }
        else                { c[0] = C0; c[1] = C1; c[2] = -1.0; }
        c[3] = C2;
    }
    if (!COMPACT) {
        if (threadIdx.x == 0) { params[b * CMP_PARAMS] = C0; params[b * CMP_PARAMS + 1] = C1; params[b * CMP_PARAMS + 2] = C2; }
        return;
    }
    block_compact(xyz, rgb, n, oxyz, orgb, on, cap, [=](int, float x, float y, float z) {
        const double p[3] = {(double)x, (double)y, (double)z};
        const double a = ((C0 * p[iu] + C1 * p[iv]) - p[id]) + C2;
        return fabs(a) < thr;
    }, L);
}
hipError_t launch_plane_filter(CloudView in, CloudOut out, int B, int cap, int axis, double thr, double* coeff, void* cscratch, hipStream_t s) {
    if (cscratch && in.xyz != out.xyz) {
        const CmpScratch c = cmp_carve(cscratch, B);
        hipLaunchKernelGGL(plane_filter_kernel<false>, dim3(B), dim3(TB), 0, s, in, out, cap, axis, thr, coeff, c.params);
        cmp_run(in, out, B, cap, PlanePred{axis == 0 ? 1 : 0, axis == 2 ? 1 : 2, axis, thr, c.params}, c, s);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(plane_filter_kernel<true>, dim3(B), dim3(TB), 0, s, in, out, cap, axis, thr, coeff, (double*)nullptr);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------ K24 end points
// pcl.get_end_points_of_road / get_end_points_of_segment (pcl.py:271-313) + width (semantic_depth.py:259):
// window -(depth+w) < z < -(depth-w) in float64; first row with min x, first row with max x.
__global__ __launch_bounds__(TB) void end_points_kernel(CloudView in, int cap, double depth, double window, RwResultDev* res) {
    const int b = blockIdx.x;
    const float* xyz = in.xyz + (size_t)b * cap * 3;
    const int n = min(in.n[b], cap);
    __shared__ unsigned long long smin[NW], smax[NW];
    const double hi = -(depth - window), lo = -(depth + window);
    // key = (monotone x key << 32) | index : min picks smallest x then smallest index;
    // for the max: (~xkey << 32) | index, min of that picks largest x then smallest index
    unsigned long long kmin = ~0ull, kmax = ~0ull;
#pragma unroll 4
    for (int i = threadIdx.x; i < n; i += TB) {
        const double z = (double)xyz[(size_t)i * 3 + 2];
        if (z < hi && z > lo) {
            const unsigned kx = f2key(xyz[(size_t)i * 3]);
            const unsigned long long a = ((unsigned long long)kx << 32) | (unsigned)i;
            const unsigned long long c = ((unsigned long long)(~kx) << 32) | (unsigned)i;
            kmin = a < kmin ? a : kmin;
            kmax = c < kmax ? c : kmax;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        unsigned long long t1 = __shfl_xor(kmin, o), t2 = __shfl_xor(kmax, o);
        kmin = t1 < kmin ? t1 : kmin;
        kmax = t2 < kmax ? t2 : kmax;
    }
    if ((threadIdx.x & 63) == 0) { smin[threadIdx.x >> 6] = kmin; smax[threadIdx.x >> 6] = kmax; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 0; w < NW; ++w) { kmin = smin[w] < kmin ? smin[w] : kmin; kmax = smax[w] < kmax ? smax[w] : kmax; }
        RwResultDev& r = res[b];
        if (kmin == ~0ull) {
            r.found = 0;
            r.width = __longlong_as_double(0x7ff8000000000000ll);
            r.x_left = r.x_right = __uint_as_float(0x7fc00000u);
            for (int j = 0; j < 3; ++j) { r.left_pt[j] = r.x_left; r.right_pt[j] = r.x_left; }
        } else {
            const unsigned il = (unsigned)(kmin & 0xFFFFFFFFull), ir = (unsigned)(kmax & 0xFFFFFFFFull);
            r.found = 1;
            for (int j = 0; j < 3; ++j) { r.left_pt[j] = xyz[(size_t)il * 3 + j]; r.right_pt[j] = xyz[(size_t)ir * 3 + j]; }
            r.x_left = r.left_pt[0];
            r.x_right = r.right_pt[0];
            r.width = fabs((double)r.x_left - (double)r.x_right);
        }
    }
}
hipError_t launch_end_points(CloudView in, int B, int cap, double depth, double window, RwResultDev* res, hipStream_t s) {
    hipLaunchKernelGGL(end_points_kernel, dim3(B), dim3(TB), 0, s, in, cap, depth, window, res);
    return hipGetLastError();
}

__global__ void record_counts_kernel(RwResultDev* res, int B, const int32_t* n_road, const int32_t* n_zcut, const int32_t* n_mad_y,
                                     const int32_t* n_mad_x, const int32_t* n_plane, const int32_t* n_sor, const int32_t* n_ror,
                                     const double* plane) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    RwResultDev& r = res[b];
    r.n_road = n_road[b]; r.n_zcut = n_zcut[b]; r.n_mad_y = n_mad_y[b]; r.n_mad_x = n_mad_x[b];
    r.n_plane = n_plane[b]; r.n_sor = n_sor[b]; r.n_ror = n_ror[b];
    for (int j = 0; j < 4; ++j) r.plane[j] = plane[(size_t)b * 4 + j];
}
hipError_t launch_record_counts(RwResultDev* res, int B, const int32_t* n_road, const int32_t* n_zcut, const int32_t* n_mad_y,
                                const int32_t* n_mad_x, const int32_t* n_plane, const int32_t* n_sor, const int32_t* n_ror,
                                const double* plane, hipStream_t s) {
    hipLaunchKernelGGL(record_counts_kernel, dim3((B + 63) / 64), dim3(64), 0, s, res, B, n_road, n_zcut, n_mad_y, n_mad_x,
                       n_plane, n_sor, n_ror, plane);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------ extract_pcls (fence chain)
// pcl.extract_pcls (pcl.py:253-268): split at np.mean of a float32 coordinate column: left = coord < mean, right = coord > mean.
// The mean is reproduced bit for bit: numpy's add.reduce walks the column in chunks of 8192 elements (the ufunc buffer
// size), sums each chunk with its pairwise routine (8 interleaved accumulators on blocks of <= 128 elements, halves
// rounded down to a multiple of 8 above that), adds the chunk sums in order, and np.mean divides by n in float32.
struct ColF32 {
    const float* p;    // first element
    int stride;        // in floats
    __device__ __forceinline__ float operator[](int i) const { return p[(size_t)i * stride]; }
};
__device__ float np_pairwise(ColF32 a, int off, int n) {
    if (n < 8) {
        float r = 0.f;
        for (int i = 0; i < n; ++i) r += a[off + i];
        return r;
    }
    if (n <= 128) {
        float r[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] = a[off + j];
        int i = 8;
        for (; i < n - (n % 8); i += 8)
#pragma unroll
            for (int j = 0; j < 8; ++j) r[j] += a[off + i + j];
        float res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; ++i) res += a[off + i];
        return res;
    }
    int n2 = n / 2;
    n2 -= n2 % 8;
    return np_pairwise(a, off, n2) + np_pairwise(a, off + n2, n - n2);
}
// np.mean(column) for a float32 column, by one workgroup
__device__ float block_np_mean(ColF32 a, int n, Lds& L) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int full = n / 8192, rem = n % 8192;
    __shared__ float csum[NW];
    __shared__ float total;
    if (tid == 0) total = 0.f;
    __syncthreads();
    for (int c0 = 0; c0 < full; c0 += NW) {
        const int c = c0 + wave;
        if (c < full) {           // one wave per 8192-chunk: lane j sums leaf j (128 elements), butterfly = the balanced recursion
            float v = np_pairwise(a, c * 8192 + lane * 128, 128);
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) v += __shfl_xor(v, o);
            if (lane == 0) csum[wave] = v;
        }
        __syncthreads();
        if (tid == 0) {
            for (int w = 0; w < NW && c0 + w < full; ++w) total = (c0 + w == 0) ? csum[w] : total + csum[w];
        }
        __syncthreads();
    }
    if (tid == 0 && rem > 0) {
        const float v = np_pairwise(a, full * 8192, rem);
        total = full == 0 ? v : total + v;
    }
    __syncthreads();
    (void)L;
    return total / (float)n;
}

__global__ __launch_bounds__(TB) void extract_pcls_kernel(CloudView in, CloudOut outl, CloudOut outr, int cap, int axis, float* mean_out) {
    const int b = blockIdx.x;
    const float* xyz = in.xyz + (size_t)b * cap * 3;
    const uint8_t* rgb = in.rgb ? in.rgb + (size_t)b * cap * 3 : nullptr;
    const int n = min(in.n[b], cap);
    __shared__ Lds L;
    const float mean = n > 0 ? block_np_mean(ColF32{xyz + axis, 3}, n, L) : __uint_as_float(0x7fc00000u);
    if (mean_out && threadIdx.x == 0) mean_out[b] = mean;
    block_compact(xyz, rgb, n, outl.xyz + (size_t)b * cap * 3, (outl.rgb && rgb) ? outl.rgb + (size_t)b * cap * 3 : nullptr, outl.n + b, cap,
                  [=](int, float x, float y, float z) { return (axis == 0 ? x : (axis == 1 ? y : z)) < mean; }, L);
    block_compact(xyz, rgb, n, outr.xyz + (size_t)b * cap * 3, (outr.rgb && rgb) ? outr.rgb + (size_t)b * cap * 3 : nullptr, outr.n + b, cap,
                  [=](int, float x, float y, float z) { return (axis == 0 ? x : (axis == 1 ? y : z)) > mean; }, L);
}
hipError_t launch_extract_pcls(CloudView in, CloudOut outl, CloudOut outr, int B, int cap, int axis, float* mean_out, hipStream_t s) {
    hipLaunchKernelGGL(extract_pcls_kernel, dim3(B), dim3(TB), 0, s, in, outl, outr, cap, axis, mean_out);
    return hipGetLastError();
}

// pcl.planes_intersection_at_certain_depth (pcl.py:212-237) for the two fence planes against the road plane, and the
// fence-to-fence distance (semantic_depth.py:321-328).  Cramer's rule in float64 (the reference inverts the 2x2 with LAPACK).
__global__ void f2f_kernel(const double* road_plane, const double* left_plane, const double* right_plane, int B, double depth,
                           const int32_t* cnt, F2fResultDev* out) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    F2fResultDev r;
    const double z = -depth;
    const double* rp = road_plane + (size_t)b * 4;
    const double* pl[2] = {left_plane + (size_t)b * 4, right_plane + (size_t)b * 4};
    double pt[2][3];
    for (int s = 0; s < 2; ++s) {
        const double a = rp[0], bb = rp[1], c = pl[s][0], d = pl[s][1];
        const double e = -(rp[2] * z + rp[3]), f = -(pl[s][2] * z + pl[s][3]);
        const double det = a * d - bb * c;
        pt[s][0] = (e * d - bb * f) / det;
        pt[s][1] = (a * f - e * c) / det;
        pt[s][2] = z;
    }
    const double dx = pt[0][0] - pt[1][0], dy = pt[0][1] - pt[1][1], dz = pt[0][2] - pt[1][2];
    r.dist = sqrt(dx * dx + dy * dy + dz * dz);
    for (int j = 0; j < 3; ++j) { r.left_pt[j] = pt[0][j]; r.right_pt[j] = pt[1][j]; }
    for (int j = 0; j < 4; ++j) { r.plane_left[j] = pl[0][j]; r.plane_right[j] = pl[1][j]; }
    for (int j = 0; j < 7; ++j) r.counts[j] = cnt[(size_t)j * B + b];
    r.ok = (r.dist == r.dist) && r.counts[5] > 0 && r.counts[6] > 0;
    out[b] = r;
}
hipError_t launch_f2f(const double* road_plane, const double* left_plane, const double* right_plane, int B, double depth,
                      const int32_t* cnt, F2fResultDev* out, hipStream_t s) {
    hipLaunchKernelGGL(f2f_kernel, dim3((B + 63) / 64), dim3(64), 0, s, road_plane, left_plane, right_plane, B, depth, cnt, out);
    return hipGetLastError();
}

__global__ void gather_planes_kernel(const RwResultDev* res, int B, double* planes) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    for (int j = 0; j < 4; ++j) planes[(size_t)b * 4 + j] = res[b].plane[j];
}
hipError_t launch_gather_planes(const RwResultDev* res, int B, double* planes, hipStream_t s) {
    hipLaunchKernelGGL(gather_planes_kernel, dim3((B + 63) / 64), dim3(64), 0, s, res, B, planes);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------ K23 Open3D filters
// [UPSTREAM Open3D legacy RemoveStatisticalOutliers / RemoveRadiusOutliers, parity unpinned — see oracle/o3d.py]
// Exact k-NN / radius counts in float64 over a uniform grid: points are binned (clamped at the grid faces,
// which keeps every lower bound valid), counting-sorted by cell, and every point searches Chebyshev shells
// of cells until its result is proven final.  d2 := (dx*dx + dy*dy) + dz*dz, no FMA.
//   statistical filter: cell size adapts to the cloud (~6 points per cell), shells until kth d2 <= (shell*cell)^2
//   radius filter     : cell = radius/4; cells entirely inside the ball are counted without touching their points,
//                       cells entirely outside are skipped, the search stops as soon as the count exceeds nb_points
constexpr int GRID_CELLS = 1 << 19;
constexpr int KMAX = 16;
constexpr int SOR_RMAX = 16;       // shells searched before the brute-force fallback
constexpr int SOR_RSOFT = 2;       // shells searched one-thread-per-query (99 % of a dense cloud end here); a query that needs more
                                   // hands its k best so far to the wave-cooperative kernel, which continues at the next shell
                                   // (RSOFT = 1 measured: per-thread kernel 2.95 -> 2.64 ms, cooperative kernel 0.55 -> 1.37 ms)
constexpr int SCAN_SEG = 2048;     // cells per scan segment
constexpr int SCAN_NSEG = GRID_CELLS / SCAN_SEG;

struct GridMeta { double ox, oy, oz, inv; double cell; int gx, gy, gz, occupied; double ext[3], mn[3], mxz; int nonfinite, pad_; };
// does the grid hold every finite point inside its NOMINAL cells along each axis (no clamping into the face layers)?  Then, with
// no non-finite point in the cloud either, a face layer is an ordinary bounded cell.
__device__ __forceinline__ bool grid_covers(const GridMeta& g) {
    const double m = 1.0 - 1e-9;
    return g.nonfinite == 0 && g.ext[0] < (double)g.gx * g.cell * m && g.ext[1] < (double)g.gy * g.cell * m && g.oz <= g.mn[2] &&
           g.mxz < g.oz + (double)g.gz * g.cell * m && g.ox <= g.mn[0] && g.oy <= g.mn[1];
}

struct O3dScratch {       // carved from one arena, per-frame strides
    GridMeta* meta;       // [B]
    int* cell_cnt;        // [B][GRID_CELLS]      counts, then scatter cursors; behind them [B][8] bounding-box accumulators
    int* cell_start;      // [B][GRID_CELLS + 1]
    int* seg_sum;         // [B][SCAN_NSEG]
    int* cell_of;         // [B][cap]
    int* sidx;            // [B][cap]  original index of the j-th sorted point
    float* sxyz;          // [B][cap][4]: the cell-sorted copy, one 16-byte load per candidate (w unused)
    double* mean_d;       // [B][cap]  (by original index)
    uint8_t* keep;        // [B][cap]
    int* hard_n;          // [B]  statistical filter: queries deferred to the wave-cooperative search (list in cell_of)
    double* hard_best;    // [B][cap][KMAX]  their k best distances after the per-thread shells (ascending, inf padded)
};
size_t o3d_scratch_bytes(int B, int cap) {
    size_t per = sizeof(GridMeta) + (size_t)GRID_CELLS * 4 + (size_t)(GRID_CELLS + 1) * 4 + SCAN_NSEG * 4 + (size_t)cap * (4 + 4 + 16 + 8 + 1) +
                 (size_t)cap * KMAX * 8;
    return (size_t)B * (per + 4 + 32) + 8192 + 256;
}
static O3dScratch carve(void* base, int B, int cap) {
    O3dScratch s;
    char* p = (char*)base;
    auto take = [&](size_t bytes) { char* r = p; p += (bytes + 255) / 256 * 256; return r; };
    s.meta = (GridMeta*)take(sizeof(GridMeta) * B);
    s.mean_d = (double*)take((size_t)B * cap * 8);
    s.cell_cnt = (int*)take((size_t)B * GRID_CELLS * 4 + (size_t)B * 32);
    s.cell_start = (int*)take((size_t)B * (GRID_CELLS + 1) * 4);
    s.seg_sum = (int*)take((size_t)B * SCAN_NSEG * 4);
    s.cell_of = (int*)take((size_t)B * cap * 4);
    s.sidx = (int*)take((size_t)B * cap * 4);
    s.sxyz = (float*)take((size_t)B * cap * 16);
    s.keep = (uint8_t*)take((size_t)B * cap);
    s.hard_n = (int*)take((size_t)B * 4);
    s.hard_best = (double*)take((size_t)B * cap * KMAX * 8);
    return s;
}

__device__ __forceinline__ int cell_coord(double p, double o, double inv, int g) {
    double t = floor((p - o) * inv);
    if (!(t >= 0.0)) t = 0.0;                 // also catches NaN
    if (t > (double)(g - 1)) t = (double)(g - 1);
    return (int)t;
}
__device__ __forceinline__ int pow2ceil(int v) { int p = 1; while (p < v) p <<= 1; return p; }

// cell size -> grid dims (powers of two with gx*gy*gz == GRID_CELLS) and origin, from the stored bounding box
__device__ void grid_layout(GridMeta& g, double cell) {
    g.cell = cell; g.inv = 1.0 / cell;
    // x and y sized to the box (clamped), z takes the rest
    const int gy = min(pow2ceil((int)fmin(g.ext[1] / cell + 2.0, 4096.0)), 64);
    const int gx = min(pow2ceil((int)fmin(g.ext[0] / cell + 2.0, 4096.0)), 256);
    g.gx = gx; g.gy = gy; g.gz = GRID_CELLS / (gx * gy);
    g.ox = g.mn[0]; g.oy = g.mn[1]; g.oz = g.mn[2];
    // scene z is negative and dense near the camera (large z): if the z range does not fit, anchor the grid at the
    // near end so the dense part is resolved and the far tail clamps into the first layer
    const double span = cell * (double)g.gz;
    if (g.ext[2] > span) g.oz = g.mxz - span + 0.5 * cell;
}

// second look at the cell size of the statistical filter: the first grid's occupancy tells how the cloud really fills
// space (a road cloud is a sheet, not a volume); aim at ~8 points per OCCUPIED cell, assuming occupancy ~ cell^2
__global__ void grid_refine_kernel(CloudView in, int cap, GridMeta* meta, int B, double occ_target) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    GridMeta g = meta[b];
    const int n = min(in.n[b], cap);
    if (g.occupied > 0 && n > 0) {
        const double avg = (double)n / (double)g.occupied;
        if (avg > 1.5 * occ_target) grid_layout(g, fmax(g.cell * sqrt(occ_target / avg), g.cell * 0.125));
    }
    g.occupied = 0;
    meta[b] = g;
}

// order-preserving map float -> unsigned (0 is below every float, -inf included): one atomicMax per component accumulates a maximum,
// and the complement of the code a minimum, from an all-zero start
__device__ __forceinline__ unsigned ord_f32(float v) { const unsigned u = __float_as_uint(v); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
__device__ __forceinline__ float unord_f32(unsigned c) { return __uint_as_float((c & 0x80000000u) ? (c & 0x7fffffffu) : ~c); }

// bounding box of the finite coordinates, every CU on it: acc[b] = {~ord(min x,y,z), ord(max x,y,z), non-finite count, -} (zeroed before)
__global__ __launch_bounds__(256) void grid_bbox_kernel(CloudView in, int cap, unsigned* acc) {
    const int b = blockIdx.y;
    const float* xyz = in.xyz + (size_t)b * cap * 3;
    const int n = min(in.n[b], cap);
    if ((int)(blockIdx.x * 256) >= n) return;
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    int bad = 0;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const float v = xyz[(size_t)i * 3 + j];
            if (v > -INFINITY && v < INFINITY) { mn[j] = fminf(mn[j], v); mx[j] = fmaxf(mx[j], v); }
            else ++bad;
        }
    unsigned* a = acc + (size_t)b * 8;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) bad += __shfl_xor(bad, o);
#pragma unroll
    for (int j = 0; j < 3; ++j) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { mn[j] = fminf(mn[j], __shfl_xor(mn[j], o)); mx[j] = fmaxf(mx[j], __shfl_xor(mx[j], o)); }
        if ((threadIdx.x & 63) == 0 && mn[j] < INFINITY) { atomicMax(a + j, ~ord_f32(mn[j])); atomicMax(a + 3 + j, ord_f32(mx[j])); }
    }
    if ((threadIdx.x & 63) == 0 && bad) atomicAdd(a + 6, (unsigned)bad);
}

// bounding box -> cell size, grid dims, origin (one thread per frame).  fixed_cell > 0: use it; else adapt to ~6 points/cell
__global__ __launch_bounds__(64) void grid_meta_kernel(CloudView in, int cap, int B, double fixed_cell, const unsigned* acc, GridMeta* meta) {
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b >= B) return;
    const int n = min(in.n[b], cap);
    const unsigned* a = acc + (size_t)b * 8;
    float mn[3], mx[3];
    double ext[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const unsigned cmin = ~a[j], cmax = a[3 + j];
        const bool any = a[j] != 0u;                               // (else: no finite coordinate)
        mn[j] = any ? unord_f32(cmin) : 0.f;
        mx[j] = any ? unord_f32(cmax) : 0.f;
        ext[j] = (double)mx[j] - (double)mn[j];
    }
    double cell = fixed_cell;
    if (!(cell > 0.0)) {
        const double vol = fmax(ext[0], 0.05) * fmax(ext[1], 0.05) * fmax(ext[2], 0.05);
        cell = cbrt(6.0 * vol / (double)max(n, 1));
        cell = fmin(fmax(cell, 0.01), 4.0);
    }
    GridMeta g;
    for (int j = 0; j < 3; ++j) { g.ext[j] = ext[j]; g.mn[j] = mn[j]; }
    g.mxz = mx[2];
    g.nonfinite = (int)a[6]; g.pad_ = 0;
    g.occupied = 0;
    grid_layout(g, cell);
    meta[b] = g;
}

// runs of equal cell ids among the lanes of a wave (a pixel-ordered cloud puts neighbouring points into the same cell): one
// atomic per run instead of one per point.  Valid lanes are a prefix of the wave.  Returns the run's first lane and length.
__device__ __forceinline__ void cell_runs(bool valid, int c, int& first, int& len) {
    const int lane = threadIdx.x & 63;
    const int pc = __shfl_up(c, 1);
    const bool start = valid && (lane == 0 || c != pc);
    const unsigned long long starts = __ballot(start);
    const int nvalid = __popcll(__ballot(valid));
    const unsigned long long upto = starts & (lane == 63 ? ~0ull : ((2ull << lane) - 1ull));
    first = upto ? 63 - __builtin_clzll(upto) : 0;
    const unsigned long long after = first == 63 ? 0ull : (starts >> (first + 1));
    const int end = after ? first + 1 + (__ffsll((long long)after) - 1) : nvalid;
    len = end - first;
}

__global__ __launch_bounds__(256) void grid_count_kernel(CloudView in, int cap, const GridMeta* meta, int* cell_cnt, int* cell_of) {
    const int b = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    const bool valid = i < min(in.n[b], cap);
    int c = 0;
    if (valid) {
        const GridMeta g = meta[b];
        const float* p = in.xyz + ((size_t)b * cap + i) * 3;
        const int cx = cell_coord((double)p[0], g.ox, g.inv, g.gx);
        const int cy = cell_coord((double)p[1], g.oy, g.inv, g.gy);
        const int cz = cell_coord((double)p[2], g.oz, g.inv, g.gz);
        c = (cz * g.gy + cy) * g.gx + cx;
        cell_of[(size_t)b * cap + i] = c;
    }
    int first, len;
    cell_runs(valid, c, first, len);
    if (valid && (int)(threadIdx.x & 63) == first) atomicAdd(&cell_cnt[(size_t)b * GRID_CELLS + c], len);
}

// exclusive scan of the per-frame cell counts in three parallel passes (segment sums, scan of sums, write-back)
__global__ __launch_bounds__(256) void grid_scan_a_kernel(const int* cell_cnt, int* seg_sum, GridMeta* meta) {
    const int b = blockIdx.y, seg = blockIdx.x, t = threadIdx.x;
    const int4* c = reinterpret_cast<const int4*>(cell_cnt + (size_t)b * GRID_CELLS + (size_t)seg * SCAN_SEG) + t * 2;
    const int4 a = c[0], d = c[1];
    int s = a.x + a.y + a.z + a.w + d.x + d.y + d.z + d.w;
    int occ = (a.x > 0) + (a.y > 0) + (a.z > 0) + (a.w > 0) + (d.x > 0) + (d.y > 0) + (d.z > 0) + (d.w > 0);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) occ += __shfl_xor(occ, o);
    if ((t & 63) == 0 && occ) atomicAdd(&meta[b].occupied, occ);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    __shared__ int ws[4];
    if ((t & 63) == 0) ws[t >> 6] = s;
    __syncthreads();
    if (t == 0) seg_sum[(size_t)b * SCAN_NSEG + seg] = ws[0] + ws[1] + ws[2] + ws[3];
}
__global__ __launch_bounds__(SCAN_NSEG) void grid_scan_b_kernel(int* seg_sum, int* cell_start) {
    const int b = blockIdx.x, t = threadIdx.x;
    __shared__ int ps[SCAN_NSEG];
    const int v = seg_sum[(size_t)b * SCAN_NSEG + t];
    ps[t] = v;
    __syncthreads();
    for (int off = 1; off < SCAN_NSEG; off <<= 1) {
        const int a = t >= off ? ps[t - off] : 0;
        __syncthreads();
        ps[t] += a;
        __syncthreads();
    }
    seg_sum[(size_t)b * SCAN_NSEG + t] = ps[t] - v;          // exclusive
    if (t == SCAN_NSEG - 1) cell_start[(size_t)b * (GRID_CELLS + 1) + GRID_CELLS] = ps[t];
}
__global__ __launch_bounds__(256) void grid_scan_c_kernel(int* cell_cnt, const int* seg_sum, int* cell_start) {
    const int b = blockIdx.y, seg = blockIdx.x, t = threadIdx.x;
    int4* c = reinterpret_cast<int4*>(cell_cnt + (size_t)b * GRID_CELLS + (size_t)seg * SCAN_SEG) + t * 2;
    const int4 a = c[0], d = c[1];
    const int v[8] = {a.x, a.y, a.z, a.w, d.x, d.y, d.z, d.w};
    int s = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) s += v[j];
    __shared__ int ps[256];
    ps[t] = s;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {
        const int x = t >= off ? ps[t - off] : 0;
        __syncthreads();
        ps[t] += x;
        __syncthreads();
    }
    int run = seg_sum[(size_t)b * SCAN_NSEG + seg] + ps[t] - s;
    int* st = cell_start + (size_t)b * (GRID_CELLS + 1) + (size_t)seg * SCAN_SEG + t * 8;
#pragma unroll
    for (int j = 0; j < 8; ++j) { st[j] = run; run += v[j]; }
    c[0] = make_int4(0, 0, 0, 0);       // counts become scatter cursors
    c[1] = make_int4(0, 0, 0, 0);
}

__global__ __launch_bounds__(256) void grid_scatter_kernel(CloudView in, int cap, int* cell_cnt, const int* cell_start,
                                                           const int* cell_of, int* sidx, float* sxyz) {
    const int b = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    const bool valid = i < min(in.n[b], cap);
    const int c = valid ? cell_of[(size_t)b * cap + i] : 0;
    int first, len;
    cell_runs(valid, c, first, len);
    const int lane = threadIdx.x & 63;
    int base = 0;
    if (valid && lane == first) base = atomicAdd(&cell_cnt[(size_t)b * GRID_CELLS + c], len);      // the run's slots in the cell
    base = __shfl(base, first);
    if (!valid) return;
    const int pos = cell_start[(size_t)b * (GRID_CELLS + 1) + c] + base + (lane - first);
    sidx[(size_t)b * cap + pos] = i;
    const float* p = in.xyz + ((size_t)b * cap + i) * 3;
    reinterpret_cast<float4*>(sxyz)[(size_t)b * cap + pos] = make_float4(p[0], p[1], p[2], 0.f);
}

__device__ __forceinline__ double dist2(double ax, double ay, double az, const float* p) {
    const double dx = ax - (double)p[0], dy = ay - (double)p[1], dz = az - (double)p[2];
    return (dx * dx + dy * dy) + dz * dz;
}

template <int CAPK>
struct TopK {
    double v[CAPK];
    double kth;      // v[k-1]
    int k;
    __device__ __forceinline__ void init(int k_) {
        k = k_;
#pragma unroll
        for (int t = 0; t < CAPK; ++t) v[t] = INFINITY;
        kth = INFINITY;
    }
    __device__ __forceinline__ void push(double d) {
        if (!(d < kth)) return;
#pragma unroll
        for (int t = 0; t < CAPK; ++t) {      // bubble the new value into the ascending array (v_min_f64 / v_max_f64;
            const double lo = fmin(d, v[t]);  // squared distances are never NaN)
            const double hi = fmax(d, v[t]);
            v[t] = lo; d = hi;
        }
        if (k == CAPK) {
            kth = v[CAPK - 1];
        } else {
            double kv = v[0];
#pragma unroll
            for (int t = 1; t < CAPK; ++t) kv = (t == k - 1) ? v[t] : kv;
            kth = kv;
        }
    }
};

// per-axis squared distance bounds from coordinate q to grid layer idx (box slightly inflated: a point may sit an ulp
// outside its nominal cell).  Clamped face layers are unbounded outward.
__device__ __forceinline__ void axis_bounds(double q, double o, double cell, int idx, int g, double& dmin, double& dmax) {
    const double eps = 1e-9 * cell;
    const double lo = o + (double)idx * cell - eps, hi = o + (double)(idx + 1) * cell + eps;
    const bool open_lo = idx == 0, open_hi = idx == g - 1;
    double a = 0.0;
    if (!open_lo && q < lo) a = lo - q;
    if (!open_hi && q > hi) a = q - hi;
    dmin = a;
    dmax = (open_lo || open_hi) ? INFINITY : fmax(q - lo, hi - q);
}

// points of shell row (dz, dy) at Chebyshev radius r around cell (cx, cy, cz): face rows are one contiguous range of the
// cell-sorted points, interior rows contribute their two end cells
template <class F>
__device__ __forceinline__ void shell_row(const GridMeta& g, const int* st, int cx, int cy, int cz, int r, int dz, int dy, F&& visit) {
    const int z = cz + dz, y = cy + dy;
    if (z < 0 || z >= g.gz || y < 0 || y >= g.gy) return;
    const int rowbase = (z * g.gy + y) * g.gx;
    const bool face = (dz == -r || dz == r || dy == -r || dy == r);
    if (face) {
        const int x0 = max(cx - r, 0), x1 = min(cx + r, g.gx - 1);
        const int e = st[rowbase + x1 + 1];
        visit(st[rowbase + x0], e);
    } else {
        const int step = r == 0 ? 1 : 2 * r;
        for (int dx = -r; dx <= r; dx += step) {
            const int x = cx + dx;
            if (x < 0 || x >= g.gx) continue;
            const int e = st[rowbase + x + 1];
            visit(st[rowbase + x], e);
        }
    }
}
// the candidates [t0, t1) of a range, four at a time: the twelve coordinate loads are issued together, so the search pays
// one memory latency per four candidates instead of one each.
// (Round 3, measured and taken out again: an f32 candidate test in front of the f64 distance -- d32 within 3e-7 of the exact d, so
//  d32 > kth (1 + 2e-6) rejects without the nine half-rate f64 operations, exactness untouched -- made the statistical filter SLOWER,
//  2.15 -> 2.42 ms per 32 frames (2.96 with the bound converted per candidate), and left the radius filter at 1.50: the search waits
//  for its gathers, not for its arithmetic; the extra compare + branch per candidate only lengthens the dependent chain.)
template <class F>
__device__ __forceinline__ void visit_points(const float* __restrict__ pts, double qx, double qy, double qz, int t0, int t1, F&& push) {
    for (int t = t0; t < t1; t += 4) {
        float c[4][3];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int tt = min(t + u, t1 - 1);
            { const float4 v4 = reinterpret_cast<const float4*>(pts)[tt]; c[u][0] = v4.x; c[u][1] = v4.y; c[u][2] = v4.z; }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (t + u < t1) push(dist2(qx, qy, qz, c[u]));
    }
}

// one thread per query for the first SOR_RSOFT shells (99 % of a dense cloud); a query whose k-th neighbour is still
// farther than the searched shells is appended to the frame's hard list (one slow lane would stall its whole wave)
template <int CAPK>
__global__ __launch_bounds__(256) void sor_knn_kernel(CloudView in, int cap, const GridMeta* meta, const int* cell_start,
                                                      const int* sidx, const float* sxyz, int k, double* mean_d, int* hard_n, int* hard_idx,
                                                      double* hard_best) {
    const int b = blockIdx.y;
    const int j = blockIdx.x * 256 + threadIdx.x;
    const int n = min(in.n[b], cap);
    if (j >= n) return;
    const GridMeta g = meta[b];
    const int* st = cell_start + (size_t)b * (GRID_CELLS + 1);
    const float* pts = sxyz + (size_t)b * cap * 4;      // float4 per point
    const float* q = pts + (size_t)j * 4;
    const double qx = q[0], qy = q[1], qz = q[2];
    const int cx = cell_coord(qx, g.ox, g.inv, g.gx), cy = cell_coord(qy, g.oy, g.inv, g.gy), cz = cell_coord(qz, g.oz, g.inv, g.gz);
    const int kk = k < n ? k : n;
    TopK<CAPK> top;
    top.init(kk);
    const int rall = max(g.gx, max(g.gy, g.gz));
    bool done = false;
    // shells 0 and 1 (every query visits them: the bound of shell 0 is zero): the cell ranges of all ten runs -- the own cell,
    // eight face rows, the two end cells of the centre row -- are fetched FIRST, so the search pays one gather latency for
    // them instead of one per row
    {
        int t0[11], t1[11];
#pragma unroll
        for (int i = 0; i < 11; ++i) { t0[i] = 0; t1[i] = 0; }
        {
            const int rb = (cz * g.gy + cy) * g.gx;
            t0[0] = st[rb + cx]; t1[0] = st[rb + cx + 1];
            if (cx - 1 >= 0) { t0[9] = st[rb + cx - 1]; t1[9] = t0[0]; }
            if (cx + 1 < g.gx) { t0[10] = t1[0]; t1[10] = st[rb + cx + 2]; }
        }
        const int x0 = max(cx - 1, 0), x1 = min(cx + 1, g.gx - 1);
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            if (i == 4) continue;
            const int z = cz + i / 3 - 1, y = cy + i % 3 - 1;
            if (z < 0 || z >= g.gz || y < 0 || y >= g.gy) continue;
            const int rb = (z * g.gy + y) * g.gx;
            const int slot = i < 4 ? i + 1 : i;
            t0[slot] = st[rb + x0]; t1[slot] = st[rb + x1 + 1];
        }
        visit_points(pts, qx, qy, qz, t0[0], t1[0], [&](double d) { top.push(d); });       // shell 0
        if (top.kth <= 0.0 || rall == 0) done = true;      // (the bound of shell 0 is zero)
        if (!done) {
            // shell 1 in the order of the generic walk below: (dz, dy) rows ascending, centre row = its two end cells
#pragma unroll
            for (int i = 0; i < 9; ++i) {
                if (i == 4) {
                    visit_points(pts, qx, qy, qz, t0[9], t1[9], [&](double d) { top.push(d); });
                    visit_points(pts, qx, qy, qz, t0[10], t1[10], [&](double d) { top.push(d); });
                } else {
                    const int slot = i < 4 ? i + 1 : i;
                    visit_points(pts, qx, qy, qz, t0[slot], t1[slot], [&](double d) { top.push(d); });
                }
            }
            const double bound = g.cell * (1.0 - 1e-9);
            if (top.kth <= bound * bound) done = true;
            if (1 >= rall) done = true;
        }
    }
    for (int r = 2; r <= SOR_RSOFT && !done; ++r) {
        for (int dz = -r; dz <= r; ++dz) {
            double zmin = 0.0, zmax;
            if (r >= 2 && cz + dz >= 0 && cz + dz < g.gz) axis_bounds(qz, g.oz, g.cell, cz + dz, g.gz, zmin, zmax);
            for (int dy = -r; dy <= r; ++dy) {
                if (r >= 2) {       // a row whose cells all lie farther than the current k-th distance cannot contribute
                    double ymin = 0.0, ymax;
                    if (cy + dy >= 0 && cy + dy < g.gy) axis_bounds(qy, g.oy, g.cell, cy + dy, g.gy, ymin, ymax);
                    if (ymin * ymin + zmin * zmin >= top.kth) continue;
                }
                shell_row(g, st, cx, cy, cz, r, dz, dy, [&](int t0, int t1) {
                    visit_points(pts, qx, qy, qz, t0, t1, [&](double d) { top.push(d); });
                });
            }
        }
        const double bound = (double)r * g.cell * (1.0 - 1e-9);
        if (top.kth <= bound * bound) done = true;    // nothing unvisited can be closer than r cells
        if (r >= rall) done = true;                    // whole grid visited
    }
    if (!done) {
        const int slot = atomicAdd(&hard_n[b], 1);
        hard_idx[(size_t)b * cap + slot] = j;
        double* hb = hard_best + ((size_t)b * cap + slot) * CAPK;
#pragma unroll
        for (int t = 0; t < CAPK; ++t) hb[t] = top.v[t];
        return;
    }
    double acc = 0.0;
#pragma unroll
    for (int t = 0; t < CAPK; ++t) if (t < kk) acc = acc + sqrt(top.v[t]);   // ascending, sequential adds
    mean_d[(size_t)b * cap + sidx[(size_t)b * cap + j]] = acc / (double)kk;
}

__device__ __forceinline__ double wave_min_f64(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const double t = __shfl_xor(v, o); v = t < v ? t : v; }
    return v;
}

// the hard queries, one WAVE each: the rows of a shell are dealt to the lanes, every lane keeps the k best of its rows
// (pruned by the wave's current k-th distance), and after each shell the 64 sorted lists are merged into the wave's k
// best.  Same candidates, same canonical d2, same ascending summation as the per-thread search: identical result.
template <int CAPK>
__global__ __launch_bounds__(256) void sor_knn_hard_kernel(CloudView in, int cap, const GridMeta* meta, const int* cell_start,
                                                           const int* sidx, const float* sxyz, int k, double* mean_d, const int* hard_n,
                                                           const int* hard_idx, const double* hard_best) {
    const int b = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int n = min(in.n[b], cap);
    const int nh = hard_n[b];
    const GridMeta g = meta[b];
    const int* st = cell_start + (size_t)b * (GRID_CELLS + 1);
    const float* pts = sxyz + (size_t)b * cap * 4;      // float4 per point
    const int kk = k < n ? k : n;
    const int rall = max(g.gx, max(g.gy, g.gz));
    for (int h = blockIdx.x * 4 + (threadIdx.x >> 6); h < nh; h += gridDim.x * 4) {
        const int j = hard_idx[(size_t)b * cap + h];
        const float* q = pts + (size_t)j * 4;
        const double qx = q[0], qy = q[1], qz = q[2];
        const int cx = cell_coord(qx, g.ox, g.inv, g.gx), cy = cell_coord(qy, g.oy, g.inv, g.gy), cz = cell_coord(qz, g.oz, g.inv, g.gz);
        TopK<CAPK> top;          // this lane's candidates of the current shell
        double best[CAPK];       // the wave's k best so far (same in every lane), ascending: shells 0..SOR_RSOFT from the
        const double* hb = hard_best + ((size_t)b * cap + h) * CAPK;      // per-thread search
#pragma unroll
        for (int t = 0; t < CAPK; ++t) best[t] = hb[t];
        double gk = best[0];     // best[kk-1]
#pragma unroll
        for (int t = 1; t < CAPK; ++t) gk = (t == kk - 1) ? best[t] : gk;
        auto merge = [&]() {     // best <- k smallest of best U all lanes' lists
            double nb[CAPK];
#pragma unroll
            for (int t = 0; t < CAPK; ++t) nb[t] = INFINITY;
            if (lane == 0) {     // lane 0 also offers the previous best list: merge it into its own list first
#pragma unroll
                for (int t = 0; t < CAPK; ++t) { top.kth = INFINITY; if (best[t] < INFINITY) top.push(best[t]); }
            }
#pragma unroll
            for (int t = 0; t < CAPK; ++t) {
                if (t < kk) {
                    const double m = wave_min_f64(top.v[0]);
                    nb[t] = m;
                    const unsigned long long owners = __ballot(top.v[0] == m);
                    if (m < INFINITY && lane == (int)__builtin_ctzll(owners)) {      // pop the head of ONE owning lane
#pragma unroll
                        for (int u = 0; u + 1 < CAPK; ++u) top.v[u] = top.v[u + 1];
                        top.v[CAPK - 1] = INFINITY;
                    }
                }
            }
#pragma unroll
            for (int t = 0; t < CAPK; ++t) best[t] = nb[t];
            double kv = best[0];
#pragma unroll
            for (int t = 1; t < CAPK; ++t) kv = (t == kk - 1) ? best[t] : kv;
            gk = kv;
        };
        bool done = false;
        for (int r = SOR_RSOFT + 1; r <= SOR_RMAX && !done; ++r) {
            top.init(kk);
            top.kth = gk;                       // prune by the wave's k-th distance; lists hold only this shell's candidates
            const int side = 2 * r + 1;
            for (int t = lane; t < side * side; t += 64) {
                const int dz = t / side - r, dy = t % side - r;
                double zmin = 0.0, ymin = 0.0, dmax;     // a row farther than the k-th distance cannot contribute
                if (cz + dz >= 0 && cz + dz < g.gz) axis_bounds(qz, g.oz, g.cell, cz + dz, g.gz, zmin, dmax);
                if (cy + dy >= 0 && cy + dy < g.gy) axis_bounds(qy, g.oy, g.cell, cy + dy, g.gy, ymin, dmax);
                if (ymin * ymin + zmin * zmin >= gk) continue;
                shell_row(g, st, cx, cy, cz, r, dz, dy, [&](int t0, int t1) {
                    visit_points(pts, qx, qy, qz, t0, t1, [&](double d) { if (d < gk) top.push(d); });
                });
            }
            merge();
            const double bound = (double)r * g.cell * (1.0 - 1e-9);
            if (gk <= bound * bound) done = true;
            if (r >= rall) done = true;
        }
        if (!done) {                             // isolated point: exact brute force over the frame
#pragma unroll
            for (int t = 0; t < CAPK; ++t) best[t] = INFINITY;
            gk = INFINITY;
            top.init(kk);
            for (int t = lane; t < n; t += 64) top.push(dist2(qx, qy, qz, pts + (size_t)t * 4));
            merge();
        }
        if (lane == 0) {
            double acc = 0.0;
#pragma unroll
            for (int t = 0; t < CAPK; ++t) if (t < kk) acc = acc + sqrt(best[t]);
            mean_d[(size_t)b * cap + sidx[(size_t)b * cap + j]] = acc / (double)kk;
        }
    }
}

template <bool COMPACT>
__global__ __launch_bounds__(TB) void sor_select_kernel(CloudView in, CloudOut out, int cap, double ratio, const double* mean_d, double* params) {
    FRAME_VIEW();
    const double* md = mean_d + (size_t)b * cap;
    double s = 0.0;
#pragma unroll 4
    for (int i = threadIdx.x; i < n; i += TB) { const double v = md[i]; if (v > 0.0) s += v; }
    const double cloud_mean = block_sum_f64(s, L) / (double)n;
    double q = 0.0;
#pragma unroll 4
    for (int i = threadIdx.x; i < n; i += TB) { const double v = md[i]; if (v > 0.0) q += (v - cloud_mean) * (v - cloud_mean); }
    const double sq = block_sum_f64(q, L);
    const double stdv = sqrt(sq / (double)(n - 1));
    const double thr = cloud_mean + ratio * stdv;
    if (!COMPACT) {
        if (threadIdx.x == 0) params[b * CMP_PARAMS] = thr;
        return;
    }
    block_compact(xyz, rgb, n, oxyz, orgb, on, cap, [=](int i, float, float, float) {
        const double v = md[i];
        return v > 0.0 && v < thr;
    }, L);
}

constexpr int ROR_RINGS = 4;       // cell = radius / 4
__global__ __launch_bounds__(256) void ror_count_kernel(CloudView in, int cap, const GridMeta* meta, const int* cell_start,
                                                        const int* sidx, const float* sxyz, int nb, double r2, uint8_t* keep) {
    const int b = blockIdx.y;
    const int j = blockIdx.x * 256 + threadIdx.x;
    const int n = min(in.n[b], cap);
    if (j >= n) return;
    const GridMeta g = meta[b];
    const int* st = cell_start + (size_t)b * (GRID_CELLS + 1);
    const float* pts = sxyz + (size_t)b * cap * 4;      // float4 per point
    const float* q = pts + (size_t)j * 4;
    const double qx = q[0], qy = q[1], qz = q[2];
    const int cx = cell_coord(qx, g.ox, g.inv, g.gx), cy = cell_coord(qy, g.oy, g.inv, g.gy), cz = cell_coord(qz, g.oz, g.inv, g.gz);
    int cnt = 0;
    // fast accept: every point of the 3 x 3 x 3 cells around the query's cell is closer than 2 sqrt(3) cell = 0.866 (r / 4) (1 + 1e-6)
    // ... = 0.433 r < r away (cells are r / 4 wide; the 1e-9 cell inflation of axis_bounds is far inside that margin), PROVIDED
    // none of them is a face layer that holds clamped or non-finite points (grid_covers).  Nine row lookups in the cell-sorted prefix array; a
    // dense cloud (every interior road point) is decided here, the exact ring walk below only sees the sparse remainder.
    const bool covers = grid_covers(g);                     // else a face layer may hold clamped (far) or non-finite points
    if (3.4641016151377545 * g.cell * (1.0 + 1e-6) < sqrt(r2) &&
        (covers || (cx >= 2 && cx + 2 < g.gx && cy >= 2 && cy + 2 < g.gy && cz >= 2 && cz + 2 < g.gz))) {
        int c27 = 0;
        const int x0 = max(cx - 1, 0), x1 = min(cx + 1, g.gx - 1);
#pragma unroll
        for (int dz = -1; dz <= 1; ++dz)
#pragma unroll
            for (int dy = -1; dy <= 1; ++dy) {
                const int z = cz + dz, y = cy + dy;
                if (z < 0 || z >= g.gz || y < 0 || y >= g.gy) continue;
                const int rb = (z * g.gy + y) * g.gx;
                c27 += st[rb + x1 + 1] - st[rb + x0];
            }
        if (c27 > nb) { keep[(size_t)b * cap + sidx[(size_t)b * cap + j]] = 1; return; }
    }
    // The exact walk over the (2 ROR_RINGS + 1)^3 cells around the query, ROW by row (dz, dy fixed, the nine x cells of the row), rows in
    // rings of growing max(|dz|, |dy|).  A row's points are one range of the cell-sorted array: an empty row costs two loads, a row of a
    // few points is tested point by point.  Otherwise the cells wholly inside the ball are one x interval around cx (the per-axis far
    // bound grows with |x - cx|; clamped face layers are never "inside") and are counted with ONE difference of the prefix array; only the
    // cells the sphere cuts are read point by point.  The x bounds of the nine cells do not depend on the row: they are computed once.
    // Same conservative bounds and the same strict d2 < r2 test per candidate as a cell-by-cell walk, and only the decision count > nb
    // leaves the kernel: the same decision.
    // (Round 3.  SQ counters of the cell-by-cell form of this walk: 684 M wave-level VALU instructions per 32 frames, ~9000 per wave -- a
    // wave is as slow as its slowest lane, 64 % of the bench cloud's queries fail the fast accept above, 2 % are true outliers that walk all
    // 729 cells, and every lane is on a path of its own: 1.52 ms.  Row-wise with one prefix difference per run of inside cells 1.36 ms, x
    // bounds hoisted 1.32 ms.  Measured and dropped: batched row loads (1.53: it is not latency); a wave per undecided query (6.7 + 0.8 ms
    // after the fast accept, 1.9 + 0.8 after ring 1, 0.42 + 1.02 after ring 2 with a row per lane); a 57-cell accept + compaction of the
    // rest in front of the walk (0.40 + 0.88 ms: the 42 gathers per query into the 2-MB-per-frame prefix array cost what they save).)
    const int xa = max(cx - ROR_RINGS, 0), xb = min(cx + ROR_RINGS, g.gx - 1);
    constexpr int NX = 2 * ROR_RINGS + 1;
    double xmn2[NX], xmx2[NX];
#pragma unroll
    for (int i = 0; i < NX; ++i) {
        double xmin = 0.0, xmax = 0.0;
        if (xa + i <= xb) axis_bounds(qx, g.ox, g.cell, xa + i, g.gx, xmin, xmax);
        xmn2[i] = xmin * xmin; xmx2[i] = xmax * xmax;
    }
    auto test_points = [&](int t0, int t1) {                     // four candidates per memory latency
        for (int t = t0; t < t1 && cnt <= nb; t += 4) {
            float c4[4][3];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int tt = min(t + u, t1 - 1);
                { const float4 v4 = reinterpret_cast<const float4*>(pts)[tt]; c4[u][0] = v4.x; c4[u][1] = v4.y; c4[u][2] = v4.z; }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) cnt += (t + u < t1) && dist2(qx, qy, qz, c4[u]) < r2;
        }
    };
    for (int rr = 0; rr <= ROR_RINGS && cnt <= nb; ++rr) {
        for (int dz = -rr; dz <= rr && cnt <= nb; ++dz) {
            const int z = cz + dz;
            if (z < 0 || z >= g.gz) continue;
            double zmin, zmax;
            axis_bounds(qz, g.oz, g.cell, z, g.gz, zmin, zmax);
            if (zmin * zmin > r2) continue;
            const int dystep = (dz == -rr || dz == rr || rr == 0) ? 1 : 2 * rr;      // rows of this ring: the two faces in z, else dy = -rr, +rr
            for (int dy = -rr; dy <= rr && cnt <= nb; dy += dystep) {
                const int y = cy + dy;
                if (y < 0 || y >= g.gy) continue;
                double ymin, ymax;
                axis_bounds(qy, g.oy, g.cell, y, g.gy, ymin, ymax);
                const double yzmin = ymin * ymin + zmin * zmin;
                if (yzmin > r2) continue;
                const double yzmax = ymax * ymax + zmax * zmax;
                const int rb = (z * g.gy + y) * g.gx;
                const int rs = st[rb + xa], re = st[rb + xb + 1];
                if (rs == re) continue;
                if (re - rs <= 8) { test_points(rs, re); continue; }
                // classify the row's cells from the bounds alone: the inside interval [i0, i1], the cut cells as a bit mask
                int i0 = NX, i1 = -1;
                unsigned cut = 0;
#pragma unroll
                for (int i = 0; i < NX; ++i) {
                    if (xa + i > xb) continue;
                    // (the same two comparisons as the cell-by-cell walk, with the row's terms moved to the other side: xmin^2 + yzmin > r2
                    //  and xmax^2 + yzmax < r2 are evaluated as written there to keep the decisions bit-identical)
                    if (xmn2[i] + yzmin > r2) continue;
                    if (xmx2[i] + yzmax < r2) { i0 = i < i0 ? i : i0; i1 = i; }
                    else cut |= 1u << i;
                }
                if (i1 >= i0) cnt += st[rb + xa + i1 + 1] - st[rb + xa + i0];
                while (cut && cnt <= nb) {
                    const int x = xa + __builtin_ctz(cut);
                    cut &= cut - 1;
                    test_points(st[rb + x], st[rb + x + 1]);
                }
            }
        }
    }
    keep[(size_t)b * cap + sidx[(size_t)b * cap + j]] = cnt > nb;    // self included, strict '<' on d2 (FLANN radius search)
}

__global__ __launch_bounds__(TB) void keep_select_kernel(CloudView in, CloudOut out, int cap, const uint8_t* keep) {
    FRAME_VIEW();
    const uint8_t* kp = keep + (size_t)b * cap;
    block_compact(xyz, rgb, n, oxyz, orgb, on, cap, [=](int i, float, float, float) { return kp[i] != 0; }, L);
}

// zero the cell counters (hipMemsetAsync of 64 MiB ran at ~270 GB/s here; 16-byte stores reach the HBM rate)
__global__ __launch_bounds__(256) void zero16_kernel(uint4* p, size_t n16) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) p[i] = uint4{0u, 0u, 0u, 0u};
}

static void build_grid(CloudView in, int B, int cap, double fixed_cell, const O3dScratch& sc, hipStream_t s, bool keep_meta = false,
                       bool occupancy_only = false) {
    hipLaunchKernelGGL(zero16_kernel, dim3(2048), dim3(256), 0, s, reinterpret_cast<uint4*>(sc.cell_cnt), (size_t)B * GRID_CELLS / 4 + (size_t)B * 2);
    if (!keep_meta) {
        unsigned* acc = reinterpret_cast<unsigned*>(sc.cell_cnt + (size_t)B * GRID_CELLS);
        hipLaunchKernelGGL(grid_bbox_kernel, dim3((unsigned)std::min((cap + 255) / 256, 64), B), dim3(256), 0, s, in, cap, acc);
        hipLaunchKernelGGL(grid_meta_kernel, dim3((B + 63) / 64), dim3(64), 0, s, in, cap, B, fixed_cell, acc, sc.meta);
    }
    dim3 grid((cap + 255) / 256, B);
    hipLaunchKernelGGL(grid_count_kernel, grid, dim3(256), 0, s, in, cap, sc.meta, sc.cell_cnt, sc.cell_of);
    hipLaunchKernelGGL(grid_scan_a_kernel, dim3(SCAN_NSEG, B), dim3(256), 0, s, sc.cell_cnt, sc.seg_sum, sc.meta);
    if (occupancy_only) return;       // the caller only wants GridMeta::occupied (cell-size refinement)
    hipLaunchKernelGGL(grid_scan_b_kernel, dim3(B), dim3(SCAN_NSEG), 0, s, sc.seg_sum, sc.cell_start);
    hipLaunchKernelGGL(grid_scan_c_kernel, dim3(SCAN_NSEG, B), dim3(256), 0, s, sc.cell_cnt, sc.seg_sum, sc.cell_start);
    hipLaunchKernelGGL(grid_scatter_kernel, grid, dim3(256), 0, s, in, cap, sc.cell_cnt, sc.cell_start, sc.cell_of, sc.sidx, sc.sxyz);
}

hipError_t launch_sor(CloudView in, CloudOut out, int B, int cap, int k, double ratio, void* scratch, double* mean_out, void* cscratch,
                      hipStream_t s) {
    if (k > KMAX) return hipErrorInvalidValue;
    O3dScratch sc = carve(scratch, B, cap);
    build_grid(in, B, cap, 0.0, sc, s, /*keep_meta=*/false, /*occupancy_only=*/true);
    const double occ_target = 6.0;      // points per occupied cell (measured on the bench cloud: 4 -> 6.2 ms, 6 -> 5.35, 8 -> 5.47, 12 -> 5.86)
    hipLaunchKernelGGL(grid_refine_kernel, dim3((B + 63) / 64), dim3(64), 0, s, in, cap, sc.meta, B, occ_target);   // re-size the cells from the
    build_grid(in, B, cap, 0.0, sc, s, /*keep_meta=*/true);                                              // measured occupancy, rebuild
    double* md = mean_out ? mean_out : sc.mean_d;
    // the top-k array is walked by every lane of a wave whenever ANY lane inserts: keep it as short as k allows
    hipMemsetAsync(sc.hard_n, 0, (size_t)B * 4, s);
    const dim3 qgrid((cap + 255) / 256, B), hgrid(64, B);
    if (k <= 10) {
        hipLaunchKernelGGL(sor_knn_kernel<10>, qgrid, dim3(256), 0, s, in, cap, sc.meta, sc.cell_start, sc.sidx, sc.sxyz, k, md, sc.hard_n, sc.cell_of, sc.hard_best);
        hipLaunchKernelGGL(sor_knn_hard_kernel<10>, hgrid, dim3(256), 0, s, in, cap, sc.meta, sc.cell_start, sc.sidx, sc.sxyz, k, md, sc.hard_n, sc.cell_of, sc.hard_best);
    } else {
        hipLaunchKernelGGL(sor_knn_kernel<KMAX>, qgrid, dim3(256), 0, s, in, cap, sc.meta, sc.cell_start, sc.sidx, sc.sxyz, k, md, sc.hard_n, sc.cell_of, sc.hard_best);
        hipLaunchKernelGGL(sor_knn_hard_kernel<KMAX>, hgrid, dim3(256), 0, s, in, cap, sc.meta, sc.cell_start, sc.sidx, sc.sxyz, k, md, sc.hard_n, sc.cell_of, sc.hard_best);
    }
    if (cscratch && in.xyz != out.xyz) {
        const CmpScratch c = cmp_carve(cscratch, B);
        hipLaunchKernelGGL(sor_select_kernel<false>, dim3(B), dim3(TB), 0, s, in, out, cap, ratio, md, c.params);
        cmp_run(in, out, B, cap, SorPred{md, cap, c.params}, c, s);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(sor_select_kernel<true>, dim3(B), dim3(TB), 0, s, in, out, cap, ratio, md, (double*)nullptr);
    return hipGetLastError();
}

hipError_t launch_ror(CloudView in, CloudOut out, int B, int cap, int nb, double radius, void* scratch, void* cscratch, hipStream_t s) {
    O3dScratch sc = carve(scratch, B, cap);
    build_grid(in, B, cap, radius / ROR_RINGS * (1.0 + 1e-6), sc, s);
    hipLaunchKernelGGL(ror_count_kernel, dim3((cap + 255) / 256, B), dim3(256), 0, s, in, cap, sc.meta, sc.cell_start, sc.sidx,
                       sc.sxyz, nb, radius * radius, sc.keep);
    if (cscratch && in.xyz != out.xyz) {
        cmp_run(in, out, B, cap, KeepPred{sc.keep, cap}, cmp_carve(cscratch, B), s);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(keep_select_kernel, dim3(B), dim3(TB), 0, s, in, out, cap, sc.keep);
    return hipGetLastError();
}

}  // namespace sd
