// Host-side helper of the output stage: the vertex rows of the ASCII PLY files the reference writes for every
// frame's clouds (semantic_depth_lib/point_cloud_2_ply.py:70, numpy.savetxt with "%f %f %f %d %d %d").  savetxt formats row by row
// in Python (0.4 s for a 150 k-point cloud); this is the same text from std::to_chars -- correctly rounded fixed notation with six
// decimals, what "%f" % float(v) prints -- on a few threads.  No handle, no GPU.
#include "../../include/semdepth.h"

#include <charconv>
#include <cmath>
#include <cstring>
#include <thread>
#include <vector>

static inline char* put_f(char* p, char* end, double v) {
    if (std::isnan(v)) { if (end - p < 3) return nullptr; std::memcpy(p, "nan", 3); return p + 3; }     // Python: 'nan' whatever the sign bit
    if (std::isinf(v)) { const char* s = v < 0 ? "-inf" : "inf"; const size_t n = std::strlen(s); if ((size_t)(end - p) < n) return nullptr; std::memcpy(p, s, n); return p + n; }
    const auto r = std::to_chars(p, end, v, std::chars_format::fixed, 6);
    return r.ec == std::errc() ? r.ptr : nullptr;
}
static inline char* put_i(char* p, char* end, int64_t v) {
    const auto r = std::to_chars(p, end, v);
    return r.ec == std::errc() ? r.ptr : nullptr;
}

// rows [lo, hi) -> buf; returns bytes written or -1 (buffer too small)
static int64_t format_rows(const double* xyz, const int64_t* rgb, int64_t lo, int64_t hi, char* buf, int64_t cap) {
    char* p = buf;
    char* const end = buf + cap;
    for (int64_t i = lo; i < hi; ++i) {
        for (int j = 0; j < 3; ++j) {
            p = put_f(p, end, xyz[3 * i + j]);
            if (!p || p == end) return -1;
            *p++ = ' ';
        }
        for (int j = 0; j < 3; ++j) {
            p = put_i(p, end, rgb[3 * i + j]);
            if (!p || p == end) return -1;
            *p++ = j == 2 ? '\n' : ' ';
        }
    }
    return p - buf;
}

extern "C" int64_t sd_ply_format_rows(const double* xyz, const int64_t* rgb, int64_t n, char* out, int64_t cap, int threads) {
    if (n < 0 || cap < 0 || (n > 0 && (!xyz || !rgb || !out))) return SD_ERR_INVALID;
    if (n == 0) return 0;
    int nt = threads > 0 ? threads : (int)std::thread::hardware_concurrency();
    if (nt < 1) nt = 1;
    if (nt > 16) nt = 16;
    if ((int64_t)nt > (n + 4095) / 4096) nt = (int)((n + 4095) / 4096);       // at least 4096 rows per thread
    if (nt == 1) {
        const int64_t w = format_rows(xyz, rgb, 0, n, out, cap);
        return w < 0 ? (int64_t)SD_ERR_INVALID : w;
    }
    // each thread formats its slice into the part of `out` that is its share of the capacity; the slices are then closed up
    std::vector<int64_t> lo(nt + 1), off(nt), len(nt);
    for (int t = 0; t <= nt; ++t) lo[t] = n * t / nt;
    std::vector<std::thread> th;
    for (int t = 0; t < nt; ++t) {
        off[t] = cap / n * lo[t];          // cap / n bytes per row are available to every row
        const int64_t share = cap / n * (lo[t + 1] - lo[t]);
        th.emplace_back([&, t, share] { len[t] = format_rows(xyz, rgb, lo[t], lo[t + 1], out + off[t], share); });
    }
    for (auto& x : th) x.join();
    int64_t w = 0;
    for (int t = 0; t < nt; ++t) {
        if (len[t] < 0) return SD_ERR_INVALID;
        if (off[t] != w) std::memmove(out + w, out + off[t], (size_t)len[t]);
        w += len[t];
    }
    return w;
}
