// Host-side helper of the input stage: PNG scanline reconstruction for the frame reader that replaces
// cv2.imread (semantic_depth.py:105, semantic_depth_cityscapes_sequence.py:123).  The DEFLATE stream is inflated by zlib
// (Python's zlib module, which releases the GIL); what remains per frame is this serial byte recurrence (PNG spec §9,
// filter types 0-4) plus the channel shuffle to OpenCV's 3-channel BGR layout.  No GPU work: a 2048 x 1024 RGB frame is
// 6 MB and the Paeth/Average predictors are a dependency chain along each row — host cores do it while the GPU computes.
#include "../../include/semdepth.h"

#include <cstdlib>
#include <cstring>

extern "C" sd_status sd_png_unfilter_bgr(const uint8_t* filtered, int height, int width, int channels, uint8_t* bgr_out) {
    if (!filtered || !bgr_out || height <= 0 || width <= 0 || channels < 1 || channels > 4) return SD_ERR_INVALID;
    const int bpp = channels, stride = width * channels;
    uint8_t* prev = static_cast<uint8_t*>(std::calloc((size_t)stride, 1));
    uint8_t* cur = static_cast<uint8_t*>(std::malloc((size_t)stride));
    if (!prev || !cur) { std::free(prev); std::free(cur); return SD_ERR_INVALID; }
    for (int y = 0; y < height; ++y) {
        const uint8_t* row = filtered + (size_t)y * (stride + 1);
        const int ft = row[0];
        const uint8_t* s = row + 1;
        switch (ft) {
            case 0: std::memcpy(cur, s, (size_t)stride); break;
            case 1:
                for (int i = 0; i < bpp; ++i) cur[i] = s[i];
                for (int i = bpp; i < stride; ++i) cur[i] = (uint8_t)(s[i] + cur[i - bpp]);
                break;
            case 2:
                for (int i = 0; i < stride; ++i) cur[i] = (uint8_t)(s[i] + prev[i]);
                break;
            case 3:
                for (int i = 0; i < bpp; ++i) cur[i] = (uint8_t)(s[i] + (prev[i] >> 1));
                for (int i = bpp; i < stride; ++i) cur[i] = (uint8_t)(s[i] + ((cur[i - bpp] + prev[i]) >> 1));
                break;
            case 4:
                for (int i = 0; i < bpp; ++i) cur[i] = (uint8_t)(s[i] + prev[i]);       // paeth(0, b, 0) = b
                for (int i = bpp; i < stride; ++i) {
                    const int a = cur[i - bpp], b = prev[i], c = prev[i - bpp];
                    const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
                    const int pr = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
                    cur[i] = (uint8_t)(s[i] + pr);
                }
                break;
            default: std::free(prev); std::free(cur); return SD_ERR_INVALID;
        }
        uint8_t* o = bgr_out + (size_t)y * width * 3;
        if (channels >= 3) {                      // RGB / RGBA -> BGR (cv2.IMREAD_COLOR drops alpha)
            for (int x = 0; x < width; ++x) { o[3 * x] = cur[bpp * x + 2]; o[3 * x + 1] = cur[bpp * x + 1]; o[3 * x + 2] = cur[bpp * x]; }
        } else {                                  // gray / gray + alpha -> replicated
            for (int x = 0; x < width; ++x) { o[3 * x] = o[3 * x + 1] = o[3 * x + 2] = cur[bpp * x]; }
        }
        uint8_t* t = prev; prev = cur; cur = t;
    }
    std::free(prev); std::free(cur);
    return SD_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// Whole-file decode and the batch reader (round 3): one native call per frame -- chunk walk, zlib inflate straight into a
// scratch buffer, scanline reconstruction, BGR shuffle, palette expansion -- and one native call per BATCH that reads and
// decodes a list of files on its own threads.  Nothing of the per-frame work runs under the Python interpreter lock any
// more (the Python-side form joined the IDAT chunks, let zlib.decompress grow its output and copied the frame twice under
// the lock: 64 threads decoded 8.8x, not 64x, what one did).
// ---------------------------------------------------------------------------------------------------------------------
#include <zlib.h>

#include <atomic>
#include <cstdio>
#include <thread>
#include <vector>

namespace {

inline uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }
const uint8_t kPngSig[8] = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1a, '\n'};

struct PngHead { int w = 0, h = 0, depth = 0, ctype = 0, interlace = 0, channels = 0; };

// IHDR of a PNG byte stream; false when it is not a PNG or not one this reader takes (8-bit, non-interlaced)
bool png_head(const uint8_t* f, size_t len, PngHead& hd) {
    if (!f || len < 8 + 25 || std::memcmp(f, kPngSig, 8) != 0) return false;
    if (be32(f + 8) != 13 || std::memcmp(f + 12, "IHDR", 4) != 0) return false;
    hd.w = (int)be32(f + 16); hd.h = (int)be32(f + 20);
    hd.depth = f[24]; hd.ctype = f[25]; hd.interlace = f[28];
    switch (hd.ctype) { case 0: hd.channels = 1; break; case 2: hd.channels = 3; break; case 3: hd.channels = 1; break;
                        case 4: hd.channels = 2; break; case 6: hd.channels = 4; break; default: return false; }
    return hd.depth == 8 && hd.interlace == 0 && hd.w > 0 && hd.h > 0;
}

sd_status png_decode(const uint8_t* f, size_t len, uint8_t* out, size_t out_cap, int* h_out, int* w_out) {
    PngHead hd;
    if (!png_head(f, len, hd)) return SD_ERR_INVALID;
    if (h_out) *h_out = hd.h;
    if (w_out) *w_out = hd.w;
    if (!out) return SD_OK;                                     // (size query)
    if (out_cap < (size_t)hd.h * hd.w * 3) return SD_ERR_INVALID;
    const size_t raw_len = (size_t)hd.h * (1 + (size_t)hd.w * hd.channels);
    std::vector<uint8_t> raw(raw_len);
    z_stream zs;
    std::memset(&zs, 0, sizeof(zs));
    if (inflateInit(&zs) != Z_OK) return SD_ERR_INVALID;
    zs.next_out = raw.data();
    zs.avail_out = (uInt)raw_len;
    const uint8_t* plte = nullptr;
    size_t plte_n = 0;
    bool done = false, ok = true;
    for (size_t p = 8; p + 12 <= len && !done;) {
        const uint32_t n = be32(f + p);
        if (p + 12 + (size_t)n > len) { ok = false; break; }
        const uint8_t* tag = f + p + 4;
        const uint8_t* body = f + p + 8;
        if (!std::memcmp(tag, "IDAT", 4)) {                     // the zlib stream runs through every IDAT chunk in file order
            zs.next_in = const_cast<Bytef*>(body);
            zs.avail_in = n;
            const int r = inflate(&zs, Z_NO_FLUSH);
            if (r != Z_OK && r != Z_STREAM_END) { ok = false; break; }
        } else if (!std::memcmp(tag, "PLTE", 4)) {
            plte = body; plte_n = n / 3;
        } else if (!std::memcmp(tag, "IEND", 4)) {
            done = true;
        }
        p += 12 + (size_t)n;
    }
    const bool full = zs.total_out == raw_len;
    inflateEnd(&zs);
    if (!ok || !full) return SD_ERR_INVALID;
    const sd_status st = sd_png_unfilter_bgr(raw.data(), hd.h, hd.w, hd.channels, out);
    if (st != SD_OK) return st;
    if (hd.ctype == 3) {                                        // palette indices (replicated by the helper) -> BGR palette entries
        if (!plte) return SD_ERR_INVALID;
        const size_t npx = (size_t)hd.h * hd.w;
        for (size_t i = 0; i < npx; ++i) {
            const size_t k = out[3 * i];
            if (k >= plte_n) return SD_ERR_INVALID;
            out[3 * i] = plte[3 * k + 2]; out[3 * i + 1] = plte[3 * k + 1]; out[3 * i + 2] = plte[3 * k];
        }
    }
    return SD_OK;
}

// PNG or baseline JPEG (host_jpeg.cpp), by signature
sd_status image_decode(const uint8_t* f, size_t len, uint8_t* out, size_t out_cap, int* h_out, int* w_out) {
    if (f && len >= 2 && f[0] == 0xFF && f[1] == 0xD8) return sd_jpeg_decode_bgr(f, len, out, out_cap, h_out, w_out);
    try {                                                       // (no exception crosses the C ABI or a worker thread: an allocation failure is a refused file)
        return png_decode(f, len, out, out_cap, h_out, w_out);
    } catch (...) {
        return SD_ERR_INVALID;
    }
}

bool read_file(const char* path, std::vector<uint8_t>& buf) {
    FILE* fp = std::fopen(path, "rb");
    if (!fp) return false;
    std::fseek(fp, 0, SEEK_END);
    const long n = std::ftell(fp);
    std::fseek(fp, 0, SEEK_SET);
    if (n <= 0) { std::fclose(fp); return false; }
    buf.resize((size_t)n);
    const size_t got = std::fread(buf.data(), 1, (size_t)n, fp);
    std::fclose(fp);
    return got == (size_t)n;
}

}  // namespace

extern "C" sd_status sd_png_decode_bgr(const uint8_t* file_host, size_t len, uint8_t* bgr_out_host, size_t out_capacity, int* height_out,
                                       int* width_out) {
    try {
        return png_decode(file_host, len, bgr_out_host, out_capacity, height_out, width_out);
    } catch (...) {
        return SD_ERR_INVALID;
    }
}

extern "C" sd_status sd_image_decode_bgr(const uint8_t* file_host, size_t len, uint8_t* bgr_out_host, size_t out_capacity, int* height_out,
                                         int* width_out) {
    return image_decode(file_host, len, bgr_out_host, out_capacity, height_out, width_out);
}

extern "C" sd_status sd_decode_files_bgr(const char* const* paths, int n, int height, int width, uint8_t* out_host, size_t frame_stride,
                                         int threads, int* status_out) {
    if (!paths || n < 0 || height <= 0 || width <= 0 || !out_host || frame_stride < (size_t)height * width * 3) return SD_ERR_INVALID;
    if (threads <= 0) threads = (int)std::thread::hardware_concurrency();
    threads = threads < 1 ? 1 : (threads > n ? (n > 0 ? n : 1) : threads);
    std::atomic<int> next(0), failed(0);
    auto work = [&]() {
        std::vector<uint8_t> file;
        for (;;) {
            const int i = next.fetch_add(1);
            if (i >= n) break;
            sd_status st = SD_ERR_NOTFOUND;
            try {
                if (paths[i] && read_file(paths[i], file)) {
                    int h = 0, w = 0;
                    st = image_decode(file.data(), file.size(), nullptr, 0, &h, &w);
                    if (st == SD_OK && (h != height || w != width)) st = SD_ERR_INVALID;     // every frame of a batch has the batch's shape
                    if (st == SD_OK) st = image_decode(file.data(), file.size(), out_host + (size_t)i * frame_stride, frame_stride, nullptr, nullptr);
                }
            } catch (...) {
                st = SD_ERR_INVALID;
            }
            if (status_out) status_out[i] = st;
            if (st != SD_OK) failed.fetch_add(1);
        }
    };
    std::vector<std::thread> pool;
    try {
        for (int t = 1; t < threads; ++t) pool.emplace_back(work);
    } catch (...) {                                             // (thread creation refused: the calling thread and the workers that exist do the batch)
    }
    work();
    for (auto& th : pool) th.join();
    return failed.load() ? SD_ERR_INVALID : SD_OK;
}
