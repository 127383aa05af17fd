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
