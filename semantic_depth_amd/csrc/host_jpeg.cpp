// Host-side baseline JPEG reader of the input stage: the frames of the reference's own example are JPEGs
// (assets/images/test_munich/test_3.jpg, read by cv2.imread at semantic_depth.py:105).  cv2.imread decodes with libjpeg(-turbo) at its
// default settings and then applies the EXIF orientation; this file restates that path for baseline / extended-sequential Huffman JPEGs
// and progressive Huffman JPEGs (SOF0 / SOF1 / SOF2 -- test_3.jpg is progressive --, 8-bit, 1 or 3 components, sampling 1x1, 2x1,
// 2x2 for the chroma: what cameras and OpenCV itself write):
//   * entropy decoding per ITU T.81 (Huffman, byte stuffing, restart intervals, interleaved and single-component scans)
//   * inverse DCT = libjpeg's jidctint.c "ISLOW" integer transform (CONST_BITS 13, PASS1_BITS 2), the default dct_method
//   * chroma upsampling = libjpeg's "fancy" triangle filters (h2v1_fancy_upsample / h2v2_fancy_upsample), the default
//   * YCbCr -> RGB = jdcolor.c's 16-bit fixed-point tables
//   * EXIF orientation 1..8 as OpenCV's ExifTransform applies it (cv2.imread without IMREAD_IGNORE_ORIENTATION)
// Output: u8 [height, width, 3] BGR.  Pinned in tests/test_frame_io.py against Pillow (libjpeg-turbo, same defaults) bit for bit.
// Arithmetic-coded, lossless, 12-bit, CMYK and other sampling ratios are refused with SD_ERR_INVALID.
#include "../../include/semdepth.h"

#include <cstdint>
#include <cstring>
#include <vector>

namespace {

const uint8_t kZigzag[64] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48,
                             41, 34, 27, 20, 13, 6, 7, 14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                             30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

struct Huff {
    bool present = false;
    uint8_t bits[17] = {0};
    uint8_t vals[256] = {0};
    // canonical decoding tables (T.81 F.2.2.3)
    int mincode[17], maxcode[18], valptr[17];
    uint16_t look[512];      // 9-bit lookahead: (length << 8) | symbol, 0 = longer than 9 bits
    // false when the code-length counts do not form a prefix code (libjpeg's JERR_BAD_HUFF_TABLE: more codes of length l than 2^l
    // leaves free) -- an over-subscribed table would index look[] past its 512 entries
    bool build() {
        int code = 0, k = 0;
        for (int l = 1; l <= 16; ++l) {
            valptr[l] = k;
            mincode[l] = code;
            code += bits[l];
            k += bits[l];
            if (code > (1 << l) || k > 256) { present = false; return false; }
            maxcode[l] = bits[l] ? code - 1 : -1;
            code <<= 1;
        }
        maxcode[17] = 0x7fffffff;
        std::memset(look, 0, sizeof(look));
        code = 0; k = 0;
        for (int l = 1; l <= 9; ++l) {
            for (int i = 0; i < bits[l]; ++i, ++k, ++code) {
                const int first = code << (9 - l), n = 1 << (9 - l);
                if (first + n > 512) { present = false; return false; }
                for (int j = 0; j < n; ++j) look[first + j] = (uint16_t)((l << 8) | vals[k]);
            }
            code <<= 1;
        }
        return true;
    }
};

struct Comp {
    int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0;
    int bw = 0, bh = 0;            // blocks per row / column of the padded plane
    int pw = 0, ph = 0;            // padded plane size in samples
    int dw = 0, dh = 0;            // downsampled (real) size in samples
    int pred = 0;
    std::vector<uint8_t> plane;
    std::vector<int16_t> coef;     // progressive: quantised coefficients of every block (natural order), accumulated over the scans
};

struct BitReader {
    const uint8_t* p; const uint8_t* end;
    uint32_t acc = 0; int n = 0;
    bool hit_marker = false;
    void reset() { acc = 0; n = 0; hit_marker = false; }
    inline void fill() {
        while (n <= 24) {
            int b = 0;
            if (!hit_marker && p < end) {
                b = *p;
                if (b == 0xFF) {
                    if (p + 1 < end && p[1] == 0x00) { p += 2; }
                    else { hit_marker = true; b = 0; }          // a marker: feed zeros, leave p on the 0xFF
                } else ++p;
            }
            acc |= (uint32_t)b << (24 - n);
            n += 8;
        }
    }
    inline int peek(int k) { if (n < k) fill(); return (int)(acc >> (32 - k)); }
    inline void skip(int k) { acc <<= k; n -= k; }
    inline int get(int k) { if (!k) return 0; const int v = peek(k); skip(k); return v; }
};

inline int extend(int v, int t) { return v < (1 << (t - 1)) ? v - (1 << t) + 1 : v; }

inline int decode_sym(BitReader& br, const Huff& h) {
    const int look = br.peek(9);
    const uint16_t e = h.look[look];
    if (e) { br.skip(e >> 8); return e & 0xff; }
    int code = br.peek(16);
    for (int l = 10; l <= 16; ++l) {
        const int c = code >> (16 - l);
        if (h.maxcode[l] >= 0 && c <= h.maxcode[l] && c >= h.mincode[l]) { br.skip(l); return h.vals[h.valptr[l] + c - h.mincode[l]]; }
    }
    return -1;
}

// jidctint.c (libjpeg 6b / libjpeg-turbo, DCTSIZE 8): accurate integer inverse DCT on dequantised coefficients, output
// level-shifted by +128 and range-limited to 0..255
#define FIXC(x) ((int32_t)((x) * 8192 + 0.5))
inline int32_t descale(int64_t x, int n) { return (int32_t)((x + ((int64_t)1 << (n - 1))) >> n); }
void idct_islow(const int32_t* in, uint8_t* out, int stride) {
    constexpr int CB = 13, P1 = 2;
    const int32_t F0_298631336 = 2446, F0_390180644 = 3196, F0_541196100 = 4433, F0_765366865 = 6270, F0_899976223 = 7373,
                  F1_175875602 = 9633, F1_501321110 = 12299, F1_847759065 = 15137, F1_961570560 = 16069, F2_053119869 = 16819,
                  F2_562915447 = 20995, F3_072711026 = 25172;
    int32_t ws[64];
    for (int c = 0; c < 8; ++c) {
        const int32_t* ip = in + c;
        int32_t* wp = ws + c;
        if (!(ip[8] | ip[16] | ip[24] | ip[32] | ip[40] | ip[48] | ip[56])) {
            const int32_t dc = ip[0] * (1 << P1);
            for (int r = 0; r < 8; ++r) wp[8 * r] = dc;
            continue;
        }
        int64_t z2 = ip[16], z3 = ip[48];
        int64_t z1 = (z2 + z3) * F0_541196100;
        int64_t tmp2 = z1 + z3 * (-F1_847759065);
        int64_t tmp3 = z1 + z2 * F0_765366865;
        z2 = ip[0]; z3 = ip[32];
        int64_t tmp0 = (z2 + z3) * (1 << CB);
        int64_t tmp1 = (z2 - z3) * (1 << CB);
        const int64_t tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
        tmp0 = ip[56]; tmp1 = ip[40]; tmp2 = ip[24]; tmp3 = ip[8];
        z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2;
        int64_t z4 = tmp1 + tmp3;
        const int64_t z5 = (z3 + z4) * F1_175875602;
        tmp0 *= F0_298631336; tmp1 *= F2_053119869; tmp2 *= F3_072711026; tmp3 *= F1_501321110;
        z1 *= -F0_899976223; z2 *= -F2_562915447; z3 *= -F1_961570560; z4 *= -F0_390180644;
        z3 += z5; z4 += z5;
        tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
        wp[0] = descale(tmp10 + tmp3, CB - P1);  wp[56] = descale(tmp10 - tmp3, CB - P1);
        wp[8] = descale(tmp11 + tmp2, CB - P1);  wp[48] = descale(tmp11 - tmp2, CB - P1);
        wp[16] = descale(tmp12 + tmp1, CB - P1); wp[40] = descale(tmp12 - tmp1, CB - P1);
        wp[24] = descale(tmp13 + tmp0, CB - P1); wp[32] = descale(tmp13 - tmp0, CB - P1);
    }
    auto clamp = [](int32_t v) { v += 128; return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); };
    for (int r = 0; r < 8; ++r) {
        const int32_t* wp = ws + 8 * r;
        uint8_t* o = out + (size_t)r * stride;
        if (!(wp[1] | wp[2] | wp[3] | wp[4] | wp[5] | wp[6] | wp[7])) {
            const uint8_t dc = clamp(descale(wp[0], P1 + 3));
            for (int c = 0; c < 8; ++c) o[c] = dc;
            continue;
        }
        int64_t z2 = wp[2], z3 = wp[6];
        int64_t z1 = (z2 + z3) * F0_541196100;
        int64_t tmp2 = z1 + z3 * (-F1_847759065);
        int64_t tmp3 = z1 + z2 * F0_765366865;
        int64_t tmp0 = ((int64_t)wp[0] + wp[4]) * (1 << CB);
        int64_t tmp1 = ((int64_t)wp[0] - wp[4]) * (1 << CB);
        const int64_t tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
        tmp0 = wp[7]; tmp1 = wp[5]; tmp2 = wp[3]; tmp3 = wp[1];
        z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2;
        int64_t z4 = tmp1 + tmp3;
        const int64_t z5 = (z3 + z4) * F1_175875602;
        tmp0 *= F0_298631336; tmp1 *= F2_053119869; tmp2 *= F3_072711026; tmp3 *= F1_501321110;
        z1 *= -F0_899976223; z2 *= -F2_562915447; z3 *= -F1_961570560; z4 *= -F0_390180644;
        z3 += z5; z4 += z5;
        tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
        constexpr int S = CB + P1 + 3;
        o[0] = clamp(descale(tmp10 + tmp3, S)); o[7] = clamp(descale(tmp10 - tmp3, S));
        o[1] = clamp(descale(tmp11 + tmp2, S)); o[6] = clamp(descale(tmp11 - tmp2, S));
        o[2] = clamp(descale(tmp12 + tmp1, S)); o[5] = clamp(descale(tmp12 - tmp1, S));
        o[3] = clamp(descale(tmp13 + tmp0, S)); o[4] = clamp(descale(tmp13 - tmp0, S));
    }
}

inline uint16_t rd16(const uint8_t* p) { return (uint16_t)((p[0] << 8) | p[1]); }

// EXIF orientation (TIFF tag 0x0112) of an APP1 "Exif\0\0" segment; 1 when absent / unreadable
int exif_orientation(const uint8_t* seg, size_t n) {
    if (n < 14 || std::memcmp(seg, "Exif\0\0", 6) != 0) return 1;
    const uint8_t* t = seg + 6;
    const size_t tn = n - 6;
    bool le;
    if (t[0] == 'I' && t[1] == 'I') le = true; else if (t[0] == 'M' && t[1] == 'M') le = false; else return 1;
    auto r16 = [&](size_t o) -> uint32_t { return o + 2 > tn ? 0u : (le ? (uint32_t)(t[o] | (t[o + 1] << 8)) : (uint32_t)((t[o] << 8) | t[o + 1])); };
    auto r32 = [&](size_t o) -> uint32_t {
        if (o + 4 > tn) return 0u;
        return le ? (uint32_t)(t[o] | (t[o + 1] << 8) | (t[o + 2] << 16) | ((uint32_t)t[o + 3] << 24))
                  : (uint32_t)(((uint32_t)t[o] << 24) | (t[o + 1] << 16) | (t[o + 2] << 8) | t[o + 3]);
    };
    if (r16(2) != 42) return 1;
    const size_t ifd = r32(4);
    const uint32_t cnt = r16(ifd);
    for (uint32_t i = 0; i < cnt; ++i) {
        const size_t e = ifd + 2 + 12 * (size_t)i;
        if (e + 12 > tn) break;
        if (r16(e) == 0x0112) {
            const uint32_t v = r16(e + 8);
            return (v >= 1 && v <= 8) ? (int)v : 1;
        }
    }
    return 1;
}

struct Decoder {
    const uint8_t* f; size_t len;
    int W = 0, H = 0, ncomp = 0, hmax = 1, vmax = 1, restart = 0, orientation = 1, adobe_transform = -1;
    bool sof = false, progressive = false;
    uint16_t qt[4][64] = {};
    bool qt_ok[4] = {false, false, false, false};
    Huff dc[4], ac[4];
    Comp comp[3];

    sd_status header_only(int* h_out, int* w_out) { const sd_status st = parse(false); if (st == SD_OK) dims(h_out, w_out); return st; }
    void dims(int* h_out, int* w_out) const {
        const bool swap = orientation >= 5;
        if (h_out) *h_out = swap ? W : H;
        if (w_out) *w_out = swap ? H : W;
    }

    size_t cap_bytes = 0;                                  // capacity of the caller's output buffer (checked at the frame header)
    sd_status parse(bool decode) {
        if (len < 4 || f[0] != 0xFF || f[1] != 0xD8) return SD_ERR_INVALID;
        size_t p = 2;
        while (p + 4 <= len) {
            if (f[p] != 0xFF) return SD_ERR_INVALID;
            while (p < len && f[p] == 0xFF) ++p;             // fill bytes
            if (p >= len) return SD_ERR_INVALID;
            const int m = f[p++];
            if (m == 0xD9) break;                            // EOI
            if (m == 0x01 || (m >= 0xD0 && m <= 0xD7)) continue;
            if (p + 2 > len) return SD_ERR_INVALID;
            const size_t n = rd16(f + p);
            if (n < 2 || p + n > len) return SD_ERR_INVALID;
            const uint8_t* s = f + p + 2;
            const size_t sn = n - 2;
            if (m == 0xDB) {                                 // DQT
                size_t o = 0;
                while (o < sn) {
                    const int pq = s[o] >> 4, tq = s[o] & 15;
                    ++o;
                    if (tq > 3 || o + (pq ? 128 : 64) > sn) return SD_ERR_INVALID;
                    for (int i = 0; i < 64; ++i) { qt[tq][kZigzag[i]] = pq ? rd16(s + o + 2 * i) : s[o + i]; }
                    o += pq ? 128 : 64;
                    qt_ok[tq] = true;
                }
            } else if (m == 0xC4) {                          // DHT
                size_t o = 0;
                while (o + 17 <= sn) {
                    const int tc = s[o] >> 4, th = s[o] & 15;
                    if (tc > 1 || th > 3) return SD_ERR_INVALID;
                    Huff& h = tc ? ac[th] : dc[th];
                    int total = 0;
                    h.bits[0] = 0;
                    for (int i = 1; i <= 16; ++i) { h.bits[i] = s[o + i]; total += s[o + i]; }
                    o += 17;
                    if (total > 256 || o + total > sn) return SD_ERR_INVALID;
                    std::memcpy(h.vals, s + o, (size_t)total);
                    o += total;
                    h.present = true;
                    if (!h.build()) return SD_ERR_INVALID;
                }
            } else if (m == 0xC0 || m == 0xC1 || m == 0xC2) {     // SOF0 / SOF1 / SOF2: baseline / extended sequential / progressive, Huffman
                if (sof) return SD_ERR_INVALID;                  // one frame header per file: a second one would re-shape planes already allocated
                progressive = m == 0xC2;
                hmax = vmax = 1;
                if (sn < 6 || s[0] != 8) return SD_ERR_INVALID;
                H = rd16(s + 1); W = rd16(s + 3); ncomp = s[5];
                if (H <= 0 || W <= 0 || (ncomp != 1 && ncomp != 3) || sn < 6 + 3 * (size_t)ncomp) return SD_ERR_INVALID;
                if ((size_t)H * (size_t)W > ((size_t)1 << 28)) return SD_ERR_INVALID;      // (256 Mpixel: refuse absurd headers before allocating planes)
                if (decode && cap_bytes < (size_t)H * W * 3) return SD_ERR_INVALID;       // the caller's buffer bounds what a header can make us allocate
                for (int i = 0; i < ncomp; ++i) {
                    comp[i].id = s[6 + 3 * i]; comp[i].h = s[7 + 3 * i] >> 4; comp[i].v = s[7 + 3 * i] & 15; comp[i].tq = s[8 + 3 * i];
                    if (comp[i].h < 1 || comp[i].v < 1 || comp[i].tq > 3) return SD_ERR_INVALID;
                    hmax = comp[i].h > hmax ? comp[i].h : hmax; vmax = comp[i].v > vmax ? comp[i].v : vmax;
                }
                if (ncomp == 1) { comp[0].h = comp[0].v = 1; hmax = vmax = 1; }
                else {
                    // luma at the full rate, both chroma planes at 1x1 of (1x1 | 2x1 | 2x2): what the fancy upsamplers cover
                    if (comp[0].h != hmax || comp[0].v != vmax || comp[1].h != 1 || comp[1].v != 1 || comp[2].h != 1 || comp[2].v != 1) return SD_ERR_INVALID;
                    if (!((hmax == 1 && vmax == 1) || (hmax == 2 && vmax == 1) || (hmax == 2 && vmax == 2))) return SD_ERR_INVALID;
                }
                sof = true;
                if (!decode) { /* keep scanning for APP1 only until SOS */ }
            } else if (m >= 0xC3 && m <= 0xCF && m != 0xC4 && m != 0xC8 && m != 0xCC) {
                return SD_ERR_INVALID;                       // lossless / differential / arithmetic
            } else if (m == 0xDD) {                          // DRI
                if (sn < 2) return SD_ERR_INVALID;
                restart = rd16(s);
            } else if (m == 0xE1) {                          // APP1: EXIF
                const int o = exif_orientation(s, sn);
                if (o != 1) orientation = o;
            } else if (m == 0xEE) {                          // APP14 Adobe: colour transform flag
                if (sn >= 12 && std::memcmp(s, "Adobe", 5) == 0) adobe_transform = s[11];
            } else if (m == 0xDA) {                          // SOS
                if (!sof) return SD_ERR_INVALID;
                if (!decode) return SD_OK;
                size_t next = 0;
                const sd_status st = scan(s, sn, p + n, next);
                if (st != SD_OK) return st;
                p = next;
                continue;
            }
            p += n;
        }
        return sof ? SD_OK : SD_ERR_INVALID;
    }

    void alloc_planes() {
        const int mcux = (W + 8 * hmax - 1) / (8 * hmax), mcuy = (H + 8 * vmax - 1) / (8 * vmax);
        for (int i = 0; i < ncomp; ++i) {
            Comp& c = comp[i];
            if (!c.plane.empty()) continue;
            c.bw = mcux * c.h; c.bh = mcuy * c.v;
            c.pw = c.bw * 8; c.ph = c.bh * 8;
            c.dw = (W * c.h + hmax - 1) / hmax; c.dh = (H * c.v + vmax - 1) / vmax;
            c.plane.assign((size_t)c.pw * c.ph, 0);
            if (progressive) c.coef.assign((size_t)c.bw * c.bh * 64, 0);
        }
    }

    // ---- progressive scans (T.81 annex G; libjpeg jdphuff.c): coefficients accumulate in Comp::coef, the inverse DCT runs at the end
    int eobrun = 0;
    bool prog_dc(BitReader& br, Comp& c, int bx, int by, int Ah, int Al) {
        int16_t* blk = (bx < c.bw && by < c.bh) ? &c.coef[((size_t)by * c.bw + bx) * 64] : nullptr;
        if (Ah == 0) {
            const Huff& hd = dc[c.td];
            if (!hd.present) return false;
            const int t = decode_sym(br, hd);
            if (t < 0 || t > 15) return false;
            c.pred += t ? extend(br.get(t), t) : 0;
            if (c.pred < -32767 || c.pred > 32767) return false;      // (a valid 8-bit stream stays within 11 bits)
            if (blk) blk[0] = (int16_t)(c.pred * (1 << Al));
        } else if (br.get(1)) {
            if (blk) blk[0] = (int16_t)(blk[0] | (1 << Al));
        }
        return true;
    }
    bool prog_ac(BitReader& br, Comp& c, int bx, int by, int Ss, int Se, int Ah, int Al) {
        if (bx >= c.bw || by >= c.bh) return false;
        int16_t* blk = &c.coef[((size_t)by * c.bw + bx) * 64];
        const Huff& ha = ac[c.ta];
        if (!ha.present) return false;
        if (Ah == 0) {
            if (eobrun > 0) { --eobrun; return true; }
            for (int k = Ss; k <= Se; ++k) {
                const int rs = decode_sym(br, ha);
                if (rs < 0) return false;
                const int r = rs >> 4, s2 = rs & 15;
                if (s2) {
                    k += r;
                    if (k > 63) return false;
                    blk[kZigzag[k]] = (int16_t)(extend(br.get(s2), s2) * (1 << Al));
                } else {
                    if (r != 15) { eobrun = (1 << r) - 1; if (r) eobrun += br.get(r); break; }
                    k += 15;
                }
            }
            return true;
        }
        const int p1 = 1 << Al, m1 = -(1 << Al);
        int k = Ss;
        if (eobrun == 0) {
            for (; k <= Se; ++k) {
                const int rs = decode_sym(br, ha);
                if (rs < 0) return false;
                int r = rs >> 4, s2 = rs & 15;
                if (s2) {
                    s2 = br.get(1) ? p1 : m1;               // (the size of a newly non-zero coefficient is always 1)
                } else if (r != 15) {
                    eobrun = 1 << r;
                    if (r) eobrun += br.get(r);
                    break;                                  // EOBr: the rest of this block only refines
                }
                do {                                        // skip r still-zero coefficients, refining the non-zero ones passed
                    int16_t& cf = blk[kZigzag[k]];
                    if (cf != 0) {
                        if (br.get(1) && (cf & p1) == 0) cf = (int16_t)(cf + (cf >= 0 ? p1 : m1));
                    } else if (--r < 0) break;
                    ++k;
                } while (k <= Se);
                if (s2 && k <= 63) blk[kZigzag[k]] = (int16_t)s2;
            }
        }
        if (eobrun > 0) {
            for (; k <= Se; ++k) {
                int16_t& cf = blk[kZigzag[k]];
                if (cf != 0 && br.get(1) && (cf & p1) == 0) cf = (int16_t)(cf + (cf >= 0 ? p1 : m1));
            }
            --eobrun;
        }
        return true;
    }
    void prog_finish() {
        int32_t tmp[64];
        for (int i = 0; i < ncomp; ++i) {
            Comp& c = comp[i];
            for (int by = 0; by < c.bh; ++by)
                for (int bx = 0; bx < c.bw; ++bx) {
                    const int16_t* blk = &c.coef[((size_t)by * c.bw + bx) * 64];
                    for (int k = 0; k < 64; ++k) tmp[k] = (int32_t)blk[k] * (int32_t)qt[c.tq][k];
                    idct_islow(tmp, c.plane.data() + (size_t)by * 8 * c.pw + (size_t)bx * 8, c.pw);
                }
        }
    }

    bool block(BitReader& br, Comp& c, int bx, int by) {
        const Huff& hd = dc[c.td];
        const Huff& ha = ac[c.ta];
        if (!hd.present || !ha.present || !qt_ok[c.tq]) return false;
        int32_t coef[64];
        std::memset(coef, 0, sizeof(coef));
        const int t = decode_sym(br, hd);
        if (t < 0 || t > 15) return false;
        const int diff = t ? extend(br.get(t), t) : 0;
        c.pred += diff;
        if (c.pred < -32767 || c.pred > 32767) return false;          // (keeps pred * qt inside int32 on crafted streams)
        coef[0] = c.pred * (int32_t)qt[c.tq][0];
        for (int k = 1; k < 64;) {
            const int rs = decode_sym(br, ha);
            if (rs < 0) return false;
            const int r = rs >> 4, s = rs & 15;
            if (!s) { if (r == 15) { k += 16; continue; } break; }
            k += r;
            if (k > 63) return false;
            const int z = kZigzag[k];
            coef[z] = extend(br.get(s), s) * (int32_t)qt[c.tq][z];
            ++k;
        }
        if (bx < c.bw && by < c.bh) idct_islow(coef, c.plane.data() + (size_t)by * 8 * c.pw + (size_t)bx * 8, c.pw);
        return true;
    }

    sd_status scan(const uint8_t* s, size_t sn, size_t data_off, size_t& next) {
        if (sn < 1) return SD_ERR_INVALID;
        const int ns = s[0];
        if (ns < 1 || ns > ncomp || sn < 1 + 2 * (size_t)ns + 3) return SD_ERR_INVALID;
        Comp* sc[3];
        for (int i = 0; i < ns; ++i) {
            Comp* c = nullptr;
            for (int j = 0; j < ncomp; ++j) if (comp[j].id == s[1 + 2 * i]) c = &comp[j];
            if (!c) return SD_ERR_INVALID;
            c->td = s[2 + 2 * i] >> 4; c->ta = s[2 + 2 * i] & 15;
            if (c->td > 3 || c->ta > 3) return SD_ERR_INVALID;
            sc[i] = c;
        }
        alloc_planes();
        const int Ss = s[1 + 2 * ns], Se = s[2 + 2 * ns], Ah = s[3 + 2 * ns] >> 4, Al = s[3 + 2 * ns] & 15;
        if (progressive && (Ss > Se || Se > 63 || (Ss == 0 && Se != 0) || (Ss > 0 && ns != 1) || Al > 13)) return SD_ERR_INVALID;
        eobrun = 0;
        BitReader br{f + data_off, f + len};
        for (int i = 0; i < ns; ++i) sc[i]->pred = 0;
        int mcus_x, mcus_y;
        if (ns == 1) { mcus_x = (sc[0]->dw + 7) / 8; mcus_y = (sc[0]->dh + 7) / 8; }       // non-interleaved: MCU = one block of the real extent
        else { mcus_x = (W + 8 * hmax - 1) / (8 * hmax); mcus_y = (H + 8 * vmax - 1) / (8 * vmax); }
        int rst_left = restart, rst_next = 0;
        for (int my = 0; my < mcus_y; ++my)
            for (int mx = 0; mx < mcus_x; ++mx) {
                if (restart && rst_left == 0) {
                    // byte-align, expect RSTn
                    br.reset();
                    const uint8_t* q = br.p;
                    while (q + 1 < br.end && !(q[0] == 0xFF && q[1] >= 0xD0 && q[1] <= 0xD7)) ++q;
                    if (q + 1 >= br.end || (q[1] & 7) != rst_next) return SD_ERR_INVALID;
                    br.p = q + 2;
                    rst_next = (rst_next + 1) & 7;
                    rst_left = restart;
                    eobrun = 0;
                    for (int i = 0; i < ns; ++i) sc[i]->pred = 0;
                }
                auto one = [&](Comp& c, int bx, int by) {
                    if (!progressive) return block(br, c, bx, by);
                    return Ss == 0 ? prog_dc(br, c, bx, by, Ah, Al) : prog_ac(br, c, bx, by, Ss, Se, Ah, Al);
                };
                if (ns == 1) {
                    if (!one(*sc[0], mx, my)) return SD_ERR_INVALID;
                } else {
                    for (int i = 0; i < ns; ++i)
                        for (int v = 0; v < sc[i]->v; ++v)
                            for (int h = 0; h < sc[i]->h; ++h)
                                if (!one(*sc[i], mx * sc[i]->h + h, my * sc[i]->v + v)) return SD_ERR_INVALID;
                }
                if (restart) --rst_left;
            }
        // the next marker
        const uint8_t* q = br.p;
        while (q + 1 < br.end && !(q[0] == 0xFF && q[1] != 0x00 && !(q[1] >= 0xD0 && q[1] <= 0xD7) && q[1] != 0xFF)) ++q;
        next = (size_t)(q - f);
        return SD_OK;
    }

    // libjpeg's fancy upsampling of one chroma plane to luma resolution (jdsample.c)
    void upsample(const Comp& c, std::vector<uint8_t>& full) const {
        full.assign((size_t)W * H, 0);
        const int cw = c.dw, chh = c.dh;
        if (hmax == 1 && vmax == 1) {
            for (int y = 0; y < H; ++y) std::memcpy(&full[(size_t)y * W], &c.plane[(size_t)y * c.pw], (size_t)W);
            return;
        }
        if (hmax == 2 && vmax == 1) {           // h2v1_fancy_upsample
            for (int y = 0; y < H; ++y) {
                const uint8_t* in = &c.plane[(size_t)y * c.pw];
                uint8_t* o = &full[(size_t)y * W];
                auto put = [&](int x, int v) { if (x < W) o[x] = (uint8_t)v; };
                if (cw == 1) { put(0, in[0]); put(1, in[0]); continue; }
                put(0, in[0]);
                put(1, (in[0] * 3 + in[1] + 2) >> 2);
                for (int i = 1; i < cw - 1; ++i) {
                    const int iv = in[i] * 3;
                    put(2 * i, (iv + in[i - 1] + 1) >> 2);
                    put(2 * i + 1, (iv + in[i + 1] + 2) >> 2);
                }
                put(2 * (cw - 1), (in[cw - 1] * 3 + in[cw - 2] + 1) >> 2);
                put(2 * (cw - 1) + 1, in[cw - 1]);
            }
            return;
        }
        // h2v2_fancy_upsample: output rows 2y (nearer to chroma row y-1) and 2y+1 (nearer to row y+1); the rows beyond the real
        // chroma extent are the replicated edge rows (jdmainct.c context rows)
        std::vector<int> sum((size_t)cw);
        for (int oy = 0; oy < H; ++oy) {
            const int y = oy >> 1;
            int yn = (oy & 1) ? y + 1 : y - 1;
            yn = yn < 0 ? 0 : (yn > chh - 1 ? chh - 1 : yn);
            const uint8_t* in0 = &c.plane[(size_t)y * c.pw];
            const uint8_t* in1 = &c.plane[(size_t)yn * c.pw];
            for (int i = 0; i < cw; ++i) sum[i] = in0[i] * 3 + in1[i];
            uint8_t* o = &full[(size_t)oy * W];
            auto put = [&](int x, int v) { if (x < W) o[x] = (uint8_t)v; };
            if (cw == 1) { put(0, (sum[0] * 4 + 8) >> 4); put(1, (sum[0] * 4 + 7) >> 4); continue; }
            put(0, (sum[0] * 4 + 8) >> 4);
            put(1, (sum[0] * 3 + sum[1] + 7) >> 4);
            for (int i = 1; i < cw - 1; ++i) {
                put(2 * i, (sum[i] * 3 + sum[i - 1] + 8) >> 4);
                put(2 * i + 1, (sum[i] * 3 + sum[i + 1] + 7) >> 4);
            }
            put(2 * (cw - 1), (sum[cw - 1] * 3 + sum[cw - 2] + 8) >> 4);
            put(2 * (cw - 1) + 1, (sum[cw - 1] * 4 + 7) >> 4);
        }
    }

    sd_status decode(uint8_t* out, size_t cap) {
        cap_bytes = cap;
        const sd_status st = parse(true);
        if (st != SD_OK) return st;
        for (int i = 0; i < ncomp; ++i) if (comp[i].plane.empty()) return SD_ERR_INVALID;
        if (cap < (size_t)W * H * 3) return SD_ERR_INVALID;
        if (progressive) {
            for (int i = 0; i < ncomp; ++i) if (!qt_ok[comp[i].tq] || comp[i].coef.empty()) return SD_ERR_INVALID;
            prog_finish();
        }
        std::vector<uint8_t> bgr;
        const bool direct = orientation == 1;
        uint8_t* dst = out;
        if (!direct) { bgr.resize((size_t)W * H * 3); dst = bgr.data(); }
        if (ncomp == 1) {
            for (int y = 0; y < H; ++y) {
                const uint8_t* in = &comp[0].plane[(size_t)y * comp[0].pw];
                uint8_t* o = dst + (size_t)y * W * 3;
                for (int x = 0; x < W; ++x) o[3 * x] = o[3 * x + 1] = o[3 * x + 2] = in[x];
            }
        } else {
            std::vector<uint8_t> cb, cr;
            upsample(comp[1], cb);
            upsample(comp[2], cr);
            // jdcolor.c build_ycc_rgb_table: SCALEBITS 16
            int crr[256], cbb[256];
            int32_t crg[256], cbg[256];
            for (int i = 0; i < 256; ++i) {
                const int x = i - 128;
                crr[i] = (int)((91881 * (int64_t)x + 32768) >> 16);       // FIX(1.40200)
                cbb[i] = (int)((116130 * (int64_t)x + 32768) >> 16);      // FIX(1.77200)
                crg[i] = -46802 * x;                                      // FIX(0.71414)
                cbg[i] = -22554 * x + 32768;                              // FIX(0.34414) + ONE_HALF
            }
            auto cl = [](int v) { return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); };
            const bool rgb_direct = adobe_transform == 0;                 // Adobe marker: the three components ARE R, G, B
            for (int y = 0; y < H; ++y) {
                const uint8_t* yy = &comp[0].plane[(size_t)y * comp[0].pw];
                const uint8_t* b = &cb[(size_t)y * W];
                const uint8_t* r = &cr[(size_t)y * W];
                uint8_t* o = dst + (size_t)y * W * 3;
                for (int x = 0; x < W; ++x) {
                    if (rgb_direct) { o[3 * x] = r[x]; o[3 * x + 1] = b[x]; o[3 * x + 2] = yy[x]; continue; }
                    const int Y = yy[x];
                    o[3 * x + 2] = cl(Y + crr[r[x]]);
                    o[3 * x + 1] = cl(Y + (int)((cbg[b[x]] + crg[r[x]]) >> 16));
                    o[3 * x] = cl(Y + cbb[b[x]]);
                }
            }
        }
        if (direct) return SD_OK;
        // EXIF orientation, as OpenCV's ExifTransform: 2 flip horizontally, 3 rotate 180, 4 flip vertically, 5 transpose,
        // 6 rotate 90 clockwise, 7 transverse, 8 rotate 90 counter-clockwise
        const int OW = orientation >= 5 ? H : W, OH = orientation >= 5 ? W : H;
        for (int oy = 0; oy < OH; ++oy)
            for (int ox = 0; ox < OW; ++ox) {
                int sx, sy;
                switch (orientation) {
                    case 2: sx = W - 1 - ox; sy = oy; break;
                    case 3: sx = W - 1 - ox; sy = H - 1 - oy; break;
                    case 4: sx = ox; sy = H - 1 - oy; break;
                    case 5: sx = oy; sy = ox; break;
                    case 6: sx = oy; sy = H - 1 - ox; break;
                    case 7: sx = W - 1 - oy; sy = H - 1 - ox; break;
                    default: sx = W - 1 - oy; sy = ox; break;      // 8
                }
                std::memcpy(out + ((size_t)oy * OW + ox) * 3, dst + ((size_t)sy * W + sx) * 3, 3);
            }
        return SD_OK;
    }
};

}  // namespace

extern "C" sd_status sd_jpeg_decode_bgr(const uint8_t* file_host, size_t len, uint8_t* bgr_out_host, size_t out_capacity, int* height_out,
                                        int* width_out) {
    if (!file_host) return SD_ERR_INVALID;
    try {                                                  // (no exception crosses the C ABI: an allocation failure is a refused file)
        Decoder d;
        d.f = file_host; d.len = len;
        if (!bgr_out_host) return d.header_only(height_out, width_out);
        const sd_status st = d.decode(bgr_out_host, out_capacity);
        if (st == SD_OK) d.dims(height_out, width_out);
        return st;
    } catch (...) {
        return SD_ERR_INVALID;
    }
}
