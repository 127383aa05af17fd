// dev tool: fuzz harness of the host image readers (CPU build under ASan + UBSan; GPU sanitizers are not available on this pool).
//   g++ -O1 -g -fsanitize=address,undefined -std=c++17 -Iinclude scripts/fuzz_decoders.cpp semantic_depth_amd/csrc/host_jpeg.cpp semantic_depth_amd/csrc/host_png.cpp -lz -lpthread -o /tmp/fuzz_decoders
//   /tmp/fuzz_decoders 3000 seed1.jpg seed2.png ...     (seeds: any small JPEG / PNG files; round 4: 13 seeds incl. the crafted PoCs x 1500 mutations, no finding)
// Round 3's byte-level mutations missed two structural holes (an over-subscribed DHT, a second SOF): kinds 6-9 below rewrite JPEG segments
// (DHT count bytes, duplicated / inserted SOF / SOS / DRI segments, SOF0 <-> SOF2 swaps); tests/test_frame_io.py::_crafted_jpegs holds the PoCs.
// Mutated JPEG / PNG files go through the C-ABI decoders from exact-size heap copies; nothing may crash or read out of bounds.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <random>
#include "semdepth.h"
static std::vector<uint8_t> readf(const char* p) { FILE* f = fopen(p, "rb"); std::vector<uint8_t> v; if (!f) return v; fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET); v.resize(n); if (fread(v.data(), 1, n, f) != (size_t)n) v.clear(); fclose(f); return v; }
int main(int argc, char** argv) {
    int iters = atoi(argv[1]);
    std::mt19937_64 rng(987654321);
    long ok = 0, bad = 0;
    std::vector<uint8_t> out(64u << 20);
    for (int a = 2; a < argc; ++a) {
        std::vector<uint8_t> base = readf(argv[a]);
        if (base.empty()) { printf("cannot read %s\n", argv[a]); return 2; }
        for (int it = 0; it < iters; ++it) {
            std::vector<uint8_t> f = base;
            int kind = rng() % 10;
            const bool jpg = f.size() > 4 && f[0] == 0xFF && f[1] == 0xD8;
            if (kind >= 6 && !jpg) kind = rng() % 6;
            if (kind == 0) f.resize(rng() % (f.size() + 1));                             // truncation
            else if (kind == 1) { int n = 1 + rng() % 8; for (int i = 0; i < n; ++i) f[rng() % f.size()] = (uint8_t)rng(); }
            else if (kind == 2) { int n = 1 + rng() % 64; for (int i = 0; i < n; ++i) f[rng() % f.size()] ^= (uint8_t)(1u << (rng() % 8)); }
            else if (kind == 3) { size_t p = rng() % f.size(), n = rng() % 64; for (size_t i = p; i < p + n && i < f.size(); ++i) f[i] = 0xff; }
            else if (kind == 4) { size_t hdr = f.size() < 700 ? f.size() : 700; int n = 1 + rng() % 6; for (int i = 0; i < n; ++i) f[rng() % hdr] = (uint8_t)rng(); }   // header region
            else if (kind == 5) { size_t p = rng() % f.size(); f.insert(f.begin() + p, (size_t)(rng() % 32), (uint8_t)rng()); }
            else {
                // walk the marker segments up to the first SOS
                struct Seg { size_t off, len; int m; };
                std::vector<Seg> segs;
                size_t p = 2;
                while (p + 4 <= f.size() && f[p] == 0xFF) {
                    const int m = f[p + 1];
                    const size_t n = ((size_t)f[p + 2] << 8) | f[p + 3];
                    if (n < 2 || p + 2 + n > f.size()) break;
                    segs.push_back({p, n + 2, m});
                    if (m == 0xDA) break;
                    p += 2 + n;
                }
                if (segs.empty()) continue;
                if (kind == 6) {                                                             // DHT: rewrite code-length counts
                    for (auto& sg : segs) if (sg.m == 0xC4 && sg.len > 21) { int n = 1 + rng() % 3; for (int i = 0; i < n; ++i) f[sg.off + 5 + rng() % 16] = (uint8_t)(rng() % 3 ? rng() % 8 : rng()); }
                } else if (kind == 7) {                                                      // duplicate a segment (SOF / SOS / DHT / DQT / DRI) somewhere later
                    const Seg sg = segs[rng() % segs.size()];
                    std::vector<uint8_t> copy(f.begin() + sg.off, f.begin() + sg.off + sg.len);
                    if ((sg.m & 0xF0) == 0xC0 && sg.m != 0xC4 && copy.size() > 9 && (rng() & 1)) { copy[5] = (uint8_t)rng(); copy[7] = (uint8_t)rng(); copy[6] = copy[8] = 0; copy[5] = 0; }   // other dims
                    const Seg at = segs[rng() % segs.size()];
                    size_t where = (rng() & 1) ? at.off : f.size() - 2;                        // before a segment, or behind the scan data
                    if (where > f.size()) where = f.size();
                    f.insert(f.begin() + where, copy.begin(), copy.end());
                } else if (kind == 8) {                                                      // SOF0 <-> SOF2, or drop the DQT
                    for (auto& sg : segs) {
                        if ((sg.m == 0xC0 || sg.m == 0xC2) && (rng() & 1)) f[sg.off + 1] = sg.m == 0xC0 ? 0xC2 : 0xC0;
                        else if (sg.m == 0xDB && (rng() % 4) == 0) f[sg.off + 1] = 0xEF;       // becomes an ignored APP15
                    }
                } else {                                                                     // a DRI with a random interval in front of the SOS
                    const Seg sg = segs.back();
                    const uint8_t dri[6] = {0xFF, 0xDD, 0, 4, (uint8_t)(rng() % 2), (uint8_t)rng()};
                    f.insert(f.begin() + sg.off, dri, dri + 6);
                }
            }
            int h = 0, w = 0;
            size_t cap = (it & 7) == 7 ? (size_t)(rng() % 4096) : out.size();              // also: too-small output buffers
            // a heap copy of exactly the file's size, so that ASan sees any read past its end
            uint8_t* fc = (uint8_t*)malloc(f.size() ? f.size() : 1);
            memcpy(fc, f.data(), f.size());
            uint8_t* oc = (uint8_t*)malloc(cap ? cap : 1);
            sd_status s = sd_image_decode_bgr(fc, f.size(), oc, cap, &h, &w);
            if (s == SD_OK) ++ok; else ++bad;
            free(oc); free(fc);
        }
    }
    printf("decoded %ld, rejected %ld\n", ok, bad);
    return 0;
}
