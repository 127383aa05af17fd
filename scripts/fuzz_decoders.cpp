// dev tool: fuzz harness of the host image readers (CPU build under ASan + UBSan; GPU sanitizers are not available on this pool).
//   g++ -O1 -g -fsanitize=address,undefined -std=c++17 -Iinclude scripts/fuzz_decoders.cpp semantic_depth_amd/csrc/host_jpeg.cpp semantic_depth_amd/csrc/host_png.cpp -lz -lpthread -o /tmp/fuzz_decoders
//   /tmp/fuzz_decoders 3000 seed1.jpg seed2.png ...     (seeds: any small JPEG / PNG files; round 3: 14 seeds x 3000 mutations, no finding)
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
            int kind = rng() % 6;
            if (kind == 0) f.resize(rng() % (f.size() + 1));                             // truncation
            else if (kind == 1) { int n = 1 + rng() % 8; for (int i = 0; i < n; ++i) f[rng() % f.size()] = (uint8_t)rng(); }
            else if (kind == 2) { int n = 1 + rng() % 64; for (int i = 0; i < n; ++i) f[rng() % f.size()] ^= (uint8_t)(1u << (rng() % 8)); }
            else if (kind == 3) { size_t p = rng() % f.size(), n = rng() % 64; for (size_t i = p; i < p + n && i < f.size(); ++i) f[i] = 0xff; }
            else if (kind == 4) { size_t hdr = f.size() < 700 ? f.size() : 700; int n = 1 + rng() % 6; for (int i = 0; i < n; ++i) f[rng() % hdr] = (uint8_t)rng(); }   // header region
            else { size_t p = rng() % f.size(); f.insert(f.begin() + p, (size_t)(rng() % 32), (uint8_t)rng()); }
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
