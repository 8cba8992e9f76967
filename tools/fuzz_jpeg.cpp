// g++ -std=c++17 -O1 -g -fsanitize=address,undefined tools/fuzz_jpeg.cpp -o /tmp/fuzz_jpeg && /tmp/fuzz_jpeg a.jpg b.jpg ...
// Mutated JPEG files (byte flips, truncation, damaged headers) against ignis_amd/csrc/host/jpeg.h under the sanitizers (round 2: 3 000 mutations of
// baseline / progressive / gray files, no finding).
#include "../ignis_amd/csrc/host/jpeg.h"
#include <random>
#include <fstream>
static std::vector<uint8_t> readAllBytes(const char* p) { std::ifstream f(p, std::ios::binary); return std::vector<uint8_t>((std::istreambuf_iterator<char>(f)), {}); }
int main(int argc, char** argv)
{
    std::mt19937 rng(99);
    long ok = 0, thrown = 0;
    for (int f = 1; f < argc; ++f) {
        std::vector<uint8_t> base = readAllBytes(argv[f]);
        for (int it = 0; it < 600; ++it) {
            std::vector<uint8_t> b = base;
            const int kind = it % 3;
            if (kind == 0) for (int k = 0; k < 1 + (int)(rng() % 6); ++k) b[rng() % b.size()] = (uint8_t)rng();
            else if (kind == 1) b.resize(2 + rng() % (b.size() - 2));
            else for (int k = 0; k < 6; ++k) b[rng() % std::min<size_t>(b.size(), 700)] = (uint8_t)rng();
            try { igh::jpg::Decoder d("mem", b); d.run(); ++ok; } catch (const std::exception&) { ++thrown; }
        }
    }
    std::printf("ok %ld thrown %ld\n", ok, thrown);
}
