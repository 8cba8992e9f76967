// g++ -std=c++17 -O1 -g -fsanitize=address,undefined tools/fuzz_float_images.cpp -o /tmp/fuzz_float_images -lz && /tmp/fuzz_float_images tests/golden/references/ref-cbox-d1-4096.exr ...
// Mutates the given OpenEXR / Radiance files (byte flips, truncation, header and payload corruption) and feeds them to the readers of
// ignis_amd/csrc/host/floatimage.h under the sanitizers: an exception is fine, a crash is not.
#include "../ignis_amd/csrc/host/floatimage.h"
#include <cstdio>
#include <random>
int main(int argc, char** argv)
{
    // mutate copies of the given files and feed them to the readers; any exception is fine, a crash is not
    std::mt19937 rng(12345);
    long ok = 0, thrown = 0;
    for (int f = 1; f < argc; ++f) {
        std::vector<uint8_t> base = igh::fimg::readAll(argv[f]);
        const std::string tmp = std::string("/tmp/fuzz_float_image_cur") + (std::string(argv[f]).find(".hdr") != std::string::npos ? ".hdr" : ".exr");
        for (int it = 0; it < 400; ++it) {
            std::vector<uint8_t> b = base;
            const int kind = it % 4;
            if (kind == 0) { // flip a few bytes
                for (int k = 0; k < 1 + (int)(rng() % 8); ++k)
                    b[rng() % b.size()] = (uint8_t)rng();
            } else if (kind == 1) { // truncate
                b.resize(rng() % b.size());
            } else if (kind == 2) { // corrupt the header region
                for (int k = 0; k < 4; ++k)
                    b[rng() % std::min<size_t>(b.size(), 400)] = (uint8_t)rng();
            } else { // corrupt after the header
                for (int k = 0; k < 16; ++k)
                    b[std::min<size_t>(b.size() - 1, 300 + rng() % (b.size() - 300 > 0 ? b.size() - 300 : 1))] ^= (uint8_t)(1u << (rng() % 8));
            }
            FILE* o = std::fopen(tmp.c_str(), "wb");
            std::fwrite(b.data(), 1, b.size(), o);
            std::fclose(o);
            try {
                igh::readFloatImage(tmp);
                ++ok;
            } catch (const std::exception&) {
                ++thrown;
            }
        }
    }
    std::printf("ok %ld thrown %ld\n", ok, thrown);
    return 0;
}
