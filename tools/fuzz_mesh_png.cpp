// g++ -std=c++17 -O1 -g -fsanitize=address,undefined -Iinclude tools/fuzz_mesh_png.cpp ignis_amd/csrc/host/mesh.cpp -o /tmp/fuzz_mesh_png -lz && /tmp/fuzz_mesh_png scenes/textures/bumpmap.png scenes/meshes/Diamond.ply scenes/meshes/Room.obj
// Mutated PNG / PLY / OBJ / Mitsuba-serialized files against the loader's readers under the sanitizers (round 2: 1 500 mutations, no finding).
#include "../ignis_amd/csrc/host/png.h"
#include "../ignis_amd/csrc/host/mesh.h"
#include <cstdio>
#include <random>
#include <fstream>
using namespace igh;
static std::vector<uint8_t> readAllBytes(const char* p) { std::ifstream f(p, std::ios::binary); return std::vector<uint8_t>((std::istreambuf_iterator<char>(f)), {}); }
int main(int argc, char** argv)
{
    std::mt19937 rng(777);
    long ok = 0, thrown = 0;
    for (int f = 1; f < argc; ++f) {
        const std::string name = argv[f];
        const std::string ext  = name.substr(name.rfind('.'));
        std::vector<uint8_t> base = readAllBytes(argv[f]);
        const std::string tmp = "/tmp/fuzz_mesh_png_cur" + ext;
        for (int it = 0; it < 300; ++it) {
            std::vector<uint8_t> b = base;
            const int kind = it % 3;
            if (kind == 0) for (int k = 0; k < 1 + (int)(rng() % 6); ++k) b[rng() % b.size()] = (uint8_t)rng();
            else if (kind == 1) b.resize(rng() % b.size());
            else for (int k = 0; k < 6; ++k) b[rng() % std::min<size_t>(b.size(), 300)] = (uint8_t)rng();
            FILE* o = std::fopen(tmp.c_str(), "wb"); std::fwrite(b.data(), 1, b.size(), o); std::fclose(o);
            try {
                if (ext == ".png") readPng(tmp);
                else if (ext == ".ply") load_ply(tmp);
                else if (ext == ".obj") load_obj(tmp);
                else load_serialized(tmp, 0);
                ++ok;
            } catch (const std::exception&) { ++thrown; }
        }
    }
    std::printf("ok %ld thrown %ld\n", ok, thrown);
}
