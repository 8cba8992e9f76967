// Fuzz harness for the PExpr compiler and the expression interpreter (ignis_amd/csrc/host/pexpr.h, include/ig_expr.h):
//   g++ -std=c++17 -O1 -g -fsanitize=address,undefined -ffp-contract=off -mfma -Iinclude -Iignis_amd/csrc/host tools/fuzz_pexpr.cpp -o /tmp/fuzz_pexpr
//   /tmp/fuzz_pexpr 200000
// Random strings over the language's alphabet and mutations of valid expressions must either compile to a program that
// ige_validate accepts and ige_run finishes, or be refused with an exception; random words must never pass ige_validate
// and then misbehave in ige_run.
#include "pexpr.h"

#include <cstdio>
#include <random>

using namespace igh::pexpr;

struct Ctx {
    ige_v4 var(int id) const { return ige_v4{ { 0.3f * id, 0.7f, -0.2f, 0.5f } }; }
    ige_v4 tex(uint32_t id, float u, float v) const { return ige_v4{ { u, v, (float)id, 1 } }; }
    ige_v4 evr(ige_v4, ige_v4, ige_v4 n) const { return n; }
};

int main(int argc, char** argv)
{
    const long n = argc > 1 ? std::atol(argv[1]) : 100000;
    std::mt19937 rng(12345);
    const char* seeds[] = {
        "mix(_tex_0((uvw).xy), max(_tex_0((uvw).xy), color(0.0, 0.0022, 0.5, 1.0)), clamp(1.0,0,1))",
        "select(checkerboard(uvw * 10.0) == 1, color(0.8,0.8,0.8, 1.0), color(0.2))",
        "norm(Nx*((2*_tex_1((uvw).xy)-color(1)).xyz).x + Ny*((2*_tex_1((uvw).xy)-color(1)).xyz).y + N*((2*_tex_1((uvw).xy)-color(1)).xyz).z)",
        "ensure_valid_reflection(Ng, V, bump(N, Nx, Ny, 1.0, (luminance(_tex_0((vec3(0.001,0,0)+uvw).xy)) - luminance(_tex_0((uvw).xy)))/0.001, 0.5))",
        "_tex_0.r * sin(uv.x * 10 * Pi) ^ 2 + 7 % 3 - -4 / 2.5e-1", "dot(N, V) > 0.5 && frontside || !(1 <= 2)", "vec3(1,2,3).zyxx == vec4(3,2,1,1)",
    };
    const char alphabet[] = "0123456789.+-*/%^()<>=!&|, abcdeNVPuvwxyz_\"'";
    const char* words[] = { "uv", "uvw", "N", "Nx", "V", "P", "sin", "mix", "color", "vec3", "select", "checkerboard", "_tex_0", "bump", "norm", "dot", "Pi", "frontside", "clamp", "luminance" };
    Env env;
    env.texture = [](const std::string& s) { return s == "_tex_0" ? 0 : (s == "_tex_1" ? 1 : -1); };
    long ok = 0, refused = 0;
    for (long i = 0; i < n; ++i) {
        std::string s;
        if (rng() % 2) {
            s = seeds[rng() % (sizeof(seeds) / sizeof(*seeds))];
            const int edits = 1 + rng() % 4;
            for (int e = 0; e < edits && !s.empty(); ++e) {
                const size_t at = rng() % s.size();
                switch (rng() % 4) {
                case 0: s.erase(at, 1 + rng() % 3); break;
                case 1: s.insert(at, 1, alphabet[rng() % (sizeof(alphabet) - 1)]); break;
                case 2: s.insert(at, words[rng() % (sizeof(words) / sizeof(*words))]); break;
                default: s[at] = alphabet[rng() % (sizeof(alphabet) - 1)]; break;
                }
            }
        } else {
            const int len = rng() % 40;
            for (int k = 0; k < len; ++k)
                if (rng() % 5 == 0)
                    s += words[rng() % (sizeof(words) / sizeof(*words))];
                else
                    s += alphabet[rng() % (sizeof(alphabet) - 1)];
        }
        try {
            Program p = compile(s, env);
            if (!ige_validate(p.code.data(), (uint32_t)p.code.size(), 0, 2)) {
                std::printf("compiled program fails validation: %s\n", s.c_str());
                return 1;
            }
            (void)ige_run(p.code.data(), Ctx{});
            ++ok;
        } catch (const std::runtime_error&) {
            ++refused;
        }
    }
    // random word streams: whatever validates must run to its END inside the table
    long valid = 0;
    for (long i = 0; i < n; ++i) {
        std::vector<uint32_t> code(1 + rng() % 24);
        for (auto& w : code) {
            w = rng();
            if (rng() % 2)
                w = (w & 0xFFFFFF00u) | (rng() % (IGE_OP_COUNT + 2));
            if (rng() % 2)
                w &= 0xFF7777FFu; // registers mostly in range
        }
        if (rng() % 2)
            code.back() = IGE_INS(IGE_END, 0, 0, 0, 0, 0);
        if (ige_validate(code.data(), (uint32_t)code.size(), 0, 2)) {
            (void)ige_run(code.data(), Ctx{});
            ++valid;
        }
    }
    std::printf("%ld compiled and ran, %ld refused, %ld random programs validated and ran\n", ok, refused, valid);
    return 0;
}
