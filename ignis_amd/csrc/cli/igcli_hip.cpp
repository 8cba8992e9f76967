// igcli_hip — C++ counterpart of the reference's command-line renderer (src/frontend/cli/main.cpp:60-185) on top of
// the two C ABIs: load a scene (igh_host.h), render spp samples per pixel in iterations of spi (igd_device.h), write
// the mean image as EXR and print the reference's statistics lines.
//
//   igcli_hip scene.json [-o out.exr] [--spp N] [--spi N] [--width W] [--height H] [--seed S] [--gpu I] [--stats]
#include "igd_device.h"
#include "igh_host.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace {
double now_ms()
{
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
// recommendSPI for a GPU target (src/runtime/Runtime.cpp:71-79)
int recommend_spi(int width, int height)
{
    const int spi = (int)std::ceil(8.0 / ((width / 1000.0) * (height / 1000.0)));
    return std::max(1, std::min(64, spi));
}
} // namespace

int main(int argc, char** argv)
{
    std::string scene_path, output = "output.exr";
    int spp = 64, spi = 0, width = 0, height = 0, seed = 0, gpu = 0;
    bool stats = false;
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        auto next = [&]() -> const char* { return i + 1 < argc ? argv[++i] : "0"; };
        if (a == "-o" || a == "--output")
            output = next();
        else if (a == "--spp")
            spp = std::atoi(next());
        else if (a == "--spi")
            spi = std::atoi(next());
        else if (a == "--width")
            width = std::atoi(next());
        else if (a == "--height")
            height = std::atoi(next());
        else if (a == "--seed")
            seed = std::atoi(next());
        else if (a == "--gpu")
            gpu = std::atoi(next());
        else if (a == "--stats")
            stats = true;
        else if (a == "-h" || a == "--help") {
            std::printf("usage: %s scene.json [-o out.exr] [--spp N] [--spi N] [--width W] [--height H] [--seed S] [--gpu I] [--stats]\n", argv[0]);
            return 0;
        } else
            scene_path = a;
    }
    if (scene_path.empty()) {
        std::fprintf(stderr, "No scene file given\n");
        return 1;
    }

    const double t_all = now_ms();
    igh_options opts{};
    opts.film_width  = width;
    opts.film_height = height;
    igh_scene* scene = igh_load_file(scene_path.c_str(), &opts);
    if (!scene) {
        std::fprintf(stderr, "Failed loading: %s\n", igh_last_error());
        return 1;
    }
    const igd_scene* tables = igh_tables(scene);
    width  = tables->film_width;
    height = tables->film_height;
    if (spi <= 0)
        spi = recommend_spi(width, height);

    igd_setup setup{};
    setup.gpu_index     = gpu;
    setup.acquire_stats = stats ? 1 : 0;
    igd_device* dev     = igd_create(&setup);
    if (!dev) {
        std::fprintf(stderr, "%s\n", igd_last_error());
        igh_free(scene);
        return 1;
    }
    int rc = igd_assign_scene(dev, tables);
    if (rc == IGD_OK)
        rc = igd_resize(dev, width, height);
    const double t_loading = now_ms() - t_all;

    const int desired_iter = std::max(1, (spp + spi - 1) / spi);
    if (spp % spi != 0)
        std::fprintf(stderr, "Given spp %d is not a multiple of the spi %d. Using spp %d instead\n", spp, spi, desired_iter * spi);
    std::fprintf(stderr, "Started rendering...\n");
    std::vector<double> samples_sec;
    double t_render = 0;
    // several iterations per call when one does not fill the GPU (igd_render_settings.iterations, bit-identical results)
    const int batch = std::max(1, std::min(64, (1 << 24) / std::max(1, width * height * spi)));
    for (int it = 0; rc == IGD_OK && it < desired_iter;) {
        const int count = std::min(batch, desired_iter - it);
        const double t0 = now_ms();
        igd_render_settings rs{};
        rs.spi = spi, rs.width = width, rs.height = height, rs.iteration = it, rs.user_seed = seed, rs.row_stride = 1, rs.iterations = count;
        rc = igd_render(dev, &rs);
        if (rc == IGD_OK)
            rc = igd_synchronize(dev); // per-call timing like the reference's blocking step()
        const double dt = now_ms() - t0;
        t_render += dt;
        for (int k = 0; k < count; ++k)
            samples_sec.push_back(1000.0 * double(spi) * width * height * count / dt);
        it += count;
    }
    if (rc != IGD_OK) {
        std::fprintf(stderr, "%s\n", igd_last_error());
        igd_destroy(dev);
        igh_free(scene);
        return 1;
    }

    const double t0    = now_ms();
    const float* fb    = igd_framebuffer_host(dev, nullptr, 1);
    const std::string s_spp = std::to_string(desired_iter * spi), s_spi = std::to_string(spi), s_seed = std::to_string(seed);
    const char* meta[] = { "igTechniqueType", "path", "igCameraType", "perspective", "igSPP", s_spp.c_str(), "igSPI", s_spi.c_str(),
                           "igSeed", s_seed.c_str(), "igTargetString", "MI355X (gfx950, HIP)", nullptr };
    const bool saved = fb && igh_save_exr(output.c_str(), fb, width, height, 1.0f / desired_iter, meta) == 0;
    const double t_saving = now_ms() - t0;
    std::fprintf(stderr, saved ? "Result saved to %s\n" : "Failed to save EXR file %s\n", output.c_str());

    if (stats) {
        igd_stats st{};
        igd_get_stats(dev, &st);
        const unsigned long long total = st.camera_rays + st.bounce_rays + st.shadow_rays;
        std::printf("Statistics:\n  Ray Count: %llu\n    Camera: %llu\n    Bounce: %llu\n    Shadow: %llu\n  Mrays/s (render time): %.1f\n",
                    total, (unsigned long long)st.camera_rays, (unsigned long long)st.bounce_rays, (unsigned long long)st.shadow_rays,
                    total / t_render / 1e3);
    }
    std::printf("  Iterations: %d\n  SPP: %d\n  SPI: %d\n  Time: %.3fs\n    Loading> %.3fs\n    Render>  %.3fs\n    Saving>  %.3fs\n",
                desired_iter, desired_iter * spi, spi, (now_ms() - t_all) / 1e3, t_loading / 1e3, t_render / 1e3, t_saving / 1e3);
    igd_destroy(dev);
    igh_free(scene);
    std::sort(samples_sec.begin(), samples_sec.end());
    std::printf("# %.3f/%.3f/%.3f (min/med/max Msamples/s)\n", samples_sec.front() * 1e-6, samples_sec[samples_sec.size() / 2] * 1e-6, samples_sec.back() * 1e-6);
    return saved ? 0 : 1;
}
