// oracle.cpp — TEST INFRASTRUCTURE ONLY (see oracle/README.md).
//
// CPU restatement of the reference CPU device's wavefront loop `cpu_trace`
// (src/artic/driver/mapping_cpu.art:719-861): 16x16 image tiles, one worker
// thread per core, per-thread SoA streams of spi*256 rays, in-place counting
// sort by entity, per-entity hit shading, stable compaction, any-hit shadow
// traversal and a serial per-tile splat. Exposed through a plain C ABI so the
// tests, __graft_entry__.smoke() and bench.py's cpu_baseline leg can call it
// with ctypes. The product (ignis_amd/) never links or loads this library.
#include "oracle_shade.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <thread>
#include <vector>

// ORACLE-ONLY switch (tests/test_photonmapper.py): store a photon's direction with a SOUND signed-normalised 16-bit encoding — scale
// 32767, the second component masked to its half of the word — instead of encode_signed_norm_16 as written (core/common.art:186-197:
// scale 65535 into an i16, which wraps beyond +-0.5, and a sign-extended second component that overwrites the first). Everything else
// of the photon mapper stays as restated. Round 3 blamed this encoding for the photon mapper's dark green channel on cycles-lights; the
// switch shows it is not: the image does not move (tests/test_photonmapper.py), a diffuse receiver only asks a photon's direction for the
// sign of a cosine. What is dark is the spot light, whose sample_emission carries 1 / spot_area (light/spot.art:41-47), exactly as in the
// light tracer. The product (ppm_core.h) has no such switch: it renders what the reference renders.
static bool g_ppm_sound_directions = false;
static int32_t sound_snorm16(float f)
{
    const float c = f < -1.0f ? -1.0f : (f > 1.0f ? 1.0f : f);
    return (int32_t)igp_round(c * 32767.0f);
}
static int32_t sound_encode_normal_32(float x, float y, float z)
{
    const float a  = igm_abs(x) + igm_abs(y) + igm_abs(z);
    const float ox = x / a, oy = y / a;
    float px = ox, py = oy;
    if (z < 0) {
        px = (1 - igm_abs(oy)) * (ox >= 0 ? 1.0f : -1.0f);
        py = (1 - igm_abs(ox)) * (oy >= 0 ? 1.0f : -1.0f);
    }
    return (int32_t)(((uint32_t)sound_snorm16(px) << 16) | ((uint32_t)sound_snorm16(py) & 0xFFFFu));
}
static void sound_decode_normal_32(int32_t val, float out[3])
{
    // decode_oct_proj as it is meant (the reflected branch with its parentheses): (1 - |y|) * sign(x)
    const float dx = (float)(int16_t)(val >> 16) / 32767.0f, dy = (float)(int16_t)val / 32767.0f;
    const float oz = 1 - igm_abs(dx) - igm_abs(dy);
    float ox = dx, oy = dy;
    if (oz < 0) {
        ox = (1 - igm_abs(dy)) * (dx >= 0 ? 1.0f : -1.0f);
        oy = (1 - igm_abs(dx)) * (dy >= 0 ? 1.0f : -1.0f);
    }
    const float inv = 1 / igm_sqrt(igm_fma(ox, ox, igm_fma(oy, oy, oz * oz)));
    out[0] = ox * inv, out[1] = oy * inv, out[2] = oz * inv;
}


using namespace oracle;

extern "C" {

typedef struct oracle_settings {
    int32_t spi;
    int32_t width, height;
    int32_t iteration, frame, seed;
    int32_t threads; // <= 0: hardware concurrency
    // optional pixel window (tile sharding); xmax/ymax <= 0 means full film
    int32_t xmin, ymin, xmax, ymax;
    // optional row sharding: only film rows row_offset, row_offset + row_stride, ... (stride <= 1: all)
    int32_t row_offset, row_stride;
} oracle_settings;

typedef struct oracle_stats {
    uint64_t camera_rays, bounce_rays, shadow_rays;
    uint64_t nodes, tris, leaves;
    uint64_t unoccluded;
    int32_t max_stack;
    int32_t threads_used;
} oracle_stats;

} // extern "C"

namespace {

struct PrimaryStream {
    std::vector<int32_t> id;
    std::vector<float> org_x, org_y, org_z, dir_x, dir_y, dir_z, tmin, tmax;
    std::vector<uint32_t> flags;
    std::vector<int32_t> ent_id, prim_id;
    std::vector<float> t, u, v;
    std::vector<uint32_t> rnd;
    std::vector<float> payload; // SoA: payload[c * capacity + i] (ShaderUtils.cpp:38)
    int capacity = 0;

    void resize(int cap)
    {
        capacity = cap;
        id.resize(cap);
        for (auto* p : { &org_x, &org_y, &org_z, &dir_x, &dir_y, &dir_z, &tmin, &tmax, &t, &u, &v })
            p->resize(cap);
        flags.resize(cap);
        ent_id.resize(cap);
        prim_id.resize(cap);
        rnd.resize(cap);
        payload.resize((size_t)cap * 7); // 6 floats of the path tracer, 7 of the volumetric one (volpathtracer.art:1-25)
    }
};

struct SecondaryStream {
    std::vector<int32_t> id;
    std::vector<float> org_x, org_y, org_z, dir_x, dir_y, dir_z, tmin, tmax;
    std::vector<uint32_t> flags;
    std::vector<int32_t> mat_id;
    std::vector<float> color_r, color_g, color_b;

    void resize(int cap)
    {
        id.resize(cap);
        for (auto* p : { &org_x, &org_y, &org_z, &dir_x, &dir_y, &dir_z, &tmin, &tmax, &color_r, &color_g, &color_b })
            p->resize(cap);
        flags.resize(cap);
        mat_id.resize(cap);
    }
};

inline Ray read_ray(const PrimaryStream& s, int i)
{
    return make_ray(Vec3{ s.org_x[i], s.org_y[i], s.org_z[i] }, Vec3{ s.dir_x[i], s.dir_y[i], s.dir_z[i] }, s.tmin[i], s.tmax[i], s.flags[i]);
}

inline void write_ray(PrimaryStream& s, int i, const Ray& r)
{
    s.org_x[i] = r.org.x, s.org_y[i] = r.org.y, s.org_z[i] = r.org.z;
    s.dir_x[i] = r.dir.x, s.dir_y[i] = r.dir.y, s.dir_z[i] = r.dir.z;
    s.tmin[i]  = r.tmin;
    s.tmax[i]  = r.tmax;
    s.flags[i] = r.flags;
}

inline PTRayPayload read_payload(const PrimaryStream& s, int i)
{
    const int c = s.capacity;
    PTRayPayload p;
    p.inv_pdf = s.payload[i];
    p.contrib = Color{ s.payload[c + i], s.payload[2 * c + i], s.payload[3 * c + i] };
    p.depth   = (int32_t)s.payload[4 * c + i];
    p.eta     = s.payload[5 * c + i];
    p.medium  = (int32_t)s.payload[6 * c + i];
    return p;
}

inline void write_payload(PrimaryStream& s, int i, const PTRayPayload& p)
{
    const int c          = s.capacity;
    s.payload[i]         = p.inv_pdf;
    s.payload[c + i]     = p.contrib.r;
    s.payload[2 * c + i] = p.contrib.g;
    s.payload[3 * c + i] = p.contrib.b;
    s.payload[4 * c + i] = (float)p.depth;
    s.payload[5 * c + i] = p.eta;
    s.payload[6 * c + i] = (float)p.medium;
}

template <typename T>
inline void swp(std::vector<T>& v, int a, int b) { std::swap(v[a], v[b]); }

// cpu_swap_primary_entry (mapping_cpu.art:23-43)
void swap_primary(PrimaryStream& s, int a, int b)
{
    swp(s.id, a, b);
    swp(s.org_x, a, b), swp(s.org_y, a, b), swp(s.org_z, a, b);
    swp(s.dir_x, a, b), swp(s.dir_y, a, b), swp(s.dir_z, a, b);
    swp(s.tmin, a, b), swp(s.tmax, a, b), swp(s.flags, a, b);
    swp(s.ent_id, a, b), swp(s.prim_id, a, b), swp(s.t, a, b), swp(s.u, a, b), swp(s.v, a, b), swp(s.rnd, a, b);
    for (int c = 0; c < 7; ++c)
        std::swap(s.payload[(size_t)c * s.capacity + a], s.payload[(size_t)c * s.capacity + b]);
}

// cpu_sort_primary (mapping_cpu.art:63-103)
int sort_primary(PrimaryStream& s, int size, std::vector<int>& ray_begins, std::vector<int>& ray_ends, int num_geometries)
{
    for (int i = 0; i <= num_geometries; ++i)
        ray_ends[i] = 0;
    auto key = [&](int i) { return s.ent_id[i] == -1 ? num_geometries : s.ent_id[i]; };
    for (int i = 0; i < size; ++i)
        ray_ends[key(i)]++;
    int n = 0;
    for (int i = 0; i <= num_geometries; ++i) {
        ray_begins[i] = n;
        n += ray_ends[i];
        ray_ends[i] = n;
    }
    for (int i = 0; i < num_geometries; ++i) {
        const int end = ray_ends[i];
        int j         = ray_begins[i];
        while (j < end) {
            const int ent = key(j);
            if (ent != i) {
                const int k = ray_begins[ent]++;
                swap_primary(s, k, j);
            } else {
                ++j;
            }
        }
    }
    return ray_ends[num_geometries - 1];
}

// cpu_compact_primary (mapping_cpu.art:205-253), scalar branch
int compact_primary(PrimaryStream& s, int size)
{
    int k = 0;
    for (int i = 0; i < size; ++i) {
        if (s.id[i] >= 0) {
            s.id[k]    = s.id[i];
            s.org_x[k] = s.org_x[i], s.org_y[k] = s.org_y[i], s.org_z[k] = s.org_z[i];
            s.dir_x[k] = s.dir_x[i], s.dir_y[k] = s.dir_y[i], s.dir_z[k] = s.dir_z[i];
            s.tmin[k] = s.tmin[i], s.tmax[k] = s.tmax[i], s.flags[k] = s.flags[i];
            s.rnd[k] = s.rnd[i];
            for (int c = 0; c < 7; ++c)
                s.payload[(size_t)c * s.capacity + k] = s.payload[(size_t)c * s.capacity + i];
            ++k;
        }
    }
    return k;
}

// cpu_compact_secondary (mapping_cpu.art:255-310), scalar branch
int compact_secondary(SecondaryStream& s, int size)
{
    int k = 0;
    for (int i = 0; i < size; ++i) {
        if (s.id[i] >= 0) {
            s.id[k]    = s.id[i];
            s.org_x[k] = s.org_x[i], s.org_y[k] = s.org_y[i], s.org_z[k] = s.org_z[i];
            s.dir_x[k] = s.dir_x[i], s.dir_y[k] = s.dir_y[i], s.dir_z[k] = s.dir_z[i];
            s.tmin[k] = s.tmin[i], s.tmax[k] = s.tmax[i], s.flags[k] = s.flags[i];
            s.mat_id[k]  = s.mat_id[i];
            s.color_r[k] = s.color_r[i], s.color_g[k] = s.color_g[i], s.color_b[k] = s.color_b[i];
            ++k;
        }
    }
    return k;
}

// (a cache line of its own, and each worker counts into a copy on its stack: the per-thread blocks used to be packed 64-byte neighbours
// that every node visit incremented — false sharing held 8 threads to 1.9 x one)
struct alignas(64) Counters {
    uint64_t camera = 0, bounce = 0, shadow = 0, unoccluded = 0;
    TraversalStats trav;
};

void camera_scale(const igd_scene& sc, int width, int height, float& sx, float& sy)
{
    const ig_camera& c = sc.camera;
    if (c.type == IG_CAMERA_FISHLENS) {
        // aspect handling of make_fishlens_camera (fishlens.art:12-37)
        const float asp = (float)width / (float)height;
        switch (c.fisheye_mode) {
        default:
        case IG_FISHEYE_CIRCULAR:
            sx = asp < 1 ? 1 : asp;
            sy = asp > 1 ? 1 : asp;
            break;
        case IG_FISHEYE_CROPPED:
            sx = asp < 1 ? 1 / asp : 1;
            sy = asp > 1 ? 1 / asp : 1;
            break;
        case IG_FISHEYE_FULL: {
            const float diameter = std::sqrt(asp * asp + 1) * (float)height;
            const float f        = diameter / (float)std::min(width, height);
            sx                   = asp < 1 ? f : f / asp;
            sy                   = asp > 1 ? f : f * asp;
        } break;
        }
        return;
    }
    // aspect = width / height unless fixed by the scene (PerspectiveCamera.cpp:41-45, OrthogonalCamera.cpp:33-35)
    const float aspect = c.aspect_ratio > 0 ? c.aspect_ratio : (float)width / (float)height;
    if (c.type == IG_CAMERA_ORTHOGONAL) {
        // make_vec2(camera_scale, camera_scale / aspect) (OrthogonalCamera.cpp:46)
        sx = c.scale;
        sy = c.scale / aspect;
        return;
    }
    // compute_scale_from_hfov / _vfov (camera/perspective.art:2-13)
    if (c.fov_is_vertical) {
        sy = std::tan(c.fov / 2);
        sx = sy * aspect;
    } else {
        sx = std::tan(c.fov / 2);
        sy = sx / aspect;
    }
}

// One tile of cpu_trace (mapping_cpu.art:731-857)
void trace_tile(const igd_scene& sc, const oracle_settings& cfg, const CameraSetup& cam, const PathTracer& pt_tech,
                int xmin, int ymin, int xmax, int ymax, float* fb, float* aov_normals, float* aov_albedo, float* aov_direct, float* aov_nee,
                PrimaryStream& primary, SecondaryStream& secondary, std::vector<int>& ray_begins, std::vector<int>& ray_ends, Counters& cnt)
{
    const int spi      = cfg.spi;
    const int capacity = primary.capacity;
    const int W = cfg.width, H = cfg.height;
    const int tile_width = xmax - xmin, tile_height = ymax - ymin;
    const int num_rays  = spi * tile_height * tile_width;
    const int E         = (int)sc.entity_count;
    const float inv_spi = 1 / (float)spi; // make_standard_accumulator (driver/accumulator.art:23-30)
    (void)H;

    // aov_di.splat / aov_nee.splat of the path tracer with "aov_mis" (pathtracer.art:133,216): standard accumulators too
    auto splat_aov = [&](float* aov, int ray_id, Color c) {
        if (!aov || !sc.technique.aov_mis)
            return;
        const int pixel = ray_id / spi;
        aov[pixel * 3 + 0] += c.r * inv_spi;
        aov[pixel * 3 + 1] += c.g * inv_spi;
        aov[pixel * 3 + 2] += c.b * inv_spi;
    };
    auto splat = [&](int ray_id, Color c) {
        const int pixel = ray_id / spi;
        fb[pixel * 3 + 0] += c.r * inv_spi;
        fb[pixel * 3 + 1] += c.g * inv_spi;
        fb[pixel * 3 + 2] += c.b * inv_spi;
    };

    int id = 0, current_size = 0;
    while (id < num_rays || current_size > 0) {
        // (Re-)generate primary rays: cpu_generate_rays (mapping_cpu.art:313-360)
        if (current_size < capacity && id < num_rays) {
            const int n = std::min(num_rays - id, capacity - current_size);
            for (int i = 0; i < n; ++i) {
                const int in_tile_id    = id + i;
                const int sample        = in_tile_id % spi;
                const int in_tile_pixel = in_tile_id / spi;
                const int in_tile_y     = in_tile_pixel / tile_width;
                const int in_tile_x     = in_tile_pixel - in_tile_y * tile_width;
                const int x = xmin + in_tile_x, y = ymin + in_tile_y;
                const int cur = current_size + i;
                Rng rnd{ create_random_seed(sample, cfg.iteration, cfg.frame, x, y, cfg.seed), 1 };
                Ray ray;
                const bool valid = generate_camera_ray(cam, rnd, cfg.iteration * spi + sample, x, y, W, cfg.height, ray);
                if (!valid)
                    ray = make_ray(Vec3{ 0, 0, 0 }, Vec3{ 0, 0, 0 }, 0, 0, 0); // make_zero_ray, id -1 (mapping_cpu.art:352-355)
                write_ray(primary, cur, ray);
                primary.id[cur]  = valid ? (y * W + x) * spi + sample : -1;
                primary.rnd[cur] = rnd.counter;
                write_payload(primary, cur, PTRayPayload{ 0, Color{ 1, 1, 1 }, 1, 1, -1 }); // (init_wireframe_raypayload: depth 1, distance 0 in the inv_pdf slot) // init_pt_raypayload (pathtracer.art:33-38), init_vpt_raypayload (volpathtracer.art:27-33)
            }
            current_size += n;
            id += n;
            cnt.camera += (uint64_t)n;
        }

        if (E == 0) {
            // only miss shading
            for (int i = 0; i < current_size; ++i) {
                Color c;
                if (primary.id[i] >= 0 && pt_tech.on_miss(read_ray(primary, i), read_payload(primary, i), c))
                    splat(primary.id[i], c);
            }
            current_size = 0;
            continue;
        }

        // cpu_traverse_primary (mapping_cpu.art:373-403)
        for (int i = 0; i < current_size; ++i) {
            const Hit hit      = traverse_scene(sc, read_ray(primary, i), false, cnt.trav);
            primary.ent_id[i]  = hit.ent_id;
            primary.prim_id[i] = hit.prim_id;
            primary.t[i]       = hit.distance;
            primary.u[i]       = hit.u;
            primary.v[i]       = hit.v;
        }

        const int total_size = current_size;
        current_size         = sort_primary(primary, current_size, ray_begins, ray_ends, E);

        // cpu_hit_shade per entity range (mapping_cpu.art:467-559,773-784)
        for (int ent = 0, begin = 0; ent < E; ++ent) {
            const int end = ray_ends[ent];
            if (begin < end) {
                const Entity entity         = load_entity(sc, ent);
                const ig_material& mat_base = sc.materials[entity.mat_id];
                for (int i = begin; i < end; ++i) {
                    const Ray ray     = read_ray(primary, i);
                    const Hit hit     = Hit{ primary.t[i], primary.u[i], primary.v[i], primary.prim_id[i], primary.ent_id[i] };
                    const int ray_id  = primary.id[i];
                    const int sample  = ray_id % spi;
                    const int pixel_l = ray_id / spi;
                    const int px = pixel_l % W, py = pixel_l / W;
                    Rng rnd{ create_random_seed(sample, cfg.iteration, cfg.frame, px, py, cfg.seed), primary.rnd[i] };

                    PTRayPayload payload      = read_payload(primary, i);
                    const SurfaceElement surf = surface_element(sc, entity, ray, hit);
                    ig_material mat_local;
                    const ig_material& mat = resolve_material(sc, mat_base, surf, vec3_neg(ray.dir), mat_local); // number expressions, per hit
                    // a bump-mapped material hands its inner BSDF a re-oriented surface (bsdf/map.art:64-67)
                    const SurfaceElement bsurf = (mat.flags & (IG_MAT_BUMP | IG_MAT_NORMALMAP | IG_MAT_EXPR_NORMAL)) ? bumped_surface(sc, mat, surf, ray) : surf;
                    // make_doublesided_bsdf (bsdf/common.art:28-46): from behind, the BSDF is built on the surface "as entered"
                    const bool ds_flip = (mat.flags & IG_MAT_DOUBLESIDED) && !surf.is_entering;
                    SurfaceElement dsurf = bsurf;
                    dsurf.is_entering    = true;
                    const Bsdf bsdf{ &mat, ds_flip ? &dsurf : &bsurf, &sc, ds_flip, vec3_neg(ray.dir) };

                    // wrap_infobuffer_renderer.on_hit (technique/internal/infobuffer.art:9-25)
                    if (aov_normals && cfg.iteration == 0 && (ray.flags & IG_RAY_FLAG_CAMERA)) {
                        const Vec3 n   = surf.local.col[2];
                        const Color al = color_saturate(bsdf.albedo(vec3_neg(ray.dir)), 1);
                        const int px_l = ray_id / spi;
                        aov_normals[px_l * 3 + 0] += n.x * inv_spi, aov_normals[px_l * 3 + 1] += n.y * inv_spi, aov_normals[px_l * 3 + 2] += n.z * inv_spi;
                        aov_albedo[px_l * 3 + 0] += al.r * inv_spi, aov_albedo[px_l * 3 + 1] += al.g * inv_spi, aov_albedo[px_l * 3 + 2] += al.b * inv_spi;
                    }

                    Color hit_color;
                    if (pt_tech.debug)
                        hit_color = pt_tech.debug_hit(ray, hit, surf, entity, bsdf, mat, entity.mat_id);
                    else if (!pt_tech.on_hit(ray, hit, surf, payload, mat, hit_color))
                        hit_color = Color{ 0, 0, 0 };
                    else
                        splat_aov(aov_direct, ray_id, hit_color);
                    splat(ray_id, hit_color);

                    const ShadowRayOut sh = pt_tech.on_shadow(ray, surf, rnd, payload, bsdf);
                    if (sh.valid) {
                        secondary.org_x[i] = sh.ray.org.x, secondary.org_y[i] = sh.ray.org.y, secondary.org_z[i] = sh.ray.org.z;
                        secondary.dir_x[i] = sh.ray.dir.x, secondary.dir_y[i] = sh.ray.dir.y, secondary.dir_z[i] = sh.ray.dir.z;
                        secondary.tmin[i]  = sh.ray.tmin;
                        secondary.tmax[i]  = sh.ray.tmax;
                        secondary.flags[i] = sh.ray.flags;
                        secondary.mat_id[i]  = entity.mat_id + 1;
                        secondary.color_r[i] = sh.color.r, secondary.color_g[i] = sh.color.g, secondary.color_b[i] = sh.color.b;
                        secondary.id[i] = ray_id;
                    } else {
                        secondary.id[i] = -1;
                    }

                    Ray new_ray;
                    if (pt_tech.wireframe ? pt_tech.wire_bounce(ray, hit, surf, payload, new_ray) : pt_tech.on_bounce(ray, surf, rnd, payload, bsdf, mat, new_ray)) {
                        write_ray(primary, i, new_ray);
                        primary.rnd[i] = rnd.counter;
                        write_payload(primary, i, payload);
                    } else {
                        primary.id[i] = -1;
                    }
                }
            }
            begin = end;
        }

        // cpu_miss_shade (mapping_cpu.art:582-617,787-790)
        {
            const int begin = ray_ends[E - 1], last = ray_ends[E];
            (void)total_size;
            for (int i = begin; i < last; ++i) {
                if (primary.id[i] < 0)
                    continue; // sample without a camera ray (masked fishlens): dropped, see generate_camera_ray
                Color c;
                if (!pt_tech.on_miss(read_ray(primary, i), read_payload(primary, i), c))
                    c = Color{ 0, 0, 0 };
                splat(primary.id[i], c);
                primary.id[i] = -1;
            }
        }

        int secondary_size = current_size;
        current_size       = compact_primary(primary, current_size);
        cnt.bounce += (uint64_t)current_size;

        secondary_size = compact_secondary(secondary, secondary_size);
        if (secondary_size > 0) {
            // cpu_traverse_secondary (mapping_cpu.art:406-435): any-hit
            for (int i = 0; i < secondary_size; ++i) {
                const Ray ray = make_ray(Vec3{ secondary.org_x[i], secondary.org_y[i], secondary.org_z[i] },
                                         Vec3{ secondary.dir_x[i], secondary.dir_y[i], secondary.dir_z[i] },
                                         secondary.tmin[i], secondary.tmax[i], secondary.flags[i]);
                const Hit hit = traverse_scene(sc, ray, true, cnt.trav);
                const int m   = secondary.mat_id[i];
                secondary.mat_id[i] = hit.prim_id < 0 ? -std::abs(m) : std::abs(m); // streams.art:112-118
            }
            cnt.shadow += (uint64_t)secondary_size;

            // serial splat of unoccluded shadow rays (mapping_cpu.art:840-853)
            for (int i = 0; i < secondary_size; ++i) {
                if (secondary.mat_id[i] < 0) {
                    splat(secondary.id[i], Color{ secondary.color_r[i], secondary.color_g[i], secondary.color_b[i] });
                    splat_aov(aov_nee, secondary.id[i], Color{ secondary.color_r[i], secondary.color_g[i], secondary.color_b[i] });
                    ++cnt.unoccluded;
                }
            }
        }
    }
}

// The light tracer (technique/lighttracer.art) path by path: a path's random numbers, vertices and splats are those of the wavefront
// pipeline (emitter -> primary traversal -> hit shading with on_shadow / on_bounce -> any-hit traversal -> on_advanced_shadow_miss
// splat), only the order in which different paths add into a pixel differs, so sums agree to rounding. One thread: splats go anywhere.
void trace_light_paths(const igd_scene& sc, const oracle_settings& cfg, const CameraSetup& cam, float* fb, Counters& cnt)
{
    const LightTracer lt(sc);
    const int spi = cfg.spi, W = cfg.width, H = cfg.height;
    const float inv_spi = 1 / (float)spi;
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x)
            for (int sample = 0; sample < spi; ++sample) {
                Rng rnd{ create_random_seed(sample, cfg.iteration, cfg.frame, x, y, cfg.seed), 1 };
                Ray ray;
                PTRayPayload payload;
                ++cnt.camera;
                if (!lt.emit(rnd, ray, payload) || sc.entity_count == 0)
                    continue;
                for (;;) {
                    const Hit hit = traverse_scene(sc, ray, false, cnt.trav);
                    if (hit.prim_id < 0)
                        break; // TechniqueNoMissFunction
                    const Entity entity        = load_entity(sc, hit.ent_id);
                    const SurfaceElement surf  = surface_element(sc, entity, ray, hit);
                    ig_material mat_local;
                    const ig_material& mat     = resolve_material(sc, sc.materials[entity.mat_id], surf, vec3_neg(ray.dir), mat_local);
                    const bool bumped          = (mat.flags & (IG_MAT_BUMP | IG_MAT_NORMALMAP | IG_MAT_EXPR_NORMAL)) != 0;
                    const SurfaceElement bsurf = bumped ? bumped_surface(sc, mat, surf, ray) : surf;
                    const bool ds_flip         = (mat.flags & IG_MAT_DOUBLESIDED) && !surf.is_entering;
                    SurfaceElement dsurf       = bsurf;
                    dsurf.is_entering          = true;
                    Bsdf bsdf{ &mat, ds_flip ? &dsurf : &bsurf, &sc, ds_flip, vec3_neg(ray.dir) };
                    bsdf.adjoint    = true;
                    bsdf.bumped     = bumped;
                    bsdf.old_normal = surf.local.col[2];

                    float nx = 0, ny = 0;
                    // (ctx.surf of the callbacks is the hit's surface element; the BSDF may sit on a re-oriented one)
                    const ShadowRayOut sh = lt.on_shadow(cam, ray, surf, payload, bsdf, nx, ny);
                    if (sh.valid) {
                        ++cnt.shadow;
                        const Hit occluder = traverse_scene(sc, sh.ray, true, cnt.trav);
                        if (occluder.prim_id < 0) {
                            ++cnt.unoccluded;
                            const int pixel = LightTracer::pixel_from_normalized(nx, ny, W, H);
                            fb[pixel * 3 + 0] += sh.color.r * inv_spi;
                            fb[pixel * 3 + 1] += sh.color.g * inv_spi;
                            fb[pixel * 3 + 2] += sh.color.b * inv_spi;
                        }
                    }
                    Ray new_ray;
                    if (!lt.on_bounce(ray, surf, rnd, payload, bsdf, new_ray))
                        break;
                    ray = new_ray;
                    ++cnt.bounce;
                }
            }
}

// The photon mapper (technique/photonmapper.art) for one iteration: the light pass path by path (make_ppm_light_emitter :141-165,
// make_ppm_light_renderer :169-245), the voxel grid (:60-110, 395-431), then the camera pass path by path (make_ppm_path_renderer
// :262-388). A light path stores at most one photon, at the slot of its index; inside a grid cell photons keep index order (the
// reference's order there is whatever its atomics produce), so that a gather's float sum is defined.
void render_photon_mapped(const igd_scene& sc, const oracle_settings& cfg, const CameraSetup& cam, float* fb, Counters& cnt)
{
    const LightTracer lt(sc); // the emitter shares everything but the payload with make_lt_emitter
    const ig_technique& tech = sc.technique;
    const int P              = tech.photon_count;
    constexpr float offset   = 0.001f;
    struct BsdfSetup { // the material's BSDF over the hit's surface (bump / two-sided wrappers as in trace_light_paths)
        SurfaceElement surf, bsurf, dsurf;
        Bsdf bsdf;
    };
    auto make_bsdf = [&](BsdfSetup& b, const ig_material& mat, const Ray& ray, bool adjoint) {
        const bool bumped = (mat.flags & (IG_MAT_BUMP | IG_MAT_NORMALMAP | IG_MAT_EXPR_NORMAL)) != 0;
        b.bsurf           = bumped ? bumped_surface(sc, mat, b.surf, ray) : b.surf;
        const bool ds_flip = (mat.flags & IG_MAT_DOUBLESIDED) && !b.surf.is_entering;
        b.dsurf             = b.bsurf;
        b.dsurf.is_entering = true;
        b.bsdf              = Bsdf{ &mat, ds_flip ? &b.dsurf : &b.bsurf, &sc, ds_flip, vec3_neg(ray.dir) };
        b.bsdf.adjoint      = adjoint;
        b.bsdf.bumped       = bumped;
        b.bsdf.old_normal   = b.surf.local.col[2];
    };

    // ---- light pass
    std::vector<igp_photon> photons((size_t)P);
    for (auto& ph : photons)
        ph.light = -1;
    for (int x = 0; x < P; ++x) {
        Rng rnd{ create_random_seed(0, cfg.iteration, cfg.frame, x, 0, cfg.seed), 1 };
        Ray ray;
        PTRayPayload pt;
        int32_t light_id = 0;
        ++cnt.camera;
        if (!lt.emit(rnd, ray, pt, &light_id) || sc.entity_count == 0)
            continue;
        for (;;) {
            const Hit hit = traverse_scene(sc, ray, false, cnt.trav);
            if (hit.prim_id < 0)
                break; // TechniqueNoMissFunction
            const Entity entity = load_entity(sc, hit.ent_id);
            BsdfSetup b;
            b.surf = surface_element(sc, entity, ray, hit);
            ig_material mat_local;
            const ig_material& mat = resolve_material(sc, sc.materials[entity.mat_id], b.surf, vec3_neg(ray.dir), mat_local);
            make_bsdf(b, mat, ray, true);
            const bool emissive = mat.light_id >= 0, all_delta = b.bsdf.is_all_delta();
            const Vec3 out_dir = vec3_neg(ray.dir);
            // on_hit (:172-195)
            if (!emissive && !all_delta) {
                const float cos_o = vec3_dot(out_dir, b.surf.local.col[2]);
                if (cos_o > flt_eps) {
                    igp_photon& ph = photons[(size_t)x];
                    ph.dir   = g_ppm_sound_directions ? sound_encode_normal_32(out_dir.x, out_dir.y, out_dir.z) : igp_encode_normal_32(out_dir.x, out_dir.y, out_dir.z);
                    ph.light = light_id;
                    ph.power = igp_encode_rgbe(pt.contrib.r, pt.contrib.g, pt.contrib.b);
                    ph.depth = pt.depth;
                    ph.pos[0] = b.surf.point.x, ph.pos[1] = b.surf.point.y, ph.pos[2] = b.surf.point.z;
                    ph.eta = pt.eta;
                }
            }
            // on_bounce (:197-227)
            if (!(all_delta && pt.depth + 2 <= tech.max_light_depth))
                break;
            BsdfSample ms;
            if (!b.bsdf.sample(rnd, out_dir, ms))
                break;
            const Color contrib = color_mul(pt.contrib, ms.color);
            if (!(color_average(contrib) > flt_eps))
                break;
            pt.contrib = contrib;
            pt.depth   = pt.depth + 1;
            pt.eta     = pt.eta * ms.eta;
            ray        = make_ray(b.surf.point, ms.in_dir, offset, flt_max, IG_RAY_FLAG_BOUNCE);
            ++cnt.bounce;
        }
    }

    // ---- grid: (cell, index) order, offsets per cell
    std::vector<std::pair<uint64_t, uint32_t>> keys;
    for (int x = 0; x < P; ++x)
        if (photons[(size_t)x].light >= 0)
            keys.push_back({ ((uint64_t)(uint32_t)igp_grid_cell(photons[(size_t)x].pos, sc.bbox_min, sc.bbox_max) << 32) | (uint32_t)x, (uint32_t)x });
    std::sort(keys.begin(), keys.end());
    std::vector<igp_photon> sorted(keys.size());
    std::vector<uint32_t> cell_offset((size_t)IGP_GRID_CELLS + 1, 0);
    for (size_t i = 0; i < keys.size(); ++i) {
        sorted[i] = photons[keys[i].second];
        ++cell_offset[(size_t)(keys[i].first >> 32) + 1];
    }
    for (size_t c = 1; c < cell_offset.size(); ++c)
        cell_offset[c] += cell_offset[c - 1];

    // ---- camera pass
    const PathTracer pt_lights(sc); // infinite-light emission only
    const float radius = igp_compute_radius(tech.merge_radius, cfg.iteration);
    const int spi = cfg.spi, W = cfg.width, H = cfg.height;
    const float inv_spi = 1 / (float)spi;
    const float clamp_value = tech.clamp;
    auto handle_color = [&](Color c) { return clamp_value > 0 ? color_saturate(c, clamp_value) : c; };
    for (int y = 0; y < H; ++y) {
        if (cfg.row_stride > 1 && y % cfg.row_stride != cfg.row_offset)
            continue;
        for (int x = 0; x < W; ++x)
            for (int sample = 0; sample < spi; ++sample) {
                Rng rnd{ create_random_seed(sample, cfg.iteration, cfg.frame, x, y, cfg.seed), 1 };
                Ray ray;
                ++cnt.camera;
                if (!generate_camera_ray(cam, rnd, cfg.iteration * spi + sample, x, y, W, H, ray))
                    continue;
                Color contrib{ 1, 1, 1 }; // init_ppm_raypayload (:131-137)
                int depth = 1, path_type = 0;
                float eta = 1, radius_payload = 0;
                float* px = fb + ((size_t)y * W + x) * 3;
                auto splat = [&](Color c) { px[0] += c.r * inv_spi, px[1] += c.g * inv_spi, px[2] += c.b * inv_spi; };
                for (;;) {
                    const Hit hit = sc.entity_count ? traverse_scene(sc, ray, false, cnt.trav) : Hit{ -1, -1, 0, 0, 0 };
                    if (hit.prim_id < 0) {
                        // on_miss (:333-360)
                        if (path_type != 1) {
                            int inflights = 0;
                            Color color{ 0, 0, 0 };
                            for (uint32_t i = 0; i < sc.infinite_light_count; ++i) {
                                Color emit;
                                if (!pt_lights.infinite_emission(sc.lights[i], ray.dir, emit))
                                    continue;
                                ++inflights;
                                color = color_add(color, color_mul(contrib, emit));
                            }
                            if (inflights > 0)
                                splat(handle_color(color));
                        }
                        break;
                    }
                    const Entity entity = load_entity(sc, hit.ent_id);
                    BsdfSetup b;
                    b.surf = surface_element(sc, entity, ray, hit);
                    ig_material mat_local;
                    const ig_material& mat = resolve_material(sc, sc.materials[entity.mat_id], b.surf, vec3_neg(ray.dir), mat_local);
                    make_bsdf(b, mat, ray, false);
                    const bool emissive = mat.light_id >= 0, all_delta = b.bsdf.is_all_delta();
                    const Vec3 out_dir = vec3_neg(ray.dir);
                    const Vec3 N       = b.surf.local.col[2];
                    // get_radius (:275-283)
                    const float actual_radius = depth > 1 ? radius_payload : igm_min(radius, hit.distance * 0.017455064f);
                    // on_hit (:285-331)
                    bool answered = false;
                    if (path_type == 0 && emissive && b.surf.is_entering) {
                        const float dot = vec3_dot(out_dir, N);
                        if (dot > flt_eps) {
                            const ig_light& light = sc.lights[mat.light_id];
                            Color emit;
                            if (light.type == IG_LIGHT_MESH_AREA)
                                emit = MeshEmitter(sc, light).radiance;
                            else if (light.type == IG_LIGHT_SPHERE)
                                emit = Color{ light.d[4], light.d[5], light.d[6] };
                            else
                                emit = PlaneEmitter(light).radiance;
                            splat(handle_color(color_mul(contrib, emit)));
                            answered = true;
                        }
                    }
                    if (!answered && depth + 1 <= tech.max_depth && !emissive && !all_delta) {
                        const float cos_o = vec3_dot(out_dir, N);
                        if (igm_abs(cos_o) > flt_eps) {
                            Color total{ 0, 0, 0 };
                            if (!(actual_radius <= flt_eps) && !sorted.empty()) {
                                const float r2    = actual_radius * actual_radius;
                                const float lo[3] = { b.surf.point.x - actual_radius, b.surf.point.y - actual_radius, b.surf.point.z - actual_radius };
                                const float hi[3] = { b.surf.point.x + actual_radius, b.surf.point.y + actual_radius, b.surf.point.z + actual_radius };
                                int32_t cmin[3], cmax[3];
                                igp_grid_pos(lo, sc.bbox_min, sc.bbox_max, cmin);
                                igp_grid_pos(hi, sc.bbox_min, sc.bbox_max, cmax);
                                for (int iz = cmin[2]; iz <= cmax[2]; ++iz)
                                    for (int iy = cmin[1]; iy <= cmax[1]; ++iy)
                                        for (int ix = cmin[0]; ix <= cmax[0]; ++ix) {
                                            const int32_t cell = igp_morton_3d(ix, iy, iz);
                                            Color cc{ 0, 0, 0 };
                                            for (uint32_t i = cell_offset[(size_t)cell]; i < cell_offset[(size_t)cell + 1]; ++i) {
                                                const igp_photon& ph = sorted[i];
                                                const Vec3 d      = vec3_sub(b.surf.point, make_vec3(ph.pos[0], ph.pos[1], ph.pos[2]));
                                                const float dist2 = vec3_dot(d, d);
                                                if (!(dist2 <= r2))
                                                    continue;
                                                float dir[3], pw[3];
                                                if (g_ppm_sound_directions)
                                                    sound_decode_normal_32(ph.dir, dir);
                                                else
                                                    igp_decode_normal_32(ph.dir, dir);
                                                const Vec3 in_dir = make_vec3(dir[0], dir[1], dir[2]);
                                                const float cos_i = vec3_dot(in_dir, N);
                                                if (depth + ph.depth <= tech.max_depth && cos_o * cos_i > flt_eps) {
                                                    igp_decode_rgbe(ph.power, pw);
                                                    const float kf  = igp_kernel(r2, dist2);
                                                    const Color mc  = b.bsdf.eval(in_dir, out_dir);
                                                    cc = color_add(cc, color_mulf(color_mul(Color{ pw[0], pw[1], pw[2] }, mc), safe_div(kf, igm_abs(cos_i))));
                                                }
                                            }
                                            total = color_add(total, cc);
                                        }
                            }
                            const float n = (float)P;
                            splat(handle_color(color_mul(contrib, Color{ total.r / n, total.g / n, total.b / n })));
                        }
                    }
                    // on_bounce (:362-399)
                    if (depth + 1 > tech.max_depth)
                        break;
                    BsdfSample ms;
                    if (!b.bsdf.sample(rnd, out_dir, ms) || ms.pdf <= flt_eps)
                        break;
                    const Color nc      = color_mul(contrib, ms.color);
                    const float rr_prob = (depth + 1 > tech.min_depth) ? russian_roulette_pbrt(color_mulf(nc, eta * eta), 0.95f) : 1.0f;
                    if (rnd.next_f32() >= rr_prob)
                        break;
                    contrib        = color_mulf(nc, 1 / rr_prob);
                    depth          = depth + 1;
                    eta            = eta * ms.eta;
                    radius_payload = actual_radius;
                    path_type      = ms.is_delta ? path_type : 1;
                    ray            = make_ray(b.surf.point, ms.in_dir, offset, flt_max, IG_RAY_FLAG_BOUNCE);
                    ++cnt.bounce;
                }
            }
    }
}

} // namespace

extern "C" {

// Renders ONE iteration into `fb` (float[height][width][3], accumulated with +=,
// exactly like the reference framebuffer: Runtime divides by the iteration count on save,
// src/runtime/Runtime.cpp:808-826).
// aov_normals / aov_albedo: optional float[height][width][3] (accumulated with +=), the info-buffer AOVs of iteration 0
// aov_direct / aov_nee: optional float[height][width][3], the "Direct Weights" / "NEE Weights" AOVs of a path tracer with aov_mis
int oracle_render_aovs(const igd_scene* sc, const oracle_settings* cfg, float* fb, oracle_stats* stats, float* aov_normals, float* aov_albedo,
                       float* aov_direct, float* aov_nee)
{
    if (!sc || !cfg || !fb || cfg->spi <= 0 || cfg->width <= 0 || cfg->height <= 0)
        return -1;

    float sx, sy;
    camera_scale(*sc, cfg->width, cfg->height, sx, sy);
    const CameraSetup cam = make_camera(sc->camera, sx, sy);
    if (sc->technique.type == IG_TECHNIQUE_LIGHTTRACER) {
        if (cfg->row_stride > 1 || cfg->xmax > 0 || cfg->ymax > 0)
            return -1; // splats land anywhere on the film: whole-film renders only
        Counters c;
        trace_light_paths(*sc, *cfg, cam, fb, c);
        if (stats) {
            stats->camera_rays += c.camera;
            stats->bounce_rays += c.bounce;
            stats->shadow_rays += c.shadow;
            stats->unoccluded += c.unoccluded;
            stats->nodes += c.trav.nodes;
            stats->tris += c.trav.tris;
            stats->leaves += c.trav.leaves;
            stats->max_stack = std::max(stats->max_stack, c.trav.max_stack);
            stats->threads_used = 1;
        }
        return 0;
    }
    if (sc->technique.type == IG_TECHNIQUE_PPM) {
        if (cfg->xmax > 0 || cfg->ymax > 0)
            return -1;
        Counters c;
        render_photon_mapped(*sc, *cfg, cam, fb, c);
        if (stats) {
            stats->camera_rays += c.camera;
            stats->bounce_rays += c.bounce;
            stats->nodes += c.trav.nodes;
            stats->tris += c.trav.tris;
            stats->leaves += c.trav.leaves;
            stats->max_stack = std::max(stats->max_stack, c.trav.max_stack);
            stats->threads_used = 1;
        }
        return 0;
    }
    PathTracer pt(*sc);
    if (pt.wireframe) {
        // camera.differential (camera/perspective.art:59-64, orthogonal.art:38-43) -> footprint_u = |dx x dy| (wireframe.art:25-26)
        const bool ortho = sc->camera.type == IG_CAMERA_ORTHOGONAL;
        const Vec3 dx = vec3_mulf(cam.view.col[0], ortho ? 1.0f : sx), dy = vec3_mulf(cam.view.col[1], ortho ? 1.0f : sy);
        pt.wire_footprint = vec3_len(vec3_cross(dx, dy));
    }

    const int tile_size = 16; // ShaderUtils.cpp:37
    const int x0 = cfg->xmax > 0 ? cfg->xmin : 0, y0 = cfg->ymax > 0 ? cfg->ymin : 0;
    const int x1 = cfg->xmax > 0 ? cfg->xmax : cfg->width, y1 = cfg->ymax > 0 ? cfg->ymax : cfg->height;
    const int tiles_x = (x1 - x0 + tile_size - 1) / tile_size;
    // with row sharding a "tile" is a 16-pixel run of one owned row
    const bool sharded = cfg->row_stride > 1;
    std::vector<int> rows;
    if (sharded)
        for (int y = y0; y < y1; ++y)
            if (y % cfg->row_stride == cfg->row_offset)
                rows.push_back(y);
    const int tiles_y   = sharded ? (int)rows.size() : (y1 - y0 + tile_size - 1) / tile_size;
    const int num_tiles = tiles_x * tiles_y;

    int threads = cfg->threads > 0 ? cfg->threads : (int)std::thread::hardware_concurrency();
    threads     = std::max(1, std::min(threads, std::max(1, num_tiles)));

    std::atomic<int> next_tile{ 0 };
    std::vector<Counters> counters((size_t)threads);
    auto worker = [&](int tid) {
        PrimaryStream primary;
        SecondaryStream secondary;
        const int capacity = cfg->spi * tile_size * tile_size; // cpu_get_stream_capacity (mapping_cpu.art:717)
        primary.resize(capacity);
        secondary.resize(capacity);
        std::vector<int> ray_begins(sc->entity_count + 2), ray_ends(sc->entity_count + 2);
        Counters local;
        for (;;) {
            const int tile = next_tile.fetch_add(1);
            if (tile >= num_tiles) {
                counters[(size_t)tid] = local;
                break;
            }
            const int tx = tile % tiles_x, ty = tile / tiles_x;
            const int xmin = x0 + tx * tile_size, ymin = sharded ? rows[(size_t)ty] : y0 + ty * tile_size;
            const int xmax = std::min(xmin + tile_size, x1), ymax = sharded ? ymin + 1 : std::min(ymin + tile_size, y1);
            trace_tile(*sc, *cfg, cam, pt, xmin, ymin, xmax, ymax, fb, aov_normals, aov_albedo, aov_direct, aov_nee, primary, secondary, ray_begins, ray_ends, local);
        }
    };

    std::vector<std::thread> pool;
    for (int t = 1; t < threads; ++t)
        pool.emplace_back(worker, t);
    worker(0);
    for (auto& th : pool)
        th.join();

    if (stats) {
        for (const auto& c : counters) {
            stats->camera_rays += c.camera;
            stats->bounce_rays += c.bounce;
            stats->shadow_rays += c.shadow;
            stats->unoccluded += c.unoccluded;
            stats->nodes += c.trav.nodes;
            stats->tris += c.trav.tris;
            stats->leaves += c.trav.leaves;
            stats->max_stack = std::max(stats->max_stack, c.trav.max_stack);
        }
        stats->threads_used = threads;
    }
    return 0;
}

// ig_photon.h on its own, for the tests' independent restatements: a photon's direction and power through their encodings, and the
// grid cell of a position
void oracle_photon_codec(const float dir[3], const float power[3], int32_t* enc_dir, int32_t* enc_power, float out_dir[3], float out_power[3])
{
    *enc_dir   = igp_encode_normal_32(dir[0], dir[1], dir[2]);
    *enc_power = igp_encode_rgbe(power[0], power[1], power[2]);
    igp_decode_normal_32(*enc_dir, out_dir);
    igp_decode_rgbe(*enc_power, out_power);
}
void oracle_set_ppm_sound_directions(int on) { g_ppm_sound_directions = on != 0; }
int32_t oracle_photon_cell(const float pos[3], const float bmin[3], const float bmax[3]) { return igp_grid_cell(pos, bmin, bmax); }
float oracle_photon_radius(float max_radius, int32_t iteration) { return igp_compute_radius(max_radius, iteration); }

int oracle_render_ex(const igd_scene* sc, const oracle_settings* cfg, float* fb, oracle_stats* stats, float* aov_normals, float* aov_albedo)
{
    return oracle_render_aovs(sc, cfg, fb, stats, aov_normals, aov_albedo, nullptr, nullptr);
}

int oracle_render(const igd_scene* sc, const oracle_settings* cfg, float* fb, oracle_stats* stats)
{
    return oracle_render_aovs(sc, cfg, fb, stats, nullptr, nullptr, nullptr, nullptr);
}

// Camera rays for ray ids [first_id, first_id + count) of one iteration, id = (y*W + x)*spi + sample
// (gpu_generate_rays id convention, mapping_gpu.art:616-669). rays: 8 floats each (org, dir, tmin, tmax).
int oracle_generate_rays(const igd_scene* sc, const oracle_settings* cfg, int64_t first_id, int64_t count, float* rays, uint32_t* rnd_counter)
{
    if (!sc || !cfg || !rays)
        return -1;
    float sx, sy;
    camera_scale(*sc, cfg->width, cfg->height, sx, sy);
    const CameraSetup cam = make_camera(sc->camera, sx, sy);
    for (int64_t i = 0; i < count; ++i) {
        const int64_t id  = first_id + i;
        const int sample  = (int)(id % cfg->spi);
        const int pixel   = (int)(id / cfg->spi);
        const int x = pixel % cfg->width, y = pixel / cfg->width;
        Rng rnd{ create_random_seed(sample, cfg->iteration, cfg->frame, x, y, cfg->seed), 1 };
        Ray r;
        if (!generate_camera_ray(cam, rnd, cfg->iteration * cfg->spi + sample, x, y, cfg->width, cfg->height, r))
            r = make_ray(Vec3{ 0, 0, 0 }, Vec3{ 0, 0, 0 }, 0, 0, 0); // no ray for this sample: make_zero_ray
        float* o    = rays + i * 8;
        o[0] = r.org.x, o[1] = r.org.y, o[2] = r.org.z;
        o[3] = r.dir.x, o[4] = r.dir.y, o[5] = r.dir.z;
        o[6] = r.tmin, o[7] = r.tmax;
        if (rnd_counter)
            rnd_counter[i] = rnd.counter;
    }
    return 0;
}

// Closest-hit (any_hit = 0) or any-hit (any_hit = 1) traversal of a ray list.
// rays: 8 floats each (org, dir, tmin, tmax). Outputs may be NULL.
int oracle_trace(const igd_scene* sc, int64_t count, const float* rays, uint32_t flags, int any_hit,
                 int32_t* ent_id, int32_t* prim_id, float* t, float* u, float* v, oracle_stats* stats)
{
    if (!sc || !rays)
        return -1;
    TraversalStats st;
    for (int64_t i = 0; i < count; ++i) {
        const float* r = rays + i * 8;
        const Ray ray  = make_ray(Vec3{ r[0], r[1], r[2] }, Vec3{ r[3], r[4], r[5] }, r[6], r[7], flags);
        const Hit hit  = traverse_scene(*sc, ray, any_hit != 0, st);
        if (ent_id) ent_id[i] = hit.ent_id;
        if (prim_id) prim_id[i] = hit.prim_id;
        if (t) t[i] = hit.distance;
        if (u) u[i] = hit.u;
        if (v) v[i] = hit.v;
    }
    if (stats) {
        stats->nodes += st.nodes;
        stats->tris += st.tris;
        stats->leaves += st.leaves;
        stats->max_stack = std::max(stats->max_stack, st.max_stack);
    }
    return 0;
}

// Brute force over every instanced triangle (no BVH): checks the BVH build + traversal.
int oracle_trace_bruteforce(const igd_scene* sc, int64_t count, const float* rays, float* t_out, int32_t* ent_out, int32_t* prim_out)
{
    if (!sc || !rays)
        return -1;
    for (int64_t i = 0; i < count; ++i) {
        const float* r = rays + i * 8;
        Ray ray        = make_ray(Vec3{ r[0], r[1], r[2] }, Vec3{ r[3], r[4], r[5] }, r[6], r[7], 0);
        float best     = ray.tmax;
        int32_t be = -1, bp = -1;
        for (uint32_t e = 0; e < sc->entity_count; ++e) {
            const Entity ent       = load_entity(*sc, (int32_t)e);
            const TriMeshView mesh = load_trimesh(*sc, ent.shape_id);
            Ray local              = transform_ray(ray, ent.local_mat);
            local.tmax             = best;
            for (int f = 0; f < mesh.num_tris; ++f) {
                auto vtx = [&](int32_t k) { return Vec3{ mesh.vertices[k * 4], mesh.vertices[k * 4 + 1], mesh.vertices[k * 4 + 2] }; };
                const Vec3 v0 = vtx(mesh.indices[f * 4]), v1 = vtx(mesh.indices[f * 4 + 1]), v2 = vtx(mesh.indices[f * 4 + 2]);
                const Vec3 e1 = vec3_sub(v2, v0), e2 = vec3_sub(v0, v1);
                const Vec3 n  = compute_stable_triangle_normal(e1, e2, vec3_sub(v1, v2));
                float t, u, v;
                if (intersect_ray_tri_mt(local, v0, e1, e2, n, t, u, v)) {
                    best       = t;
                    local.tmax = t;
                    be         = (int32_t)e;
                    bp         = f;
                }
            }
        }
        if (t_out) t_out[i] = best;
        if (ent_out) ent_out[i] = be;
        if (prim_out) prim_out[i] = bp;
    }
    return 0;
}

// ---- unit-level entry points for the golden-vector tests
int oracle_intersect_tri(const float org[3], const float dir[3], float tmin, float tmax,
                         const float v0[3], const float e1[3], const float e2[3], const float n[3], float out_tuv[3])
{
    const Ray ray = make_ray(Vec3{ org[0], org[1], org[2] }, Vec3{ dir[0], dir[1], dir[2] }, tmin, tmax, 0);
    return intersect_ray_tri_mt(ray, Vec3{ v0[0], v0[1], v0[2] }, Vec3{ e1[0], e1[1], e1[2] }, Vec3{ e2[0], e2[1], e2[2] }, Vec3{ n[0], n[1], n[2] },
                                out_tuv[0], out_tuv[1], out_tuv[2])
               ? 1
               : 0;
}

int oracle_intersect_box(const float org[3], const float dir[3], float tmin, float tmax, const float bmin[3], const float bmax[3], float out[2])
{
    const Ray ray = make_ray(Vec3{ org[0], org[1], org[2] }, Vec3{ dir[0], dir[1], dir[2] }, tmin, tmax, 0);
    intersect_ray_box(ray, bmin, bmax, out[0], out[1]);
    return out[0] <= out[1] ? 1 : 0;
}

uint32_t oracle_random_seed(int32_t sample, int32_t iter, int32_t frame, int32_t x, int32_t y, int32_t user)
{
    return create_random_seed(sample, iter, frame, x, y, user);
}

void oracle_random_f32(uint32_t seed, uint32_t first_counter, int32_t count, float* out_f, uint32_t* out_raw)
{
    Rng a{ seed, first_counter }, b{ seed, first_counter };
    for (int i = 0; i < count; ++i) {
        if (out_f) out_f[i] = a.next_f32();
        if (out_raw) out_raw[i] = b.next_u32();
    }
}

// which: 0 sin, 1 cos, 2 acos, 3 asin
void oracle_detmath(int which, int64_t n, const float* x, float* y)
{
    for (int64_t i = 0; i < n; ++i) {
        switch (which) {
        case 0: y[i] = igm_sin(x[i]); break;
        case 1: y[i] = igm_cos(x[i]); break;
        case 2: y[i] = igm_acos(x[i]); break;
        case 4: y[i] = igm_atan2(x[2 * i], x[2 * i + 1]); break; // x holds (y, x) pairs
        case 5: y[i] = igm_exp(x[i]); break;
        default: y[i] = igm_asin(x[i]); break;
        }
    }
}

// mat3x3_align_vectors (core/matrix.art:261-284) applied to v: out = M * v, m9 = M column by column
void oracle_align_vectors(const float a[3], const float b[3], const float v[3], float out[3], float m9[9])
{
    const Mat3x3 m = mat3x3_align_vectors(Vec3{ a[0], a[1], a[2] }, Vec3{ b[0], b[1], b[2] });
    const Vec3 r   = mat3x3_mul(m, Vec3{ v[0], v[1], v[2] });
    out[0] = r.x, out[1] = r.y, out[2] = r.z;
    for (int c = 0; c < 3; ++c)
        m9[3 * c] = m.col[c].x, m9[3 * c + 1] = m.col[c].y, m9[3 * c + 2] = m.col[c].z;
}

// ensure_valid_reflection (core/sampling.art:118-166)
void oracle_ensure_valid_reflection(const float ng[3], const float i[3], const float n[3], float out[3])
{
    const Vec3 r = ensure_valid_reflection(Vec3{ ng[0], ng[1], ng[2] }, Vec3{ i[0], i[1], i[2] }, Vec3{ n[0], n[1], n[2] });
    out[0] = r.x, out[1] = r.y, out[2] = r.z;
}

// VNDF-GGX (core/microfacet.art:403-425): sampled normal, its pdf, and D(h) * h.z for the identity frame
void oracle_vndf_ggx(float alpha_u, float alpha_v, uint32_t seed, const float wo[3], float normal[3], float* pdf, float* d_times_cos)
{
    Mat3x3 id;
    id.col[0] = Vec3{ 1, 0, 0 }, id.col[1] = Vec3{ 0, 1, 0 }, id.col[2] = Vec3{ 0, 0, 1 };
    const GGX g{ id, alpha_u, alpha_v };
    Rng rnd{ seed, 0 };
    const Vec3 w = Vec3{ wo[0], wo[1], wo[2] };
    const Vec3 m = g.sample(rnd, w);
    normal[0] = m.x, normal[1] = m.y, normal[2] = m.z;
    *pdf         = g.pdf(w, m);
    *d_times_cos = g.D(m) * m.z;
}

// make_image_texture lookup (texture/image.art) of bitmap texture `tex_id` of the scene at n (u, v) pairs
void oracle_image_lookup(const igd_scene* scene, int32_t tex_id, int64_t n, const float* uv, float* rgb)
{
    for (int64_t i = 0; i < n; ++i) {
        const Color c = image_lookup(*scene, scene->textures[tex_id], Vec2{ uv[2 * i], uv[2 * i + 1] });
        rgb[3 * i] = c.r, rgb[3 * i + 1] = c.g, rgb[3 * i + 2] = c.b;
    }
}

// BSDF probe for the tests: material `mat_id` of the scene on a flat surface with normal +z (shading frame = identity,
// texture coordinate (0.5, 0.5)), `entering` as given. mode 0: eval + pdf for n pairs (wi[n][3], wo fixed);
// mode 1: n samples from RNG seed `seed` (counter 1): wi, pdf, colour (weight = eval / pdf), eta; rejected samples get pdf 0.
int oracle_bsdf_probe(const igd_scene* sc, int32_t mat_id, int32_t entering, int32_t mode, const float wo[3], int64_t n, uint32_t seed,
                      float* wi, float* pdf, float* color, float* eta)
{
    if (!sc || mat_id < 0 || (uint32_t)mat_id >= sc->material_count)
        return -1;
    SurfaceElement surf{};
    surf.is_entering  = entering != 0;
    surf.point        = make_vec3(0, 0, 0);
    surf.face_normal  = make_vec3(0, 0, 1);
    surf.tex_coords   = Vec2{ 0.5f, 0.5f };
    surf.local.col[0] = make_vec3(1, 0, 0);
    surf.local.col[1] = make_vec3(0, 1, 0);
    surf.local.col[2] = make_vec3(0, 0, 1);
    const ig_material& mat = sc->materials[mat_id];
    const Bsdf bsdf{ &mat, &surf, sc };
    const Vec3 out_dir = make_vec3(wo[0], wo[1], wo[2]);
    Rng rnd{ seed, 1 };
    for (int64_t i = 0; i < n; ++i) {
        if (mode == 0) {
            const Vec3 in_dir = make_vec3(wi[i * 3], wi[i * 3 + 1], wi[i * 3 + 2]);
            const Color c     = bsdf.eval(in_dir, out_dir);
            color[i * 3] = c.r, color[i * 3 + 1] = c.g, color[i * 3 + 2] = c.b;
            pdf[i] = bsdf.pdf(in_dir, out_dir);
        } else {
            BsdfSample bs{};
            if (bsdf.sample(rnd, out_dir, bs)) {
                wi[i * 3] = bs.in_dir.x, wi[i * 3 + 1] = bs.in_dir.y, wi[i * 3 + 2] = bs.in_dir.z;
                pdf[i] = bs.pdf;
                color[i * 3] = bs.color.r, color[i * 3 + 1] = bs.color.g, color[i * 3 + 2] = bs.color.b;
                if (eta)
                    eta[i] = bs.eta;
            } else {
                pdf[i] = 0;
                color[i * 3] = color[i * 3 + 1] = color[i * 3 + 2] = 0;
                if (eta)
                    eta[i] = 1;
            }
        }
    }
    return 0;
}

// The warps the reference's test_warp.art checks for bijectivity (core/warp.art), forward direction, as the shading code calls them.
// which 0: square_to_concentric_disk(a, b) -> out[0..1]; 1: equal_area_square_to_sphere(a, b) -> out[0..2];
// 2: dir_from_spherical(theta = a, phi = b) as sphere_unmap_uv evaluates it -> out[0..2]; 3: spherical_from_dir of the direction
// (a = theta, b = phi) names, through sphere_map_uv -> out[0] = theta, out[1] = phi
void oracle_warp(int32_t which, float a, float b, float out[3])
{
    out[0] = out[1] = out[2] = 0;
    if (which == 0) {
        square_to_concentric_disk(a, b, out[0], out[1]);
    } else if (which == 1) {
        const Vec3 d = equal_area_square_to_sphere(a, b);
        out[0] = d.x, out[1] = d.y, out[2] = d.z;
    } else if (which == 2) {
        // sphere_unmap_uv returns (dir.y, -dir.x, dir.z) of dir_from_spherical(v pi, u 2 pi): undo the swizzle
        const Vec3 s = sphere_unmap_uv(Vec2{ b / (2 * flt_pi), a / flt_pi });
        out[0] = -s.y, out[1] = s.x, out[2] = s.z;
    } else {
        const float st = igm_sin(a);
        const Vec3 dir = make_vec3(st * igm_cos(b), st * igm_sin(b), igm_cos(a));
        // sphere_map_uv applies spherical_from_dir to (dir.y, -dir.x, dir.z): hand it the direction whose swizzle is `dir`
        float u, v;
        sphere_map_uv(make_vec3(-dir.y, dir.x, dir.z), u, v);
        out[0] = v * flt_pi, out[1] = u * (2 * flt_pi);
    }
}

// interval::binary_search (core/interval.art:7-23) over an integer array: mode 0: pred(i) = arr[i] <= value, 1: arr[i] < value
int32_t oracle_interval_search(const int32_t* arr, int32_t size, int32_t value, int32_t mode)
{
    return interval_binary_search(size, [&](int32_t i) { return mode == 0 ? arr[i] <= value : arr[i] < value; });
}

// CDF probes for the tests (core/cdf.art). `data` omits the leading zero, as the device buffers do.
// mode 0: sample_discrete(u) -> off, pdf; 1: sample_continuous(u) -> off, pos, pdf; 2: pdf_continuous(x = u) -> off, pdf
void oracle_cdf1d(const float* data, int32_t func_size, int32_t mode, float u, int32_t* off, float* pos, float* pdf)
{
    const Cdf1D cdf{ data, func_size };
    *pos = 0;
    if (mode == 0)
        *off = cdf.sample_discrete(u, *pdf);
    else if (mode == 1)
        *pos = cdf.sample_continuous(u, *off, *pdf);
    else
        *pdf = cdf.pdf_continuous(u, *off);
}

// 2D: table = marginal (size_y) then conditionals (size_x per row). mode 1: sample_continuous(u) -> pos, pdf; 2: pdf_continuous(pos = u)
void oracle_cdf2d(const float* data, int32_t size_x, int32_t size_y, int32_t mode, const float u[2], float pos[2], float* pdf)
{
    const Cdf2D cdf{ data, size_x, size_y };
    if (mode == 1) {
        const Vec2 p = cdf.sample_continuous(u[0], u[1], *pdf);
        pos[0] = p.x, pos[1] = p.y;
    } else {
        *pdf = cdf.pdf_continuous(Vec2{ u[0], u[1] });
    }
}

int oracle_hardware_threads(void) { return (int)std::thread::hardware_concurrency(); }

} // extern "C"
