// oracle_shade.h — TEST INFRASTRUCTURE ONLY.
//
// CPU restatement of the reference's shading side of the hot path: RNG, camera
// ray generation, surface reconstruction, BSDFs, lights, the path tracer
// callbacks. Each function cites the reference file:line it follows.
#pragma once

#include "oracle_core.h"
#include "ig_expr.h"
#include "ig_photon.h"

#include <algorithm>
#include <cstring>
#include <functional>
#include <limits>

namespace oracle {

// ---- core/random.art
static inline uint32_t hash_init() { return 0x811C9DC5u; } // random.art:4
// random.art:7-13 (FNV-1a, bytewise)
static inline uint32_t hash_combine(uint32_t h, uint32_t d)
{
    h = (h * 16777619u) ^ (d & 0xFF);
    h = (h * 16777619u) ^ ((d >> 8) & 0xFF);
    h = (h * 16777619u) ^ ((d >> 16) & 0xFF);
    h = (h * 16777619u) ^ ((d >> 24) & 0xFF);
    return h;
}
// random.art:15-24
static inline uint32_t sample_tea_u32(uint32_t v0, uint32_t v1)
{
    uint32_t sum = 0;
    for (int i = 0; i < 4; ++i) {
        sum += 0x9e3779b9u;
        v0 += ((v1 << 4) + 0xa341316cu) ^ (v1 + sum) ^ ((v1 >> 5) + 0xc8013ea4u);
        v1 += ((v0 << 4) + 0xad90777du) ^ (v0 + sum) ^ ((v0 >> 5) + 0x7e95761eu);
    }
    return v1;
}
// random.art:34-43
static inline uint32_t create_random_seed(int32_t sample, int32_t iter, int32_t frame, int32_t x, int32_t y, int32_t user)
{
    uint32_t hash = hash_init();
    hash          = hash_combine(hash, (uint32_t)sample);
    hash          = hash_combine(hash, (uint32_t)iter);
    hash          = hash_combine(hash, (uint32_t)frame);
    hash          = hash_combine(hash, (uint32_t)x);
    hash          = hash_combine(hash, (uint32_t)y);
    hash          = hash_combine(hash, (uint32_t)user);
    return hash;
}

// random.art:45-87
struct Rng {
    uint32_t seed;
    uint32_t counter;
    uint32_t next_u32() { return sample_tea_u32(seed, counter++); }
    float next_f32()
    {
        const uint32_t x = next_u32();
        return igm_float((x & 0x7FFFFFu) | 0x3F800000u) - 1;
    }
    // next_i32(s, e), inclusive range, rejection sampling (random.art:46-61)
    int32_t next_i32(int32_t s, int32_t e)
    {
        const uint32_t range     = (uint32_t)(e - s);
        const uint32_t rng_range = 0xFFFFFFFFu;
        if (rng_range == range)
            return (int32_t)next_u32() + s;
        const uint32_t erange  = range + 1;
        const uint32_t scaling = rng_range / erange;
        const uint32_t past    = erange * scaling;
        uint32_t ret           = next_u32();
        while (ret >= past)
            ret = next_u32();
        return (int32_t)(ret / scaling) + s;
    }
};

// ---- camera/{perspective,orthogonal,fishlens}.art + driver/camera.art + driver/emitter.art
struct CameraSetup {
    Vec3 eye, dir;
    Mat3x3 view; // right, up, dir
    float sx, sy; // perspective / orthogonal: scale; fishlens: (xasp, yasp)
    float tmin, tmax;
    int type;
    bool mask;
    float aperture_radius, focal_length;
    int pixel_sampler;
};

// ---- sampler/pixel_sampler.art
// permute_element (core/common.art:302-335): Kensler's hashed permutation of [0, l)
static inline uint32_t permute_element(uint32_t i, uint32_t l, uint32_t seed)
{
    uint32_t w = l - 1;
    if (w == 0)
        return 0;
    w |= w >> 1;
    w |= w >> 2;
    w |= w >> 4;
    w |= w >> 8;
    w |= w >> 16;
    do {
        i ^= seed;
        i *= 0xe170893du;
        i ^= seed >> 16;
        i ^= (i & w) >> 4;
        i ^= seed >> 8;
        i *= 0x0929eb3fu;
        i ^= seed >> 23;
        i ^= (i & w) >> 1;
        i *= 1 | seed >> 27;
        i *= 0x6935fa69u;
        i ^= (i & w) >> 11;
        i *= 0x74dcb303u;
        i ^= (i & w) >> 2;
        i *= 0x9e501cc3u;
        i ^= (i & w) >> 2;
        i *= 0xc860a3dfu;
        i &= w;
        i ^= i >> 5;
    } while (i >= l);
    return (i + seed) % l;
}

// radical_inverse (pixel_sampler.art:37-54)
static inline float radical_inverse(uint32_t index, uint32_t base)
{
    const uint32_t limit  = 0xFFFFFFFFu / base - base;
    const float inv_base  = 1.0f / (float)base;
    float inv_base_n      = 1;
    uint32_t reversed     = 0;
    while (index != 0 && reversed < limit) {
        const uint32_t next  = index / base;
        const uint32_t digit = index - next * base;
        reversed             = reversed * base + digit;
        inv_base_n *= inv_base;
        index = next;
    }
    return igm_min((float)reversed * inv_base_n, 1 - flt_eps);
}

// inverse_radical_inverse (:56-64)
static inline uint32_t inverse_radical_inverse(uint32_t inv, uint32_t base, uint32_t digits)
{
    uint32_t index = 0;
    for (uint32_t i = 0; i < digits; ++i) {
        const uint32_t digit = inv % base;
        inv /= base;
        index = index * base + digit;
    }
    return index;
}

// extended_gcd / multiplicative_inverse (:77-90): the remainder is the signed one, as written
static inline void extended_gcd(uint32_t a, uint32_t b, int32_t& x, int32_t& y)
{
    if (b == 0) {
        x = 1, y = 0;
        return;
    }
    const int32_t d = (int32_t)(a / b);
    int32_t xx, yy;
    extended_gcd(b, a % b, xx, yy);
    x = yy;
    y = (int32_t)((uint32_t)xx - (uint32_t)d * (uint32_t)yy);
}

struct HaltonSetup { // :92-99 (bases 2 and 3), without the buffer: halton_offset() recomputes what it holds
    uint32_t scale_x, scale_y, exp_x, exp_y;
    int32_t mul_inv_x, mul_inv_y;
};

static inline HaltonSetup setup_halton(int width, int height)
{
    HaltonSetup h;
    // compute_halton_base_info (:66-75)
    h.scale_x = 1, h.exp_x = 0;
    while (h.scale_x < (uint32_t)width)
        h.scale_x *= 2, ++h.exp_x;
    h.scale_y = 1, h.exp_y = 0;
    while (h.scale_y < (uint32_t)height)
        h.scale_y *= 3, ++h.exp_y;
    int32_t x, y;
    extended_gcd(h.scale_x, h.scale_y, x, y);
    h.mul_inv_x = x % (int32_t)h.scale_y;
    extended_gcd(h.scale_y, h.scale_x, x, y);
    h.mul_inv_y = x % (int32_t)h.scale_x;
    return h;
}

// the value setup_halton_pixel_sampler stores in "__halton_offset" for pixel (x, y) (:127-140); i32 arithmetic wraps
static inline int32_t halton_offset(const HaltonSetup& h, int x, int y)
{
    const uint32_t stride = h.scale_x * h.scale_y;
    if (stride <= 1)
        return 0;
    const uint32_t dx = inverse_radical_inverse((uint32_t)x, 2, h.exp_x);
    const uint32_t dy = inverse_radical_inverse((uint32_t)y, 3, h.exp_y);
    const uint32_t a  = (dx * (stride / h.scale_x)) * (uint32_t)h.mul_inv_x;
    const uint32_t b  = (dy * (stride / h.scale_y)) * (uint32_t)h.mul_inv_y;
    return (int32_t)(a + b) % (int32_t)stride;
}

// PixelSampler (:1): position of sample `index` (= iter * spi + sample, emitter.art:9) inside pixel (x, y)
static inline void pixel_sample(int sampler, Rng& rnd, int index, int x, int y, int width, int height, float& rx, float& ry)
{
    if (sampler == IG_PIXEL_SAMPLER_MJITT) {
        // make_mjitt_pixel_sampler(4, 4) (:13-34)
        const uint32_t bin_x = 4, bin_y = 4;
        const uint32_t seed  = hash_combine(hash_combine(0x811C9DC5u, (uint32_t)x), (uint32_t)y);
        const uint32_t idx   = (uint32_t)index;
        const float sx       = (float)permute_element(idx % bin_x, bin_x, seed * 0xa511e9b3u);
        const float sy       = (float)permute_element(idx / bin_x, bin_y, seed * 0x63d83595u);
        const float jx       = rnd.next_f32();
        const float jy       = rnd.next_f32();
        rx                   = (sx + (sy + jx) / (float)bin_y) / (float)bin_x;
        ry                   = (sy + (sx + jy) / (float)bin_x) / (float)bin_y;
    } else if (sampler == IG_PIXEL_SAMPLER_HALTON) {
        // make_halton_pixel_sampler (:152-167); the generator is not advanced
        const HaltonSetup h   = setup_halton(width, height);
        const uint32_t stride = h.scale_x * h.scale_y;
        const uint32_t hindex = (uint32_t)(halton_offset(h, x, y) + (int32_t)((uint32_t)index * stride));
        rx                    = radical_inverse(hindex >> h.exp_x, 2);
        ry                    = radical_inverse(hindex / h.scale_y, 3);
    } else {
        // make_uniform_pixel_sampler (:4-10)
        rx = rnd.next_f32();
        ry = rnd.next_f32();
    }
}

// make_perspective_camera (perspective.art:29-42), make_perspective_dof_camera (:69-84),
// make_orthogonal_camera (orthogonal.art:14-26), make_fishlens_camera (fishlens.art:8-37).
// (sx, sy) comes from the host (camera_scale in oracle.cpp), evaluated with libm.
static inline CameraSetup make_camera(const ig_camera& c, float sx, float sy)
{
    CameraSetup s;
    s.eye          = Vec3{ c.eye[0], c.eye[1], c.eye[2] };
    const Vec3 dir = Vec3{ c.dir[0], c.dir[1], c.dir[2] };
    const Vec3 up  = Vec3{ c.up[0], c.up[1], c.up[2] };
    s.dir          = dir;
    s.view.col[0]  = vec3_normalize(vec3_cross(dir, up));
    s.view.col[1]  = up;
    s.view.col[2]  = dir;
    s.sx           = sx;
    s.sy           = sy;
    s.tmin         = c.near_clip;
    s.tmax         = c.far_clip;
    s.type         = c.type;
    s.mask         = c.fisheye_mask != 0;
    s.aperture_radius = c.aperture_radius;
    s.focal_length    = c.focal_length;
    s.pixel_sampler   = c.pixel_sampler;
    return s;
}

// square_to_concentric_disk (core/warp.art:2-22)
static inline void square_to_concentric_disk(float px, float py, float& ox, float& oy)
{
    const float a = 2 * px - 1;
    const float b = 2 * py - 1;
    if (a == 0 && b == 0) {
        ox = 0, oy = 0;
    } else if (a * a > b * b) {
        const float phi = (flt_pi / 4) * safe_div(b, a);
        ox = igm_cos(phi) * a;
        oy = igm_sin(phi) * a;
    } else {
        const float phi = (flt_pi / 2) - (flt_pi / 4) * safe_div(a, b);
        ox = igm_cos(phi) * b;
        oy = igm_sin(phi) * b;
    }
}

// make_camera_emitter (emitter.art:6-16) + the pixel sampler (pixel_sampler.art)
// + make_pixelcoord_from_xy (camera.art:21-29) + Camera::generate_ray. Returns false when the camera
// yields no ray for the sample (masked fishlens, fishlens.art:44): cpu_generate_rays then stores a zero
// ray with id -1 (mapping_cpu.art:352-355). The reference goes on to miss-shade that entry and splats
// the result to pixel -1 / spi; the restatement (and the HIP device) drop the sample instead.
static inline bool generate_camera_ray(const CameraSetup& cam, Rng& rnd, int index, int x, int y, int w, int h, Ray& out)
{
    float rx, ry;
    pixel_sample(cam.pixel_sampler, rnd, index, x, y, w, h, rx, ry);
    const float nx = 2 * ((float)x + rx) / ((float)w) - 1;
    const float ny = 1 - 2 * ((float)y + ry) / ((float)h);
    if (cam.type == IG_CAMERA_ORTHOGONAL) {
        // orthogonal.art:19-22
        const Vec3 pos = vec3_add(mat3x3_mul(cam.view, make_vec3(cam.sx * nx, cam.sy * ny, 0)), cam.eye);
        out            = make_ray(pos, cam.dir, cam.tmin, cam.tmax, IG_RAY_FLAG_CAMERA);
        return true;
    }
    if (cam.type == IG_CAMERA_FISHLENS) {
        // compute_d (fishlens.art:39-53), fov = pi
        const float fx    = nx * cam.sx;
        const float fy    = ny * cam.sy;
        const float r     = igm_sqrt(fx * fx + fy * fy);
        const float theta = r * flt_pi / 2;
        if (cam.mask && r > 1)
            return false;
        const float sT = igm_sin(theta);
        const float cT = igm_cos(theta);
        const float sP = r < flt_eps ? 0 : fy / r;
        const float cP = r < flt_eps ? 0 : fx / r;
        out            = make_ray(cam.eye, mat3x3_mul(cam.view, make_vec3(sT * cP, sT * sP, cT)), cam.tmin, cam.tmax, IG_RAY_FLAG_CAMERA);
        return true;
    }
    const Vec3 d = vec3_normalize(mat3x3_mul(cam.view, make_vec3(cam.sx * nx, cam.sy * ny, 1)));
    if (cam.aperture_radius > flt_eps) {
        // gen_ray of make_perspective_dof_camera (perspective.art:73-84; chosen at PerspectiveCamera.cpp:50)
        const Vec3 focus_pos = vec3_mulf(d, cam.focal_length);
        const float u0       = rnd.next_f32();
        const float u1       = rnd.next_f32();
        float ax, ay;
        square_to_concentric_disk(u0, u1, ax, ay);
        const Vec3 ap = mat3x3_mul(cam.view, make_vec3(ax * cam.aperture_radius, ay * cam.aperture_radius, 0));
        out           = make_ray(vec3_add(cam.eye, ap), vec3_normalize(vec3_sub(focus_pos, ap)), cam.tmin, cam.tmax, IG_RAY_FLAG_CAMERA);
        return true;
    }
    out = make_ray(cam.eye, d, cam.tmin, cam.tmax, IG_RAY_FLAG_CAMERA);
    return true;
}

// ---- driver/entity.art:12-28
struct Entity {
    Mat3x4 local_mat, global_mat;
    Mat3x3 normal_mat;
    int32_t shape_id, mat_id;
};

static inline Entity load_entity(const igd_scene& sc, int32_t id)
{
    const float* d = sc.entities + (size_t)id * IG_ENTITY_FLOATS;
    Entity e;
    for (int c = 0; c < 4; ++c) {
        e.local_mat.col[c]  = Vec3{ d[c * 3], d[c * 3 + 1], d[c * 3 + 2] };
        e.global_mat.col[c] = Vec3{ d[12 + c * 3], d[12 + c * 3 + 1], d[12 + c * 3 + 2] };
    }
    for (int c = 0; c < 3; ++c)
        e.normal_mat.col[c] = Vec3{ d[24 + c * 3], d[24 + c * 3 + 1], d[24 + c * 3 + 2] };
    std::memcpy(&e.shape_id, d + 33, 4);
    std::memcpy(&e.mat_id, d + 34, 4);
    return e;
}

// ---- shapes/trimesh.art:76-103 (make_trimesh_from_buffer / load_trimesh)
struct TriMeshView {
    const float* vertices;  // 4 floats each
    const float* normals;   // 4 floats each
    const int32_t* indices; // 4 ints each
    const float* texcoords; // 2 floats each
    int32_t num_tris;
};

static inline TriMeshView load_trimesh(const igd_scene& sc, int32_t shape_id)
{
    const uint8_t* base = sc.shape_data + sc.shape_lookups[shape_id].offset;
    int32_t hdr[4];
    std::memcpy(hdr, base, 16);
    const float* f = reinterpret_cast<const float*>(base);
    TriMeshView m;
    const int v_start   = 12;
    const int n_start   = v_start + hdr[1] * 4;
    const int ind_start = n_start + hdr[2] * 4;
    const int tex_start = ind_start + hdr[0] * 4;
    m.vertices  = f + v_start;
    m.normals   = f + n_start;
    m.indices   = reinterpret_cast<const int32_t*>(f + ind_start);
    m.texcoords = f + tex_start;
    m.num_tris  = hdr[0];
    return m;
}

// ---- driver/surface_element.art
struct SurfaceElement {
    bool is_entering;
    Vec3 point, face_normal;
    float area, inv_area;
    Vec2 prim_coords, tex_coords;
    Mat3x3 local;
};

// core/triangle.art:31-44
static inline Vec3 compute_stable_triangle_normal(Vec3 e1, Vec3 e2, Vec3 e3)
{
    const float x12 = e1.z * e2.y, y12 = e1.x * e2.z, z12 = e1.y * e2.x;
    const float x23 = e2.z * e3.y, y23 = e2.x * e3.z, z23 = e2.y * e3.x;
    const Vec3 c12  = make_vec3(e1.y * e2.z - x12, e1.z * e2.x - y12, e1.x * e2.y - z12);
    const Vec3 c23  = make_vec3(e2.y * e3.z - x23, e2.z * e3.x - y23, e2.x * e3.y - z23);
    return make_vec3(igm_abs(x12) < igm_abs(x23) ? c12.x : c23.x,
                     igm_abs(y12) < igm_abs(y23) ? c12.y : c23.y,
                     igm_abs(z12) < igm_abs(z23) ? c12.z : c23.z);
}

// make_trimesh_shape.surface_element (trimesh.art:14-40) with
// make_standard_pointmapperset (pointmapper.art:28-36) and make_triangle (triangle.art:12-29)
// ---- shapes/sphere.art
struct Sphere {
    Vec3 origin;
    float radius;
};
static inline Sphere load_sphere(const igd_scene& sc, int32_t shape_id) // sphere.art:96-103
{
    const float* f = reinterpret_cast<const float*>(sc.shape_data + sc.shape_lookups[shape_id].offset);
    return Sphere{ make_vec3(f[0], f[1], f[2]), f[3] };
}
// sphere_unmap_uv (sphere.art:8-13) over dir_from_spherical (theta from +z, phi in the xy plane)
static inline Vec3 sphere_unmap_uv(Vec2 uv)
{
    const float theta = uv.y * flt_pi;
    const float phi   = uv.x * 2 * flt_pi;
    const float st    = igm_sin(theta);
    const Vec3 dir    = make_vec3(st * igm_cos(phi), st * igm_sin(phi), igm_cos(theta));
    return make_vec3(dir.y, -dir.x, dir.z);
}
// sphere_compute_surface_element_for_normal (sphere.art:30-46); `area` = compute_ellipsoid_area, evaluated by the loader
static inline void sphere_surface_for_normal(const Entity& entity, const Sphere& sphere, Vec3 normal, Vec3& point, Vec3& face_normal)
{
    const Vec3 p = vec3_add(sphere.origin, vec3_mulf(normal, sphere.radius));
    face_normal  = vec3_normalize(mat3x3_mul(entity.normal_mat, normal));
    point        = mat3x4_transform_point(entity.global_mat, p);
}

// make_sphere_shape.surface_element (sphere.art:52-73). The area is only read by area lights, which carry their own.
static inline SurfaceElement sphere_surface_element(const igd_scene& sc, const Entity& entity, const Ray& ray, const Hit& hit)
{
    const Sphere sphere = load_sphere(sc, entity.shape_id);
    SurfaceElement s;
    s.point          = vec3_add(ray.org, vec3_mulf(ray.dir, hit.distance));
    const Vec3 dir   = vec3_sub(s.point, mat3x4_transform_point(entity.global_mat, sphere.origin));
    const float len  = vec3_len(dir);
    const Vec3 n     = vec3_mulf(dir, 1 / len);
    s.is_entering    = true;
    s.face_normal    = n;
    s.area           = 0;
    s.inv_area       = 0;
    s.prim_coords    = Vec2{ hit.u, hit.v };
    s.tex_coords     = s.prim_coords;
    s.local          = make_orthonormal_mat3x3(n);
    return s;
}

static inline SurfaceElement surface_element(const igd_scene& sc, const Entity& entity, const Ray& ray, const Hit& hit)
{
    if (sc.shape_lookups[entity.shape_id].type_id == IG_SHAPE_SPHERE)
        return sphere_surface_element(sc, entity, ray, hit);
    const TriMeshView mesh = load_trimesh(sc, entity.shape_id);
    const int32_t i0 = mesh.indices[hit.prim_id * 4 + 0], i1 = mesh.indices[hit.prim_id * 4 + 1], i2 = mesh.indices[hit.prim_id * 4 + 2];
    auto vtx = [&](int32_t i) { return Vec3{ mesh.vertices[i * 4], mesh.vertices[i * 4 + 1], mesh.vertices[i * 4 + 2] }; };
    auto nrm = [&](int32_t i) { return Vec3{ mesh.normals[i * 4], mesh.normals[i * 4 + 1], mesh.normals[i * 4 + 2] }; };
    auto tex = [&](int32_t i) { return Vec2{ mesh.texcoords[i * 2], mesh.texcoords[i * 2 + 1] }; };

    const Vec3 v0 = mat3x4_transform_point(entity.global_mat, vtx(i0));
    const Vec3 v1 = mat3x4_transform_point(entity.global_mat, vtx(i1));
    const Vec3 v2 = mat3x4_transform_point(entity.global_mat, vtx(i2));
    const Vec3 e1 = vec3_sub(v2, v0);
    const Vec3 e2 = vec3_sub(v0, v1);
    const Vec3 e3 = vec3_sub(v1, v2);
    const Vec3 n  = compute_stable_triangle_normal(e1, e2, e3);
    const float nn          = vec3_len(n);
    const Vec3 face_normal  = vec3_mulf(n, 1 / nn);
    const float area        = nn / 2;
    const Vec3 normal       = vec3_normalize(mat3x3_mul(entity.normal_mat, vec3_lerp2(nrm(i0), nrm(i1), nrm(i2), hit.u, hit.v)));
    const bool is_entering  = vec3_dot(ray.dir, face_normal) <= 0;
    const Vec2 tex_coords   = vec2_lerp2(tex(i0), tex(i1), tex(i2), hit.u, hit.v);

    SurfaceElement s;
    s.is_entering = is_entering;
    s.point       = vec3_add(ray.org, vec3_mulf(ray.dir, hit.distance));
    s.face_normal = is_entering ? face_normal : vec3_neg(face_normal);
    s.area        = area;
    s.inv_area    = safe_div(1, area);
    s.prim_coords = Vec2{ hit.u, hit.v };
    s.tex_coords  = tex_coords;
    s.local       = make_orthonormal_mat3x3(is_entering ? normal : vec3_neg(normal));
    return s;
}

// ---- colours (core/color.art:14-36)
static inline Color color_mul(Color a, Color b) { return Color{ a.r * b.r, a.g * b.g, a.b * b.b }; }
static inline Color color_mulf(Color c, float f) { return Color{ c.r * f, c.g * f, c.b * f }; }
static inline Color color_add(Color a, Color b) { return Color{ a.r + b.r, a.g + b.g, a.b + b.b }; }
static inline float color_average(Color c) { return (c.r + c.g + c.b) / 3; }
static inline float color_max_component(Color c) { return igm_max(c.r, igm_max(c.g, c.b)); } // vec3_max_value, vector.art:38,113
static inline Color color_saturate(Color a, float f) { return Color{ igm_min(a.r, f), igm_min(a.g, f), igm_min(a.b, f) }; }

// ---- core/sampling.art:12-20,62-70
struct DirSample {
    Vec3 dir;
    float pdf;
};
static inline float cosine_hemisphere_pdf(float c) { return c / flt_pi; }
static inline DirSample sample_cosine_hemisphere(float u, float v)
{
    const float c   = safe_sqrt(v);
    const float s   = safe_sqrt(1 - v);
    const float phi = 2 * flt_pi * u;
    DirSample d;
    d.dir = make_vec3(s * igm_cos(phi), s * igm_sin(phi), c);
    d.pdf = cosine_hemisphere_pdf(c);
    return d;
}
static inline float positive_cos(Vec3 a, Vec3 b)
{
    const float c = vec3_dot(a, b);
    return c >= 0 ? c : 0.0f;
} // common.art:292-295

// ---- core/fresnel.art:7-27
static inline float fresnel_factor(float eta, float cos_i, float cos_t)
{
    const float R_s = safe_div(eta * cos_i - cos_t, eta * cos_i + cos_t);
    const float R_p = safe_div(cos_i - eta * cos_t, cos_i + eta * cos_t);
    return clampf((R_s * R_s + R_p * R_p) * 0.5f, 0, 1);
}
static inline float snell(float eta, float cos_i) { return 1 - (1 - cos_i * cos_i) * eta * eta; }
static inline bool fresnel(float eta, float cos_i, float& cos_t_out, float& factor)
{
    const float eta2   = cos_i < 0 ? 1 / eta : eta;
    const float cos2_t = snell(eta2, cos_i);
    if (cos2_t <= 0.0f)
        return false;
    const float cos_t = igm_sqrt(cos2_t);
    cos_t_out         = cos_i < 0 ? -cos_t : cos_t;
    factor            = fresnel_factor(eta2, igm_abs(cos_i), cos_t);
    return true;
}

// ---- core/fresnel.art:29-36
static inline float conductor_factor(float n, float k, float cos_i)
{
    const float f   = n * n + k * k;
    const float d1  = f * cos_i * cos_i;
    const float d2  = 2.0f * n * cos_i;
    const float R_s = safe_div(d1 - d2, d1 + d2);
    const float R_p = safe_div(f - d2 + cos_i * cos_i, f + d2 + cos_i * cos_i);
    return clampf((R_s * R_s + R_p * R_p) * 0.5f, 0, 1);
}

static inline float absolute_cos(Vec3 a, Vec3 b) { return igm_abs(vec3_dot(a, b)); } // common.art:302
static inline Vec3 vec3_halfway(Vec3 a, Vec3 b) { return vec3_normalize(vec3_add(a, b)); } // vector.art:142

// ---- GGX microfacet model (core/microfacet.art:158-199) and the VNDF sampler of Dupuy & Benyoub
// (microfacet.art:372-404), distribution make_vndf_ggx_distribution (:404-425)
struct GGX {
    Mat3x3 local;
    float alpha_u, alpha_v;

    float D(Vec3 m) const // ndf_ggx
    {
        const float cosZ = vec3_dot(local.col[2], m);
        const float cosX = vec3_dot(local.col[0], m);
        const float cosY = vec3_dot(local.col[1], m);
        const float kx   = cosX / alpha_u;
        const float ky   = cosY / alpha_v;
        const float k    = kx * kx + ky * ky + cosZ * cosZ;
        return safe_div(1, flt_pi * alpha_u * alpha_v * k * k);
    }
    float G1(Vec3 w) const // g_1_smith
    {
        const float cosZ = vec3_dot(local.col[2], w);
        if (igm_abs(cosZ) <= flt_eps)
            return 0;
        const float cosX = vec3_dot(local.col[0], w);
        const float cosY = vec3_dot(local.col[1], w);
        const float kx   = alpha_u * cosX;
        const float ky   = alpha_v * cosY;
        const float a2   = kx * kx + ky * ky;
        if (a2 <= flt_eps)
            return 1;
        const float k2    = a2 / (cosZ * cosZ);
        const float denom = 1 + igm_sqrt(1 + k2);
        return 2 / denom;
    }
    float pdf(Vec3 w, Vec3 h) const // pdf_vndf_ggx
    {
        const float cosZ = absolute_cos(local.col[2], w);
        return safe_div(G1(w) * absolute_cos(w, h) * D(h), cosZ);
    }
    Vec3 sample(Rng& rnd, Vec3 vN) const // sample_vndf_ggx
    {
        const Vec3 vL = make_vec3(vec3_dot(local.col[0], vN), vec3_dot(local.col[1], vN), vec3_dot(local.col[2], vN)); // shading::to_local
        const Vec3 sL = vec3_normalize(make_vec3(alpha_u * vL.x, alpha_v * vL.y, vL.z));
        const float u0 = rnd.next_f32();
        const float u1 = rnd.next_f32();
        const float phi      = 2 * flt_pi * u0;
        const float z        = (1 - u1) * (1 + sL.z) - sL.z;
        const float sinTheta = igm_sqrt(clampf(1 - z * z, 0, 1));
        const float x        = sinTheta * igm_cos(phi);
        const float y        = sinTheta * igm_sin(phi);
        const Vec3 h         = vec3_add(make_vec3(x, y, z), vL);
        const Vec3 Nh        = vec3_normalize(make_vec3(h.x * alpha_u, h.y * alpha_v, h.z));
        // shading::to_world
        return vec3_add(vec3_add(vec3_mulf(local.col[0], Nh.x), vec3_mulf(local.col[1], Nh.y)), vec3_mulf(local.col[2], Nh.z));
    }
};

// make_checkerboard_texture with the identity transform (texture/checkerboard.art, core/math.art:88-91)
static inline float math_wrap(float v, float mn, float mx)
{
    const float range = mx - mn;
    return range <= flt_eps ? mn : v - (range * igm_floor((v - mn) / range));
}
static inline Color checkerboard(const ig_material& mat, Vec2 uv)
{
    const float sx      = uv.x * mat.q[6];
    const float sy      = uv.y * mat.q[7];
    const bool parity_x = ((int32_t)math_wrap(sx, 0, 2) % 2) == 0;
    const bool parity_y = ((int32_t)math_wrap(sy, 0, 2) % 2) == 0;
    return (parity_x ^ parity_y) ? Color{ mat.q[0], mat.q[1], mat.q[2] } : Color{ mat.q[3], mat.q[4], mat.q[5] };
}

// ---- bitmap textures (texture/image.art, driver/image.art:9-16, mapping_cpu.art:980-992)
static inline Color image_pixel(const igd_scene& sc, const ig_texture& t, int32_t x, int32_t y)
{
    const uint8_t* base = sc.texture_data + t.offset;
    if (t.channels & IG_TEX_FLOAT_BIT) {
        // unpacked images (driver/image.art:1-7: the floats device.load_image returns, one or four per pixel)
        const uint32_t nc = t.channels & 0xFFu;
        float c[4];
        std::memcpy(c, base + sizeof(float) * nc * (size_t)(y * (int32_t)t.width + x), sizeof(float) * nc);
        return nc == 1 ? Color{ c[0], c[0], c[0] } : Color{ c[0], c[1], c[2] };
    }
    if (t.channels == 1) {
        const float g = (float)base[y * (int32_t)t.width + x] / 255; // image_mono_unpack
        return Color{ g, g, g };
    }
    uint32_t packed;
    std::memcpy(&packed, base + 4 * (size_t)(y * (int32_t)t.width + x), 4);
    // image_rgba_unpack (alpha is not part of Color here)
    return Color{ (float)(packed & 0x000000FFu) / 255, (float)((packed & 0x0000FF00u) >> 8) / 255, (float)((packed & 0x00FF0000u) >> 16) / 255 };
}
static inline int32_t image_border(uint32_t mode, int32_t x, int32_t w)
{
    if (mode == IG_WRAP_CLAMP) // image.art:9-15
        return x < 0 ? 0 : (x > w - 1 ? w - 1 : x);
    if (mode == IG_WRAP_MIRROR) { // image.art:28-40
        const int32_t t = x < 0 ? -1 - x : x;
        const int32_t i = t / w;
        const int32_t k = t - i * w;
        return (i & 1) == 0 ? w - 1 - k : k;
    }
    const int32_t t = x % w; // image.art:17-26
    return t < 0 ? t + w : t;
}
static inline Color color_lerp(Color a, Color b, float t) // color.art:17-21
{
    return Color{ (1 - t) * a.r + t * b.r, (1 - t) * a.g + t * b.g, (1 - t) * a.b + t * b.b };
}
// make_image_texture with an identity transform (image.art:158-163) over the three filters (image.art:85-156)
static inline Color image_lookup(const igd_scene& sc, const ig_texture& t, Vec2 uv)
{
    const int32_t W = (int32_t)t.width, H = (int32_t)t.height;
    auto px = [&](int32_t x, int32_t y) { return image_pixel(sc, t, image_border(t.wrap_u, x, W), image_border(t.wrap_v, y, H)); };
    if (t.filter == IG_TEX_NEAREST) {
        const float u = uv.x * (float)W, v = uv.y * (float)H; // map_uv_to_image_pixel_nearest
        return px((int32_t)igm_floor(u), (int32_t)igm_floor(v));
    }
    const float u    = uv.x * (float)W - 0.5f; // map_uv_to_image_pixel
    const float v    = uv.y * (float)H - 0.5f;
    const int32_t ix = (int32_t)igm_floor(u), iy = (int32_t)igm_floor(v);
    const float fx = u - igm_floor(u), fy = v - igm_floor(v); // math::fract
    if (t.filter == IG_TEX_BILINEAR)
        return color_lerp(color_lerp(px(ix, iy), px(ix + 1, iy), fx), color_lerp(px(ix, iy + 1), px(ix + 1, iy + 1), fx), fy);
    // bicubic B-spline through two bilinear-like taps per axis (image.art:105-156)
    auto w0 = [](float a) { return (a * (a * (-a + 3) - 3) + 1) / 6; };
    auto w1 = [](float a) { return (a * a * (3 * a - 6) + 4) / 6; };
    auto w2 = [](float a) { return (a * (a * (-3 * a + 3) + 3) + 1) / 6; };
    auto w3 = [](float a) { return (a * a * a) / 6; };
    auto g0 = [&](float a) { return w0(a) + w1(a); };
    auto g1 = [&](float a) { return w2(a) + w3(a); };
    auto h0 = [&](float a) { return (w1(a) / g0(a)) - 1; };
    auto h1 = [&](float a) { return (w3(a) / g1(a)) + 1; };
    const float g0x = g0(fx), g0y = g0(fy), g1x = g1(fx), g1y = g1(fy);
    const int32_t ix0 = (int32_t)igm_floor((float)ix + h0(fx) + 0.5f);
    const int32_t iy0 = (int32_t)igm_floor((float)iy + h0(fy) + 0.5f);
    const int32_t ix1 = (int32_t)igm_floor((float)ix + h1(fx) + 0.5f);
    const int32_t iy1 = (int32_t)igm_floor((float)iy + h1(fy) + 0.5f);
    const Color p00 = color_mulf(px(ix0, iy0), g0x * g0y);
    const Color p10 = color_mulf(px(ix1, iy0), g1x * g0y);
    const Color p01 = color_mulf(px(ix0, iy1), g0x * g1y);
    const Color p11 = color_mulf(px(ix1, iy1), g1x * g1y);
    const Color a{ p00.r + p10.r, p00.g + p10.g, p00.b + p10.b }, c{ p01.r + p11.r, p01.g + p11.g, p01.b + p11.b };
    return Color{ a.r + c.r, a.g + c.g, a.b + c.b };
}

// ---- bump mapping (bsdf/map.art:36-42,64-67; MapBSDF.cpp:44-47; core/sampling.art:118-166; core/matrix.art:261-284)
static inline Vec3 ensure_valid_reflection(Vec3 Ng, Vec3 I, Vec3 N)
{
    const Vec3 R          = vec3_reflect(I, N);
    const float threshold = igm_min(0.9f * vec3_dot(Ng, I), 0.01f);
    if (vec3_dot(Ng, R) >= threshold)
        return N;
    const float NdotNg = vec3_dot(N, Ng);
    const Vec3 X       = vec3_normalize(vec3_sub(N, vec3_mulf(Ng, NdotNg)));
    const float Ix = vec3_dot(I, X), Iz = vec3_dot(I, Ng);
    const float Ix2 = Ix * Ix, Iz2 = Iz * Iz;
    const float a   = Ix2 + Iz2;
    const float b   = safe_sqrt(Ix2 * (a - threshold * threshold));
    const float c   = Iz * threshold + a;
    const float fac = 0.5f / a;
    const float N1_z2 = fac * (b + c), N2_z2 = fac * (-b + c);
    const bool valid1 = (N1_z2 > 1e-5f) && (N1_z2 <= (1.0f + 1e-5f));
    const bool valid2 = (N2_z2 > 1e-5f) && (N2_z2 <= (1.0f + 1e-5f));
    Vec2 Nn;
    if (valid1 && valid2) {
        const Vec2 N1{ safe_sqrt(1 - N1_z2), safe_sqrt(N1_z2) };
        const Vec2 N2{ safe_sqrt(1 - N2_z2), safe_sqrt(N2_z2) };
        const float R1 = 2 * (N1.x * Ix + N1.y * Iz) * N1.y - Iz;
        const float R2 = 2 * (N2.x * Ix + N2.y * Iz) * N2.y - Iz;
        const bool valid3 = R1 >= 1e-5f, valid4 = R2 >= 1e-5f;
        if (valid3 && valid4)
            Nn = R1 < R2 ? N1 : N2;
        else
            Nn = R1 > R2 ? N1 : N2;
    } else if (valid1 || valid2) {
        const float Nz2 = valid1 ? N1_z2 : N2_z2;
        Nn              = Vec2{ safe_sqrt(1 - Nz2), safe_sqrt(Nz2) };
    } else {
        Nn = Vec2{ 0, 1 };
    }
    return vec3_add(vec3_mulf(X, Nn.x), vec3_mulf(Ng, Nn.y));
}
static inline Mat3x3 mat3x3_align_vectors(Vec3 a, Vec3 b)
{
    const Vec3 axis  = vec3_cross(b, a);
    const float cosA = vec3_dot(a, b);
    Mat3x3 m;
    if (cosA <= -1) {
        m.col[0] = Vec3{ -1, 0, 0 }, m.col[1] = Vec3{ 0, -1, 0 }, m.col[2] = Vec3{ 0, 0, -1 };
        return m;
    }
    const float k = 1 / (1 + cosA);
    m.col[0] = Vec3{ (axis.x * axis.x * k) + cosA, (axis.y * axis.x * k) - axis.z, (axis.z * axis.x * k) + axis.y };
    m.col[1] = Vec3{ (axis.x * axis.y * k) + axis.z, (axis.y * axis.y * k) + cosA, (axis.z * axis.y * k) - axis.x };
    m.col[2] = Vec3{ (axis.x * axis.z * k) - axis.y, (axis.y * axis.z * k) + axis.x, (axis.z * axis.z * k) + cosA };
    return m;
}
// ---- shading expressions: the scene's PExpr strings, compiled by the loader into the bytecode of include/ig_expr.h
// (the reference transpiles them into the shader instead, src/runtime/loader/Transpiler.cpp:960-1230). The variables
// are those of sInternalVariables (Transpiler.cpp:338-363) the table format carries; texture alpha reads as 1.
struct ExprContext {
    const igd_scene* scene;
    const SurfaceElement* surf;
    Vec3 view; // "V" / "Rd" = vec3_neg(ctx.ray.dir)
    static ige_v4 pack(Vec3 a) { return ige_v4{ { a.x, a.y, a.z, 0.0f } }; }
    ige_v4 var(int id) const
    {
        switch (id) {
        case IGE_VAR_UVW: return ige_v4{ { surf->tex_coords.x, surf->tex_coords.y, 0.0f, 0.0f } }; // vec2_to_3(surf.tex_coords, 0), shading_context.art:38
        case IGE_VAR_P: return pack(surf->point);
        case IGE_VAR_V: return pack(view);
        case IGE_VAR_N: return pack(surf->local.col[2]);
        case IGE_VAR_NG: return pack(surf->face_normal);
        case IGE_VAR_NX: return pack(surf->local.col[0]);
        case IGE_VAR_NY: return pack(surf->local.col[1]);
        default: {
            const float f = surf->is_entering ? 1.0f : 0.0f;
            return ige_v4{ { f, f, f, f } };
        }
        }
    }
    ige_v4 tex(uint32_t id, float u, float v) const
    {
        const Color c = image_lookup(*scene, scene->textures[id], Vec2{ u, v });
        return ige_v4{ { c.r, c.g, c.b, 1.0f } };
    }
    ige_v4 evr(ige_v4 ng, ige_v4 v, ige_v4 n) const
    {
        return pack(ensure_valid_reflection(make_vec3(ng.v[0], ng.v[1], ng.v[2]), make_vec3(v.v[0], v.v[1], v.v[2]), make_vec3(n.v[0], n.v[1], n.v[2])));
    }
};
static inline Vec3 eval_expression(const igd_scene& sc, int32_t start, const SurfaceElement& surf, Vec3 view)
{
    const ige_v4 r = ige_run(sc.expr_code + start, ExprContext{ &sc, &surf, view });
    return make_vec3(r.v[0], r.v[1], r.v[2]);
}

// The surface the inner BSDF of a bump-mapped material sees: make_bumpmap -> make_normal_set (adjoint == false
// on camera paths, so transform_surf_bsdf changes nothing else).
static inline SurfaceElement bumped_surface(const igd_scene& sc, const ig_material& mat, const SurfaceElement& surf, const Ray& ray)
{
    Vec3 N;
    if (mat.flags & IG_MAT_EXPR_NORMAL) {
        // "transform" BSDF (TransformBSDF.cpp:43-46): make_normal_set with the normal expression
        N = eval_expression(sc, mat.tex_id, surf, vec3_neg(ray.dir));
    } else if (mat.flags & IG_MAT_NORMALMAP) {
        const ig_texture& t = sc.textures[mat.tex_id];
        // make_normalmap (bsdf/map.art:55-61): normal given as [0, 1] RGB; mat3x3_left_mul = (col_i . v)
        const Color c  = image_lookup(sc, t, surf.tex_coords);
        const Vec3 nt  = vec3_normalize(make_vec3(2 * c.r - 1, 2 * c.g - 1, 2 * c.b - 1));
        const Vec3 oN  = make_vec3(vec3_dot(surf.local.col[0], nt), vec3_dot(surf.local.col[1], nt), vec3_dot(surf.local.col[2], nt));
        const float st = mat.p[11];
        N = st != 1 ? vec3_normalize(vec3_add(surf.local.col[2], vec3_mulf(vec3_sub(oN, surf.local.col[2]), st))) : oN;
    } else {
        const ig_texture& t = sc.textures[mat.tex_id];
        const float delta = 0.001f; // texture_dx / texture_dy (texture/common.art:33-43)
        const Color c0    = image_lookup(sc, t, surf.tex_coords);
        const Color cx    = image_lookup(sc, t, Vec2{ surf.tex_coords.x + delta, surf.tex_coords.y });
        const Color cy    = image_lookup(sc, t, Vec2{ surf.tex_coords.x, surf.tex_coords.y + delta });
        const float dx    = (cx.r - c0.r) * (1 / delta);
        const float dy    = (cy.r - c0.r) * (1 / delta);
        N = vec3_normalize(vec3_sub(surf.local.col[2], vec3_mulf(vec3_add(vec3_mulf(surf.local.col[0], dx), vec3_mulf(surf.local.col[1], dy)), mat.p[11])));
    }
    const Vec3 n        = ensure_valid_reflection(surf.face_normal, vec3_neg(ray.dir), vec3_normalize(N));
    const Mat3x3 trans  = mat3x3_align_vectors(surf.local.col[2], n);
    SurfaceElement out  = surf;
    out.local.col[0]    = mat3x3_mul(trans, surf.local.col[0]); // mat3x3_matmul(trans, local), matrix.art:124-127
    out.local.col[1]    = mat3x3_mul(trans, surf.local.col[1]);
    out.local.col[2]    = mat3x3_mul(trans, surf.local.col[2]);
    return out;
}

// ---- principled BSDF (bsdf/principled.art). All directions of the closure are in the shading frame.
static inline float lerpf(float a, float b, float k) { return (1 - k) * a + k * b; } // common.art:237
static inline float color_luminance(Color c) { return c.r * 0.2126f + c.g * 0.7152f + c.b * 0.0722f; } // color.art:29,81-83
static inline float schlick_approx(float f) // fresnel.art:88-91
{
    const float s = clampf(1 - f, 0, 1);
    return (s * s) * (s * s) * s;
}
static inline float schlick_r0(float eta) // fresnel.art:101-104
{
    const float factor = clampf((eta - 1) / (eta + 1), -1, 1);
    return factor * factor;
}
static inline float fresnel_dielectric(float eta, float cos_i) // math.art:120-123
{
    float cos_t, factor;
    return fresnel(eta, cos_i, cos_t, factor) ? factor : 1.0f;
}
static inline float halfway_reflective_jacobian(float c) { return safe_div(1, 4 * c); } // shading.art:69
static inline float halfway_refractive_jacobian(float eta, float cos_i, float cos_o)    // shading.art:71-74
{
    const float jacob_d = cos_i + cos_o * eta;
    return safe_div(eta * eta * cos_i, jacob_d * jacob_d);
}
static inline Vec3 vec3_halfway_refractive(Vec3 a, Vec3 b, float eta) { return vec3_normalize(vec3_add(a, vec3_mulf(b, eta))); } // vector.art:142
static inline bool is_positive_hemisphere(Vec3 v) { return v.z >= 0; }                                                          // shading.art:62
static inline bool is_same_hemisphere(Vec3 a, Vec3 b) { return is_positive_hemisphere(a) == is_positive_hemisphere(b); }        // shading.art:63
static inline Vec3 make_same_hemisphere(Vec3 a, Vec3 b) { return is_same_hemisphere(a, b) ? b : vec3_neg(b); }                  // shading.art:65
static inline Vec3 make_positive_hemisphere(Vec3 v) { return is_positive_hemisphere(v) ? v : vec3_neg(v); }                     // shading.art:66

struct Principled {
    // principled::Closure (principled.art:22-43)
    Mat3x3 local;
    bool is_entering;
    Color base_color;
    float reflective_ior, refractive_ior, diffuse_transmission, specular_transmission, specular_tint;
    float roughness_u, roughness_v, flatness, metallic, sheen, sheen_tint, clearcoat, clearcoat_gloss, clearcoat_roughness;
    bool thin, clearcoat_top_only;
    float reflective_eta, refractive_eta;

    static constexpr float micro_eps   = 1e-5f; // principled.art:243-244
    static constexpr float grazing_eps = 1e-5f;

    // make_principled_bsdf (principled.art:236-268)
    Principled(const ig_material& m, const SurfaceElement& surf, Color base)
    {
        local                 = surf.local;
        is_entering           = surf.is_entering;
        base_color            = base;
        reflective_ior        = m.p[3];
        refractive_ior        = m.p[4];
        diffuse_transmission  = m.p[5];
        specular_transmission = m.p[6];
        specular_tint         = m.p[7];
        roughness_u           = igm_max(1e-3f, m.p[8]);
        roughness_v           = igm_max(1e-3f, m.p[9]);
        flatness              = m.p[10];
        metallic              = m.r[0];
        sheen                 = m.r[1];
        sheen_tint            = m.r[2];
        clearcoat             = m.r[3];
        clearcoat_gloss       = m.r[4];
        clearcoat_roughness   = m.r[5];
        thin                  = (m.flags & IG_MAT_THIN) != 0;
        clearcoat_top_only    = (m.flags & IG_MAT_CLEARCOAT_ALL) == 0;
        reflective_eta        = (is_entering || thin) ? 1 / reflective_ior : reflective_ior;
        refractive_eta        = (is_entering || thin) ? 1 / refractive_ior : refractive_ior;
    }

    static Mat3x3 identity()
    {
        Mat3x3 m;
        m.col[0] = make_vec3(1, 0, 0), m.col[1] = make_vec3(0, 1, 0), m.col[2] = make_vec3(0, 0, 1);
        return m;
    }
    Vec3 to_local(Vec3 v) const { return make_vec3(vec3_dot(local.col[0], v), vec3_dot(local.col[1], v), vec3_dot(local.col[2], v)); }
    Vec3 to_world(Vec3 v) const { return vec3_add(vec3_add(vec3_mulf(local.col[0], v.x), vec3_mulf(local.col[1], v.y)), vec3_mulf(local.col[2], v.z)); }

    // tint_color / sheen_tint_color (principled.art:52-60)
    static Color tint_color(Color c)
    {
        const float lum = color_luminance(c);
        return lum <= flt_eps ? Color{ 1, 1, 1 } : color_mulf(c, safe_div(1, lum));
    }
    // getMicro / getReflectionMicro / getRefractionMicro (principled.art:62-78)
    static GGX micro(float ru, float rv) { return GGX{ identity(), igm_max(1e-3f, ru * ru), igm_max(1e-3f, rv * rv) }; }
    GGX reflection_micro() const { return micro(roughness_u, roughness_v); }
    GGX refraction_micro() const
    {
        if (thin)
            return micro(clampf((0.65f * refractive_ior - 0.35f) * roughness_u, 0, 1), clampf((0.65f * refractive_ior - 0.35f) * roughness_v, 0, 1));
        return micro(roughness_u, roughness_v);
    }

    // evalDisneyFresnelTerm (principled.art:80-95)
    Color fresnel_term(Vec3 wo, Vec3 wi, Vec3 h) const
    {
        const float HdV = absolute_cos(wo, h);
        const float HdL = absolute_cos(wi, h);
        if (HdV * HdL <= flt_eps)
            return Color{ 0, 0, 0 };
        const float f1v = fresnel_dielectric(reflective_eta, HdV);
        const Color f1{ f1v, f1v, f1v };
        const Color color = tint_color(base_color);
        const Color a     = color_lerp(Color{ 1, 1, 1 }, color, specular_tint);
        const Color r0    = color_lerp(color_mulf(a, schlick_r0(reflective_eta)), base_color, metallic);
        const float s     = schlick_approx(HdL); // schlick(r0, white, HdL), fresnel.art:93-97
        const Color f2{ r0.r + (1 - r0.r) * s, r0.g + (1 - r0.g) * s, r0.b + (1 - r0.b) * s };
        return color_lerp(f1, f2, metallic);
    }
    // evalSubsurfaceTerm (principled.art:97-108)
    float subsurface_term(Vec3 wo, Vec3 wi, Vec3 h) const
    {
        const float r2    = roughness_u * roughness_v;
        const float HdotL = vec3_dot(wi, h);
        const float fss90 = HdotL * HdotL * r2;
        const float aNdL  = igm_abs(wi.z);
        const float aNdV  = igm_abs(wo.z);
        const float lk    = schlick_approx(aNdL);
        const float vk    = schlick_approx(aNdV);
        const float fss   = (1 - lk + fss90 * lk) * (1 - vk + fss90 * vk);
        return 1.25f * (fss * (1 / (aNdL + aNdV + 1e-5f) - 0.5f) + 0.5f);
    }
    // evalSheenTerm (principled.art:110-113)
    Color sheen_term(Vec3 wi) const
    {
        const float lk = schlick_approx(igm_abs(wi.z));
        return color_mulf(color_lerp(Color{ 1, 1, 1 }, tint_color(base_color), sheen_tint), sheen * lk * igm_abs(wi.z));
    }
    // evalDiffuseTerm (principled.art:115-128)
    float diffuse_term(Vec3 wo, Vec3 wi, Vec3 h) const
    {
        const float lk    = schlick_approx(igm_abs(wi.z));
        const float vk    = schlick_approx(igm_abs(wo.z));
        const float diff  = (1 - 0.5f * lk) * (1 - 0.5f * vk);
        const float VdotL = absolute_cos(wi, wo);
        const float rr    = (VdotL + 1) * (roughness_u + roughness_v) / 2;
        const float retro = rr * (lk + vk + lk * vk * (rr - 1));
        const float ss    = thin ? 1 - flatness + subsurface_term(wo, wi, h) * flatness : 1.0f;
        return flt_inv_pi * (diff + retro) * ss * igm_abs(wi.z);
    }
    // evalTranslucentTerm (principled.art:130-137)
    float translucent_term(Vec3 wo, Vec3 wi) const
    {
        const float lk   = schlick_approx(igm_abs(wi.z));
        const float vk   = schlick_approx(igm_abs(wo.z));
        const float diff = (1 - 0.5f * lk) * (1 - 0.5f * vk);
        return flt_inv_pi * diff * igm_abs(wi.z);
    }
    // evalReflectionTerm (principled.art:139-148)
    Color reflection_term(Vec3 wo, Vec3 wi, Vec3 h) const
    {
        const GGX m       = reflection_micro();
        const Color F     = fresnel_term(wo, wi, h);
        const float D     = m.D(h);
        const float G     = m.G1(wi) * m.G1(wo);
        const float jacob = halfway_reflective_jacobian(wo.z);
        return color_mulf(F, igm_abs(D * G * jacob));
    }
    // evalRefractionTerm (principled.art:150-176)
    Color refraction_term(Vec3 wo, Vec3 wi, Vec3 h) const
    {
        if (thin) {
            const float fterm = fresnel_dielectric(refractive_eta, igm_abs(wo.z));
            const float F     = fterm + (1 - fterm) * fterm / (fterm + 1);
            return color_mulf(Color{ igm_sqrt(base_color.r), igm_sqrt(base_color.g), igm_sqrt(base_color.b) }, 1 - F);
        }
        const GGX m       = refraction_micro();
        const float HdI   = vec3_dot(wi, h);
        const float HdO   = vec3_dot(wo, h);
        const float F     = fresnel_dielectric(refractive_eta, igm_abs(HdO));
        const float D     = m.D(h);
        const float G     = m.G1(wi) * m.G1(wo);
        const float jacob = halfway_refractive_jacobian(refractive_eta, HdI, HdO);
        const float norm  = igm_abs(safe_div(HdO * jacob, wo.z));
        return color_mulf(base_color, (1 - F) * D * G * norm);
    }
    // evalClearcoatTerm (principled.art:178-191)
    Color clearcoat_term(Vec3 wo, Vec3 wi, Vec3 h) const
    {
        const float F0   = 0.04f;
        const float R    = 0.25f;
        const float R2   = igm_max(0.001f, clearcoat_roughness * (1 - clearcoat_gloss) + 0.01f * clearcoat_gloss);
        const float aHdL = absolute_cos(wi, h);
        const float d    = GGX{ identity(), R2, R2 }.D(h);
        const float f    = F0 + (1 - F0) * schlick_approx(aHdL); // schlick_f, fresnel.art:99
        const GGX gm{ identity(), R, R };
        const float g     = gm.G1(wi) * gm.G1(wo);
        const float jacob = halfway_reflective_jacobian(wo.z);
        const float v     = igm_abs(R * d * f * g * jacob * wi.z);
        return Color{ v, v, v };
    }

    struct Lobes {
        float diff_refl, diff_trans, spec_refl, spec_trans;
    };
    // calcLobeDistribution (principled.art:200-233)
    Lobes lobes(Vec3 wo) const
    {
        const float metallic_in   = clampf(metallic, 0, 1);
        const float diff_trans_in = clampf(diffuse_transmission, 0, 1);
        const float spec_trans_in = clampf(specular_transmission, 0, 1);
        const float abs_gen       = color_luminance(base_color);
        const float abs_spec      = lerpf(1, color_luminance(tint_color(base_color)), specular_tint);
        const float diff_refl     = clampf(abs_gen * (1 - metallic_in) * (1 - spec_trans_in), 0, 1);
        const float F             = fresnel_dielectric(refractive_eta, igm_abs(wo.z));
        const float spec_refl     = clampf(abs_spec * (1 - F) + F, 0, 1);
        const bool has_transmission = diff_trans_in > 0 || spec_trans_in > 0;
        if (!has_transmission) {
            const float norm = diff_refl + spec_refl;
            if (norm > flt_eps)
                return Lobes{ diff_refl / norm, 0, spec_refl / norm, 0 };
            return Lobes{ 1, 0, 0, 0 };
        }
        const float diff_trans = clampf(abs_gen * diff_trans_in * diff_refl, 0, 1);
        const float spec_trans = clampf((1 - F) * abs_gen * (1 - metallic_in) * spec_trans_in, 0, 1);
        const float norm       = diff_refl + spec_refl + diff_trans + spec_trans;
        if (norm > flt_eps)
            return Lobes{ diff_refl / norm, diff_trans / norm, spec_refl / norm, spec_trans / norm };
        return Lobes{ 1, 0, 0, 0 };
    }

    // eval (principled.art:270-334)
    Color eval(Vec3 in_dir, Vec3 out_dir) const
    {
        const Vec3 wo = to_local(out_dir);
        const Vec3 wi = to_local(in_dir);
        const bool is_transmission = !is_same_hemisphere(wi, wo);
        const Vec3 h = make_same_hemisphere(wo, is_transmission ? vec3_halfway_refractive(wi, wo, refractive_eta) : vec3_halfway(wi, wo));
        const bool in_front         = is_entering == is_positive_hemisphere(wi);
        const bool out_front        = is_entering == is_positive_hemisphere(wo);
        const bool upper_hemisphere = in_front && out_front;
        const float aNdL = igm_abs(wi.z);
        if (aNdL <= grazing_eps)
            return Color{ 0, 0, 0 };
        Color contrib{ 0, 0, 0 };
        const float diffuse_weight = (thin ? 1.0f : 1 - clampf(metallic, 0, 1)) * (1 - clampf(specular_transmission, 0, 1));
        const float trans_weight   = (1 - clampf(metallic, 0, 1)) * clampf(specular_transmission, 0, 1);
        const float spec_weight    = 1;
        if (!is_transmission) {
            if (diffuse_weight > 0)
                contrib = color_add(contrib, color_mulf(base_color, diffuse_term(wo, wi, h) * diffuse_weight));
            if (sheen > 0)
                contrib = color_add(contrib, color_mulf(sheen_term(wi), diffuse_weight));
            contrib = color_add(contrib, color_mulf(reflection_term(wo, wi, h), spec_weight));
            if ((!clearcoat_top_only || upper_hemisphere) && clearcoat > 0)
                contrib = color_add(contrib, color_mulf(clearcoat_term(wo, wi, h), clearcoat));
        } else {
            if (thin && diffuse_transmission > 0)
                contrib = color_add(contrib, color_mulf(base_color, translucent_term(wo, wi) * diffuse_transmission));
            if (specular_transmission > 0)
                contrib = color_add(contrib, color_mulf(refraction_term(wo, wi, h), trans_weight));
        }
        return contrib;
    }

    // diffPdf_local / specReflPdf_local / specTransPdf_local (principled.art:336-359)
    static float bound_spec_pdf(float v) { return v > micro_eps ? v : 0.0f; }
    static float diff_pdf_local(Vec3 wi) { return cosine_hemisphere_pdf(igm_abs(wi.z)); }
    float spec_refl_pdf_local(Vec3 wo, Vec3 wi) const
    {
        const Vec3 pwo      = make_positive_hemisphere(wo);
        const Vec3 pwi      = make_positive_hemisphere(wi);
        const GGX m         = reflection_micro();
        const Vec3 H        = vec3_halfway(pwo, pwi);
        const float cos_h_o = vec3_dot(pwo, H);
        return igm_abs(bound_spec_pdf(m.pdf(pwo, H)) * halfway_reflective_jacobian(cos_h_o));
    }
    float spec_trans_pdf_local(Vec3 wo, Vec3 wi) const
    {
        const Vec3 pwo      = make_positive_hemisphere(wo);
        const Vec3 pwi      = vec3_neg(make_positive_hemisphere(wi));
        const GGX m         = refraction_micro();
        const Vec3 H        = vec3_halfway_refractive(pwi, pwo, refractive_eta);
        const float cos_h_i = vec3_dot(pwi, H);
        const float cos_h_o = vec3_dot(pwo, H);
        return igm_abs(bound_spec_pdf(m.pdf(pwo, H)) * halfway_refractive_jacobian(refractive_eta, cos_h_i, cos_h_o));
    }
    // pdf (principled.art:361-377)
    float pdf(Vec3 in_dir, Vec3 out_dir) const
    {
        const Vec3 wo = to_local(out_dir);
        const Vec3 wi = to_local(in_dir);
        if (igm_abs(wo.z) <= grazing_eps || igm_abs(wi.z) <= grazing_eps)
            return 0;
        const Lobes l        = lobes(wo);
        const float diff_pdf = diff_pdf_local(wi);
        if (is_same_hemisphere(wo, wi))
            return l.diff_refl * diff_pdf + l.spec_refl * spec_refl_pdf_local(wo, wi);
        if (thin)
            return l.diff_trans * diff_pdf + l.spec_trans;
        return l.diff_trans * diff_pdf + l.spec_trans * spec_trans_pdf_local(wo, wi);
    }

    // sample (principled.art:382-476). Returns false for reject_bsdf_sample().
    bool sample(Rng& rnd, Vec3 out_dir, Vec3& in_dir, float& pdf_out, Color& color, float& eta, bool adjoint = false) const
    {
        const Vec3 wo = to_local(out_dir);
        if (igm_abs(wo.z) <= grazing_eps)
            return false;
        const Lobes l    = lobes(wo);
        const float pick = rnd.next_f32();
        Vec3 dir;
        float spdf;
        if (pick < l.diff_refl) {
            const float u     = rnd.next_f32();
            const float v     = rnd.next_f32();
            const DirSample s = sample_cosine_hemisphere(u, v);
            dir               = make_same_hemisphere(wo, s.dir);
            spdf              = s.pdf * l.diff_refl + spec_refl_pdf_local(wo, dir) * l.spec_refl;
        } else if (pick < l.diff_refl + l.diff_trans) {
            const float u     = rnd.next_f32();
            const float v     = rnd.next_f32();
            const DirSample s = sample_cosine_hemisphere(u, v);
            dir               = vec3_neg(make_same_hemisphere(wo, s.dir));
            spdf              = s.pdf * l.diff_trans + spec_trans_pdf_local(wo, dir) * l.spec_trans;
        } else if (pick < l.diff_refl + l.diff_trans + l.spec_trans) {
            if (thin) {
                dir  = vec3_neg(wo);
                spdf = l.spec_trans;
            } else {
                const Vec3 pwo   = make_positive_hemisphere(wo);
                const GGX m      = refraction_micro();
                const Vec3 n     = m.sample(rnd, pwo);
                const float mpdf = m.pdf(pwo, n);
                if (mpdf <= micro_eps || vec3_len2(n) <= flt_eps)
                    return false;
                const Vec3 oH       = vec3_normalize(n);
                const Vec3 H        = igm_signbit(vec3_dot(oH, pwo)) ? vec3_neg(oH) : oH;
                const float cos_h_o = vec3_dot(pwo, H);
                float cos_t, factor;
                if (fresnel(refractive_eta, cos_h_o, cos_t, factor)) {
                    const Vec3 pwi = vec3_normalize(vec3_refract(pwo, H, refractive_eta, cos_h_o, cos_t));
                    if (!is_same_hemisphere(pwo, pwi) && cos_h_o > flt_eps && -pwi.z > grazing_eps) {
                        dir  = vec3_neg(make_same_hemisphere(wo, pwi));
                        spdf = igm_abs(mpdf * halfway_refractive_jacobian(refractive_eta, vec3_dot(pwi, H), cos_h_o)) * l.spec_trans + diff_pdf_local(dir) * l.diff_trans;
                    } else {
                        return false;
                    }
                } else { // total reflection
                    const Vec3 pwi = vec3_normalize(vec3_reflect(pwo, H));
                    if (is_same_hemisphere(pwo, pwi) && cos_h_o > flt_eps && pwi.z > grazing_eps) {
                        dir  = make_same_hemisphere(wo, pwi);
                        spdf = mpdf * halfway_reflective_jacobian(cos_h_o) * l.spec_trans + diff_pdf_local(dir) * l.diff_trans;
                    } else {
                        return false;
                    }
                }
            }
        } else {
            const Vec3 pwo   = make_positive_hemisphere(wo);
            const GGX m      = reflection_micro();
            const Vec3 n     = m.sample(rnd, pwo);
            const float mpdf = m.pdf(pwo, n);
            if (mpdf <= micro_eps || vec3_len2(n) <= flt_eps)
                return false;
            const Vec3 oH       = vec3_normalize(n);
            const Vec3 H        = igm_signbit(vec3_dot(oH, pwo)) ? vec3_neg(oH) : oH;
            const float cos_h_o = vec3_dot(pwo, H);
            const Vec3 pwi      = vec3_normalize(vec3_reflect(pwo, H));
            if (is_same_hemisphere(pwo, pwi) && cos_h_o > flt_eps && pwi.z > grazing_eps) {
                dir  = make_same_hemisphere(wo, pwi);
                spdf = igm_abs(mpdf * halfway_reflective_jacobian(cos_h_o)) * l.spec_refl + diff_pdf_local(dir) * l.diff_refl;
            } else {
                return false;
            }
        }
        if (!(spdf > flt_eps && (igm_abs(spdf) <= 3.402823466e+38f)))
            return false;
        eta     = (thin || is_same_hemisphere(wo, dir)) ? 1.0f : refractive_eta;
        in_dir  = to_world(dir);
        pdf_out = spdf;
        // principled.art:471: light paths carry 1 / eta^2 across a refraction
        const float spread = (adjoint && !thin && !is_same_hemisphere(wo, dir)) ? 1 / (refractive_eta * refractive_eta) : 1.0f;
        color   = color_mulf(eval(in_dir, out_dir), spread / spdf);
        return true;
    }
};

// ---- BSDFs (driver/bsdf.art)
struct BsdfSample {
    Vec3 in_dir;
    float pdf;
    Color color;
    float eta;
    bool is_delta;
};

// make_rough_dielectric_bsdf (bsdf/dielectric.art:64-191) over the VNDF-GGX distribution of the surface frame
struct RoughDielectric {
    Vec3 N;
    float eta, pdf_eps;
    Color ks, kt;
    GGX micro;
    static constexpr float cos_eps = 1e-5f;

    RoughDielectric(const ig_material& m, const SurfaceElement& surf)
    {
        N       = surf.local.col[2];
        eta     = surf.is_entering ? m.p[0] / m.p[1] : m.p[1] / m.p[0];
        pdf_eps = m.p[8];
        ks      = Color{ m.p[2], m.p[3], m.p[4] };
        kt      = Color{ m.p[5], m.p[6], m.p[7] };
        micro   = GGX{ surf.local, m.p[9], m.p[10] };
    }
    Color eval(Vec3 in_dir, Vec3 out_dir) const
    {
        const float cos_i = vec3_dot(N, in_dir);
        const float cos_o = vec3_dot(N, out_dir);
        if (igm_abs(cos_i * cos_o) <= cos_eps)
            return Color{ 0, 0, 0 };
        const bool is_transmission = igm_signbit(cos_i * cos_o);
        const Vec3 H        = is_transmission ? vec3_halfway_refractive(in_dir, out_dir, eta) : vec3_halfway(in_dir, out_dir);
        const float cos_h_i = vec3_dot(H, in_dir);
        const float cos_h_o = vec3_dot(H, out_dir);
        if (igm_abs(cos_h_i * cos_h_o) <= cos_eps)
            return Color{ 0, 0, 0 };
        const float fterm = fresnel_dielectric(eta, igm_abs(cos_h_o));
        const float D     = micro.D(H);
        const float G     = micro.G1(in_dir) * micro.G1(out_dir);
        if (!is_transmission) {
            const float jacob = halfway_reflective_jacobian(cos_o);
            return color_mulf(ks, fterm * D * G * igm_abs(jacob));
        }
        const float jacob = halfway_refractive_jacobian(eta, cos_h_i, cos_h_o);
        const float norm  = igm_abs(safe_div(cos_h_o * jacob, cos_o));
        return color_mulf(kt, (1 - fterm) * D * G * norm);
    }
    float pdf(Vec3 in_dir, Vec3 out_dir) const
    {
        const float cos_i = vec3_dot(N, in_dir);
        const float cos_o = vec3_dot(N, out_dir);
        if (igm_abs(cos_i * cos_o) <= cos_eps)
            return 0;
        const bool is_transmission = igm_signbit(cos_i * cos_o);
        const Vec3 H        = is_transmission ? vec3_halfway_refractive(in_dir, out_dir, eta) : vec3_halfway(in_dir, out_dir);
        const float cos_h_i = vec3_dot(H, in_dir);
        const float cos_h_o = vec3_dot(H, out_dir);
        if (igm_abs(cos_h_i * cos_h_o) <= cos_eps)
            return 0;
        const float fterm = fresnel_dielectric(eta, igm_abs(cos_h_o));
        const float mpdf  = micro.pdf(out_dir, H);
        if (mpdf <= pdf_eps)
            return 0;
        if (!is_transmission)
            return fterm * mpdf * igm_abs(halfway_reflective_jacobian(cos_h_o));
        return (1 - fterm) * mpdf * igm_abs(halfway_refractive_jacobian(eta, cos_h_i, cos_h_o));
    }
    bool sample(Rng& rnd, Vec3 out_dir, BsdfSample& s, bool adjoint = false) const
    {
        const float cos_o = vec3_dot(N, out_dir);
        if (igm_abs(cos_o) <= cos_eps)
            return false;
        const Vec3 m     = micro.sample(rnd, out_dir);
        const float mpdf = micro.pdf(out_dir, m);
        if (vec3_len2(m) <= flt_eps || mpdf <= pdf_eps)
            return false;
        const Vec3 oH       = vec3_normalize(m);
        const Vec3 H        = igm_signbit(vec3_dot(oH, out_dir)) ? vec3_neg(oH) : oH;
        const float cos_h_o = vec3_dot(H, out_dir);
        if (igm_abs(cos_h_o) <= cos_eps)
            return false;
        float cos_t = 0, factor = 1;
        if (!fresnel(eta, cos_h_o, cos_t, factor)) {
            cos_t  = 0;
            factor = 1;
        }
        Vec3 in_dir;
        float sel_pdf;
        if (rnd.next_f32() > factor) {
            in_dir            = vec3_normalize(vec3_refract(out_dir, H, eta, cos_h_o, cos_t));
            const float jacob = halfway_refractive_jacobian(eta, vec3_dot(H, in_dir), cos_h_o);
            sel_pdf           = (1 - factor) * igm_abs(jacob);
        } else {
            in_dir            = vec3_normalize(vec3_reflect(out_dir, H));
            const float jacob = halfway_reflective_jacobian(cos_h_o);
            sel_pdf           = factor * igm_abs(jacob);
        }
        const float cos_i          = vec3_dot(N, in_dir);
        const float f_pdf          = mpdf * sel_pdf;
        const bool is_transmission = igm_signbit(cos_i * cos_o);
        s.in_dir   = in_dir;
        s.pdf      = f_pdf;
        s.color    = color_mulf(eval(in_dir, out_dir), safe_div((is_transmission && adjoint) ? 1 / (eta * eta) : 1.0f, f_pdf)); // dielectric.art:181-185
        s.eta      = !is_transmission ? 1.0f : eta;
        s.is_delta = false;
        return true;
    }
};

// fresnel_diffuse_factor (core/fresnel.art:42-63)
static inline float fresnel_diffuse_factor(float eta)
{
    if (eta < 1)
        return -1.4399f * (eta * eta) + 0.7099f * eta + 0.6681f + 0.0636f / eta;
    const float ieta1 = 1 / eta;
    const float ieta2 = ieta1 * ieta1;
    const float ieta3 = ieta2 * ieta1;
    const float ieta4 = ieta3 * ieta1;
    const float ieta5 = ieta4 * ieta1;
    return 0.919317f - 3.4793f * ieta1 + 6.75335f * ieta2 - 7.80989f * ieta3 + 4.98554f * ieta4 - 1.36881f * ieta5;
}

// make_plastic_bsdf (bsdf/plastic.art:2-41): make_join_bsdf (bsdf/mix.art:4-65) of a lambertian with an inner-scattering
// factor and the conductor with eta = black, k = white (PlasticBSDF.cpp:36-39) — the rough conductor
// (conductor.art:47-116) or, without roughness, the mirror (conductor.art:2-10, chosen at :131-135) —, mixed by the
// dielectric Fresnel term of the outgoing direction.
struct Plastic {
    Mat3x3 local;
    Color kd, ks;
    float eta, fdr;
    bool smooth;
    GGX micro;

    Plastic(const ig_material& m, const SurfaceElement& surf, Color diffuse)
    {
        local  = surf.local;
        kd     = diffuse;
        ks     = Color{ m.p[6], m.p[7], m.p[8] };
        eta    = m.p[3] / m.p[4];
        fdr    = fresnel_diffuse_factor(eta);
        smooth = (m.flags & IG_MAT_SMOOTH) != 0;
        micro  = GGX{ surf.local, m.p[9], m.p[10] };
    }
    Vec3 N() const { return local.col[2]; }
    float diff_scattering(float cos_i) const
    {
        const float fi = fresnel_dielectric(eta, cos_i);
        return (1 - fi) * eta * eta / (1 - fdr);
    }
    float mix(Vec3 out_dir) const { return fresnel_dielectric(eta, absolute_cos(out_dir, N())); }

    // lobe 0: diffuse_extra, lobe 1: the coating
    Color lobe_eval(int lobe, Vec3 in_dir, Vec3 out_dir) const
    {
        if (lobe == 0)
            return color_mulf(color_mulf(kd, positive_cos(in_dir, N()) * flt_inv_pi), diff_scattering(absolute_cos(in_dir, N())));
        if (smooth)
            return Color{ 0, 0, 0 };
        const float cos_o = absolute_cos(out_dir, N());
        const float cos_i = absolute_cos(in_dir, N());
        if (cos_o <= flt_eps || cos_i <= flt_eps)
            return Color{ 0, 0, 0 };
        const Vec3 H   = vec3_halfway(in_dir, out_dir);
        const float D  = micro.D(H);
        const float G  = micro.G1(in_dir) * micro.G1(out_dir);
        const float f  = conductor_factor(0, 1, absolute_cos(out_dir, H));
        const Color F{ f, f, f }, IF{ 1 - f, 1 - f, 1 - f };
        const Color c{ 0.0f * IF.r + ks.r * F.r, 0.0f * IF.g + ks.g * F.g, 0.0f * IF.b + ks.b * F.b };
        return color_mulf(c, D * G / (4 * cos_o));
    }
    float lobe_pdf(int lobe, Vec3 in_dir, Vec3 out_dir) const
    {
        if (lobe == 0)
            return cosine_hemisphere_pdf(positive_cos(in_dir, N()));
        if (smooth)
            return 0;
        const Vec3 H        = vec3_halfway(in_dir, out_dir);
        const float cos_h_o = absolute_cos(out_dir, H);
        return micro.pdf(out_dir, H) * safe_div(1, 4 * cos_h_o);
    }
    bool lobe_sample(int lobe, Rng& rnd, Vec3 out_dir, BsdfSample& s) const
    {
        if (lobe == 0) {
            const float u      = rnd.next_f32();
            const float v      = rnd.next_f32();
            const DirSample ds = sample_cosine_hemisphere(u, v);
            s.in_dir           = mat3x3_mul(local, ds.dir);
            s.pdf              = ds.pdf;
            s.color            = color_mulf(kd, diff_scattering(absolute_cos(s.in_dir, N())));
            s.eta              = 1;
            s.is_delta         = false;
            return true;
        }
        if (smooth) {
            s.in_dir   = vec3_reflect(out_dir, N());
            s.pdf      = 1;
            s.color    = ks;
            s.eta      = 1;
            s.is_delta = true;
            return true;
        }
        const float cos_o = absolute_cos(out_dir, N());
        if (cos_o <= flt_eps)
            return false;
        const Vec3 m     = micro.sample(rnd, out_dir);
        const float mpdf = micro.pdf(out_dir, m);
        if (vec3_len2(m) <= flt_eps)
            return false;
        const Vec3 oH     = vec3_normalize(m);
        const Vec3 H      = igm_signbit(vec3_dot(oH, out_dir)) ? vec3_neg(oH) : oH;
        const Vec3 in_dir = vec3_reflect(out_dir, H);
        if (absolute_cos(in_dir, N()) <= flt_eps)
            return false;
        const float jacob = 1 / (4 * absolute_cos(out_dir, H));
        s.in_dir   = in_dir;
        s.pdf      = mpdf * jacob;
        s.color    = color_mulf(lobe_eval(1, in_dir, out_dir), safe_div(1, s.pdf));
        s.eta      = 1;
        s.is_delta = false;
        return true;
    }

    // make_join_bsdf with eval_f = color_lerp (mix.art:5-22)
    Color eval(Vec3 in_dir, Vec3 out_dir) const { return color_lerp(lobe_eval(0, in_dir, out_dir), lobe_eval(1, in_dir, out_dir), mix(out_dir)); }
    float pdf(Vec3 in_dir, Vec3 out_dir) const { return lerpf(lobe_pdf(0, in_dir, out_dir), lobe_pdf(1, in_dir, out_dir), mix(out_dir)); }
    // sample_mat (mix.art:28-38)
    bool sample_lobe(int first, float t, Rng& rnd, Vec3 out_dir, BsdfSample& s) const
    {
        if (!lobe_sample(first, rnd, out_dir, s))
            return false;
        const int second = 1 - first;
        const float p    = lerpf(s.pdf, lobe_pdf(second, s.in_dir, out_dir), t);
        const Color c    = color_lerp(color_mulf(s.color, s.pdf), lobe_eval(second, s.in_dir, out_dir), t);
        s.pdf            = p;
        s.color          = color_mulf(c, safe_div(1, p));
        return true;
    }
    bool sample(Rng& rnd, Vec3 out_dir, BsdfSample& s) const // mix.art:40-55
    {
        const float k = mix(out_dir);
        if (rnd.next_f32() < 1 - k) {
            if (sample_lobe(0, k, rnd, out_dir, s))
                return true;
            return sample_lobe(1, k, rnd, out_dir, s);
        }
        if (sample_lobe(1, 1 - k, rnd, out_dir, s))
            return true;
        return sample_lobe(0, 1 - k, rnd, out_dir, s);
    }
};

// ---- Radiance's BRTDfunc with constant arguments and the Roos glazing model on top of it (bsdf/rad.art:7-56):
// make_add_bsdf (bsdf/mix.art:68) = make_join_bsdf with colour addition, nested two levels deep over four leaf BSDFs.
struct RadLeaf {
    int kind; // 0 make_lambertian_bsdf (diffuse.art:2-12), 1 make_lambertian_transmission_bsdf (:14-24), 2 make_mirror_bsdf (conductor.art:2-10),
              // 3 make_perfect_refraction_bsdf (dielectric.art:3-11)
    Color c;
    static float negative_cos(Vec3 a, Vec3 b)
    {
        const float cs = vec3_dot(a, b);
        return cs <= 0 ? cs : 0.0f; // core/common.art:265-268
    }
    Color eval(const Mat3x3& local, Vec3 in_dir) const
    {
        if (kind == 0)
            return color_mulf(c, positive_cos(in_dir, local.col[2]) * flt_inv_pi);
        if (kind == 1)
            return color_mulf(c, -negative_cos(in_dir, local.col[2]) * flt_inv_pi);
        return Color{ 0, 0, 0 };
    }
    float pdf(const Mat3x3& local, Vec3 in_dir) const
    {
        if (kind == 0)
            return positive_cos(in_dir, local.col[2]) / flt_pi; // cosine_hemisphere_pdf
        if (kind == 1)
            return -negative_cos(in_dir, local.col[2]) / flt_pi;
        return 0;
    }
    void sample(const Mat3x3& local, Rng& rnd, Vec3 out_dir, BsdfSample& s) const
    {
        s.eta = 1;
        if (kind <= 1) {
            const float u       = rnd.next_f32();
            const float v       = rnd.next_f32();
            const DirSample smp = sample_cosine_hemisphere(u, v);
            const Vec3 gdir     = mat3x3_mul(local, smp.dir);
            s.in_dir   = kind == 0 ? gdir : vec3_neg(gdir);
            s.pdf      = smp.pdf;
            s.color    = c;
            s.is_delta = false;
        } else {
            s.in_dir   = kind == 2 ? vec3_reflect(out_dir, local.col[2]) : vec3_neg(out_dir);
            s.pdf      = 1;
            s.color    = c;
            s.is_delta = true;
        }
    }
};
// make_join_bsdf (mix.art:4-65) with eval_f = color_add and a constant sampling probability
template <class A, class B>
struct RadAdd {
    A a;
    B b;
    float k;
    Color eval(const Mat3x3& l, Vec3 in_dir) const { return color_add(a.eval(l, in_dir), b.eval(l, in_dir)); }
    float pdf(const Mat3x3& l, Vec3 in_dir) const
    {
        if (k <= 0)
            return a.pdf(l, in_dir);
        if (k >= 1)
            return b.pdf(l, in_dir);
        return lerpf(a.pdf(l, in_dir), b.pdf(l, in_dir), k);
    }
    template <class F, class S>
    static void sample_mat(const F& first, const S& second, float t, const Mat3x3& l, Rng& rnd, Vec3 out_dir, BsdfSample& s)
    {
        first.sample(l, rnd, out_dir, s);
        const float p = lerpf(s.pdf, second.pdf(l, s.in_dir), t);
        const Color c = color_add(color_mulf(s.color, s.pdf), second.eval(l, s.in_dir));
        s.pdf         = p;
        s.color       = color_mulf(c, safe_div(1, p));
    }
    void sample(const Mat3x3& l, Rng& rnd, Vec3 out_dir, BsdfSample& s) const // (every leaf always delivers a sample: no fallback to the other side)
    {
        if (rnd.next_f32() < 1 - k)
            sample_mat(a, b, k, l, rnd, out_dir, s);
        else
            sample_mat(b, a, 1 - k, l, rnd, out_dir, s);
    }
};
using RadBrtd = RadAdd<RadAdd<RadLeaf, RadLeaf>, RadAdd<RadLeaf, RadLeaf>>;
static inline RadBrtd make_rad_brtd(bool is_entering, Color refl_spec, Color trns_spec, Color refl_f_diff_plus_direct, Color refl_b_diff_plus_direct, Color trns_diff)
{
    const Color refl_diff = is_entering ? refl_f_diff_plus_direct : refl_b_diff_plus_direct;
    RadBrtd r;
    r.b.a = RadLeaf{ 2, refl_spec };
    r.b.b = RadLeaf{ 3, trns_spec };
    r.b.k = safe_div(color_average(trns_spec), color_average(refl_spec) + color_average(trns_spec));
    r.a.a = RadLeaf{ 0, refl_diff };
    r.a.b = RadLeaf{ 1, trns_diff };
    r.a.k = safe_div(color_average(trns_diff), color_average(refl_diff) + color_average(trns_diff));
    const float sum_spec = color_average(color_add(refl_spec, trns_spec));
    const float sum_diff = color_average(color_add(refl_diff, trns_diff));
    r.k = safe_div(sum_spec, sum_diff + sum_spec);
    return r;
}
// make_rad_roos_bsdf (rad.art:36-56): (tau, rf) from the cosine between ray and shading normal
static inline void rad_roos_factors(const ig_material& m, float cosN, float& rf, float& tau)
{
    const float trns_w = m.p[0], trns_p = m.p[1], trns_q = m.p[2], refl_w = m.p[3], refl_p = m.p[4], refl_q = m.p[5];
    const float a = 8;
    auto b     = [](float q) { return 0.25f / q; };
    auto c     = [&](float, float q) { return 1 - a - b(q); };
    auto alpha = [](float q) { return 5.2f + 0.7f * q; };
    const float beta = 2;
    auto gamma = [](float p, float q) { return (5.26f + 0.06f * p) + (0.73f + 0.04f * p) * q; };
    const float z = igm_acos(igm_abs(clampf(cosN, -1, 1))) * 0.636619772368f;
    tau = trns_w * (1 - a * igm_pow(z, alpha(trns_q)) - b(trns_q) * igm_pow(z, beta) - c(trns_p, trns_q) * igm_pow(z, gamma(trns_p, trns_q)));
    rf  = refl_w + (1 - refl_w) * igm_pow(z, gamma(refl_p, refl_q));
}

// IG_MAT_EXPR_NUMBERS (ig_tables.h; ShadingTree::addNumber with an expression): the record with its number expressions evaluated at
// this hit, in `local`
static inline const ig_material& resolve_material(const igd_scene& sc, const ig_material& mat, const SurfaceElement& surf, Vec3 view, ig_material& local)
{
    if (!(mat.flags & IG_MAT_EXPR_NUMBERS))
        return mat;
    local = mat;
    uint32_t at;
    std::memcpy(&at, &mat.r[7], 4);
    const uint32_t* lst = sc.expr_code + at;
    for (uint32_t i = 0; i < lst[0]; ++i) {
        float aspect;
        std::memcpy(&aspect, &lst[2 + 3 * i], 4);
        ig_material_set_number(&local, lst[1 + 3 * i], aspect, eval_expression(sc, (int32_t)lst[3 + 3 * i], surf, view).x);
    }
    return local;
}

struct Bsdf {
    const ig_material* mat;
    const SurfaceElement* surf;
    const igd_scene* scene = nullptr; // bitmap reflectance lookups
    // make_doublesided_bsdf (bsdf/common.art:28-46) on a surface hit from behind: `surf` already has is_entering = true (the
    // caller's copy), every direction is negated on the way in and the sampled one on the way out
    bool flip = false;
    Vec3 view{ 0, 0, 0 }; // -ray.dir for the "V" of a colour expression (zero inside a blend: the loader keeps V out of those)
    // light paths (the light tracer): bsdf.sample(rnd, out_dir, adjoint = true); `bumped`: the BSDF sits on a re-oriented surface
    // (transform_surf_bsdf, bsdf/map.art:16-32) whose original shading normal was old_normal
    bool adjoint = false, bumped = false;
    Vec3 old_normal{ 0, 0, 0 };
    Bsdf unflipped() const
    {
        Bsdf b = *this;
        b.flip = false;
        return b;
    }

    // plastic: mat1.is_all_delta & mat2.is_all_delta with a diffuse mat1 (mix.art:63)
    // the weight of a blend / mask: a constant or a number expression (BlendBSDF.cpp:40, MaskBSDF.cpp:30-55)
    float weight() const { return (mat->flags & IG_MAT_EXPR_WEIGHT) ? eval_expression(*scene, mat->tex_id, *surf, view).x : mat->p[0]; }
    bool is_rad() const { return mat->bsdf_type == IG_BSDF_RAD_BRTD || mat->bsdf_type == IG_BSDF_RAD_ROOS; }
    // make_rad_brtdfunc_bsdf / make_rad_roos_bsdf (bsdf/rad.art) from the material record; the Roos model's cosN is
    // -dot(ctx.ray.dir, ctx.surf.local.col(2)) (RadRoosBSDF.cpp:28)
    RadBrtd rad() const
    {
        const Color td{ mat->q[0], mat->q[1], mat->q[2] };
        if (mat->bsdf_type == IG_BSDF_RAD_ROOS) {
            float rf, tau;
            rad_roos_factors(*mat, vec3_dot(view, surf->local.col[2]), rf, tau);
            const Color black{ 0, 0, 0 };
            return make_rad_brtd(surf->is_entering, Color{ rf, rf, rf }, Color{ tau, tau, tau }, color_add(Color{ mat->p[6], mat->p[7], mat->p[8] }, black),
                                 color_add(Color{ mat->p[9], mat->p[10], mat->p[11] }, black), td);
        }
        return make_rad_brtd(surf->is_entering, Color{ mat->p[0], mat->p[1], mat->p[2] }, Color{ mat->p[3], mat->p[4], mat->p[5] }, Color{ mat->p[6], mat->p[7], mat->p[8] },
                             Color{ mat->p[9], mat->p[10], mat->p[11] }, td);
    }
    bool is_all_delta() const
    {
        if (mat->bsdf_type == IG_BSDF_BLEND) // mat1.is_all_delta & mat2.is_all_delta (mix.art:63)
            return inner(0).is_all_delta() && inner(1).is_all_delta();
        return mat->bsdf_type == IG_BSDF_DIELECTRIC || mat->bsdf_type == IG_BSDF_TRANSPARENT || (mat->bsdf_type == IG_BSDF_CONDUCTOR && (mat->flags & IG_MAT_SMOOTH));
    }
    // the two BSDFs a blend mixes (make_mix_bsdf, bsdf/mix.art:4-68); they see the same surface
    Bsdf inner(int i) const
    {
        Bsdf b{ &scene->materials[mat->pad[i]], surf, scene };
        b.adjoint = adjoint;
        return b;
    }

    Color kd() const
    {
        if (mat->flags & IG_MAT_EXPR_COLOR) { // vec4_to_color / vec3_to_color of the expression, a number as grey (Transpiler.cpp:1290-1302)
            const Vec3 c = eval_expression(*scene, mat->tex_refl, *surf, view);
            return Color{ c.x, c.y, c.z };
        }
        if (mat->flags & IG_MAT_IMAGE)
            return image_lookup(*scene, scene->textures[mat->tex_refl], surf->tex_coords);
        if (mat->flags & IG_MAT_CHECKER)
            return checkerboard(*mat, surf->tex_coords);
        return Color{ mat->p[0], mat->p[1], mat->p[2] };
    }
    GGX ggx() const { return GGX{ surf->local, mat->p[9], mat->p[10] }; }

    // fresnelTerm of make_rough_conductor_bsdf (bsdf/conductor.art:118-125)
    Color conductor_fresnel(float cosTheta) const
    {
        return Color{ conductor_factor(mat->p[0], mat->p[3], cosTheta), conductor_factor(mat->p[1], mat->p[4], cosTheta), conductor_factor(mat->p[2], mat->p[5], cosTheta) };
    }

    // make_orennayar_bsdf.eval (bsdf/diffuse.art:28-39): p[3] alpha
    Color orennayar_eval(Vec3 in_dir, Vec3 out_dir) const
    {
        const Vec3 N   = surf->local.col[2];
        const float a2 = mat->p[3] * mat->p[3];
        const float p1 = positive_cos(in_dir, N);
        const float p2 = positive_cos(out_dir, N);
        const float sv = -p1 * p2 + positive_cos(out_dir, in_dir);
        const float t  = sv <= flt_eps ? 1.0f : igm_max(flt_eps, igm_max(p1, p2));
        const float A  = 1 - 0.5f * a2 / (a2 + 0.33f);
        const float B  = 0.45f * a2 / (a2 + 0.09f);
        const float C  = 0.17f * a2 / (a2 + 0.13f);
        const Color k  = kd();
        return color_mulf(color_add(color_mulf(k, (A + (B * sv / t)) / flt_pi), color_mul(k, color_mulf(k, C / flt_pi))), p1);
    }
    // fastlog2 / fastpow2 / fastpow (core/common.art:71-90): bit tricks over float and integer arithmetic only
    static float fastpow(float x, float p)
    {
        const uint32_t vx = igm_bits(x);
        const float z     = igm_float((vx & 0x007FFFFFu) | 0x3f000000u);
        const float y     = (float)vx * 1.1920928955078125e-7f;
        const float lg    = y - 124.22551499f - 1.498030302f * z - 1.72587999f / (0.3520887068f + z);
        const float q     = p * lg;
        const float off   = q < 0 ? 1.0f : 0.0f;
        const float clipp = q < -126 ? -126.0f : q;
        const int w       = (int)clipp;
        const float zz    = clipp - (float)w + off;
        const int v       = (int)((float)(1u << 23) * (clipp + 121.2740575f + 27.7280233f / (4.84252568f - zz) - 1.49012907f * zz));
        return igm_float((uint32_t)v);
    }
    // make_phong_bsdf (bsdf/phong.art:1-22): p[0..2] ks, p[3] ns
    Color phong_eval(Vec3 in_dir, Vec3 out_dir) const
    {
        const Vec3 N     = surf->local.col[2];
        const float ns   = mat->p[3];
        const float cosI = positive_cos(in_dir, N);
        const float c    = positive_cos(in_dir, vec3_reflect(out_dir, N));
        return color_mulf(Color{ mat->p[0], mat->p[1], mat->p[2] }, cosI * fastpow(c, ns) * (ns + 2) / (2 * flt_pi));
    }
    float phong_pdf(Vec3 in_dir, Vec3 out_dir) const
    {
        const float ns = mat->p[3];
        const float c  = positive_cos(in_dir, vec3_reflect(out_dir, surf->local.col[2]));
        return fastpow(c, ns) * (ns + 1) * (1 / (2 * flt_pi)); // cosine_power_hemisphere_pdf (core/sampling.art:79-81)
    }
    bool phong_sample(Rng& rnd, Vec3 out_dir, BsdfSample& s) const
    {
        const Vec3 N    = surf->local.col[2];
        const float ns  = mat->p[3];
        const Vec3 refl = vec3_reflect(out_dir, N);
        const float u   = rnd.next_f32();
        const float v   = rnd.next_f32();
        // sample_cosine_power_hemisphere (core/sampling.art:84-96)
        const float c   = igm_min(fastpow(v, 1 / (ns + 1)), 1.0f);
        const float sn  = igm_sqrt(1 - c * c);
        const float phi = 2 * flt_pi * u;
        const float pck = c != 0 ? v / c : 0.0f;
        s.pdf           = pck * (ns + 1) * (1 / (2 * flt_pi));
        s.in_dir        = mat3x3_mul(make_orthonormal_mat3x3(refl), make_vec3(sn * igm_cos(phi), sn * igm_sin(phi), c));
        const float cs  = positive_cos(s.in_dir, N);
        s.color         = color_mulf(Color{ mat->p[0], mat->p[1], mat->p[2] }, cs * (ns + 2) / (ns + 1));
        s.eta           = 1;
        s.is_delta      = false;
        return true;
    }

    // Bsdf::albedo per model (what wrap_infobuffer_renderer splats, technique/internal/infobuffer.art:13-21)
    Color albedo(Vec3 out_dir) const
    {
        if (flip)
            return unflipped().albedo(vec3_neg(out_dir));
        if (is_rad()) { // make_join_bsdf.albedo (mix.art:56-61): colour lerps of the leaves' colours
            const RadBrtd r = rad();
            return color_lerp(color_lerp(r.a.a.c, r.a.b.c, r.a.k), color_lerp(r.b.a.c, r.b.b.c, r.b.k), r.k);
        }
        const Vec3 N = surf->local.col[2];
        if (mat->bsdf_type == IG_BSDF_PHONG) // ks (phong.art:20)
            return Color{ mat->p[0], mat->p[1], mat->p[2] };
        if (mat->bsdf_type == IG_BSDF_TRANSPARENT) // make_perfect_refraction_bsdf: kt (dielectric.art:9)
            return Color{ mat->p[0], mat->p[1], mat->p[2] };
        if (mat->bsdf_type == IG_BSDF_DIELECTRIC && (mat->flags & IG_MAT_THIN)) // make_thin_dielectric_bsdf (dielectric.art:60)
            return Color{ mat->p[2], mat->p[3], mat->p[4] };
        if (mat->bsdf_type == IG_BSDF_DIELECTRIC || mat->bsdf_type == IG_BSDF_ROUGH_DIELECTRIC) // dielectric.art:35,190
            return color_lerp(Color{ mat->p[2], mat->p[3], mat->p[4] }, Color{ mat->p[5], mat->p[6], mat->p[7] }, 0.5f);
        if (mat->bsdf_type == IG_BSDF_CONDUCTOR) { // compute_albedo (conductor.art:50-56), kd = black
            const Color F = conductor_fresnel(absolute_cos(out_dir, N));
            return Color{ 0.0f * (1 - F.r) + mat->p[6] * F.r, 0.0f * (1 - F.g) + mat->p[7] * F.g, 0.0f * (1 - F.b) + mat->p[8] * F.b };
        }
        if (mat->bsdf_type == IG_BSDF_PLASTIC) { // make_join_bsdf.albedo (mix.art:56-61)
            const Plastic pl(*mat, *surf, kd());
            Color coat = pl.ks; // make_mirror_bsdf (conductor.art:8)
            if (!pl.smooth) {
                const float f = conductor_factor(0, 1, absolute_cos(out_dir, N));
                coat          = Color{ 0.0f * (1 - f) + pl.ks.r * f, 0.0f * (1 - f) + pl.ks.g * f, 0.0f * (1 - f) + pl.ks.b * f };
            }
            return color_lerp(kd(), coat, pl.mix(out_dir));
        }
        if (mat->bsdf_type == IG_BSDF_BLEND) // mix.art:56-61
            return color_lerp(inner(0).albedo(out_dir), inner(1).albedo(out_dir), weight());
        return kd(); // lambertian (diffuse.art:10), principled base colour (principled.art:478)
    }

    // make_lambertian_bsdf (bsdf/diffuse.art:2-13); make_rough_base_conductor_bsdf (bsdf/conductor.art:70-84);
    // delta BSDFs evaluate to black (dielectric.art:16-17)
    Color eval(Vec3 in_dir, Vec3 out_dir) const
    {
        if (flip)
            return unflipped().eval(vec3_neg(in_dir), vec3_neg(out_dir));
        if (is_rad())
            return rad().eval(surf->local, in_dir);
        if (mat->bsdf_type == IG_BSDF_BLEND) // eval_f = color_lerp (mix.art:5-8,68)
            return color_lerp(inner(0).eval(in_dir, out_dir), inner(1).eval(in_dir, out_dir), weight());
        if (mat->bsdf_type == IG_BSDF_PHONG)
            return phong_eval(in_dir, out_dir);
        if (mat->bsdf_type == IG_BSDF_PRINCIPLED)
            return Principled(*mat, *surf, kd()).eval(in_dir, out_dir);
        if (mat->bsdf_type == IG_BSDF_PLASTIC)
            return Plastic(*mat, *surf, kd()).eval(in_dir, out_dir);
        if (mat->bsdf_type == IG_BSDF_ROUGH_DIELECTRIC)
            return RoughDielectric(*mat, *surf).eval(in_dir, out_dir);
        if (mat->bsdf_type == IG_BSDF_DIFFUSE) {
            if (mat->p[3] > flt_eps) // make_diffuse_bsdf (diffuse.art:52-58): a roughness selects Oren-Nayar
                return orennayar_eval(in_dir, out_dir);
            return color_mulf(kd(), positive_cos(in_dir, surf->local.col[2]) * flt_inv_pi);
        }
        if (mat->bsdf_type == IG_BSDF_CONDUCTOR) {
            const Vec3 N      = surf->local.col[2];
            const float cos_o = absolute_cos(out_dir, N);
            const float cos_i = absolute_cos(in_dir, N);
            if (cos_o <= flt_eps || cos_i <= flt_eps)
                return Color{ 0, 0, 0 };
            const GGX g   = ggx();
            const Vec3 H  = vec3_halfway(in_dir, out_dir);
            const float D = g.D(H);
            const float G = g.G1(in_dir) * g.G1(out_dir);
            const Color F = conductor_fresnel(absolute_cos(out_dir, H));
            const Color IF{ 1 - F.r, 1 - F.g, 1 - F.b };
            const Color kdz{ 0, 0, 0 }; // kd = black for conductors
            const Color ks{ mat->p[6], mat->p[7], mat->p[8] };
            const Color c = Color{ kdz.r * IF.r + ks.r * F.r, kdz.g * IF.g + ks.g * F.g, kdz.b * IF.b + ks.b * F.b };
            return color_mulf(c, D * G / (4 * cos_o));
        }
        return Color{ 0, 0, 0 };
    }
    float pdf(Vec3 in_dir, Vec3 out_dir) const
    {
        if (flip)
            return unflipped().pdf(vec3_neg(in_dir), vec3_neg(out_dir));
        if (is_rad())
            return rad().pdf(surf->local, in_dir);
        if (mat->bsdf_type == IG_BSDF_BLEND) { // mix.art:10-22 with a constant weight
            const float k = weight();
            if (k <= 0)
                return inner(0).pdf(in_dir, out_dir);
            if (k >= 1)
                return inner(1).pdf(in_dir, out_dir);
            return lerpf(inner(0).pdf(in_dir, out_dir), inner(1).pdf(in_dir, out_dir), k);
        }
        if (mat->bsdf_type == IG_BSDF_PHONG)
            return phong_pdf(in_dir, out_dir);
        if (mat->bsdf_type == IG_BSDF_PRINCIPLED)
            return Principled(*mat, *surf, kd()).pdf(in_dir, out_dir);
        if (mat->bsdf_type == IG_BSDF_PLASTIC)
            return Plastic(*mat, *surf, kd()).pdf(in_dir, out_dir);
        if (mat->bsdf_type == IG_BSDF_ROUGH_DIELECTRIC)
            return RoughDielectric(*mat, *surf).pdf(in_dir, out_dir);
        if (mat->bsdf_type == IG_BSDF_DIFFUSE)
            return cosine_hemisphere_pdf(positive_cos(in_dir, surf->local.col[2]));
        if (mat->bsdf_type == IG_BSDF_CONDUCTOR) {
            const Vec3 H        = vec3_halfway(in_dir, out_dir);
            const float cos_h_o = absolute_cos(out_dir, H);
            const float jacob   = safe_div(1, 4 * cos_h_o);
            return ggx().pdf(out_dir, H) * jacob;
        }
        return 0;
    }
    // shading_normal_adjoint (bsdf/map.art:1-7)
    static float shading_normal_adjoint(Vec3 in_dir, Vec3 out_dir, Vec3 ns, Vec3 ng)
    {
        const float ons = positive_cos(out_dir, ns), ins = positive_cos(in_dir, ns);
        const float ong = positive_cos(out_dir, ng), ing = positive_cos(in_dir, ng);
        return (ins <= flt_eps || ong <= flt_eps) ? 0.0f : (ons / ins) * (ing / ong);
    }
    bool sample(Rng& rnd, Vec3 out_dir, BsdfSample& s) const
    {
        if (flip) {
            if (!unflipped().sample(rnd, vec3_neg(out_dir), s))
                return false;
            s.in_dir = vec3_neg(s.in_dir);
            return true;
        }
        if (adjoint && bumped) { // transform_surf_bsdf.sample with adjoint (bsdf/map.art:19-30)
            Bsdf plain   = *this;
            plain.bumped = false;
            if (!plain.sample(rnd, out_dir, s))
                return false;
            s.color = color_mulf(s.color, shading_normal_adjoint(s.in_dir, out_dir, surf->local.col[2], old_normal));
            return true;
        }
        if (mat->bsdf_type == IG_BSDF_BLEND) {
            // make_join_bsdf.sample (mix.art:27-55)
            const Bsdf m1 = inner(0), m2 = inner(1);
            auto sample_mat = [&](const Bsdf& first, const Bsdf& second, float t) {
                if (!first.sample(rnd, out_dir, s))
                    return false;
                const float p = lerpf(s.pdf, second.pdf(s.in_dir, out_dir), t);
                const Color c = color_lerp(color_mulf(s.color, s.pdf), second.eval(s.in_dir, out_dir), t);
                s.pdf         = p;
                s.color       = color_mulf(c, safe_div(1, p));
                return true;
            };
            const float k = weight();
            if (rnd.next_f32() < 1 - k)
                return sample_mat(m1, m2, k) || sample_mat(m2, m1, k);
            return sample_mat(m2, m1, 1 - k) || sample_mat(m1, m2, 1 - k);
        }
        if (is_rad()) {
            rad().sample(surf->local, rnd, out_dir, s);
            return true;
        }
        if (mat->bsdf_type == IG_BSDF_TRANSPARENT) { // make_perfect_refraction_bsdf.sample (dielectric.art:6-8)
            s.in_dir   = vec3_neg(out_dir);
            s.pdf      = 1;
            s.color    = Color{ mat->p[0], mat->p[1], mat->p[2] };
            s.eta      = 1;
            s.is_delta = true;
            return true;
        }
        if (mat->bsdf_type == IG_BSDF_PHONG)
            return phong_sample(rnd, out_dir, s);
        if (mat->bsdf_type == IG_BSDF_PRINCIPLED) {
            s.is_delta = false;
            return Principled(*mat, *surf, kd()).sample(rnd, out_dir, s.in_dir, s.pdf, s.color, s.eta, adjoint);
        }
        if (mat->bsdf_type == IG_BSDF_PLASTIC)
            return Plastic(*mat, *surf, kd()).sample(rnd, out_dir, s);
        if (mat->bsdf_type == IG_BSDF_ROUGH_DIELECTRIC)
            return RoughDielectric(*mat, *surf).sample(rnd, out_dir, s, adjoint);
        if (mat->bsdf_type == IG_BSDF_DIFFUSE) {
            const float u      = rnd.next_f32();
            const float v      = rnd.next_f32();
            const DirSample ds = sample_cosine_hemisphere(u, v);
            s.in_dir           = mat3x3_mul(surf->local, ds.dir);
            s.pdf              = ds.pdf;
            s.color            = mat->p[3] > flt_eps ? color_mulf(orennayar_eval(s.in_dir, out_dir), 1 / ds.pdf) : kd(); // diffuse.art:46
            s.eta              = 1;
            s.is_delta         = false;
            return true;
        }
        if (mat->bsdf_type == IG_BSDF_CONDUCTOR && (mat->flags & IG_MAT_SMOOTH)) {
            // delta branch of make_rough_base_conductor_bsdf (bsdf/conductor.art:56-68): compute_albedo(out_dir), kd = black
            const Vec3 N      = surf->local.col[2];
            const float cos_o = absolute_cos(out_dir, N);
            const Color F     = Color{ conductor_factor(mat->p[0], mat->p[3], cos_o), conductor_factor(mat->p[1], mat->p[4], cos_o), conductor_factor(mat->p[2], mat->p[5], cos_o) };
            const Color IF    = Color{ 1 - F.r, 1 - F.g, 1 - F.b };
            s.in_dir   = vec3_reflect(out_dir, N);
            s.pdf      = 1;
            s.color    = Color{ 0.0f * IF.r + mat->p[6] * F.r, 0.0f * IF.g + mat->p[7] * F.g, 0.0f * IF.b + mat->p[8] * F.b };
            s.eta      = 1;
            s.is_delta = true;
            return true;
        }
        if (mat->bsdf_type == IG_BSDF_CONDUCTOR) {
            // make_rough_base_conductor_bsdf.sample (bsdf/conductor.art:93-114)
            const Vec3 N      = surf->local.col[2];
            const float cos_o = absolute_cos(out_dir, N);
            if (cos_o <= flt_eps)
                return false;
            const GGX g      = ggx();
            const Vec3 m     = g.sample(rnd, out_dir);
            const float mpdf = g.pdf(out_dir, m);
            if (vec3_len2(m) <= flt_eps)
                return false;
            const Vec3 oH = vec3_normalize(m);
            const Vec3 H  = igm_signbit(vec3_dot(oH, out_dir)) ? vec3_neg(oH) : oH;
            const Vec3 in_dir = vec3_reflect(out_dir, H);
            const float cos_i = absolute_cos(in_dir, N);
            if (cos_i <= flt_eps)
                return false;
            const float cos_h_o = absolute_cos(out_dir, H);
            const float jacob   = 1 / (4 * cos_h_o);
            s.in_dir   = in_dir;
            s.pdf      = mpdf * jacob;
            s.color    = color_mulf(eval(in_dir, out_dir), safe_div(1, s.pdf));
            s.eta      = 1;
            s.is_delta = false;
            return true;
        }
        if (mat->flags & IG_MAT_THIN) {
            // make_thin_dielectric_bsdf (bsdf/dielectric.art:40-61): always from outside to inside
            const float kk    = mat->p[0] / mat->p[1];
            const Vec3 Nn     = surf->local.col[2];
            const float fterm = fresnel_dielectric(kk, absolute_cos(out_dir, Nn));
            const float F     = fterm + (1 - fterm) * fterm / (fterm + 1);
            if (rnd.next_f32() > F) {
                s.in_dir = vec3_neg(out_dir);
                s.color  = Color{ mat->p[5], mat->p[6], mat->p[7] };
            } else {
                s.in_dir = vec3_normalize(vec3_reflect(out_dir, Nn));
                s.color  = Color{ mat->p[2], mat->p[3], mat->p[4] };
            }
            s.pdf      = 1;
            s.eta      = 1;
            s.is_delta = true;
            return true;
        }
        // make_pure_dielectric_bsdf (bsdf/dielectric.art:15-37); n1 = ext_ior, n2 = int_ior
        // (runtime/bsdf/DielectricBSDF.cpp:31-38)
        const float n1 = mat->p[0], n2 = mat->p[1];
        const Color ks = Color{ mat->p[2], mat->p[3], mat->p[4] };
        const Color kt = Color{ mat->p[5], mat->p[6], mat->p[7] };
        const float k  = surf->is_entering ? n1 / n2 : n2 / n1;
        const Vec3 n   = surf->local.col[2];
        const float cos_o = vec3_dot(out_dir, n);
        float cos_t = 0, factor = 1;
        if (!fresnel(k, cos_o, cos_t, factor)) {
            cos_t  = 0;
            factor = 1;
        }
        if (rnd.next_f32() > factor) {
            s.in_dir = vec3_refract(out_dir, n, k, cos_o, cos_t);
            s.pdf    = 1;
            s.color  = color_mulf(kt, adjoint ? k * k : 1.0f); // adjoint_term (dielectric.art:28)
            s.eta    = k;
        } else {
            s.in_dir = vec3_reflect(out_dir, n);
            s.pdf    = 1;
            s.color  = ks;
            s.eta    = 1;
        }
        s.is_delta = true;
        return true;
    }
};

// ---- lights
struct DirectLightSample {
    Vec3 pos, dir;
    Color intensity;
    float pdf_value;
    bool pdf_is_area; // make_area_pdf vs make_solid_pdf (driver/pdf.art:24-38)
    bool pdf_is_delta;
    float cos, dist;
};

struct SQ {
    Vec3 o, n;
    float x0, y0, z0, x1, y1, b0, b1, k, s;
};

// make_plane_area_emitter (light/area.art:124-257), parameters from the
// "SimplePlaneLight" record (area.art:416-440)
struct PlaneEmitter {
    Vec3 origin, x_axis, y_axis, normal;
    float area, inv_area, width, height;
    Vec3 ex, ey;
    Vec2 t0, t1, t2, t3;
    Color radiance;

    explicit PlaneEmitter(const ig_light& l)
    {
        const float* d = l.d;
        origin   = Vec3{ d[0], d[1], d[2] };
        x_axis   = Vec3{ d[4], d[5], d[6] };
        y_axis   = Vec3{ d[8], d[9], d[10] };
        normal   = Vec3{ d[3], d[7], d[11] };
        t0       = Vec2{ d[12], d[13] };
        t1       = Vec2{ d[14], d[15] };
        t2       = Vec2{ d[16], d[17] };
        t3       = Vec2{ d[18], d[19] };
        radiance = Color{ d[20], d[21], d[22] };
        area     = d[23];
        inv_area = safe_div(1, area);
        width    = vec3_len(x_axis);
        height   = vec3_len(y_axis);
        ex       = vec3_mulf(x_axis, 1 / width);
        ey       = vec3_mulf(y_axis, 1 / height);
    }

    static float safe_acos(float a) { return igm_acos(clampf(a, -1, 1)); }

    // area.art:133-181
    SQ compute_sq(Vec3 from_point) const
    {
        const Vec3 dir  = vec3_sub(origin, from_point);
        const float x0  = vec3_dot(dir, ex);
        const float y0  = vec3_dot(dir, ey);
        const float z0_ = vec3_dot(dir, normal);
        const float x1  = x0 + width;
        const float y1  = y0 + height;

        const bool pos = !igm_signbit(z0_);
        const float z0 = pos ? -z0_ : z0_;
        const Vec3 n   = pos ? vec3_neg(normal) : normal;

        const float diff[4] = { x0 - x1, y1 - y0, x1 - x0, y0 - y1 };
        const float nzi[4]  = { y0 * diff[0], x1 * diff[1], y1 * diff[2], x0 * diff[3] };
        float nz[4];
        for (int i = 0; i < 4; ++i)
            nz[i] = nzi[i] / igm_sqrt((diff[i] * diff[i]) * (z0 * z0) + nzi[i] * nzi[i]);

        const float g0 = safe_acos(-nz[0] * nz[1]);
        const float g1 = safe_acos(-nz[1] * nz[2]);
        const float g2 = safe_acos(-nz[2] * nz[3]);
        const float g3 = safe_acos(-nz[3] * nz[0]);

        SQ sq;
        sq.o  = from_point;
        sq.n  = n;
        sq.x0 = x0;
        sq.y0 = y0;
        sq.z0 = z0;
        sq.x1 = x1;
        sq.y1 = y1;
        sq.b0 = nz[0];
        sq.b1 = nz[2];
        sq.k  = 2 * flt_pi - g2 - g3;
        sq.s  = g0 + g1 - sq.k;
        return sq;
    }

    // area.art:183-222; returns the sampled point, solid-angle pdf and weight
    void sample_direct(Vec2 uv, Vec3 from_point, Vec3& p, float& pdf_s, float& weight) const
    {
        const SQ sq = compute_sq(from_point);

        const float au = igm_fma(uv.x, sq.s, sq.k);
        const float fu = igm_fma(igm_cos(au), sq.b0, -sq.b1) / igm_sin(au);
        const float cu = clampf(igm_copysign(1.0f, fu) / igm_sqrt(sum_of_prod(fu, fu, sq.b0, sq.b0)), -1, 1);

        const float xu = clampf(-(cu * sq.z0) / igm_sqrt(igm_fma(-cu, cu, 1.0f)), sq.x0, sq.x1);

        const float d   = igm_sqrt(sum_of_prod(xu, xu, sq.z0, sq.z0));
        const float h0  = sq.y0 / igm_sqrt(sum_of_prod(d, d, sq.y0, sq.y0));
        const float h1  = sq.y1 / igm_sqrt(sum_of_prod(d, d, sq.y1, sq.y1));
        const float hv  = igm_fma(uv.y, h1 - h0, h0);
        const float hv2 = hv * hv;
        const float yv  = (hv2 < 1 - 1e-6f) ? (hv * d) / igm_sqrt(1 - hv2) : sq.y1;

        p      = vec3_add(sq.o, vec3_add(vec3_mulf(ex, xu), vec3_add(vec3_mulf(ey, yv), vec3_mulf(sq.n, sq.z0))));
        pdf_s  = safe_div(1, sq.s);
        weight = sq.s;
    }

    float pdf_direct(Vec3 from_point) const { return safe_div(1, compute_sq(from_point).s); } // area.art:224-228
};

// make_area_light.sample_direct (light/area.art:10-26) over the plane emitter
static inline DirectLightSample sample_direct_plane(const ig_light& l, Rng& rnd, const SurfaceElement& from_surf)
{
    const PlaneEmitter pe(l);
    Vec2 uv;
    uv.x = rnd.next_f32();
    uv.y = rnd.next_f32();
    Vec3 p;
    float pdf_s, weight;
    pe.sample_direct(uv, from_surf.point, p, pdf_s, weight);
    const Vec3 dir_  = vec3_sub(p, from_surf.point);
    const float dist = vec3_len(dir_);
    const Vec3 dir   = vec3_mulf(dir_, safe_div(1, dist));
    const float cos  = vec3_dot(dir, pe.normal) * (from_surf.is_entering ? -1.0f : 1.0f);
    DirectLightSample s;
    s.pos          = p;
    s.dir          = dir;
    s.intensity    = color_mulf(pe.radiance, weight);
    s.pdf_value    = pdf_s;
    s.pdf_is_area  = false;
    s.pdf_is_delta = false;
    s.cos          = cos;
    s.dist         = dist;
    return s;
}

// make_point_light.sample_direct (light/point.art:3-8)
// make_spot_light.sample_direct (light/spot.art:8-42)
static inline DirectLightSample sample_direct_spot(const ig_light& l, const SurfaceElement& from_surf)
{
    const Vec3 pos = Vec3{ l.d[0], l.d[1], l.d[2] }, dir = Vec3{ l.d[4], l.d[5], l.d[6] };
    const float cosCutoffAngle = l.d[3], cosFalloffAngle = l.d[7];
    const float blendRange = cosFalloffAngle - cosCutoffAngle;
    const Vec3 out_dir_    = vec3_sub(pos, from_surf.point);
    const float dist       = vec3_len(out_dir_);
    const Vec3 out_dir     = vec3_mulf(out_dir_, safe_div(1, dist));
    const Vec3 to_surf     = vec3_neg(out_dir);
    const float cos_angle  = vec3_dot(to_surf, dir); // eval_dir(vec3_neg(out_dir))
    float factor;
    if (blendRange <= flt_eps) {
        factor = cos_angle <= cosCutoffAngle ? 0.0f : 1.0f;
    } else {
        const float x = clampf((cos_angle - cosCutoffAngle) / blendRange, 0, 1);
        factor        = x * x * (3 - 2 * x); // smoothstep (core/common.art:241)
    }
    DirectLightSample s;
    s.pos          = pos;
    s.dir          = out_dir;
    s.intensity    = Color{ l.d[8] * factor, l.d[9] * factor, l.d[10] * factor };
    s.pdf_value    = vec3_dot(to_surf, dir) > cosCutoffAngle ? 1.0f : 0.0f; // check_valid
    s.pdf_is_area  = true;
    s.pdf_is_delta = false;
    s.cos          = -vec3_dot(out_dir, dir);
    s.dist         = dist;
    return s;
}

// make_directional_light.sample_direct (light/directional.art:6): make_delta_pdf -> value 1 in either measure
static inline DirectLightSample sample_direct_directional(const ig_light& l, const SurfaceElement& from_surf, float scene_radius)
{
    const Vec3 dir = Vec3{ l.d[0], l.d[1], l.d[2] };
    DirectLightSample s;
    s.pos          = vec3_add(from_surf.point, vec3_mulf(dir, -scene_radius));
    s.dir          = vec3_neg(dir);
    s.intensity    = Color{ l.d[4], l.d[5], l.d[6] };
    s.pdf_value    = 1;
    s.pdf_is_area  = false;
    s.pdf_is_delta = true;
    s.cos          = 1;
    s.dist         = scene_radius;
    return s;
}

static inline DirectLightSample sample_direct_point(const ig_light& l, const SurfaceElement& from_surf)
{
    const Vec3 pos   = Vec3{ l.d[0], l.d[1], l.d[2] };
    const Vec3 dir_  = vec3_sub(pos, from_surf.point);
    const float dist = vec3_len(dir_);
    DirectLightSample s;
    s.pos          = pos;
    s.dir          = vec3_mulf(dir_, safe_div(1, dist));
    s.intensity    = Color{ l.d[4], l.d[5], l.d[6] };
    s.pdf_value    = 1;
    s.pdf_is_area  = true;
    s.pdf_is_delta = false;
    s.cos          = 1;
    s.dist         = dist;
    return s;
}

// pdf.as_solid (driver/pdf.art:19-45)
static inline float pdf_as_solid(float value, bool is_area, float cos, float dist2) { return is_area ? value * dist2 / cos : value; }

// make_environment_light -> make_environment_light_function_spherical with a constant colour
// (light/env.art:83-108,161-164); equal_area_square_to_sphere (core/warp.art:63-91)
static inline Vec3 equal_area_square_to_sphere(float px, float py)
{
    const float u  = 2 * px - 1;
    const float v  = 2 * py - 1;
    const float au = igm_abs(u);
    const float av = igm_abs(v);
    const float signedDistance = 1 - (au + av);
    const float d  = igm_abs(signedDistance);
    const float r  = 1 - d;
    const float phi      = (r == 0 ? 1.0f : (av - au) / r + 1) * flt_pi / 4;
    const float cosTheta = igm_copysign(1 - r * r, signedDistance);
    const float sinTheta = safe_sqrt(2 - r * r) * r;
    const float cosPhi   = igm_copysign(igm_cos(phi), u);
    const float sinPhi   = igm_copysign(igm_sin(phi), v);
    return make_vec3(cosPhi * sinTheta, sinPhi * sinTheta, cosTheta);
}

static inline DirectLightSample sample_direct_env(const ig_light& l, Rng& rnd, const SurfaceElement& from_surf, float scene_radius)
{
    const float u   = rnd.next_f32();
    const float v   = rnd.next_f32();
    const Vec3 dir  = equal_area_square_to_sphere(u, v);
    const float pdf = 1 / (4 * flt_pi); // equal_area_sphere_pdf
    DirectLightSample s;
    s.pos          = vec3_add(from_surf.point, vec3_mulf(dir, scene_radius));
    s.dir          = dir;
    s.intensity    = color_mulf(Color{ l.d[0], l.d[1], l.d[2] }, 1 / pdf);
    s.pdf_value    = pdf;
    s.pdf_is_area  = false;
    s.pdf_is_delta = false;
    s.cos          = 1.0f;
    s.dist         = scene_radius;
    return s;
}

// ---- core/cdf.art:43-159 (marginal / conditional tables) and core/interval.art:7-23
static inline int32_t interval_binary_search(int32_t size, const std::function<bool(int32_t)>& pred)
{
    int32_t first = 0, len = size;
    while (len > 0) {
        const int32_t half   = len / 2;
        const int32_t middle = first + half;
        if (pred(middle)) {
            first = middle + 1;
            len -= half + 1;
        } else {
            len = half;
        }
    }
    return std::min(std::max(first - 1, 0), size - 1);
}

// make_cdf_1d over a buffer that omits the leading 0 (cdf.art:43-73)
struct Cdf1D {
    const float* data;
    int32_t func_size;

    float get(int32_t i) const { return i == 0 ? 0.0f : data[i - 1]; }
    float pdf_discrete(int32_t x) const { return get(x + 1) - get(x); }
    int32_t sample_discrete(float u, float& pdf) const
    {
        const int32_t off = std::min(interval_binary_search(func_size + 1, [&](int32_t i) { return get(i) <= u; }), func_size - 1);
        pdf               = pdf_discrete(off);
        return off;
    }
    float pdf_continuous(float x, int32_t& off) const
    {
        off = std::min(std::max((int32_t)(x * (float)func_size), 0), func_size - 1);
        return pdf_discrete(off) * (float)func_size;
    }
    // returns the position in [0, 1]
    float sample_continuous(float u, int32_t& off, float& pdf) const
    {
        float dpdf;
        off             = sample_discrete(u, dpdf);
        const float rem = safe_div(u - get(off), dpdf);
        pdf             = dpdf * (float)func_size;
        return clampf(((float)off + rem) / (float)func_size, 0, 1);
    }
};

// make_cdf_2d_from_buffer (cdf.art:108-159): the marginal (size_y entries) comes first, then one conditional per row
struct Cdf2D {
    const float* data;
    int32_t size_x, size_y;

    Cdf1D marginal() const { return Cdf1D{ data, size_y }; }
    Cdf1D conditional(int32_t row) const { return Cdf1D{ data + size_y + (size_t)row * size_x, size_x }; }
    Vec2 sample_continuous(float ux, float uy, float& pdf) const
    {
        int32_t oy, ox;
        float p1, p2;
        const float py = marginal().sample_continuous(uy, oy, p1);
        const float px = conditional(oy).sample_continuous(ux, ox, p2);
        pdf            = p1 * p2;
        return Vec2{ px, py };
    }
    float pdf_continuous(Vec2 pos) const
    {
        int32_t oy, ox;
        const float p1 = marginal().pdf_continuous(pos.y, oy);
        const float p2 = conditional(oy).pdf_continuous(pos.x, ox);
        return p1 * p2;
    }
};

// ---- light/env.art:11-21,109-157 (make_environment_light_textured), core/warp.art:44-57
static inline Vec3 switch_env_up(Vec3 v) { return make_vec3(v.x, v.z, v.y); }
static inline Vec2 map_env_uv(Vec3 dir)
{
    const float theta = igm_acos(dir.z); // spherical_from_dir
    float phi         = igm_atan2(dir.y, dir.x);
    if (phi < 0)
        phi = phi + 2 * flt_pi;
    const float v = theta / flt_pi;
    const float u = phi / (2 * flt_pi);
    const float r = u + 0.25f;
    return Vec2{ r - igm_floor(r), 1 - v };
}

struct TexturedEnv {
    const igd_scene& sc;
    Color scale;
    Mat3x3 transform;
    const ig_texture* tex;
    Cdf2D cdf;

    TexturedEnv(const igd_scene& scene, const ig_light& l)
        : sc(scene)
    {
        scale = Color{ l.d[0], l.d[1], l.d[2] };
        for (int c = 0; c < 3; ++c)
            transform.col[c] = make_vec3(l.d[3 + c * 3], l.d[4 + c * 3], l.d[5 + c * 3]);
        uint32_t ints[4];
        std::memcpy(ints, &l.d[12], sizeof(ints));
        tex = &sc.textures[ints[0]];
        cdf = Cdf2D{ sc.cdf_data + ints[1], (int32_t)ints[2], (int32_t)ints[3] };
    }
    // sample_dir (env.art:112-123): note that the intensity is the bare texture value, without `scale`
    void sample_dir(Rng& rnd, Vec3& dir, Color& intensity, float& pdf_dir) const
    {
        const float u0 = rnd.next_f32();
        const float u1 = rnd.next_f32();
        if (cdf.size_x == 0) {
            // "cdf": "none" (EnvironmentLight.cpp:60,89-96): make_environment_light (env.art:161-164), i.e. the spherical function
            // environment (:79-93) over scale * tex: uniform directions, `scale` included
            dir       = equal_area_square_to_sphere(u0, u1);
            intensity = emission(dir);
            pdf_dir   = 1 / (4 * flt_pi);
            return;
        }
        float pdf;
        const Vec2 pos    = cdf.sample_continuous(u0, u1, pdf);
        intensity         = image_lookup(sc, *tex, pos);
        const float theta = (1 - pos.y) * flt_pi;
        const float phi   = (pos.x - 0.25f) * 2 * flt_pi;
        const float st = igm_sin(theta), ct = igm_cos(theta);
        const Vec3 d   = make_vec3(st * igm_cos(phi), st * igm_sin(phi), ct); // dir_from_spherical
        const float sinTheta = safe_sqrt(1 - d.z * d.z);                      // shading::sin_theta
        pdf_dir              = safe_div(pdf, sinTheta * flt_pi * flt_pi * 2);
        const Vec3 e         = switch_env_up(d);
        dir = make_vec3(vec3_dot(transform.col[0], e), vec3_dot(transform.col[1], e), vec3_dot(transform.col[2], e)); // mat3x3_left_mul
    }
    Vec3 local_dir(Vec3 ray_dir) const { return switch_env_up(mat3x3_mul(transform, ray_dir)); }
    float pdf(Vec3 ray_dir) const // env.art:125-131
    {
        if (cdf.size_x == 0)
            return 1 / (4 * flt_pi); // equal_area_sphere_pdf (env.art:101)
        const Vec3 ldir      = local_dir(ray_dir);
        const float sinTheta = safe_sqrt(1 - ldir.z * ldir.z);
        return safe_div(cdf.pdf_continuous(map_env_uv(ldir)), sinTheta * flt_pi * flt_pi * 2);
    }
    Color emission(Vec3 ray_dir) const // env.art:145-150
    {
        return color_mul(scale, image_lookup(sc, *tex, map_env_uv(local_dir(ray_dir))));
    }
};

static inline DirectLightSample sample_direct_env_textured(const igd_scene& sc, const ig_light& l, Rng& rnd, const SurfaceElement& from_surf)
{
    const TexturedEnv env(sc, l);
    Vec3 dir;
    Color intensity;
    float pdf_dir;
    env.sample_dir(rnd, dir, intensity, pdf_dir);
    DirectLightSample s;
    s.pos          = vec3_add(from_surf.point, vec3_mulf(dir, sc.scene_radius));
    s.dir          = dir;
    s.intensity    = color_mulf(intensity, 1 / pdf_dir);
    s.pdf_value    = pdf_dir;
    s.pdf_is_area  = false;
    s.pdf_is_delta = false;
    s.cos          = 1.0f;
    s.dist         = sc.scene_radius;
    return s;
}

// ---- make_shape_area_emitter (light/area.art:58-103) under make_area_light (area.art:10-43) with a constant colour:
// a uniformly chosen triangle of the emissive entity's mesh, a uniform point on it
struct MeshEmitter {
    Entity entity;
    TriMeshView mesh;
    Color radiance;

    MeshEmitter(const igd_scene& sc, const ig_light& l)
        : entity(load_entity(sc, l.entity_id))
        , mesh(load_trimesh(sc, entity.shape_id))
        , radiance(Color{ l.d[0], l.d[1], l.d[2] })
    {
    }
    // shape.surface_element_for_point (shapes/trimesh.art:41-68): what the light needs of it
    void surface(int32_t f, float u, float v, Vec3& point, Vec3& face_normal, float& area) const
    {
        const int32_t i0 = mesh.indices[f * 4 + 0], i1 = mesh.indices[f * 4 + 1], i2 = mesh.indices[f * 4 + 2];
        auto vtx = [&](int32_t i) { return mat3x4_transform_point(entity.global_mat, Vec3{ mesh.vertices[i * 4], mesh.vertices[i * 4 + 1], mesh.vertices[i * 4 + 2] }); };
        const Vec3 v0 = vtx(i0), v1 = vtx(i1), v2 = vtx(i2);
        const Vec3 n   = compute_stable_triangle_normal(vec3_sub(v2, v0), vec3_sub(v0, v1), vec3_sub(v1, v2)); // make_triangle (core/triangle.art:12-44)
        const float nn = vec3_len(n);
        face_normal    = vec3_mulf(n, 1 / nn);
        area           = nn / 2;
        point          = vec3_lerp2(v0, v1, v2, u, v);
    }
    // the (triangle, barycentrics) a uv pair addresses (area.art:62-65)
    void address(Vec2 uv, int32_t& f, float& u, float& v) const
    {
        const float ux = uv.x * (float)mesh.num_tris;
        f              = std::min((int32_t)ux, mesh.num_tris - 1);
        const float a = ux - (float)f, b = uv.y;
        if (a + b > 1) // sample_triangle (core/sampling.art:34-36)
            u = 1 - a, v = 1 - b;
        else
            u = a, v = b;
    }
    // pdf_direct(surf.prim_coords, .) (area.art:39,74-82): the reference feeds the hit's barycentrics through the same
    // addressing as a sample's uv, which is what is restated here
    float pdf_area(Vec2 uv) const
    {
        int32_t f;
        float u, v, area;
        Vec3 p, n;
        address(uv, f, u, v);
        surface(f, u, v, p, n, area);
        return safe_div(1, area) / (float)mesh.num_tris;
    }
};

static inline DirectLightSample sample_direct_mesh(const igd_scene& sc, const ig_light& l, Rng& rnd, const SurfaceElement& from_surf)
{
    const MeshEmitter me(sc, l);
    const float ux = rnd.next_f32();
    const float uy = rnd.next_f32();
    int32_t f;
    float u, v, area;
    Vec3 point, face_normal;
    me.address(Vec2{ ux, uy }, f, u, v);
    me.surface(f, u, v, point, face_normal, area);
    const float pdfv   = safe_div(1, area) / (float)me.mesh.num_tris;
    const float weight = area * (float)me.mesh.num_tris;
    const Vec3 dir_    = vec3_sub(point, from_surf.point);
    const float dist   = vec3_len(dir_);
    const Vec3 dir     = vec3_mulf(dir_, safe_div(1, dist));
    DirectLightSample s;
    s.pos          = point;
    s.dir          = dir;
    s.intensity    = color_mulf(me.radiance, weight);
    s.pdf_value    = pdfv;
    s.pdf_is_area  = true;
    s.pdf_is_delta = false;
    s.cos          = vec3_dot(dir, face_normal) * (from_surf.is_entering ? -1.0f : 1.0f);
    s.dist         = dist;
    return s;
}

// make_sphere_area_emitter (light/area.art:259-317) for IG_LIGHT_SPHERE
struct SphereEmitter {
    Entity entity;
    Sphere sphere;
    Color radiance;
    float area, inv_area;
    SphereEmitter(const igd_scene& sc, const ig_light& l)
        : entity(load_entity(sc, l.entity_id))
        , sphere(Sphere{ make_vec3(l.d[0], l.d[1], l.d[2]), l.d[3] })
        , radiance(Color{ l.d[4], l.d[5], l.d[6] })
        , area(l.d[7])
        , inv_area(safe_div(1, l.d[7]))
    {
    }
};

static inline DirectLightSample sample_direct_sphere(const igd_scene& sc, const ig_light& l, Rng& rnd, const SurfaceElement& from_surf)
{
    const SphereEmitter se(sc, l);
    const float ux = rnd.next_f32();
    const float uy = rnd.next_f32();
    const Vec3 glb_org = mat3x4_transform_point(se.entity.global_mat, se.sphere.origin);
    // a uniform point on the sphere, mirrored to the near side when it fell on the far one (area.art:268-294)
    Vec3 point, face_normal;
    sphere_surface_for_normal(se.entity, se.sphere, equal_area_square_to_sphere(ux, uy), point, face_normal);
    const float los = vec3_len2(vec3_sub(from_surf.point, glb_org));
    const float lps = vec3_len2(vec3_sub(from_surf.point, point));
    if (!(lps <= los)) {
        const Vec3 po   = vec3_sub(glb_org, point);
        const Vec3 np   = vec3_add(point, vec3_mulf(po, 2));
        const Vec3 norm = vec3_normalize(vec3_sub(np, glb_org));
        // pmset.to_local_normal (driver/pointmapper.art:31): mat3x3_left_mul(normal_mat, n) / |diag(normal_mat)|^2 — as written
        // there ("TODO: Really? Invalid scaling..."): the result is not a unit vector, so the mirrored point lies INSIDE the
        // sphere (at a third of the radius for an unscaled entity) and its shadow ray is stopped by the sphere itself
        const Mat3x3& m = se.entity.normal_mat;
        const Vec3 diag = make_vec3(m.col[0].x, m.col[1].y, m.col[2].z);
        const Vec3 ln   = vec3_mulf(make_vec3(vec3_dot(norm, m.col[0]), vec3_dot(norm, m.col[1]), vec3_dot(norm, m.col[2])), 1 / vec3_len2(diag));
        sphere_surface_for_normal(se.entity, se.sphere, ln, point, face_normal);
    }
    const Vec3 dir_  = vec3_sub(point, from_surf.point);
    const float dist = vec3_len(dir_);
    const Vec3 dir   = vec3_mulf(dir_, safe_div(1, dist));
    DirectLightSample s;
    s.pos          = point;
    s.dir          = dir;
    s.intensity    = color_mulf(se.radiance, se.area);
    s.pdf_value    = se.inv_area;
    s.pdf_is_area  = true;
    s.pdf_is_delta = false;
    s.cos          = vec3_dot(dir, face_normal) * (from_surf.is_entering ? -1.0f : 1.0f);
    s.dist         = dist;
    return s;
}

// ---- light/cie.art:1-41: CIE sky radiance functions (direction in the light's Y-up frame)
struct CieSky {
    int kind;
    bool has_ground;
    Color zenith, ground, scale;
    float ground_brightness, zenith_brightness, c2;
    Vec3 sun_dir;
    Mat3x3 transform;

    explicit CieSky(const ig_light& l)
    {
        kind              = l.pad[0];
        has_ground        = l.pad[1] != 0;
        zenith            = Color{ l.d[0], l.d[1], l.d[2] };
        ground            = Color{ l.d[3], l.d[4], l.d[5] };
        ground_brightness = l.d[6];
        zenith_brightness = l.d[7];
        c2                = l.d[8];
        sun_dir           = make_vec3(l.d[9], l.d[10], l.d[11]);
        scale             = Color{ l.d[12], l.d[13], l.d[14] };
        for (int c = 0; c < 3; ++c)
            transform.col[c] = make_vec3(l.d[15 + c * 3], l.d[16 + c * 3], l.d[17 + c * 3]);
    }
    // cie_wmean (cie.art:1-7); pow(x, 10) as x^8 * x^2 (exact products instead of libm powf)
    static Color wmean(float cos_theta, Color c1, Color c2)
    {
        const float x  = cos_theta + 1.01f;
        const float x2 = x * x;
        const float x4 = x2 * x2;
        const float a  = (x4 * x4) * x2;
        const float f1 = a * a / (a * a + 1);
        const float f2 = 1 / (a * a + 1);
        return color_add(color_mulf(c1, f1), color_mulf(c2, f2));
    }
    // perez::eval (perez.art:235-242)
    static float perez_eval(float cos_sun, float cos_theta, const float p[5])
    {
        const float sun_a = igm_acos(cos_sun);
        const float A     = 1 + p[0] * igm_exp(p[1] / igm_max(1e-5f, cos_theta));
        const float B     = 1 + p[2] * igm_exp(p[3] * sun_a) + p[4] * cos_sun * cos_sun;
        return A * B;
    }
    Color radiance(Vec3 dir) const
    {
        const float cos_theta = dir.y;
        if (kind == IG_CIE_PEREZ) {
            // sky_function of make_perez_light_raw (perez.art:293-298): (a, b, c) sit in the three brightness slots, (d, e) in `scale`
            const float p[5]  = { ground_brightness, zenith_brightness, c2, scale.r, scale.g };
            const float sun_c = clampf(vec3_dot(dir, sun_dir), -1, 1);
            return wmean(cos_theta, color_mulf(zenith, perez_eval(sun_c, cos_theta, p)), ground);
        }
        if (!has_ground && cos_theta < 0)
            return Color{ 0, 0, 0 };
        if (kind == IG_CIE_UNIFORM || kind == IG_CIE_CLOUDY) {
            // make_cie_sky_light (cie.art:13-21)
            const bool cloudy = kind == IG_CIE_CLOUDY;
            const float c1    = cloudy ? (1 + 2 * cos_theta) / 3 : 1.0f;
            const float k2    = cloudy ? 0.777777777f : 1.0f;
            return wmean(cos_theta, color_mulf(zenith, c1), color_mulf(ground, ground_brightness * k2));
        }
        // make_cie_sunny_light (cie.art:24-41)
        const float cos_gamma = vec3_dot(dir, sun_dir);
        const float gamma     = igm_acos(clampf(cos_gamma, -1, 1));
        float c1;
        if (kind == IG_CIE_CLEAR) {
            c1 = (0.91f + 10 * igm_exp(-3 * gamma) + 0.45f * cos_gamma * cos_gamma) * (cos_theta >= 0.01f ? 1 - igm_exp(-0.32f / cos_theta) : 1.0f);
        } else {
            const float theta  = igm_acos(clampf(cos_theta, -1, 1));
            const float stheta = igm_acos(clampf(sun_dir.y, -1, 1));
            c1 = ((1.35f * igm_sin(5.631f - 3.59f * theta) + 3.12f) * igm_sin(4.396f - 2.6f * stheta) + 6.37f - theta) / 2.326f
                 * igm_exp(gamma * (-0.563f) * ((2.629f - theta) * (1.562f - stheta) + 0.812f));
        }
        return color_mul(scale, wmean(cos_theta, color_mulf(zenith, zenith_brightness * c1), color_mulf(ground, ground_brightness * c2)));
    }
    // emission / pdf_direct of make_environment_light_function_{hemi,spherical} (env.art:24-105); half = !has_ground
    Color emission(Vec3 ray_dir) const
    {
        const Vec3 local_dir = mat3x3_mul(transform, ray_dir);
        if (!has_ground)
            return local_dir.y > flt_eps ? radiance(local_dir) : Color{ 0, 0, 0 };
        return radiance(local_dir);
    }
    float pdf(Vec3 ray_dir) const
    {
        if (has_ground)
            return 1 / (4 * flt_pi);
        const Vec3 local_dir = mat3x3_mul(transform, ray_dir);
        return local_dir.y > flt_eps ? cosine_hemisphere_pdf(local_dir.y) : 0.0f;
    }
};

static inline DirectLightSample sample_direct_cie(const igd_scene& sc, const ig_light& l, Rng& rnd, const SurfaceElement& from_surf)
{
    const CieSky sky(l);
    const float u = rnd.next_f32();
    const float v = rnd.next_f32();
    DirectLightSample s;
    if (!sky.has_ground) {
        // hemi: cosine sample around Y, radiance looked up with the untransformed direction (env.art:31-37)
        const DirSample ds = sample_cosine_hemisphere(u, v);
        const Vec3 dir     = switch_env_up(ds.dir);
        s.intensity        = color_mulf(sky.radiance(dir), 1 / ds.pdf);
        s.dir              = make_vec3(vec3_dot(sky.transform.col[0], dir), vec3_dot(sky.transform.col[1], dir), vec3_dot(sky.transform.col[2], dir));
        s.pdf_value        = ds.pdf;
    } else {
        // spherical (env.art:79-93)
        const Vec3 dir  = equal_area_square_to_sphere(u, v);
        const float pdf = 1 / (4 * flt_pi);
        s.intensity     = color_mulf(sky.radiance(mat3x3_mul(sky.transform, dir)), 1 / pdf);
        s.dir           = dir;
        s.pdf_value     = pdf;
    }
    s.pos          = vec3_add(from_surf.point, vec3_mulf(s.dir, sc.scene_radius));
    s.pdf_is_area  = false;
    s.pdf_is_delta = false;
    s.cos          = 1.0f;
    s.dist         = sc.scene_radius;
    return s;
}

// ---- light/sun.art:8-48 (make_sun_light, not handled as delta), core/sampling.art:106-116
struct SunLight {
    Vec3 dir; // scene to light
    float cos_angle;
    Color radiance;
    explicit SunLight(const ig_light& l)
        : dir(make_vec3(l.d[0], l.d[1], l.d[2]))
        , cos_angle(l.d[3])
        , radiance(Color{ l.d[4], l.d[5], l.d[6] })
    {
    }
    float dir_pdf() const { return safe_div(1, 2 * flt_pi * (1 - cos_angle)); } // uniform_cone_pdf
    bool hits(Vec3 towards_light) const { return vec3_dot(dir, towards_light) >= cos_angle; }
};

// the sun of make_perez_light_raw (perez.art:301-317): direction in scene space d[27..29], half-angle cosine d[14], radiance d[24..26]
static inline SunLight perez_sun(const ig_light& l)
{
    ig_light s = l;
    s.d[0] = l.d[27], s.d[1] = l.d[28], s.d[2] = l.d[29];
    s.d[3] = l.d[14];
    s.d[4] = l.d[24], s.d[5] = l.d[25], s.d[6] = l.d[26];
    return SunLight(s);
}

static inline DirectLightSample sample_direct_sun(const SunLight& sun, Rng& rnd)
{
    const float u  = rnd.next_f32();
    const float v  = rnd.next_f32();
    const float c1 = 1 - sun.cos_angle; // sample_uniform_cone
    float px, py;
    square_to_concentric_disk(u, v, px, py);
    const float n2 = px * px + py * py;
    const float z  = sun.cos_angle + c1 * (1 - n2);
    const float k  = safe_sqrt(c1 * (2 - c1 * n2));
    const Vec3 local = make_vec3(px * k, py * k, z);
    const Vec3 ndir  = mat3x3_mul(make_orthonormal_mat3x3(vec3_neg(sun.dir)), local);
    const float inv_pdf = 2 * flt_pi * (1 - sun.cos_angle);
    DirectLightSample s;
    s.pos          = make_vec3(0, 0, 0);
    s.dir          = vec3_neg(ndir);
    s.intensity    = color_mulf(sun.radiance, inv_pdf);
    s.pdf_value    = sun.dir_pdf();
    s.pdf_is_area  = false;
    s.pdf_is_delta = false;
    s.cos          = local.z;
    s.dist         = std::numeric_limits<float>::infinity();
    return s;
}

// ---- light/light_hierarchy.art:14-96 over the table built by the host (LightHierarchy.cpp)
struct HierEntry {
    Vec3 pos, dir;
    float flux;
    int32_t id;
    bool has_dir, is_leaf;
};
static inline HierEntry hier_load(const igd_scene& sc, int32_t id)
{
    const float* e = sc.light_hierarchy + (size_t)id * 8;
    int32_t index;
    std::memcpy(&index, e + 7, 4);
    HierEntry h;
    h.pos     = Vec3{ e[0], e[1], e[2] };
    h.dir     = Vec3{ e[4], e[5], e[6] };
    h.flux    = igm_abs(e[3]);
    h.id      = index < 0 ? -index - 1 : index;
    h.has_dir = !igm_signbit(e[3]);
    h.is_leaf = index >= 0;
    return h;
}
static inline float hier_cost(const HierEntry& e, Vec3 pos)
{
    const Vec3 cdir   = vec3_sub(e.pos, pos);
    const float dist2 = vec3_len2(cdir);
    const float cos_d = e.has_dir ? igm_abs(vec3_dot(e.dir, vec3_normalize(cdir))) : 1.0f;
    return safe_div(e.flux * cos_d, dist2);
}
static inline float hier_left_prop(const HierEntry& l, const HierEntry& r, Vec3 pos)
{
    const float cl = hier_cost(l, pos);
    const float cr = hier_cost(r, pos);
    return 1 / (1 + cr / cl);
}
static inline int32_t hier_sample(const igd_scene& sc, Rng& rnd, Vec3 pos, float& pdf)
{
    pdf           = 1;
    HierEntry ent = hier_load(sc, 0);
    while (!ent.is_leaf) {
        const HierEntry left  = hier_load(sc, ent.id);
        const HierEntry right = hier_load(sc, ent.id + 1);
        const float prop      = hier_left_prop(left, right, pos);
        const bool is_left    = rnd.next_f32() < prop;
        ent                   = is_left ? left : right;
        pdf *= is_left ? prop : 1 - prop;
    }
    return ent.id;
}
static inline float hier_pdf(const igd_scene& sc, int32_t finite_id, Vec3 pos)
{
    uint32_t code = sc.light_codes[finite_id];
    float pdf     = 1;
    HierEntry ent = hier_load(sc, 0);
    while (!ent.is_leaf) {
        const HierEntry left  = hier_load(sc, ent.id);
        const HierEntry right = hier_load(sc, ent.id + 1);
        const float prop      = hier_left_prop(left, right, pos);
        const bool is_left    = (code & 0x1) == 0;
        ent                   = is_left ? left : right;
        pdf *= is_left ? prop : 1 - prop;
        code >>= 1;
    }
    return pdf;
}

// ---- technique/pathtracer.art
struct PTRayPayload {
    float inv_pdf;
    Color contrib;
    int32_t depth;
    float eta;
    int32_t medium; // VPTRayPayload (volpathtracer.art:1-7); the path tracer does not use it
};

// ---- medium/homogeneous.art:1-58, driver/medium.art (make_vacuum_medium), phase/henyeygreenstein.art
struct MediumSampleOut {
    Vec3 pos;
    Color color; // transmittance / pdf
};
struct Medium {
    bool vacuum     = true;
    bool scattering = false;
    Color sigma_t{ 0, 0, 0 };
    int sigma_ind   = 0;
    float sigma_t_p = 0;
    float g         = 0;

    Medium() = default;
    explicit Medium(const ig_medium& m)
    {
        vacuum = m.type == IG_MEDIUM_VACUUM;
        if (vacuum)
            return;
        const Color sigma_a{ m.sigma_a[0], m.sigma_a[1], m.sigma_a[2] }, sigma_s{ m.sigma_s[0], m.sigma_s[1], m.sigma_s[2] };
        sigma_t    = color_add(sigma_a, sigma_s);
        scattering = !(igm_abs(sigma_s.r) <= 1e-4f && igm_abs(sigma_s.g) <= 1e-4f && igm_abs(sigma_s.b) <= 1e-4f); // is_black_eps(sigma_s, 1e-4)
        // vec3_min_index (core/vector.art:110)
        sigma_ind = sigma_t.r < sigma_t.g ? (sigma_t.r < sigma_t.b ? 0 : 2) : (sigma_t.g < sigma_t.b ? 1 : 2);
        sigma_t_p = sigma_ind == 0 ? sigma_t.r : (sigma_ind == 1 ? sigma_t.g : sigma_t.b);
        g         = m.g;
    }
    Color eval_tr(float t) const { return Color{ igm_exp(-sigma_t.r * t), igm_exp(-sigma_t.g * t), igm_exp(-sigma_t.b * t) }; }
    Color eval(Vec3 p_start, Vec3 p_end) const
    {
        if (vacuum)
            return Color{ 1, 1, 1 };
        return eval_tr(vec3_len(vec3_sub(p_end, p_start)));
    }
    Color eval_inf() const
    {
        if (vacuum)
            return Color{ 1, 1, 1 };
        if (!scattering) {
            const bool clear = igm_abs(sigma_t.r) <= 1e-4f && igm_abs(sigma_t.g) <= 1e-4f && igm_abs(sigma_t.b) <= 1e-4f;
            return clear ? Color{ 1, 1, 1 } : Color{ 0, 0, 0 };
        }
        return Color{ 0, 0, 0 };
    }
    // sample (homogeneous.art:38-52); vacuum and non-scattering media reject without touching the generator
    bool sample(Rng& rnd, Vec3 p_start, Vec3 p_end, MediumSampleOut& out) const
    {
        if (vacuum || !scattering)
            return false;
        const float eps   = 1e-3f;
        const Vec3 dir_u  = vec3_sub(p_end, p_start);
        const float dist  = vec3_len(dir_u);
        const float ndist = igm_min(dist, -igm_log(1 - rnd.next_f32() * 0.99999f) / sigma_t_p);
        if (igm_abs(dist - ndist) <= eps)
            return false;
        const Vec3 dir  = vec3_mulf(dir_u, safe_div(1, dist));
        out.pos         = vec3_add(p_start, vec3_mulf(dir, ndist));
        const Color tr  = eval_tr(ndist);
        const float pdf = (sigma_ind == 0 ? tr.r : (sigma_ind == 1 ? tr.g : tr.b)) * sigma_t_p;
        out.color       = Color{ tr.r / pdf, tr.g / pdf, tr.b / pdf };
        return true;
    }
    // make_henyeygreenstein_phase(g).sample (henyeygreenstein.art:20-38): weight 1; the anisotropic branch returns the direction
    // in the sampling frame itself (it is not rotated about out_dir), as written there
    Vec3 sample_phase(Rng& rnd) const
    {
        if (igm_abs(g) <= 1e-3f) {
            const float u = rnd.next_f32();
            const float v = rnd.next_f32();
            const float c = 2 * v - 1; // sample_uniform_sphere (core/sampling.art:42-47)
            const float sn = safe_sqrt(1 - c * c);
            const float phi = 2 * flt_pi * u;
            return make_vec3(sn * igm_cos(phi), sn * igm_sin(phi), c);
        }
        const float sqr_term  = (1 - g * g) / (1 + g - 2 * g * rnd.next_f32());
        const float cos_theta = -(1 + g * g - sqr_term * sqr_term) / (2 * g);
        const float sin_theta = igm_sqrt(igm_max(0.0f, 1 - cos_theta * cos_theta));
        const float phi       = 2 * flt_pi * rnd.next_f32();
        return make_vec3(sin_theta * igm_cos(phi), sin_theta * igm_sin(phi), cos_theta);
    }
};

static inline float russian_roulette_pbrt(Color c, float clamp) { return clampf(color_max_component(c), 0.05f, clamp); } // pathtracer.art:5

struct ShadowRayOut {
    bool valid;
    Ray ray;
    Color color;
};

struct PathTracer {
    const igd_scene& sc;
    int32_t max_path_len, min_path_len;
    float clamp_value;
    bool enable_nee;
    static constexpr float offset = 0.001f;

    explicit PathTracer(const igd_scene& s)
        : sc(s)
        , max_path_len(s.technique.max_depth)
        , min_path_len(s.technique.min_depth)
        , clamp_value(s.technique.clamp)
        , enable_nee(s.technique.nee != 0)
        , ambient_occlusion(s.technique.type == IG_TECHNIQUE_AO)
        , volumetric(s.technique.type == IG_TECHNIQUE_VOLPATH)
        , debug(s.technique.type == IG_TECHNIQUE_DEBUG)
        , wireframe(s.technique.type == IG_TECHNIQUE_WIREFRAME)
    {
    }
    bool debug; // make_debug_renderer: only on_hit
    // make_wireframe_renderer (technique/wireframe.art:21-73): on_hit and on_bounce; footprint_u = |dx x dy| of camera.differential
    // (perspective: (right scale.x, up scale.y), orthogonal: (right, up)), set by the caller that knows the film's scale
    bool wireframe;
    float wire_footprint = 0;
    // is_edge_hit (wireframe.art:24-31); the payload's distance travels in PTRayPayload.inv_pdf
    bool is_edge_hit(const Hit& hit, const SurfaceElement& surf, float add_distance, float& edge_t) const
    {
        const float w  = clampf(0, 1, 1 - hit.u - hit.v); // clampf(v = 0, l = 1, u = ...) as written: min(1 - u - v, 1)
        edge_t         = igm_min(hit.u, igm_min(hit.v, w)); // vec3_min_value
        const float fp = (hit.distance + add_distance) * wire_footprint;
        const float cond = 0.01f * fp * igm_sqrt(surf.inv_area);
        return edge_t <= cond;
    }
    // make_volume_path_renderer (technique/volpathtracer.art:37-260): the same callbacks with a current medium
    bool volumetric;
    // get_medium (volpathtracer.art:46-49) over the media table of LoaderMedium::generate (LoaderMedium.cpp:89-111): unknown ids are vacuum
    Medium get_medium(int32_t id) const { return (volumetric && id >= 0 && (uint32_t)id < sc.media_count) ? Medium(sc.media[id]) : Medium(); }
    // make_ao_renderer (technique/aotracer.art:1-24): only on_shadow does anything
    bool ambient_occlusion;

    Color handle_color(Color c) const { return clamp_value > 0 ? color_saturate(c, clamp_value) : c; }

    int32_t n_inf() const { return (int32_t)sc.infinite_light_count; }
    int32_t n_fin() const { return (int32_t)(sc.light_count - sc.infinite_light_count); }
    bool use_hierarchy() const { return sc.technique.light_selector == IG_SELECTOR_HIERARCHY && n_fin() > 0 && sc.light_hierarchy != nullptr; }

    // pick_light_id (light/light_selector.art:18-24)
    static int32_t pick_light_id(Rng& rnd, int32_t num_lights) { return num_lights <= 1 ? 0 : rnd.next_i32(0, num_lights - 1); }

    // LightSelector::sample: make_uniform_light_selector (light_selector.art:26-46) or
    // make_hierarchy_light_selector (:80-110) + make_light_hierarchy (light_hierarchy.art:103-123).
    // Returns the index into sc.lights.
    bool use_cdf() const { return sc.technique.light_selector == IG_SELECTOR_SIMPLE && n_fin() > 0 && sc.light_cdf != nullptr; }
    // make_cdf_light_selector (light_selector.art:48-78)
    int32_t select_light_cdf(Rng& rnd, float& pdf) const
    {
        const Cdf1D cdf{ sc.light_cdf, n_fin() };
        if (n_inf() == 0)
            return cdf.sample_discrete(rnd.next_f32(), pdf);
        const float pdf_infinite = 1 / (float)n_inf();
        const float ratio        = 0.5f;
        const float q            = rnd.next_f32();
        if (q < ratio) {
            const int32_t id = pick_light_id(rnd, n_inf());
            pdf              = pdf_infinite * ratio;
            return id;
        }
        float p;
        const int32_t fid = cdf.sample_discrete(rnd.next_f32(), p);
        pdf               = p * (1 - ratio);
        return n_inf() + fid;
    }

    int32_t select_light(Rng& rnd, Vec3 from_pos, float& pdf) const
    {
        if (use_cdf())
            return select_light_cdf(rnd, pdf);
        if (!use_hierarchy()) {
            const int32_t num = (int32_t)sc.light_count;
            pdf               = num == 0 ? 1.0f : 1 / (float)num;
            return pick_light_id(rnd, num);
        }
        auto hsample = [&](float& p) -> int32_t {
            if (n_fin() == 1) {
                p = 1;
                return 0;
            }
            return hier_sample(sc, rnd, from_pos, p);
        };
        if (n_inf() == 0) {
            const int32_t fid = hsample(pdf);
            return fid;
        }
        const float pdf_infinite = 1 / (float)n_inf();
        const float ratio        = 0.5f;
        const float q            = rnd.next_f32();
        if (q < ratio) {
            const int32_t id = pick_light_id(rnd, n_inf());
            pdf              = pdf_infinite * ratio;
            return id;
        }
        float p;
        const int32_t fid = hsample(p);
        pdf               = p * (1 - ratio);
        return n_inf() + fid;
    }

    // LightSelector::pdf for light index `li` seen from `from_pos`
    float select_pdf(int32_t li, Vec3 from_pos) const
    {
        if (use_cdf()) {
            const Cdf1D cdf{ sc.light_cdf, n_fin() };
            if (n_inf() == 0)
                return cdf.pdf_discrete(li);
            return li < n_inf() ? (1 / (float)n_inf()) * 0.5f : cdf.pdf_discrete(li - n_inf()) * (1 - 0.5f);
        }
        if (!use_hierarchy())
            return sc.light_count == 0 ? 1.0f : 1 / (float)sc.light_count;
        const bool infinite = li < n_inf();
        auto hpdf           = [&]() { return n_fin() == 1 ? 1.0f : hier_pdf(sc, li - n_inf(), from_pos); };
        if (n_inf() == 0)
            return hpdf();
        const float ratio = 0.5f;
        if (infinite)
            return (1 / (float)n_inf()) * ratio;
        return hpdf() * (1 - ratio);
    }

    // on_shadow (pathtracer.art:52-117)
    ShadowRayOut on_shadow(const Ray& ray, const SurfaceElement& surf, Rng& rnd, const PTRayPayload& pt, const Bsdf& bsdf) const
    {
        if (debug || wireframe) {
            ShadowRayOut none;
            none.valid = false;
            return none;
        }
        if (ambient_occlusion) {
            // a sample of make_lambertian_bsdf(ctx.surf, white) (bsdf/diffuse.art:2-12): colour = kd, whatever the direction
            ShadowRayOut ao;
            const float u       = rnd.next_f32();
            const float v       = rnd.next_f32();
            const DirSample smp = sample_cosine_hemisphere(u, v);
            ao.valid            = true;
            ao.color            = Color{ 1, 1, 1 };
            ao.ray              = make_ray(surf.point, mat3x3_mul(surf.local, smp.dir), offset, flt_max, IG_RAY_FLAG_BOUNCE);
            return ao;
        }
        ShadowRayOut out;
        out.valid = false;
        if (!enable_nee)
            return out;
        if (bsdf.is_all_delta() || sc.light_count == 0)
            return out;
        if (pt.depth + 1 > max_path_len)
            return out;

        float light_select_pdf;
        const int id          = select_light(rnd, surf.point, light_select_pdf);
        const ig_light& light = sc.lights[id];

        DirectLightSample ls;
        bool delta = false, infinite = false;
        switch (light.type) {
        case IG_LIGHT_PLANE:
            ls = sample_direct_plane(light, rnd, surf);
            break;
        case IG_LIGHT_POINT:
            ls    = sample_direct_point(light, surf);
            delta = true;
            break;
        case IG_LIGHT_SPOT:
            ls    = sample_direct_spot(light, surf);
            delta = true;
            break;
        case IG_LIGHT_DIRECTIONAL:
            ls       = sample_direct_directional(light, surf, sc.scene_radius);
            delta    = true;
            infinite = true;
            break;
        case IG_LIGHT_ENV_TEXTURED:
            ls       = sample_direct_env_textured(sc, light, rnd, surf);
            infinite = true;
            break;
        case IG_LIGHT_SUN:
            ls       = sample_direct_sun(SunLight(light), rnd);
            infinite = true;
            break;
        case IG_LIGHT_PEREZ: {
            // sample_direct of make_perez_light_raw (perez.art:304-308): the sun's sample, carrying the sky seen in its direction too
            const CieSky sky(light);
            ls           = sample_direct_sun(perez_sun(light), rnd);
            const Vec3 d = make_vec3(vec3_dot(sky.transform.col[0], ls.dir), vec3_dot(sky.transform.col[1], ls.dir), vec3_dot(sky.transform.col[2], ls.dir));
            ls.intensity = color_add(ls.intensity, color_mulf(sky.radiance(d), 1 / ls.pdf_value));
            infinite     = true;
            break;
        }
        case IG_LIGHT_CIE:
            ls       = sample_direct_cie(sc, light, rnd, surf);
            infinite = true;
            break;
        case IG_LIGHT_MESH_AREA:
            ls = sample_direct_mesh(sc, light, rnd, surf);
            break;
        case IG_LIGHT_SPHERE:
            ls = sample_direct_sphere(sc, light, rnd, surf);
            break;
        default:
            ls       = sample_direct_env(light, rnd, surf, sc.scene_radius);
            infinite = true;
            break;
        }

        const float pdf_l_s = pdf_as_solid(ls.pdf_value, ls.pdf_is_area, ls.cos, ls.dist * ls.dist) * light_select_pdf;
        if (pdf_l_s <= flt_eps)
            return out;

        const Vec3 in_dir  = ls.dir;
        const Vec3 out_dir = vec3_neg(ray.dir);

        // volpathtracer.art:39-41: the last interaction was a medium one; transmittance from the ray origin to this hit
        const bool was_medium_interaction = volumetric && igm_signbit(pt.inv_pdf);
        const Medium medium               = get_medium(pt.medium);
        const Color hitvol                = medium.eval(ray.org, surf.point);

        if (ls.cos > flt_eps) {
            float mis;
            if (delta || was_medium_interaction) {
                mis = 1;
            } else {
                const float pdf_e_s = bsdf.pdf(in_dir, out_dir);
                mis                 = 1 / (1 + pdf_e_s / pdf_l_s);
            }
            const float factor  = ls.pdf_value / pdf_l_s;
            const Color contrib = handle_color(color_mulf(color_mul(ls.intensity, color_mul(pt.contrib, bsdf.eval(in_dir, out_dir))), mis * factor));
            if (color_average(contrib) <= flt_eps)
                return out;

            out.valid = true;
            out.color = contrib;
            if (infinite)
                out.ray = make_ray(surf.point, in_dir, offset, flt_max, IG_RAY_FLAG_SHADOW);
            else
                out.ray = make_ray(surf.point, vec3_sub(ls.pos, surf.point), offset, 1 - offset, IG_RAY_FLAG_SHADOW);
            if (volumetric) // volpathtracer.art:71-83: the medium the path is in, up to the hit and on towards the light
                out.color = color_mul(contrib, color_mul(hitvol, infinite ? medium.eval_inf() : medium.eval(surf.point, ls.pos)));
        }
        return out;
    }

    // colormap::palette (core/colormap.art:68-92)
    static Color palette(int32_t i)
    {
        static const float c[23][3] = {
            { 0.450000f, 0.376630f, 0.112500f }, { 0.112500f, 0.450000f, 0.405978f }, { 0.112500f, 0.450000f, 0.229891f }, { 0.450000f, 0.112500f, 0.376630f },
            { 0.435326f, 0.450000f, 0.112500f }, { 0.112500f, 0.141848f, 0.450000f }, { 0.435326f, 0.112500f, 0.450000f }, { 0.112500f, 0.450000f, 0.141848f },
            { 0.347283f, 0.450000f, 0.112500f }, { 0.450000f, 0.112500f, 0.200543f }, { 0.112500f, 0.229891f, 0.450000f }, { 0.450000f, 0.288587f, 0.112500f },
            { 0.347283f, 0.112500f, 0.450000f }, { 0.450000f, 0.112500f, 0.288587f }, { 0.450000f, 0.112500f, 0.112500f }, { 0.450000f, 0.200543f, 0.112500f },
            { 0.171196f, 0.450000f, 0.112500f }, { 0.112500f, 0.450000f, 0.317935f }, { 0.259239f, 0.450000f, 0.112500f }, { 0.259239f, 0.112500f, 0.450000f },
            { 0.112500f, 0.405978f, 0.450000f }, { 0.171196f, 0.112500f, 0.450000f }, { 0.112500f, 0.317935f, 0.450000f }
        };
        const int k = i % 23;
        return Color{ c[k][0], c[k][1], c[k][2] };
    }
    // on_hit of make_debug_renderer (technique/debugtracer.art:3-140); the other callbacks do nothing
    Color debug_hit(const Ray& ray, const Hit& hit, const SurfaceElement& surf, const Entity& entity, const Bsdf& bsdf, const ig_material& mat, int32_t mat_id) const
    {
        auto absv    = [](Vec3 n) { return Color{ igm_abs(n.x), igm_abs(n.y), igm_abs(n.z) }; };
        // to_local_normal of make_standard_pointmapperset (driver/pointmapper.art:31)
        auto local_n = [&](Vec3 n) {
            const Vec3 d = make_vec3(entity.normal_mat.col[0].x, entity.normal_mat.col[1].y, entity.normal_mat.col[2].z);
            const Vec3 v = make_vec3(vec3_dot(entity.normal_mat.col[0], n), vec3_dot(entity.normal_mat.col[1], n), vec3_dot(entity.normal_mat.col[2], n));
            return vec3_normalize(vec3_mulf(v, 1 / vec3_dot(d, d)));
        };
        const Color yes{ 0, 0, 1 }, no{ 1, 0, 0 }; // true_color = blue, false_color = red (core/color.art:74-75)
        const int32_t inner = (mat.pad[2] & 0xFFFF) - 1, outer = ((mat.pad[2] >> 16) & 0xFFFF) - 1;
        switch (sc.technique.debug_mode) {
        case 1: return absv(surf.local.col[0]);
        case 2: return absv(surf.local.col[1]);
        case 3: return absv(surf.face_normal);
        case 4: return absv(local_n(surf.local.col[2]));
        case 5: return absv(local_n(surf.local.col[0]));
        case 6: return absv(local_n(surf.local.col[1]));
        case 7: return absv(local_n(surf.face_normal));
        case 8: return Color{ igm_abs(surf.tex_coords.x), igm_abs(surf.tex_coords.y), 0 };
        case 9: return Color{ igm_abs(hit.u), igm_abs(hit.v), 0 };
        case 10: return Color{ surf.point.x, surf.point.y, surf.point.z };
        case 11: {
            const Vec3 p = mat3x4_transform_point(entity.local_mat, surf.point);
            return Color{ p.x, p.y, p.z };
        }
        case 12: { // make_normalized_pointmapper (pointmapper.art:4-8) over the shape's bounding box (trimesh.art:89, sphere.art:74)
            const Vec3 lp = mat3x4_transform_point(entity.local_mat, surf.point);
            Vec3 lo, hi;
            const uint8_t* base = sc.shape_data + sc.shape_lookups[entity.shape_id].offset;
            if (sc.shape_lookups[entity.shape_id].type_id == IG_SHAPE_SPHERE) {
                float d[4];
                std::memcpy(d, base, 16); // origin, radius: make_centered_bbox(origin, 2 * radius)
                lo = make_vec3(d[0] - d[3], d[1] - d[3], d[2] - d[3]);
                hi = make_vec3(d[0] + d[3], d[1] + d[3], d[2] + d[3]);
            } else {
                float d[12];
                std::memcpy(d, base, 48);
                lo = make_vec3(d[4], d[5], d[6]);
                hi = make_vec3(d[8], d[9], d[10]);
            }
            return Color{ safe_div(lp.x - lo.x, hi.x - lo.x), safe_div(lp.y - lo.y, hi.y - lo.y), safe_div(lp.z - lo.z, hi.z - lo.z) };
        }
        case 13: return Color{ hit.distance, hit.distance, hit.distance };
        case 14: {
            float area = surf.area;
            if (sc.shape_lookups[entity.shape_id].type_id == IG_SHAPE_SPHERE) {
                // compute_ellipsoid_area (shapes/sphere.art:21-28), which the sphere's surface element carries
                const Sphere sp = load_sphere(sc, entity.shape_id);
                const float l1  = vec3_len2(vec3_mulf(entity.global_mat.col[0], sp.radius));
                const float l2  = vec3_len2(vec3_mulf(entity.global_mat.col[1], sp.radius));
                const float l3  = vec3_len2(vec3_mulf(entity.global_mat.col[2], sp.radius));
                const float P   = 1.6f;
                area = 4 * flt_pi * igm_pow((igm_pow(l1 * l2, P / 2) + igm_pow(l1 * l3, P / 2) + igm_pow(l2 * l3, P / 2)) / 3, 1 / P);
            }
            return Color{ area, area, area };
        }
        case 15: return Color{ (float)hit.prim_id, (float)hit.prim_id, (float)hit.prim_id };
        case 16: return palette(hit.prim_id);
        case 17: return Color{ (float)hit.ent_id, (float)hit.ent_id, (float)hit.ent_id };
        case 18: return palette(hit.ent_id);
        case 19: return Color{ (float)mat_id, (float)mat_id, (float)mat_id };
        case 20: return palette(mat_id);
        case 21: return mat.light_id >= 0 ? yes : no;
        case 22: return bsdf.is_all_delta() ? yes : no;
        case 23: return surf.is_entering ? yes : no;
        case 24: { // DEBUG_CHECK_BSDF: red / orange / yellow / blue = neither / only the pdf / only the weight / both agree; pink = no sample
            const Color verdict[4] = { Color{ 1, 0, 0 }, Color{ 1, 0.5f, 0 }, Color{ 1, 1, 0 }, Color{ 0, 0, 1 } };
            const Vec3 out_dir     = vec3_neg(ray.dir);
            if (bsdf.is_all_delta()) {
                const Vec3 r      = vec3_reflect(out_dir, surf.local.col[2]);
                const Color evl   = bsdf.eval(r, out_dir);
                const float pdf   = bsdf.pdf(r, out_dir);
                const int pdf_ok  = igm_abs(0 - pdf) <= flt_eps ? 1 : 0;
                const int w_ok    = igm_abs(0 - evl.r) + igm_abs(0 - evl.g) + igm_abs(0 - evl.b) <= flt_eps ? 1 : 0;
                return verdict[(w_ok << 1) | pdf_ok];
            }
            Rng tmp{ hash_combine(hash_combine(hash_combine(0x811C9DC5u, igm_bits(hit.distance)), igm_bits(hit.u)), igm_bits(hit.v)), 1 };
            BsdfSample ms;
            if (!bsdf.sample(tmp, out_dir, ms))
                return Color{ 1, 0, 1 };
            const float pdf  = bsdf.pdf(ms.in_dir, out_dir);
            const Color evl  = color_mulf(bsdf.eval(ms.in_dir, out_dir), safe_div(1, pdf));
            const int pdf_ok = igm_abs(ms.pdf - pdf) <= 0.001f ? 1 : 0;
            const int w_ok   = igm_abs(ms.color.r - evl.r) + igm_abs(ms.color.g - evl.g) + igm_abs(ms.color.b - evl.b) <= 0.001f ? 1 : 0;
            return verdict[(w_ok << 1) | pdf_ok];
        }
        case 25: return bsdf.albedo(vec3_neg(ray.dir));
        case 26: return inner < 0 ? Color{ 0, 0, 0 } : palette(inner);
        case 27: return outer < 0 ? Color{ 0, 0, 0 } : palette(outer);
        default: return absv(surf.local.col[2]);
        }
    }

    // on_hit (pathtracer.art:119-139) with make_emissive_material (driver/material.art:22-30)
    bool on_hit(const Ray& ray, const Hit& hit, const SurfaceElement& surf, const PTRayPayload& pt, const ig_material& mat, Color& out) const
    {
        if (ambient_occlusion || debug)
            return false;
        if (wireframe) { // wireframe.art:33-43: color_lerp(white, black, t)
            float t;
            if (!is_edge_hit(hit, surf, pt.inv_pdf, t))
                return false;
            const float c = (1 - t) * 1.0f + t * 0.0f;
            out           = Color{ c, c, c };
            return true;
        }
        if (mat.light_id >= 0 && surf.is_entering) {
            const float dot = -vec3_dot(ray.dir, surf.local.col[2]);
            if (dot > flt_eps) {
                const ig_light& light = sc.lights[mat.light_id];
                Color emit;
                float pdf_s;
                if (light.type == IG_LIGHT_MESH_AREA) {
                    const MeshEmitter me(sc, light);
                    emit  = me.radiance;
                    pdf_s = me.pdf_area(surf.prim_coords) * (hit.distance * hit.distance) / dot; // Pdf::as_solid (driver/pdf.art:19-38)
                } else if (light.type == IG_LIGHT_SPHERE) {
                    emit  = Color{ light.d[4], light.d[5], light.d[6] };
                    pdf_s = safe_div(1, light.d[7]) * (hit.distance * hit.distance) / dot; // make_area_pdf(inv_area).as_solid
                } else {
                    const PlaneEmitter pe(light);
                    emit  = pe.radiance;            // light.emission(ctx)
                    pdf_s = pe.pdf_direct(ray.org); // solid-angle pdf: as_solid is the identity
                }
                if (volumetric) {
                    // volpathtracer.art:97-109: medium interactions (negative inv_pdf) count as pdf-less, the emission is attenuated
                    const float inv_pdf = igm_max(0.0f, pt.inv_pdf);
                    const float mis     = enable_nee ? 1 / (1 + inv_pdf * select_pdf(mat.light_id, ray.org) * pdf_s) : 1.0f;
                    const Color vol     = get_medium(pt.medium).eval(ray.org, surf.point);
                    out                 = handle_color(color_mulf(color_mul(pt.contrib, color_mul(emit, vol)), mis));
                    return true;
                }
                const float mis = enable_nee ? 1 / (1 + pt.inv_pdf * select_pdf(mat.light_id, ray.org) * pdf_s) : 1.0f;
                out             = handle_color(color_mulf(color_mul(pt.contrib, emit), mis));
                return true;
            }
        }
        return false;
    }

    // Light::emission of an infinite, non-delta light (what on_miss evaluates); false: delta light
    bool infinite_emission(const ig_light& light, Vec3 dir, Color& emit) const
    {
        if (light.type == IG_LIGHT_ENV_TEXTURED) {
            emit = TexturedEnv(sc, light).emission(dir);
        } else if (light.type == IG_LIGHT_CIE) {
            emit = CieSky(light).emission(dir);
        } else if (light.type == IG_LIGHT_PEREZ) {
            const SunLight sun = perez_sun(light);
            const CieSky sky(light);
            const Vec3 d = make_vec3(vec3_dot(sky.transform.col[0], dir), vec3_dot(sky.transform.col[1], dir), vec3_dot(sky.transform.col[2], dir));
            emit         = color_add(sun.hits(dir) ? sun.radiance : Color{ 0, 0, 0 }, sky.radiance(d));
        } else if (light.type == IG_LIGHT_SUN) {
            const SunLight sun(light);
            emit = sun.hits(dir) ? sun.radiance : Color{ 0, 0, 0 };
        } else if (light.type == IG_LIGHT_ENV) {
            emit = Color{ light.d[0], light.d[1], light.d[2] };
        } else {
            return false;
        }
        return true;
    }

    // on_miss (pathtracer.art:141-168): sum over infinite, non-delta lights
    bool on_miss(const Ray& ray, const PTRayPayload& pt, Color& out) const
    {
        if (ambient_occlusion || debug || wireframe)
            return false;
        int inflights = 0;
        Color color   = Color{ 0, 0, 0 };
        for (uint32_t i = 0; i < sc.infinite_light_count; ++i) {
            const ig_light& light = sc.lights[i];
            if (light.type != IG_LIGHT_ENV && light.type != IG_LIGHT_ENV_TEXTURED && light.type != IG_LIGHT_SUN && light.type != IG_LIGHT_CIE && light.type != IG_LIGHT_PEREZ)
                continue; // delta lights
            ++inflights;
            Color emit;
            float pdf_s;
            if (light.type == IG_LIGHT_ENV_TEXTURED) {
                const TexturedEnv env(sc, light);
                emit  = env.emission(ray.dir);
                pdf_s = env.pdf(ray.dir);
            } else if (light.type == IG_LIGHT_CIE) {
                const CieSky sky(light);
                emit  = sky.emission(ray.dir);
                pdf_s = sky.pdf(ray.dir);
            } else if (light.type == IG_LIGHT_PEREZ) {
                // emission / pdf_direct of make_perez_light_raw (perez.art:314 over sun.art:31-45)
                const SunLight sun = perez_sun(light);
                const CieSky sky(light);
                const bool hit = sun.hits(ray.dir);
                const Vec3 d   = make_vec3(vec3_dot(sky.transform.col[0], ray.dir), vec3_dot(sky.transform.col[1], ray.dir), vec3_dot(sky.transform.col[2], ray.dir));
                emit           = color_add(hit ? sun.radiance : Color{ 0, 0, 0 }, sky.radiance(d));
                pdf_s          = hit ? sun.dir_pdf() : 0.0f;
            } else if (light.type == IG_LIGHT_SUN) {
                const SunLight sun(light); // sun.art:31-45
                const bool hit = sun.hits(ray.dir);
                emit           = hit ? sun.radiance : Color{ 0, 0, 0 };
                pdf_s          = hit ? sun.dir_pdf() : 0.0f;
            } else {
                emit  = Color{ light.d[0], light.d[1], light.d[2] };
                pdf_s = 1 / (4 * flt_pi); // equal_area_sphere_pdf (env.art:101)
            }
            if (volumetric) {
                // volpathtracer.art:131-138
                const float mis = enable_nee ? 1 / (1 + igm_max(0.0f, pt.inv_pdf) * select_pdf((int32_t)i, ray.org) * pdf_s) : 1.0f;
                const Color c   = handle_color(color_mulf(color_mul(pt.contrib, color_mul(emit, get_medium(pt.medium).eval_inf())), mis));
                color           = color_add(color, c);
                continue;
            }
            const float mis   = enable_nee ? 1 / (1 + pt.inv_pdf * select_pdf((int32_t)i, ray.org) * pdf_s) : 1.0f;
            const Color c     = handle_color(color_mulf(color_mul(pt.contrib, emit), mis));
            color             = Color{ color.r + c.r, color.g + c.g, color.b + c.b };
        }
        if (inflights > 0) {
            out = color;
            return true;
        }
        return false;
    }

    // on_bounce (pathtracer.art:170-210)
    // on_bounce of make_wireframe_renderer (wireframe.art:45-63): past a hit that is not on an edge the ray goes straight on
    bool wire_bounce(const Ray& ray, const Hit& hit, const SurfaceElement& surf, PTRayPayload& pt, Ray& new_ray) const
    {
        float t;
        if (is_edge_hit(hit, surf, pt.inv_pdf, t))
            return false;
        pt.depth   = pt.depth + 1;
        pt.inv_pdf = pt.inv_pdf + hit.distance;
        new_ray    = make_ray(surf.point, ray.dir, offset, flt_max, IG_RAY_FLAG_BOUNCE);
        return true;
    }

    bool on_bounce(const Ray& ray, const SurfaceElement& surf, Rng& rnd, PTRayPayload& pt, const Bsdf& bsdf, const ig_material& mat, Ray& new_ray) const
    {
        if (ambient_occlusion || debug)
            return false;
        if (pt.depth + 1 > max_path_len)
            return false;

        const Vec3 out_dir = vec3_neg(ray.dir);
        if (volumetric) {
            // on_bounce of make_volume_path_renderer (volpathtracer.art:155-247)
            const Medium medium = get_medium(pt.medium);
            MediumSampleOut msmp;
            if (medium.sample(rnd, ray.org, surf.point, msmp)) {
                const Vec3 in_dir   = medium.sample_phase(rnd);
                const Color contrib = color_mul(pt.contrib, msmp.color); // phase weight 1
                const float rr_prob = (pt.depth + 1 > min_path_len) ? russian_roulette_pbrt(color_mulf(contrib, pt.eta * pt.eta), 0.95f) : 1.0f;
                if (rnd.next_f32() >= rr_prob)
                    return false;
                pt.inv_pdf = -1; // "the last interaction was a medium"
                pt.contrib = color_mulf(contrib, 1 / rr_prob);
                pt.depth   = pt.depth + 1;
                new_ray    = make_ray(msmp.pos, in_dir, 0, flt_max, IG_RAY_FLAG_BOUNCE);
                return true;
            }
            BsdfSample ms;
            if (!bsdf.sample(rnd, out_dir, ms))
                return false;
            if (ms.pdf <= flt_eps)
                return false;
            const Color vol_contrib = color_mul(medium.eval(ray.org, surf.point), pt.contrib);
            const Color contrib     = color_mul(vol_contrib, ms.color);
            const float rr_prob     = (pt.depth + 1 > min_path_len) ? russian_roulette_pbrt(color_mulf(contrib, pt.eta * pt.eta), 0.95f) : 1.0f;
            if (rnd.next_f32() >= rr_prob)
                return false;
            // a transmission enters the medium on the other side of the interface (make_medium_interface.pick, driver/medium.art:34-38)
            const bool is_transmission = igm_signbit(vec3_dot(surf.local.col[2], ms.in_dir));
            const int32_t inner = (mat.pad[2] & 0xFFFF) - 1, outer = ((mat.pad[2] >> 16) & 0xFFFF) - 1;
            pt.medium  = is_transmission ? (surf.is_entering ? inner : outer) : pt.medium;
            pt.inv_pdf = ms.is_delta ? 0 : 1 / ms.pdf;
            pt.contrib = color_mulf(contrib, 1 / rr_prob);
            pt.depth   = pt.depth + 1;
            pt.eta     = pt.eta * ms.eta;
            new_ray    = make_ray(surf.point, ms.in_dir, offset, flt_max, IG_RAY_FLAG_BOUNCE);
            return true;
        }
        BsdfSample ms;
        if (!bsdf.sample(rnd, out_dir, ms))
            return false;
        if (ms.pdf <= flt_eps)
            return false;

        const Color contrib = color_mul(pt.contrib, ms.color);
        const float rr_prob = (pt.depth + 1 > min_path_len) ? russian_roulette_pbrt(color_mulf(contrib, pt.eta * pt.eta), 0.95f) : 1.0f;
        if (rnd.next_f32() >= rr_prob)
            return false;

        const float inv_pdf     = ms.is_delta ? 0 : 1 / ms.pdf;
        const Color new_contrib = color_mulf(contrib, 1 / rr_prob);

        pt.inv_pdf = inv_pdf;
        pt.contrib = new_contrib;
        pt.depth   = pt.depth + 1;
        pt.eta     = pt.eta * ms.eta;
        new_ray    = make_ray(surf.point, ms.in_dir, offset, flt_max, IG_RAY_FLAG_BOUNCE);
        return true;
    }
};

// ================================================================================================================
// Light tracer (technique/lighttracer.art): paths start on a light (make_lt_emitter, :35-62), every non-delta vertex is
// connected to the camera (on_shadow, :75-113) and the unoccluded connection is splatted into the pixel the vertex projects
// to (on_advanced_shadow_miss, :116-120; the "advanced" shadow kernels of driver/mapping_gpu.art:293-333 only exist to hand
// that callback its secondary payload). Bounces sample the BSDF with adjoint = true (:123-163).

struct EmissionSample { // make_emission_sample (light/light.art)
    Vec3 pos, dir;
    Color intensity;
    float pdf_area, pdf_dir, cos;
};

// env_sample_pos (light/env.art:2-6): a point of the disc of radius scene_radius that faces the scene from direction `dir`
static inline void env_sample_pos(const igd_scene& sc, Rng& rnd, Vec3 dir, Vec3& pos, float& pdf)
{
    const float r = sc.scene_radius; // bbox_radius(scene_bbox) * 1.01
    const float u = rnd.next_f32();
    const float v = rnd.next_f32();
    float dx, dy;
    square_to_concentric_disk(u, v, dx, dy); // sample_uniform_disk (core/sampling.art:101-103)
    const Vec3 bmin = make_vec3(sc.bbox_min[0], sc.bbox_min[1], sc.bbox_min[2]), bmax = make_vec3(sc.bbox_max[0], sc.bbox_max[1], sc.bbox_max[2]);
    const Vec3 center = vec3_add(bmin, vec3_mulf(vec3_sub(bmax, bmin), 0.5f)); // bbox_center (core/bbox.art:22)
    pos = vec3_add(center, vec3_add(vec3_mulf(dir, r), mat3x3_mul(make_orthonormal_mat3x3(dir), make_vec3(dx * r, dy * r, 0))));
    pdf = 1 / (flt_pi * r * r);
}

// Light::sample_emission of the light types the light tracer is lowered for
static inline bool sample_emission(const igd_scene& sc, const ig_light& l, Rng& rnd, EmissionSample& e)
{
    switch (l.type) {
    case IG_LIGHT_POINT: { // light/point.art:9-12
        const float u = rnd.next_f32();
        const float v = rnd.next_f32();
        const float c = 2 * v - 1, sn = safe_sqrt(1 - c * c), phi = 2 * flt_pi * u; // sample_uniform_sphere (core/sampling.art:42-47)
        const float pdf = 1 / (4 * flt_pi);
        e = EmissionSample{ make_vec3(l.d[0], l.d[1], l.d[2]), make_vec3(sn * igm_cos(phi), sn * igm_sin(phi), c), color_mulf(Color{ l.d[4], l.d[5], l.d[6] }, 1 / pdf), 1, pdf, 1 };
        return true;
    }
    case IG_LIGHT_SPOT: { // light/spot.art:8-47
        const Vec3 pos = make_vec3(l.d[0], l.d[1], l.d[2]), dir = make_vec3(l.d[4], l.d[5], l.d[6]);
        const float cosCutoffAngle = l.d[3], cosFalloffAngle = l.d[7];
        const float blendRange  = cosFalloffAngle - cosCutoffAngle;
        const float spot_radius = igm_sqrt(1 - cosCutoffAngle * cosCutoffAngle) / cosCutoffAngle;
        const float spot_area   = flt_pi * spot_radius * spot_radius;
        const float u = rnd.next_f32();
        const float v = rnd.next_f32();
        const float c1 = 1 - cosCutoffAngle; // sample_uniform_cone (core/sampling.art:109-116)
        float px, py;
        square_to_concentric_disk(u, v, px, py);
        const float n2  = px * px + py * py;
        const float z   = cosCutoffAngle + c1 * (1 - n2);
        const float sc2 = safe_sqrt(c1 * (2 - c1 * n2));
        const float pdf = safe_div(1, 2 * flt_pi * (1 - cosCutoffAngle));
        const Vec3 out_dir    = mat3x3_mul(make_orthonormal_mat3x3(dir), make_vec3(px * sc2, py * sc2, z));
        const float cos_angle = vec3_dot(out_dir, dir);
        float factor;
        if (blendRange <= flt_eps) {
            factor = cos_angle <= cosCutoffAngle ? 0.0f : 1.0f;
        } else {
            const float x = clampf((cos_angle - cosCutoffAngle) / blendRange, 0, 1);
            factor        = x * x * (3 - 2 * x);
        }
        const Color color = color_mulf(color_mulf(Color{ l.d[8], l.d[9], l.d[10] }, factor), 1 / (spot_area * pdf));
        e = EmissionSample{ pos, out_dir, color, 1, spot_area * pdf, z };
        return true;
    }
    case IG_LIGHT_PLANE:
    case IG_LIGHT_SPHERE:
    case IG_LIGHT_MESH_AREA: { // make_area_light.sample_emission (light/area.art:26-37)
        const float u0 = rnd.next_f32();
        const float u1 = rnd.next_f32();
        Vec3 point, normal;
        float area_pdf;
        Color radiance;
        if (l.type == IG_LIGHT_SPHERE) { // make_sphere_area_emitter.sample_emission (area.art:296-299): local frame around the face normal
            const SphereEmitter se(sc, l);
            sphere_surface_for_normal(se.entity, se.sphere, equal_area_square_to_sphere(u0, u1), point, normal);
            area_pdf = se.inv_area;
            radiance = se.radiance;
        } else if (l.type == IG_LIGHT_PLANE) { // make_plane_area_emitter.sample (area.art:230-250)
            const PlaneEmitter pe(l);
            point    = vec3_add(vec3_add(vec3_mulf(pe.x_axis, u0), vec3_mulf(pe.y_axis, u1)), pe.origin);
            normal   = pe.normal;
            area_pdf = pe.inv_area;
            radiance = pe.radiance;
        } else { // make_shape_area_emitter.sample (area.art:62-72) over shape.surface_element_for_point (shapes/trimesh.art:41-68)
            const MeshEmitter me(sc, l);
            int32_t f;
            float bu, bv, area;
            Vec3 fn;
            me.address(Vec2{ u0, u1 }, f, bu, bv);
            me.surface(f, bu, bv, point, fn, area);
            const int32_t i0 = me.mesh.indices[f * 4 + 0], i1 = me.mesh.indices[f * 4 + 1], i2 = me.mesh.indices[f * 4 + 2];
            auto nrm = [&](int32_t i) { return Vec3{ me.mesh.normals[i * 4], me.mesh.normals[i * 4 + 1], me.mesh.normals[i * 4 + 2] }; };
            normal   = vec3_normalize(mat3x3_mul(me.entity.normal_mat, vec3_lerp2(nrm(i0), nrm(i1), nrm(i2), bu, bv)));
            area_pdf = safe_div(1, area) / (float)me.mesh.num_tris;
            radiance = me.radiance;
        }
        const float u2 = rnd.next_f32();
        const float u3 = rnd.next_f32();
        const DirSample smp = sample_cosine_hemisphere(u2, u3);
        const float weight  = safe_div(1, area_pdf * smp.pdf);
        e = EmissionSample{ point, mat3x3_mul(make_orthonormal_mat3x3(normal), smp.dir), color_mulf(radiance, weight), area_pdf, smp.pdf, smp.dir.z };
        return true;
    }
    case IG_LIGHT_DIRECTIONAL: { // light/directional.art:7-10
        const Vec3 dir = make_vec3(l.d[0], l.d[1], l.d[2]);
        Vec3 pos;
        float pos_pdf;
        env_sample_pos(sc, rnd, vec3_neg(dir), pos, pos_pdf);
        e = EmissionSample{ pos, dir, color_mulf(Color{ l.d[4], l.d[5], l.d[6] }, safe_div(1, pos_pdf)), pos_pdf, 1, 1 };
        return true;
    }
    case IG_LIGHT_SUN: { // make_sun_light.sample_emission (light/sun.art:24-29)
        const SunLight sun(l);
        const float u  = rnd.next_f32();
        const float v  = rnd.next_f32();
        const float c1 = 1 - sun.cos_angle; // sample_uniform_cone
        float px, py;
        square_to_concentric_disk(u, v, px, py);
        const float n2 = px * px + py * py;
        const float z  = sun.cos_angle + c1 * (1 - n2);
        const float k  = safe_sqrt(c1 * (2 - c1 * n2));
        const Vec3 ndir     = mat3x3_mul(make_orthonormal_mat3x3(vec3_neg(sun.dir)), make_vec3(px * k, py * k, z));
        const float inv_pdf = 2 * flt_pi * (1 - sun.cos_angle);
        Vec3 pos;
        float pos_pdf;
        env_sample_pos(sc, rnd, vec3_neg(ndir), pos, pos_pdf);
        e = EmissionSample{ pos, ndir, color_mulf(sun.radiance, safe_div(inv_pdf, pos_pdf)), pos_pdf, sun.dir_pdf(), z };
        return true;
    }
    case IG_LIGHT_ENV: { // make_environment_light_function_spherical.sample_emission (light/env.art:87-93), constant colour, identity transform
        const float u   = rnd.next_f32();
        const float v   = rnd.next_f32();
        const Vec3 dir  = equal_area_square_to_sphere(u, v);
        const float pdf = 1 / (4 * flt_pi);
        Vec3 pos;
        float pos_pdf;
        env_sample_pos(sc, rnd, dir, pos, pos_pdf);
        e = EmissionSample{ pos, vec3_neg(dir), color_mulf(Color{ l.d[0], l.d[1], l.d[2] }, safe_div(1, pos_pdf * pdf)), pos_pdf, pdf, 1.0f };
        return true;
    }
    case IG_LIGHT_ENV_TEXTURED: { // make_environment_light_textured.sample_emission (light/env.art:141-145); "cdf": "none": the spherical function environment (:87-93)
        const TexturedEnv env(sc, l);
        Vec3 dir;
        Color intensity;
        float pdf_dir;
        env.sample_dir(rnd, dir, intensity, pdf_dir);
        Vec3 pos;
        float pos_pdf;
        env_sample_pos(sc, rnd, dir, pos, pos_pdf);
        e = EmissionSample{ pos, vec3_neg(dir), color_mulf(intensity, safe_div(1, pos_pdf * pdf_dir)), pos_pdf, pdf_dir, 1.0f };
        return true;
    }
    case IG_LIGHT_CIE: { // make_environment_light_function_{hemi, spherical}.sample_emission (light/env.art:38-46,87-93) over the sky function
        const CieSky sky(l);
        const float u = rnd.next_f32();
        const float v = rnd.next_f32();
        Vec3 gdir;
        Color intensity;
        float pdf;
        if (!sky.has_ground) {
            const DirSample ds = sample_cosine_hemisphere(u, v);
            const Vec3 dir     = switch_env_up(ds.dir);
            pdf                = ds.pdf;
            intensity          = sky.radiance(dir);
            gdir               = make_vec3(vec3_dot(sky.transform.col[0], dir), vec3_dot(sky.transform.col[1], dir), vec3_dot(sky.transform.col[2], dir));
        } else {
            gdir      = equal_area_square_to_sphere(u, v);
            pdf       = 1 / (4 * flt_pi);
            intensity = sky.radiance(mat3x3_mul(sky.transform, gdir));
        }
        Vec3 pos;
        float pos_pdf;
        env_sample_pos(sc, rnd, gdir, pos, pos_pdf);
        e = EmissionSample{ pos, vec3_neg(gdir), color_mulf(intensity, safe_div(1, pdf * pos_pdf)), pos_pdf, pdf, 1.0f };
        return true;
    }
    case IG_LIGHT_PEREZ: { // make_perez_light_raw.sample_emission (light/perez.art:309-313): the sun's sample (sun.art:24-29) plus the sky seen against it
        const SunLight sun = perez_sun(l);
        const float u  = rnd.next_f32();
        const float v  = rnd.next_f32();
        const float c1 = 1 - sun.cos_angle;
        float px, py;
        square_to_concentric_disk(u, v, px, py);
        const float n2 = px * px + py * py;
        const float z  = sun.cos_angle + c1 * (1 - n2);
        const float k  = safe_sqrt(c1 * (2 - c1 * n2));
        const Vec3 ndir     = mat3x3_mul(make_orthonormal_mat3x3(vec3_neg(sun.dir)), make_vec3(px * k, py * k, z));
        const float inv_pdf = 2 * flt_pi * (1 - sun.cos_angle);
        Vec3 pos;
        float pos_pdf;
        env_sample_pos(sc, rnd, vec3_neg(ndir), pos, pos_pdf);
        const CieSky sky(l);
        const Vec3 to_sky = vec3_neg(ndir);
        const Vec3 d      = make_vec3(vec3_dot(sky.transform.col[0], to_sky), vec3_dot(sky.transform.col[1], to_sky), vec3_dot(sky.transform.col[2], to_sky));
        const Color c     = color_add(color_mulf(sun.radiance, safe_div(inv_pdf, pos_pdf)), color_mulf(sky.radiance(d), 1 / (pos_pdf * sun.dir_pdf())));
        e = EmissionSample{ pos, ndir, c, pos_pdf, sun.dir_pdf(), z };
        return true;
    }
    default:
        return false;
    }
}

struct LightTracer {
    const igd_scene& sc;
    const PathTracer selector; // light selection only
    int32_t max_path_len, min_path_len;
    float clamp_value;
    static constexpr float offset = 0.001f;

    explicit LightTracer(const igd_scene& s)
        : sc(s)
        , selector(s)
        , max_path_len(s.technique.max_depth)
        , min_path_len(s.technique.min_depth)
        , clamp_value(s.technique.clamp)
    {
    }
    Color handle_color(Color c) const { return clamp_value > 0 ? color_saturate(c, clamp_value) : c; }

    // make_lt_emitter (lighttracer.art:35-62); false: no ray for this sample
    bool emit(Rng& rnd, Ray& ray, PTRayPayload& payload, int32_t* light_id = nullptr) const
    {
        if (sc.light_count == 0)
            return false;
        float light_pdf;
        const int32_t li  = selector.select_light(rnd, make_vec3(0, 0, 0), light_pdf);
        if (light_id)
            *light_id = li; // make_ppm_light_emitter (photonmapper.art:141-165): the same emitter, the payload also names the light
        const ig_light& l = sc.lights[li];
        EmissionSample es;
        if (!sample_emission(sc, l, rnd, es))
            return false;
        const bool infinite = li < (int32_t)sc.infinite_light_count;
        ray                 = make_ray(es.pos, es.dir, infinite ? 0.0f : offset, flt_max, IG_RAY_FLAG_LIGHT);
        payload             = PTRayPayload{ 0, color_mulf(es.intensity, safe_div(igm_abs(es.cos), light_pdf * 1.0f)), 1, 1, -1 };
        return true;
    }

    // camera.sample_pixel of make_perspective_camera (camera/perspective.art:16-26,43-57): no test for points behind the eye, as written
    static bool sample_pixel(const CameraSetup& cam, Vec3 pos, Vec3& dir, float& nx, float& ny)
    {
        if (cam.type != IG_CAMERA_PERSPECTIVE || cam.aperture_radius > 0)
            return false;
        const Vec3 d  = vec3_sub(pos, cam.eye);
        const Vec3 un = make_vec3(vec3_dot(cam.view.col[0], d), vec3_dot(cam.view.col[1], d), vec3_dot(cam.view.col[2], d)); // mat3x3_left_mul
        nx = un.x / (un.z * cam.sx);
        ny = un.y / (un.z * cam.sy);
        if (!(nx >= -1 && nx <= 1 && ny >= -1 && ny <= 1))
            return false;
        dir = vec3_sub(cam.eye, pos);
        return true;
    }
    // make_pixelcoord_from_normalized (driver/camera.art:45-57)
    static int pixel_from_normalized(float nx, float ny, int w, int h)
    {
        const int x = std::min((int)igm_floor((float)w * (nx + 1) / 2), w - 1);
        const int y = std::min((int)igm_floor((float)h * (1 - ny) / 2), h - 1);
        return y * w + x;
    }

    // on_shadow (lighttracer.art:75-113)
    ShadowRayOut on_shadow(const CameraSetup& cam, const Ray& ray, const SurfaceElement& surf, const PTRayPayload& pt, const Bsdf& bsdf, float& nx, float& ny) const
    {
        ShadowRayOut out;
        out.valid = false;
        if (bsdf.is_all_delta())
            return out;
        if (pt.depth + 1 > max_path_len)
            return out;
        Vec3 cam_dir;
        if (!sample_pixel(cam, surf.point, cam_dir, nx, ny))
            return out;
        const Vec3 in_dir  = vec3_normalize(cam_dir);
        const Vec3 out_dir = vec3_neg(ray.dir);
        const float cos_o  = vec3_dot(out_dir, surf.local.col[2]);
        const float cos_i  = vec3_dot(in_dir, surf.local.col[2]);
        if (!(cos_o * cos_i > flt_eps))
            return out;
        const float d2     = vec3_len2(cam_dir);
        const float factor = safe_div(cos_i, cos_o * d2);
        // camera_sample.weight = make_gray_color(image_area) with image_area = 1 (perspective.art:36,47-51)
        out.color = handle_color(color_mulf(color_mul(Color{ 1, 1, 1 }, color_mul(pt.contrib, bsdf.eval(out_dir, in_dir))), factor));
        out.ray   = make_ray(surf.point, cam_dir, offset, 1 - offset, IG_RAY_FLAG_SHADOW);
        out.valid = true;
        return out;
    }

    // on_bounce (lighttracer.art:123-163): the path tracer's with adjoint = true (the Bsdf carries the flag)
    bool on_bounce(const Ray& ray, const SurfaceElement& surf, Rng& rnd, PTRayPayload& pt, const Bsdf& bsdf, Ray& new_ray) const
    {
        if (pt.depth + 1 > max_path_len)
            return false;
        const Vec3 out_dir = vec3_neg(ray.dir);
        BsdfSample ms;
        if (!bsdf.sample(rnd, out_dir, ms))
            return false;
        if (ms.pdf <= flt_eps)
            return false;
        const Color contrib = color_mul(pt.contrib, ms.color);
        const float rr_prob = (pt.depth + 1 > min_path_len) ? russian_roulette_pbrt(color_mulf(contrib, pt.eta * pt.eta), 0.95f) : 1.0f;
        if (rnd.next_f32() >= rr_prob)
            return false;
        pt.contrib = color_mulf(contrib, 1 / rr_prob);
        pt.depth   = pt.depth + 1;
        pt.eta     = pt.eta * ms.eta;
        new_ray    = make_ray(surf.point, ms.in_dir, offset, flt_max, IG_RAY_FLAG_BOUNCE);
        return true;
    }
};

} // namespace oracle
