// oracle_core.h — TEST INFRASTRUCTURE ONLY (see oracle/README.md).
//
// CPU restatement of the reference's traversal + intersection core, scalar
// (vector width 1) CPU-device semantics. Each function cites the reference
// file:line it follows. Nothing under ignis_amd/ includes this file.
//
// Float rules: compiled with -ffp-contract=off; fused
// multiply-adds appear only where written (vec3_dot is an fma chain in the
// reference itself, src/artic/core/vector.art:96-98; the slab test is written
// as fma because the reference's -ffast-math build contracts it). Transcendental
// functions come from include/ig_detmath.h.
#pragma once

#include "ig_detmath.h"
#include "ig_tables.h"

#include <cstdint>
#include <cstring>

namespace oracle {

static constexpr float flt_eps = 1.1920928955e-07f; // common.art:3
static constexpr float flt_max = 3.4028234664e+38f; // common.art:4
static constexpr float flt_pi  = 3.14159265359f;    // common.art:7
static constexpr float flt_inv_pi = 0.31830988618379067154f;

struct Vec2 {
    float x, y;
};
struct Vec3 {
    float x, y, z;
};
struct Color {
    float r, g, b;
};
struct Mat3x3 {
    Vec3 col[3];
};
struct Mat3x4 {
    Vec3 col[4];
};

// ---- vector.art
static inline Vec3 make_vec3(float x, float y, float z) { return Vec3{ x, y, z }; }
static inline Vec3 vec3_add(Vec3 a, Vec3 b) { return Vec3{ a.x + b.x, a.y + b.y, a.z + b.z }; }
static inline Vec3 vec3_sub(Vec3 a, Vec3 b) { return Vec3{ a.x - b.x, a.y - b.y, a.z - b.z }; }
static inline Vec3 vec3_mul(Vec3 a, Vec3 b) { return Vec3{ a.x * b.x, a.y * b.y, a.z * b.z }; }
static inline Vec3 vec3_neg(Vec3 a) { return Vec3{ -a.x, -a.y, -a.z }; }
static inline Vec3 vec3_mulf(Vec3 a, float t) { return Vec3{ a.x * t, a.y * t, a.z * t }; }
// vector.art:97
static inline float vec3_dot(Vec3 a, Vec3 b) { return igm_fma(a.x, b.x, igm_fma(a.y, b.y, a.z * b.z)); }
// vector.art:102-105
static inline Vec3 vec3_cross(Vec3 a, Vec3 b) { return Vec3{ a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x }; }
static inline float vec3_len2(Vec3 v) { return vec3_dot(v, v); }
static inline float vec3_len(Vec3 v) { return igm_sqrt(vec3_len2(v)); }
static inline Vec3 vec3_normalize(Vec3 v) { return vec3_mulf(v, 1 / vec3_len(v)); } // vector.art:138
static inline Vec3 vec3_reflect(Vec3 v, Vec3 n) { return vec3_sub(vec3_mulf(n, 2 * vec3_dot(n, v)), v); } // vector.art:123
static inline Vec3 vec3_refract(Vec3 v, Vec3 n, float eta, float cos_i, float cos_t) { return vec3_sub(vec3_mulf(n, eta * cos_i - cos_t), vec3_mulf(v, eta)); } // vector.art:126
// common.art:238, vector.art:152-156
static inline float lerp2(float a, float b, float c, float k1, float k2) { return (1 - k1 - k2) * a + k1 * b + k2 * c; }
static inline Vec3 vec3_lerp2(Vec3 a, Vec3 b, Vec3 c, float u, float v) { return Vec3{ lerp2(a.x, b.x, c.x, u, v), lerp2(a.y, b.y, c.y, u, v), lerp2(a.z, b.z, c.z, u, v) }; }
static inline Vec2 vec2_lerp2(Vec2 a, Vec2 b, Vec2 c, float u, float v) { return Vec2{ lerp2(a.x, b.x, c.x, u, v), lerp2(a.y, b.y, c.y, u, v) }; }
static inline float lerp(float a, float b, float k) { return (1 - k) * a + k * b; }
static inline Vec2 vec2_lerp(Vec2 a, Vec2 b, float k) { return Vec2{ lerp(a.x, b.x, k), lerp(a.y, b.y, k) }; }

// ---- matrix.art
// mat3x4_mul with a (v,1) / (v,0) vector: vec4_dot per row (matrix.art:120-123,246-247, vector.art:98)
static inline float vec4_dot(float ax, float ay, float az, float aw, float bx, float by, float bz, float bw)
{
    return igm_fma(ax, bx, igm_fma(ay, by, igm_fma(az, bz, aw * bw)));
}
static inline Vec3 mat3x4_transform_point(const Mat3x4& m, Vec3 v)
{
    return Vec3{ vec4_dot(m.col[0].x, m.col[1].x, m.col[2].x, m.col[3].x, v.x, v.y, v.z, 1),
                 vec4_dot(m.col[0].y, m.col[1].y, m.col[2].y, m.col[3].y, v.x, v.y, v.z, 1),
                 vec4_dot(m.col[0].z, m.col[1].z, m.col[2].z, m.col[3].z, v.x, v.y, v.z, 1) };
}
static inline Vec3 mat3x4_transform_direction(const Mat3x4& m, Vec3 v)
{
    return Vec3{ vec4_dot(m.col[0].x, m.col[1].x, m.col[2].x, m.col[3].x, v.x, v.y, v.z, 0),
                 vec4_dot(m.col[0].y, m.col[1].y, m.col[2].y, m.col[3].y, v.x, v.y, v.z, 0),
                 vec4_dot(m.col[0].z, m.col[1].z, m.col[2].z, m.col[3].z, v.x, v.y, v.z, 0) };
}
// matrix.art:110-113 (rows dotted with v)
static inline Vec3 mat3x3_mul(const Mat3x3& m, Vec3 v)
{
    return Vec3{ vec3_dot(Vec3{ m.col[0].x, m.col[1].x, m.col[2].x }, v),
                 vec3_dot(Vec3{ m.col[0].y, m.col[1].y, m.col[2].y }, v),
                 vec3_dot(Vec3{ m.col[0].z, m.col[1].z, m.col[2].z }, v) };
}
// matrix.art:24-32 (Duff et al.)
static inline Mat3x3 make_orthonormal_mat3x3(Vec3 n)
{
    const float sign = igm_copysign(1.0f, n.z);
    const float a    = -1 / (sign + n.z);
    const float b    = n.x * n.y * a;
    Mat3x3 m;
    m.col[0] = Vec3{ 1 + sign * n.x * n.x * a, sign * b, -sign * n.x };
    m.col[1] = Vec3{ b, sign + n.y * n.y * a, -n.y };
    m.col[2] = n;
    return m;
}

// ---- common.art
static inline float prodsign(float x, float y) { return igm_float(igm_bits(x) ^ (igm_bits(y) & 0x80000000u)); } // common.art:210
// common.art:212-215
static inline float safe_rcp(float x)
{
    const float min_rcp = 1e-8f;
    if ((x > 0 ? x : -x) < min_rcp)
        return prodsign(flt_max, x);
    return 1 / x;
}
static inline float safe_div(float a, float b) { return igm_abs(b) <= flt_eps ? 0.0f : a / b; } // common.art:263
static inline float safe_sqrt(float a) { return igm_sqrt(igm_max(0.0f, a)); }                    // common.art:265
static inline float clampf(float v, float l, float u) { return igm_min(u, igm_max(l, v)); }       // common.art:261
// common.art:285-290
static inline float sum_of_prod(float a, float b, float c, float d)
{
    const float cd  = c * d;
    const float sum = igm_fma(a, b, cd);
    const float err = igm_fma(c, d, -cd);
    return sum + err;
}

// ---- traversal/ray.art
struct Ray {
    Vec3 org, dir, inv_dir, inv_org;
    float tmin, tmax;
    uint32_t flags;
};

// ray.art:27-39
static inline Ray make_ray(Vec3 org, Vec3 dir, float tmin, float tmax, uint32_t flags)
{
    Ray r;
    r.org     = org;
    r.dir     = dir;
    r.inv_dir = make_vec3(safe_rcp(dir.x), safe_rcp(dir.y), safe_rcp(dir.z));
    r.inv_org = vec3_neg(vec3_mul(org, r.inv_dir));
    r.tmin    = tmin;
    r.tmax    = tmax;
    r.flags   = flags;
    return r;
}

// ray.art:51
static inline bool check_ray_visibility(const Ray& ray, uint32_t flags)
{
    return (ray.flags & IG_RAY_FLAG_TYPE_MASK) == ((ray.flags & flags) & IG_RAY_FLAG_TYPE_MASK);
}

// ray.art:56-59: direction is not normalised, so distances stay global
static inline Ray transform_ray(const Ray& ray, const Mat3x4& m)
{
    return make_ray(mat3x4_transform_point(m, ray.org), mat3x4_transform_direction(m, ray.dir), ray.tmin, ray.tmax, ray.flags);
}

// ---- traversal/intersection.art
struct Hit {
    float distance;
    float u, v;
    int32_t prim_id;
    int32_t ent_id;
};

static inline Hit invalid_hit(float tmax) { return Hit{ tmax, 0, 0, -1, -1 }; } // intersection.art:70

// intersection.art:223-234, unordered variant with float min/max
// (CPU device: make_cpu_*_min_max float flavour, src/runtime/shader/ShaderUtils.cpp:26-28)
static inline void intersect_ray_box(const Ray& ray, const float bmin[3], const float bmax[3], float& entry, float& exit)
{
    const float t0x = igm_fma(ray.inv_dir.x, bmin[0], ray.inv_org.x);
    const float t0y = igm_fma(ray.inv_dir.y, bmin[1], ray.inv_org.y);
    const float t0z = igm_fma(ray.inv_dir.z, bmin[2], ray.inv_org.z);
    const float t1x = igm_fma(ray.inv_dir.x, bmax[0], ray.inv_org.x);
    const float t1y = igm_fma(ray.inv_dir.y, bmax[1], ray.inv_org.y);
    const float t1z = igm_fma(ray.inv_dir.z, bmax[2], ray.inv_org.z);
    // fmaxmaxf(fminf(t0x,t1x), fminf(t0y,t1y), fminmaxf(t0z,t1z,tmin))
    entry = igm_max(igm_max(igm_min(t0x, t1x), igm_min(t0y, t1y)), igm_max(igm_min(t0z, t1z), ray.tmin));
    // fminminf(fmaxf(t0x,t1x), fmaxf(t0y,t1y), fmaxminf(t0z,t1z,tmax))
    exit = igm_min(igm_min(igm_max(t0x, t1x), igm_max(t0y, t1y)), igm_min(igm_max(t0z, t1z), ray.tmax));
}

// intersection.art:74-106, no backface culling (trimesh.art:133), scalar check_if_none
static inline bool intersect_ray_tri_mt(const Ray& ray, Vec3 v0, Vec3 e1, Vec3 e2, Vec3 n, float& t_out, float& u_out, float& v_out)
{
    const Vec3 c           = vec3_sub(v0, ray.org);
    const Vec3 r           = vec3_cross(c, ray.dir);
    const float det        = vec3_dot(n, ray.dir);
    const float abs_det    = igm_abs(det);
    const uint32_t sgn_det = igm_bits(det) & 0x80000000u;

    const float u = igm_float(igm_bits(vec3_dot(r, e1)) ^ sgn_det);
    const float v = igm_float(igm_bits(vec3_dot(r, e2)) ^ sgn_det);

    bool mask = u >= 0;
    mask &= v >= 0;
    mask &= u + v <= abs_det;
    mask &= det != 0;
    if (!mask)
        return false;

    const float t = igm_float(igm_bits(vec3_dot(c, n)) ^ sgn_det);
    mask &= t >= abs_det * ray.tmin;
    mask &= t <= abs_det * ray.tmax;
    if (!mask)
        return false;

    const float rcp = 1 / abs_det;
    t_out           = t * rcp;
    u_out           = igm_max(u * rcp, 0.0f);
    v_out           = igm_max(v * rcp, 0.0f);
    return true;
}

// ---- traversal/stack.art:52-123 — top in "registers" + 64-entry array, sentinel 0
struct Stack {
    int32_t nodes[64];
    float tmins[64];
    int32_t node = 0;
    float tmin   = flt_max;
    int ptr      = -1;
    int max_ptr  = -1;

    void push(int32_t n, float t)
    {
        ++ptr;
        if (ptr > max_ptr)
            max_ptr = ptr;
        nodes[ptr] = node;
        tmins[ptr] = tmin;
        node       = n;
        tmin       = t;
    }
    void push_after(int32_t n, float t)
    {
        ++ptr;
        if (ptr > max_ptr)
            max_ptr = ptr;
        nodes[ptr] = n;
        tmins[ptr] = t;
    }
    void pop(int32_t& n, float& t)
    {
        n    = node;
        t    = tmin;
        node = nodes[ptr];
        tmin = tmins[ptr];
        --ptr;
    }
    bool is_empty() const { return node == 0; }
};

struct TraversalStats {
    uint64_t nodes  = 0; // inner nodes fetched (prim + scene BVH)
    uint64_t tris   = 0; // triangle tests
    uint64_t leaves = 0; // entity leaves tested
    int32_t max_stack = 0;
};

// View of one shape's BVH inside the "trimesh_primbvh" fix table
// (make_cpu_trimesh_bvh_table, shapes/trimesh.art:201-219, vector width >= 8 branch)
struct PrimBvh {
    const ig_node8* nodes;
    const ig_tri4* tris;
};

static inline PrimBvh prim_bvh_at(const igd_scene& sc, uint64_t offset_floats)
{
    const uint8_t* header = sc.primbvh + offset_floats * 4;
    uint32_t node_count;
    std::memcpy(&node_count, header, 4);
    PrimBvh b;
    b.nodes = reinterpret_cast<const ig_node8*>(header + 16);
    b.tris  = reinterpret_cast<const ig_tri4*>(header + 16 + (size_t)node_count * sizeof(ig_node8));
    return b;
}

// cpu_traverse_helper_prim, traversal/mapping_cpu.art:282-419 with vector_width = 1
static inline Hit traverse_prim(Ray ray, const PrimBvh& bvh, bool any_hit, TraversalStats& st)
{
    Hit hit         = invalid_hit(ray.tmax);
    bool terminated = false;
    Stack stack;
    stack.push(1, ray.tmin);

    for (;;) {
        // Cull nodes
        bool exit_all = false;
        for (;;) {
            if (stack.is_empty()) {
                exit_all = true;
                break;
            }
            const bool active = (stack.tmin <= ray.tmax) & !terminated;
            if (active)
                break;
            int32_t n;
            float t;
            stack.pop(n, t);
        }
        if (exit_all)
            break;

        // Intersect inner nodes
        bool culled = false;
        while (stack.node > 0) {
            int32_t node_id;
            float node_t;
            stack.pop(node_id, node_t);
            const ig_node8& node = bvh.nodes[node_id - 1];
            ++st.nodes;

            bool pushed = false;
            for (int i = 0; i < 8; ++i) {
                const int32_t child_id = node.child[i];
                if (child_id == 0)
                    break;
                const float bmin[3] = { node.bounds[0][i], node.bounds[2][i], node.bounds[4][i] };
                const float bmax[3] = { node.bounds[1][i], node.bounds[3][i], node.bounds[5][i] };
                float tentry, texit;
                intersect_ray_box(ray, bmin, bmax, tentry, texit);
                const bool miss = texit < tentry;
                if (!miss) {
                    if (any_hit || stack.tmin > tentry)
                        stack.push(child_id, tentry);
                    else
                        stack.push_after(child_id, tentry);
                    pushed = true;
                }
            }
            if (!pushed) {
                culled = true;
                break;
            }
        }
        if (culled)
            continue;

        if (stack.node < 0) {
            bool active = (stack.tmin <= ray.tmax) & !terminated;
            int32_t leaf;
            float leaf_t;
            stack.pop(leaf, leaf_t);
            int32_t prim_id = ~leaf;
            // An inactive leaf (entry behind the current hit) has no observable effect in the
            // reference: its items are iterated with every test masked out.
            while (active) {
                const ig_tri4& tri = bvh.tris[prim_id++];
                for (int i = 0; i < 4; ++i) {
                    if (tri.prim_id[i] == -1)
                        break;
                    if (active) {
                        ++st.tris;
                        float t, u, v;
                        if (intersect_ray_tri_mt(ray,
                                                 Vec3{ tri.v0[0][i], tri.v0[1][i], tri.v0[2][i] },
                                                 Vec3{ tri.e1[0][i], tri.e1[1][i], tri.e1[2][i] },
                                                 Vec3{ tri.e2[0][i], tri.e2[1][i], tri.e2[2][i] },
                                                 Vec3{ tri.n[0][i], tri.n[1][i], tri.n[2][i] }, t, u, v)) {
                            hit      = Hit{ t, u, v, tri.prim_id[i] & 0x7FFFFFFF, -1 };
                            ray.tmax = t;
                            if (any_hit) {
                                terminated = true;
                                active     = false;
                            }
                        }
                    }
                    if (any_hit && terminated)
                        goto done;
                }
                if (tri.prim_id[3] < 0)
                    break;
            }
        }
    }
done:
    if (stack.max_ptr > st.max_stack)
        st.max_stack = stack.max_ptr;
    return hit;
}

static inline Mat3x4 leaf_local(const ig_entity_leaf1& l)
{
    Mat3x4 m;
    for (int c = 0; c < 4; ++c)
        m.col[c] = Vec3{ l.local[c * 3 + 0], l.local[c * 3 + 1], l.local[c * 3 + 2] };
    return m;
}

// sphere_map_uv (shapes/sphere.art:1-6) over spherical_from_dir (core/common.art: theta = acos(z), phi = atan2(y, x) in [0, 2 pi))
static inline void sphere_map_uv(Vec3 dir, float& u, float& v)
{
    const Vec3 d      = make_vec3(dir.y, -dir.x, dir.z);
    const float theta = igm_acos(d.z);
    float phi         = igm_atan2(d.y, d.x);
    if (phi < 0)
        phi += 2 * flt_pi;
    v = theta / flt_pi;
    u = phi / (2 * flt_pi);
}

// intersect_sphere (shapes/sphere.art:107-137): the ray direction need not be normalised
static inline Hit intersect_sphere(Vec3 origin, float radius, const Ray& ray)
{
    const Vec3 L   = vec3_sub(ray.org, origin);
    const float S  = -vec3_dot(L, ray.dir);
    const float D2 = vec3_len2(ray.dir);
    const float L2 = vec3_len2(L);
    const float R2 = radius * radius * D2;
    const float M2 = L2 * D2 - S * S;
    if ((S < 0) || (M2 > R2))
        return invalid_hit(ray.tmax);
    const float Q   = igm_sqrt(R2 - M2);
    const float t0_ = (S - Q) / D2;
    const float t1_ = (S + Q) / D2;
    const float t0 = t0_ > t1_ ? t1_ : t0_, t1 = t0_ > t1_ ? t0_ : t1_;
    const float tmin = t0 < ray.tmin ? t1 : t0;
    if (tmin >= ray.tmin && tmin <= ray.tmax) {
        const Vec3 dir = vec3_mulf(vec3_add(L, vec3_mulf(ray.dir, tmin)), 1 / radius);
        Hit h;
        h.distance = tmin;
        sphere_map_uv(dir, h.u, h.v);
        h.prim_id = 0;
        h.ent_id  = -1; // InvalidHitId, set by the scene traversal
        return h;
    }
    return invalid_hit(ray.tmax);
}

// One SceneGeometry of the scene (driver/scene.art, TraversalShader.cpp:73-95): a scene BVH over the entities of one shape
// provider and that provider's local handler. kind 0: triangle meshes (prim BVH traversal), kind 1: analytic spheres.
struct SceneGeometry {
    const ig_node8* nodes;
    uint32_t node_count;
    const ig_entity_leaf1* leaves;
    int kind;
};

// cpu_traverse_helper, traversal/mapping_cpu.art:421-518 with vector_width = 1; `hit` = init_hit (invalid_hit(ray.tmax) for the
// first geometry, the previous geometry's result afterwards, driver/mapping_cpu.art:385-403)
static inline Hit traverse_geometry(const igd_scene& sc, const SceneGeometry& geom, Ray ray, Hit hit, bool any_hit, TraversalStats& st)
{
    bool terminated = false;
    if (geom.node_count == 0)
        return hit;

    Stack stack;
    stack.push(1, ray.tmin);

    for (;;) {
        bool exit_all = false;
        for (;;) {
            if (stack.is_empty()) {
                exit_all = true;
                break;
            }
            const bool active = (stack.tmin <= ray.tmax) & !terminated;
            if (active)
                break;
            int32_t n;
            float t;
            stack.pop(n, t);
        }
        if (exit_all)
            break;

        bool culled = false;
        while (stack.node > 0) {
            int32_t node_id;
            float node_t;
            stack.pop(node_id, node_t);
            const ig_node8& node = geom.nodes[node_id - 1];
            ++st.nodes;

            bool pushed = false;
            for (int i = 0; i < 8; ++i) {
                const int32_t child_id = node.child[i];
                if (child_id == 0)
                    break;
                const float bmin[3] = { node.bounds[0][i], node.bounds[2][i], node.bounds[4][i] };
                const float bmax[3] = { node.bounds[1][i], node.bounds[3][i], node.bounds[5][i] };
                float tentry, texit;
                intersect_ray_box(ray, bmin, bmax, tentry, texit);
                const bool miss = texit < tentry;
                if (!miss) {
                    if (any_hit || stack.tmin > tentry)
                        stack.push(child_id, tentry);
                    else
                        stack.push_after(child_id, tentry);
                    pushed = true;
                }
            }
            if (!pushed) {
                culled = true;
                break;
            }
        }
        if (culled)
            continue;

        if (stack.node < 0) {
            bool active = (stack.tmin <= ray.tmax) & !terminated;
            int32_t leaf_ref;
            float leaf_t;
            stack.pop(leaf_ref, leaf_t);
            int32_t ref_id = ~leaf_ref;
            // Inactive run: the reference still calls handle_local but discards the result.
            while (active) {
                const ig_entity_leaf1& leaf = geom.leaves[ref_id++];
                ++st.leaves;
                if (check_ray_visibility(ray, leaf.flags)) {
                    // intersect_ray_box_single_section, intersection.art:247-256
                    float entry, exit;
                    intersect_ray_box(ray, leaf.min, leaf.max, entry, exit);
                    if ((entry <= exit) & (exit >= 0)) {
                        if (entry <= hit.distance) {
                            const Ray local_ray  = transform_ray(ray, leaf_local(leaf));
                            Hit local_hit;
                            if (geom.kind == 1) {
                                // make_scene_local_handler_sphere (shapes/sphere.art:139-148)
                                const float* sp = reinterpret_cast<const float*>(sc.shape_data + sc.shape_lookups[leaf.shape_id].offset);
                                local_hit       = intersect_sphere(make_vec3(sp[0], sp[1], sp[2]), sp[3], local_ray);
                            } else {
                                const uint64_t off = ((uint64_t)(uint32_t)leaf.user[1] << 32) | (uint64_t)(uint32_t)leaf.user[0];
                                local_hit          = traverse_prim(local_ray, prim_bvh_at(sc, off), any_hit, st);
                            }
                            if (active) {
                                if (local_hit.prim_id != -1) {
                                    if (local_hit.distance <= hit.distance) {
                                        hit        = local_hit;
                                        hit.ent_id = leaf.entity_id & 0x7FFFFFFF;
                                        ray.tmax   = hit.distance;
                                        if (any_hit) {
                                            terminated = true;
                                            active     = false;
                                        }
                                    }
                                }
                            }
                        }
                    }
                }
                if (any_hit && terminated)
                    goto done;
                if (leaf.entity_id < 0)
                    break;
            }
        }
    }
done:
    if (stack.max_ptr > st.max_stack)
        st.max_stack = stack.max_ptr;
    return hit;
}

// cpu_traverse_primary / _secondary (driver/mapping_cpu.art:377-435): every geometry in turn, each starting from the hit so far.
// Triangle meshes first, then spheres (the reference iterates an unordered container of providers; the order only matters
// for exactly equal distances).
static inline Hit traverse_scene(const igd_scene& sc, Ray ray, bool any_hit, TraversalStats& st)
{
    Hit hit = invalid_hit(ray.tmax);
    hit     = traverse_geometry(sc, SceneGeometry{ sc.scene_nodes, sc.scene_node_count, sc.scene_leaves, 0 }, ray, hit, any_hit, st);
    if (sc.sphere_node_count != 0 && !(any_hit && hit.prim_id != -1)) {
        ray.tmax = hit.distance; // ray.tmax follows the hit inside a traversal; carried over here as the reference re-reads the hit
        hit      = traverse_geometry(sc, SceneGeometry{ sc.sphere_nodes, sc.sphere_node_count, sc.sphere_leaves, 1 }, ray, hit, any_hit, st);
    }
    return hit;
}

} // namespace oracle
