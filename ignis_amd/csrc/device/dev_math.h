// dev_math.h — device-side vector / ray math for the gfx950 kernels.
//
// Arithmetic follows the reference stdlib expression by expression so results
// are reproducible (compile with -ffp-contract=off; fused multiply-adds appear
// only where the reference writes fmaf: vec*_dot, src/artic/core/vector.art:96-98,
// or where its -ffast-math build contracts, the slab test). Citations are to
// src/artic/... of the reference.
#pragma once

#include <hip/hip_runtime.h>

#include "ig_detmath.h"
#include "ig_tables.h"

#define IG_DEV __device__ __forceinline__

namespace igdev {

constexpr float kFltEps = 1.1920928955e-07f; // core/common.art:3
constexpr float kFltMax = 3.4028234664e+38f; // core/common.art:4
constexpr float kPi     = 3.14159265359f;    // core/common.art:7
constexpr float kInvPi  = 0.31830988618379067154f;

struct f2 {
    float x, y;
};
struct f3 {
    float x, y, z;
};
struct m33 {
    f3 c0, c1, c2; // columns
};
struct m34 {
    f3 c0, c1, c2, c3; // columns
};

IG_DEV f3 mk3(float x, float y, float z) { return f3{ x, y, z }; }
IG_DEV f3 operator+(f3 a, f3 b) { return f3{ a.x + b.x, a.y + b.y, a.z + b.z }; }
IG_DEV f3 operator-(f3 a, f3 b) { return f3{ a.x - b.x, a.y - b.y, a.z - b.z }; }
IG_DEV f3 operator-(f3 a) { return f3{ -a.x, -a.y, -a.z }; }
IG_DEV f3 operator*(f3 a, f3 b) { return f3{ a.x * b.x, a.y * b.y, a.z * b.z }; }
IG_DEV f3 operator*(f3 a, float s) { return f3{ a.x * s, a.y * s, a.z * s }; }

// core/vector.art:97
IG_DEV float dot3(f3 a, f3 b) { return igm_fma(a.x, b.x, igm_fma(a.y, b.y, a.z * b.z)); }
// core/vector.art:102-105
IG_DEV f3 cross3(f3 a, f3 b) { return f3{ a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x }; }
IG_DEV float len3(f3 v) { return igm_sqrt(dot3(v, v)); }
IG_DEV f3 normalize3(f3 v) { return v * (1 / len3(v)); } // core/vector.art:138

// core/vector.art:98 with b = (v, w)
IG_DEV float dot4w(float ax, float ay, float az, float aw, f3 v, float w)
{
    return igm_fma(ax, v.x, igm_fma(ay, v.y, igm_fma(az, v.z, aw * w)));
}
// core/matrix.art:120-123,246-247
IG_DEV f3 xform_point(const m34& m, f3 v)
{
    return f3{ dot4w(m.c0.x, m.c1.x, m.c2.x, m.c3.x, v, 1), dot4w(m.c0.y, m.c1.y, m.c2.y, m.c3.y, v, 1), dot4w(m.c0.z, m.c1.z, m.c2.z, m.c3.z, v, 1) };
}
IG_DEV f3 xform_dir(const m34& m, f3 v)
{
    return f3{ dot4w(m.c0.x, m.c1.x, m.c2.x, m.c3.x, v, 0), dot4w(m.c0.y, m.c1.y, m.c2.y, m.c3.y, v, 0), dot4w(m.c0.z, m.c1.z, m.c2.z, m.c3.z, v, 0) };
}
// core/matrix.art:110-113
IG_DEV f3 mul33(const m33& m, f3 v)
{
    return f3{ dot3(f3{ m.c0.x, m.c1.x, m.c2.x }, v), dot3(f3{ m.c0.y, m.c1.y, m.c2.y }, v), dot3(f3{ m.c0.z, m.c1.z, m.c2.z }, v) };
}
// core/matrix.art:24-32
IG_DEV m33 orthonormal_basis(f3 n)
{
    const float sign = igm_copysign(1.0f, n.z);
    const float a    = -1 / (sign + n.z);
    const float b    = n.x * n.y * a;
    m33 m;
    m.c0 = f3{ 1 + sign * n.x * n.x * a, sign * b, -sign * n.x };
    m.c1 = f3{ b, sign + n.y * n.y * a, -n.y };
    m.c2 = n;
    return m;
}

// 1 / x where the sources write a reciprocal (shade_core.h). IG_FAST_RCP: v_rcp_f32 + one Newton step, which IS the correctly rounded
// quotient for 2^-100 <= |x| <= 2^100 (every bit pattern compared, tools/rcp_exhaustive.hip, profiles/r03_rcp_exhaustive.txt), with the
// compiler's division behind a branch for the rest: 3 + 4 instructions instead of 11. Measured in round 6 (profiles/r06_experiment_ab.txt
// section 6); the default build divides.
#ifndef IG_FAST_RCP
#define IG_FAST_RCP 0
#endif
IG_DEV float igm_rcp(float x)
{
#if IG_FAST_RCP
    const float r  = __builtin_amdgcn_rcpf(x);
    float q        = igm_fma(r, igm_fma(-x, r, 1.0f), r);
    const float ax = igm_abs(x);
    if (__builtin_expect(!(ax >= 0x1p-100f && ax <= 0x1p100f), 0))
        q = 1 / x;
    return q;
#else
    return 1 / x;
#endif
}

// core/common.art:210-215
IG_DEV float safe_rcp(float x)
{
    const float ax = x > 0 ? x : -x;
    if (ax < 1e-8f)
        return igm_float(igm_bits(kFltMax) ^ (igm_bits(x) & 0x80000000u));
    return 1 / x;
}
IG_DEV float safe_div(float a, float b) { return igm_abs(b) <= kFltEps ? 0.0f : a / b; } // core/common.art:263
IG_DEV float safe_sqrt(float a) { return igm_sqrt(igm_max(0.0f, a)); }                   // core/common.art:265
// core/common.art:285-290
IG_DEV float sum_of_prod(float a, float b, float c, float d)
{
    const float cd  = c * d;
    const float sum = igm_fma(a, b, cd);
    const float err = igm_fma(c, d, -cd);
    return sum + err;
}

// Ray with the precomputed slab-test terms (traversal/ray.art:9-39)
struct RayT {
    f3 org, dir, inv_dir, inv_org;
};

IG_DEV RayT make_ray_terms(f3 org, f3 dir)
{
    RayT r;
    r.org     = org;
    r.dir     = dir;
    r.inv_dir = f3{ safe_rcp(dir.x), safe_rcp(dir.y), safe_rcp(dir.z) };
    r.inv_org = -(org * r.inv_dir);
    return r;
}

// traversal/intersection.art:223-234, unordered, IEEE min/max (make_amdgpu_min_max,
// traversal/mapping_gpu.art:63)
IG_DEV void slab_test(const RayT& r, float tmin, float tmax, float lox, float hix, float loy, float hiy, float loz, float hiz, float& entry, float& exit)
{
    const float t0x = igm_fma(r.inv_dir.x, lox, r.inv_org.x);
    const float t0y = igm_fma(r.inv_dir.y, loy, r.inv_org.y);
    const float t0z = igm_fma(r.inv_dir.z, loz, r.inv_org.z);
    const float t1x = igm_fma(r.inv_dir.x, hix, r.inv_org.x);
    const float t1y = igm_fma(r.inv_dir.y, hiy, r.inv_org.y);
    const float t1z = igm_fma(r.inv_dir.z, hiz, r.inv_org.z);
    entry = igm_max(igm_max(igm_min(t0x, t1x), igm_min(t0y, t1y)), igm_max(igm_min(t0z, t1z), tmin));
    exit  = igm_min(igm_min(igm_max(t0x, t1x), igm_max(t0y, t1y)), igm_min(igm_max(t0z, t1z), tmax));
}

// traversal/intersection.art:74-106 (Moeller-Trumbore, no culling)
IG_DEV bool tri_test(const RayT& r, float tmin, float tmax, f3 v0, f3 e1, f3 e2, f3 n, float& t_out, float& u_out, float& v_out)
{
    const f3 c         = v0 - r.org;
    const f3 rr        = cross3(c, r.dir);
    const float det    = dot3(n, r.dir);
    const float adet   = igm_abs(det);
    const uint32_t sgn = igm_bits(det) & 0x80000000u;
    const float u      = igm_float(igm_bits(dot3(rr, e1)) ^ sgn);
    const float v      = igm_float(igm_bits(dot3(rr, e2)) ^ sgn);
    if (!((u >= 0) & (v >= 0) & (u + v <= adet) & (det != 0)))
        return false;
    const float t = igm_float(igm_bits(dot3(c, n)) ^ sgn);
    if (!((t >= adet * tmin) & (t <= adet * tmax)))
        return false;
    const float rcp = 1 / adet;
    t_out           = t * rcp;
    u_out           = igm_max(u * rcp, 0.0f);
    v_out           = igm_max(v * rcp, 0.0f);
    return true;
}

// The same test in two parts, for callers that test under a lane mask (traverse_core.h): the first part is arithmetic every lane
// of the region runs (identical operations, hence identical t / u / v where the test passes); the division and the three products
// of the second part are run by the lanes that hit only.
struct TriCandidate {
    float t, u, v, adet; // not yet divided by |det|
};
IG_DEV bool tri_test_candidate(const RayT& r, float tmin, float tmax, f3 v0, f3 e1, f3 e2, f3 n, TriCandidate& c_out)
{
    const f3 c         = v0 - r.org;
    const f3 rr        = cross3(c, r.dir);
    const float det    = dot3(n, r.dir);
    const float adet   = igm_abs(det);
    const uint32_t sgn = igm_bits(det) & 0x80000000u;
    const float u      = igm_float(igm_bits(dot3(rr, e1)) ^ sgn);
    const float v      = igm_float(igm_bits(dot3(rr, e2)) ^ sgn);
    const float t      = igm_float(igm_bits(dot3(c, n)) ^ sgn);
    c_out              = TriCandidate{ t, u, v, adet };
    return (u >= 0) & (v >= 0) & (u + v <= adet) & (det != 0) & (t >= adet * tmin) & (t <= adet * tmax);
}
IG_DEV void tri_test_finish(const TriCandidate& c, float& t_out, float& u_out, float& v_out)
{
    const float rcp = 1 / c.adet;
    t_out           = c.t * rcp;
    u_out           = igm_max(c.u * rcp, 0.0f);
    v_out           = igm_max(c.v * rcp, 0.0f);
}

} // namespace igdev
