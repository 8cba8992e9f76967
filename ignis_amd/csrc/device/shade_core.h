// shade_core.h — per-ray shading for the gfx950 kernels: RNG, surface reconstruction, BSDFs,
// lights and the path-tracer callbacks, shared by the wavefront kernel (shade.hip) and the tail
// kernel (tail.hip). Arithmetic follows src/artic/{core,bsdf,light,technique} expression by
// expression (citations inline); transcendental functions come from include/ig_detmath.h.
#pragma once

#include <type_traits>

#include "dev_math.h"
#include "kernels.h"
#include "ig_expr.h"

namespace igdev {

// ---------------------------------------------------------------- RNG (core/random.art)

IG_DEV uint32_t fnv_step(uint32_t h, uint32_t d) // random.art:7-13
{
    h = (h * 16777619u) ^ (d & 0xFF);
    h = (h * 16777619u) ^ ((d >> 8) & 0xFF);
    h = (h * 16777619u) ^ ((d >> 16) & 0xFF);
    h = (h * 16777619u) ^ ((d >> 24) & 0xFF);
    return h;
}

IG_DEV uint32_t make_seed(int sample, int iter, int frame, int x, int y, int user) // random.art:34-43
{
    uint32_t h = 0x811C9DC5u;
    h          = fnv_step(h, (uint32_t)sample);
    h          = fnv_step(h, (uint32_t)iter);
    h          = fnv_step(h, (uint32_t)frame);
    h          = fnv_step(h, (uint32_t)x);
    h          = fnv_step(h, (uint32_t)y);
    h          = fnv_step(h, (uint32_t)user);
    return h;
}

IG_DEV uint32_t tea4(uint32_t v0, uint32_t v1) // random.art:15-24
{
    uint32_t sum = 0;
#ifdef IG_EXP_TEA_ROUNDS
    constexpr int kRounds = IG_EXP_TEA_ROUNDS; // experiment (wrong images): the share of the generator in the shading kernels
#else
    constexpr int kRounds = 4;
#endif
#pragma unroll
    for (int i = 0; i < kRounds; ++i) {
        sum += 0x9e3779b9u;
        v0 += ((v1 << 4) + 0xa341316cu) ^ (v1 + sum) ^ ((v1 >> 5) + 0xc8013ea4u);
        v1 += ((v0 << 4) + 0xad90777du) ^ (v0 + sum) ^ ((v0 >> 5) + 0x7e95761eu);
    }
    return v1;
}

struct Tea {
    uint32_t seed, counter;
    IG_DEV uint32_t u32() { return tea4(seed, counter++); }
    IG_DEV float f32() { return igm_float((u32() & 0x7FFFFFu) | 0x3F800000u) - 1; } // random.art:64-70
    IG_DEV int range(int s, int e)                                                    // next_i32, random.art:46-61,71-73
    {
        const uint32_t range = (uint32_t)(e - s);
        if (range == 0xFFFFFFFFu)
            return (int)u32() + s;
        const uint32_t erange  = range + 1;
        const uint32_t scaling = 0xFFFFFFFFu / erange;
        const uint32_t past    = erange * scaling;
        uint32_t r             = u32();
        while (r >= past)
            r = u32();
        return (int)(r / scaling) + s;
    }
};

// ---------------------------------------------------------------- shading helpers

struct Surf { // driver/surface_element.art
    bool entering;
    f3 point, face_normal;
    f2 tex;
    m33 local;
};

IG_DEV f3 stable_normal(f3 e1, f3 e2, f3 e3) // core/triangle.art:31-44
{
    const float x12 = e1.z * e2.y, y12 = e1.x * e2.z, z12 = e1.y * e2.x;
    const float x23 = e2.z * e3.y, y23 = e2.x * e3.z, z23 = e2.y * e3.x;
    const f3 c12    = f3{ e1.y * e2.z - x12, e1.z * e2.x - y12, e1.x * e2.y - z12 };
    const f3 c23    = f3{ e2.y * e3.z - x23, e2.z * e3.x - y23, e2.x * e3.y - z23 };
    return f3{ igm_abs(x12) < igm_abs(x23) ? c12.x : c23.x, igm_abs(y12) < igm_abs(y23) ? c12.y : c23.y, igm_abs(z12) < igm_abs(z23) ? c12.z : c23.z };
}

IG_DEV float lerp2(float a, float b, float c, float k1, float k2) { return (1 - k1 - k2) * a + k1 * b + k2 * c; } // core/common.art:238

IG_DEV f3 ld3(const float* p) { return f3{ p[0], p[1], p[2] }; }
IG_DEV f3 ld3v(const float* p) // 16-byte aligned (x, y, z, pad) record
{
    const float4 v = *reinterpret_cast<const float4*>(p);
    return f3{ v.x, v.y, v.z };
}

// make_trimesh_shape.surface_element (shapes/trimesh.art:14-40), entity table (driver/entity.art:12-28),
// point mappers (driver/pointmapper.art:28-36)
template <bool SPHERES>
IG_DEV Surf surface_element(const DevScene& sc, int ent_id, int prim_id, f3 org, f3 dir, float t, float u, float v)
{
    // entity record = 9 x 16 bytes; words 12..35 hold toGlobal 3x4, the normal 3x3, shape id, material id
    const float4* e = reinterpret_cast<const float4*>(sc.entities + (size_t)ent_id * IG_ENTITY_FLOATS);
    const float4 r3 = e[3], r4 = e[4], r5 = e[5], r6 = e[6], r7 = e[7], r8 = e[8];
    m34 global;
    global.c0 = f3{ r3.x, r3.y, r3.z }, global.c1 = f3{ r3.w, r4.x, r4.y }, global.c2 = f3{ r4.z, r4.w, r5.x }, global.c3 = f3{ r5.y, r5.z, r5.w };
    m33 nmat;
    nmat.c0 = f3{ r6.x, r6.y, r6.z }, nmat.c1 = f3{ r6.w, r7.x, r7.y }, nmat.c2 = f3{ r7.z, r7.w, r8.x };
    if (SPHERES) {
        const uint4 ext = sc.entity_ext[ent_id];
        if (ext.y == 0xFFFFFFFFu) {
            // analytic sphere: make_sphere_shape.surface_element (shapes/sphere.art:52-73); ext.x = its {centre, radius} record.
            // (only in the full kernel variants, which igd_assign_scene selects for scenes with spheres)
            const float4 sp = *reinterpret_cast<const float4*>(sc.shape_data + ext.x);
            Surf s;
            s.point         = org + dir * t;
            const f3 d      = s.point - xform_point(global, f3{ sp.x, sp.y, sp.z });
            const float len = len3(d);
            const f3 n      = d * (igm_rcp(len));
            s.tex           = f2{ u, v };
            s.entering      = true;
            s.face_normal   = n;
            s.local         = orthonormal_basis(n);
            return s;
        }
    }
    // the triangle's record (igd_assign_scene gathers what the shape's index record points at — header {faces, vertices, normals,
    // texcoords}, then vertices, normals, indices, texcoords: TriMeshProvider.cpp:575-596 — into six rows per triangle)
    const float4* rec = sc.prim_records + ((size_t)sc.entity_rec[ent_id] + (size_t)prim_id * 6);
    const float4 a0 = rec[0], a1 = rec[1], a2 = rec[2], a3 = rec[3], a4 = rec[4], a5 = rec[5];

    const f3 v0 = xform_point(global, f3{ a0.x, a0.y, a0.z });
    const f3 v1 = xform_point(global, f3{ a1.x, a1.y, a1.z });
    const f3 v2 = xform_point(global, f3{ a2.x, a2.y, a2.z });
    const f3 e1 = v2 - v0, e2 = v0 - v1, e3 = v1 - v2;
    const f3 n  = stable_normal(e1, e2, e3); // make_triangle, core/triangle.art:12-29
    const float nn = len3(n);
    const f3 fn    = n * (igm_rcp(nn));

    const f3 n0 = f3{ a3.x, a3.y, a3.z }, n1 = f3{ a4.x, a4.y, a4.z }, n2 = f3{ a5.x, a5.y, a5.z };
    const f3 ln = f3{ lerp2(n0.x, n1.x, n2.x, u, v), lerp2(n0.y, n1.y, n2.y, u, v), lerp2(n0.z, n1.z, n2.z, u, v) };
    const f3 sn = normalize3(mul33(nmat, ln));

    const f2 t0 = f2{ a0.w, a1.w }, t1 = f2{ a2.w, a3.w }, t2 = f2{ a4.w, a5.w };

    Surf s;
    s.tex         = f2{ lerp2(t0.x, t1.x, t2.x, u, v), lerp2(t0.y, t1.y, t2.y, u, v) };
    s.entering    = dot3(dir, fn) <= 0;
    s.point       = org + dir * t;
    s.face_normal = s.entering ? fn : -fn;
    s.local       = orthonormal_basis(s.entering ? sn : -sn);
    return s;
}

// SurfaceElement.inv_area of a triangle hit (make_triangle, core/triangle.art:12-29: area = |n| / 2), for the wireframe technique only:
// kept out of Surf so that the other kernels do not carry it
IG_DEV float triangle_inv_area(const DevScene& sc, int ent_id, int prim_id)
{
    const float4* e = reinterpret_cast<const float4*>(sc.entities + (size_t)ent_id * IG_ENTITY_FLOATS);
    const float4 r3 = e[3], r4 = e[4], r5 = e[5];
    m34 global;
    global.c0 = f3{ r3.x, r3.y, r3.z }, global.c1 = f3{ r3.w, r4.x, r4.y }, global.c2 = f3{ r4.z, r4.w, r5.x }, global.c3 = f3{ r5.y, r5.z, r5.w };
    const uint4 ext    = sc.entity_ext[ent_id];
    const float* verts = reinterpret_cast<const float*>(sc.shape_data + ext.x);
    const float* inds  = reinterpret_cast<const float*>(sc.shape_data + ext.z);
    const int4 tri     = *reinterpret_cast<const int4*>(inds + prim_id * 4);
    const f3 v0 = xform_point(global, ld3v(verts + tri.x * 4));
    const f3 v1 = xform_point(global, ld3v(verts + tri.y * 4));
    const f3 v2 = xform_point(global, ld3v(verts + tri.z * 4));
    const float nn = len3(stable_normal(v2 - v0, v0 - v1, v1 - v2));
    return safe_div(1, nn / 2);
}

struct Col {
    float r, g, b;
};
IG_DEV Col operator*(Col a, Col b) { return Col{ a.r * b.r, a.g * b.g, a.b * b.b }; }
IG_DEV Col operator*(Col a, float f) { return Col{ a.r * f, a.g * f, a.b * f }; }

IG_DEV float clampf(float v, float l, float u) { return igm_min(u, igm_max(l, v)); } // core/common.art:261
IG_DEV float pos_cos(f3 a, f3 b)
{
    const float c = dot3(a, b);
    return c >= 0 ? c : 0.0f;
} // core/common.art:292-295

// Spherical-rectangle sampling of a plane emitter (Urena et al.), light/area.art:124-257
struct PlaneLight {
    f3 origin, normal, ex, ey;
    float width, height;
    Col radiance;

    IG_DEV explicit PlaneLight(const ig_light& l)
    {
        const float* d = l.d;
        origin         = f3{ d[0], d[1], d[2] };
        const f3 xa    = f3{ d[4], d[5], d[6] };
        const f3 ya    = f3{ d[8], d[9], d[10] };
        normal         = f3{ d[3], d[7], d[11] };
        radiance       = Col{ d[20], d[21], d[22] };
        width          = len3(xa);
        height         = len3(ya);
        ex             = xa * (igm_rcp(width));
        ey             = ya * (igm_rcp(height));
    }

    struct SQ {
        f3 n;
        float x0, y0, z0, x1, y1, b0, b1, k, s;
    };

    IG_DEV static float sacos(float a) { return igm_acos(clampf(a, -1, 1)); }

    IG_DEV SQ sq(f3 from) const // area.art:133-181
    {
        const f3 dir    = origin - from;
        const float x0  = dot3(dir, ex);
        const float y0  = dot3(dir, ey);
        const float z0_ = dot3(dir, normal);
        const float x1  = x0 + width;
        const float y1  = y0 + height;
        const bool pos  = !igm_signbit(z0_);
        SQ q;
        q.z0 = pos ? -z0_ : z0_;
        q.n  = pos ? -normal : normal;

        const float d0 = x0 - x1, d1 = y1 - y0, d2 = x1 - x0, d3 = y0 - y1;
        const float m0 = y0 * d0, m1 = x1 * d1, m2 = y1 * d2, m3 = x0 * d3;
        const float zz = q.z0 * q.z0;
        const float nz0 = m0 / igm_sqrt((d0 * d0) * zz + m0 * m0);
        const float nz1 = m1 / igm_sqrt((d1 * d1) * zz + m1 * m1);
        const float nz2 = m2 / igm_sqrt((d2 * d2) * zz + m2 * m2);
        const float nz3 = m3 / igm_sqrt((d3 * d3) * zz + m3 * m3);

        const float g0 = sacos(-nz0 * nz1);
        const float g1 = sacos(-nz1 * nz2);
        const float g2 = sacos(-nz2 * nz3);
        const float g3 = sacos(-nz3 * nz0);

        q.x0 = x0, q.y0 = y0, q.x1 = x1, q.y1 = y1;
        q.b0 = nz0;
        q.b1 = nz2;
        q.k  = 2 * kPi - g2 - g3;
        q.s  = g0 + g1 - q.k;
        return q;
    }

    IG_DEV void sample(float ux, float uy, f3 from, f3& p, float& pdf_s, float& weight) const // area.art:183-222
    {
        const SQ q     = sq(from);
        const float au = igm_fma(ux, q.s, q.k);
        const float fu = igm_fma(igm_cos(au), q.b0, -q.b1) / igm_sin(au);
        const float cu = clampf(igm_copysign(1.0f, fu) / igm_sqrt(sum_of_prod(fu, fu, q.b0, q.b0)), -1, 1);
        const float xu = clampf(-(cu * q.z0) / igm_sqrt(igm_fma(-cu, cu, 1.0f)), q.x0, q.x1);
        const float d   = igm_sqrt(sum_of_prod(xu, xu, q.z0, q.z0));
        const float h0  = q.y0 / igm_sqrt(sum_of_prod(d, d, q.y0, q.y0));
        const float h1  = q.y1 / igm_sqrt(sum_of_prod(d, d, q.y1, q.y1));
        const float hv  = igm_fma(uy, h1 - h0, h0);
        const float hv2 = hv * hv;
        const float yv  = (hv2 < 1 - 1e-6f) ? (hv * d) / igm_sqrt(1 - hv2) : q.y1;
        p      = from + (ex * xu + (ey * yv + q.n * q.z0));
        pdf_s  = safe_div(1, q.s);
        weight = q.s;
    }

    IG_DEV float pdf(f3 from) const { return safe_div(1, sq(from).s); } // area.art:224-228
};

// core/fresnel.art:7-27
IG_DEV float fresnel_factor(float eta, float cos_i, float cos_t)
{
    const float rs = safe_div(eta * cos_i - cos_t, eta * cos_i + cos_t);
    const float rp = safe_div(cos_i - eta * cos_t, cos_i + eta * cos_t);
    return clampf((rs * rs + rp * rp) * 0.5f, 0, 1);
}

IG_DEV bool fresnel(float eta, float cos_i, float& cos_t, float& factor)
{
    const float eta2   = cos_i < 0 ? igm_rcp(eta) : eta;
    const float cos2_t = 1 - (1 - cos_i * cos_i) * eta2 * eta2; // snell
    if (cos2_t <= 0.0f)
        return false;
    const float ct = igm_sqrt(cos2_t);
    cos_t          = cos_i < 0 ? -ct : ct;
    factor         = fresnel_factor(eta2, igm_abs(cos_i), ct);
    return true;
}

// ---------------------------------------------------------------- BSDFs

// core/fresnel.art:29-36
IG_DEV float conductor_factor(float n, float k, float cos_i)
{
    const float f  = n * n + k * k;
    const float d1 = f * cos_i * cos_i;
    const float d2 = 2.0f * n * cos_i;
    const float rs = safe_div(d1 - d2, d1 + d2);
    const float rp = safe_div(f - d2 + cos_i * cos_i, f + d2 + cos_i * cos_i);
    return clampf((rs * rs + rp * rp) * 0.5f, 0, 1);
}

IG_DEV float abs_cos(f3 a, f3 b) { return igm_abs(dot3(a, b)); } // core/common.art:302

// GGX model (core/microfacet.art:158-199) + VNDF sampling with spherical caps (:372-404)
struct Ggx {
    m33 local;
    float au, av;

    IG_DEV float D(f3 m) const
    {
        const float cz = dot3(local.c2, m), cx = dot3(local.c0, m), cy = dot3(local.c1, m);
        const float kx = cx / au, ky = cy / av;
        const float k  = kx * kx + ky * ky + cz * cz;
        return safe_div(1, kPi * au * av * k * k);
    }
    IG_DEV float G1(f3 w) const
    {
        const float cz = dot3(local.c2, w);
        if (igm_abs(cz) <= kFltEps)
            return 0;
        const float cx = dot3(local.c0, w), cy = dot3(local.c1, w);
        const float kx = au * cx, ky = av * cy;
        const float a2 = kx * kx + ky * ky;
        if (a2 <= kFltEps)
            return 1;
        const float k2 = a2 / (cz * cz);
        return 2 / (1 + igm_sqrt(1 + k2));
    }
    IG_DEV float pdf(f3 w, f3 h) const { return safe_div(G1(w) * abs_cos(w, h) * D(h), abs_cos(local.c2, w)); }
    IG_DEV f3 sample(Tea& rnd, f3 vN) const
    {
        const f3 vL    = f3{ dot3(local.c0, vN), dot3(local.c1, vN), dot3(local.c2, vN) };
        const f3 sL    = normalize3(f3{ au * vL.x, av * vL.y, vL.z });
        const float u0 = rnd.f32();
        const float u1 = rnd.f32();
        const float phi = 2 * kPi * u0;
        const float z   = (1 - u1) * (1 + sL.z) - sL.z;
        const float st  = igm_sqrt(clampf(1 - z * z, 0, 1));
        const f3 h      = f3{ st * igm_cos(phi), st * igm_sin(phi), z } + vL;
        const f3 Nh     = normalize3(f3{ h.x * au, h.y * av, h.z });
        return (local.c0 * Nh.x + local.c1 * Nh.y) + local.c2 * Nh.z;
    }
};

// make_checkerboard_texture, identity transform (texture/checkerboard.art; math::wrap core/math.art:88-91)
IG_DEV float wrapf(float v, float mn, float mx)
{
    const float range = mx - mn;
    return range <= kFltEps ? mn : v - (range * igm_floor((v - mn) / range));
}

// ---- bitmap textures (texture/image.art, driver/image.art:9-16): packed 8-bit texels, divided by 255 on fetch
IG_DEV Col image_pixel(const DevScene& sc, const ig_texture& t, int x, int y)
{
    const uint8_t* base = sc.texture_data + t.offset;
    if (t.channels & IG_TEX_FLOAT_BIT) { // float images (driver/image.art:1-7 over device.load_image): as stored
        if ((t.channels & 0xFFu) == 1) {
            const float g = reinterpret_cast<const float*>(base)[y * (int)t.width + x];
            return Col{ g, g, g };
        }
        const float4 c = reinterpret_cast<const float4*>(base)[y * (int)t.width + x];
        return Col{ c.x, c.y, c.z };
    }
    if (t.channels == 1) {
        const float g = (float)base[y * (int)t.width + x] / 255;
        return Col{ g, g, g };
    }
    const uint32_t packed = reinterpret_cast<const uint32_t*>(base)[y * (int)t.width + x];
    return Col{ (float)(packed & 0xFFu) / 255, (float)((packed >> 8) & 0xFFu) / 255, (float)((packed >> 16) & 0xFFu) / 255 };
}
IG_DEV int image_border(uint32_t mode, int x, int w) // image.art:9-40
{
    if (mode == IG_WRAP_CLAMP)
        return x < 0 ? 0 : (x > w - 1 ? w - 1 : x);
    if (mode == IG_WRAP_MIRROR) {
        const int t = x < 0 ? -1 - x : x;
        const int i = t / w;
        const int k = t - i * w;
        return (i & 1) == 0 ? w - 1 - k : k;
    }
    const int t = x % w;
    return t < 0 ? t + w : t;
}
IG_DEV Col lerp_col(Col a, Col b, float t) // core/color.art:17-21
{
    return Col{ (1 - t) * a.r + t * b.r, (1 - t) * a.g + t * b.g, (1 - t) * a.b + t * b.b };
}
IG_DEV float cubic_w0(float a) { return (a * (a * (-a + 3) - 3) + 1) / 6; } // image.art:107-118
IG_DEV float cubic_w1(float a) { return (a * a * (3 * a - 6) + 4) / 6; }
IG_DEV float cubic_w2(float a) { return (a * (a * (-3 * a + 3) + 3) + 1) / 6; }
IG_DEV float cubic_w3(float a) { return (a * a * a) / 6; }
IG_DEV float cubic_g0(float a) { return cubic_w0(a) + cubic_w1(a); }
IG_DEV float cubic_g1(float a) { return cubic_w2(a) + cubic_w3(a); }
IG_DEV float cubic_h0(float a) { return (cubic_w1(a) / cubic_g0(a)) - 1; }
IG_DEV float cubic_h1(float a) { return (cubic_w3(a) / cubic_g1(a)) + 1; }

// make_image_texture with an identity transform (image.art:158-163) over the three filters (image.art:85-156)
IG_DEV Col image_lookup(const DevScene& sc, const ig_texture& t, f2 uv)
{
    const int W = (int)t.width, H = (int)t.height;
    if (t.filter == IG_TEX_NEAREST) {
        const float u = uv.x * (float)W, v = uv.y * (float)H;
        return image_pixel(sc, t, image_border(t.wrap_u, (int)igm_floor(u), W), image_border(t.wrap_v, (int)igm_floor(v), H));
    }
    const float u = uv.x * (float)W - 0.5f;
    const float v = uv.y * (float)H - 0.5f;
    const int ix = (int)igm_floor(u), iy = (int)igm_floor(v);
    const float fx = u - igm_floor(u), fy = v - igm_floor(v);
    if (t.filter == IG_TEX_BILINEAR) {
        const int x0 = image_border(t.wrap_u, ix, W), x1 = image_border(t.wrap_u, ix + 1, W);
        const int y0 = image_border(t.wrap_v, iy, H), y1 = image_border(t.wrap_v, iy + 1, H);
        return lerp_col(lerp_col(image_pixel(sc, t, x0, y0), image_pixel(sc, t, x1, y0), fx),
                        lerp_col(image_pixel(sc, t, x0, y1), image_pixel(sc, t, x1, y1), fx), fy);
    }
    const float g0x = cubic_g0(fx), g0y = cubic_g0(fy), g1x = cubic_g1(fx), g1y = cubic_g1(fy);
    const int x0 = image_border(t.wrap_u, (int)igm_floor((float)ix + cubic_h0(fx) + 0.5f), W);
    const int y0 = image_border(t.wrap_v, (int)igm_floor((float)iy + cubic_h0(fy) + 0.5f), H);
    const int x1 = image_border(t.wrap_u, (int)igm_floor((float)ix + cubic_h1(fx) + 0.5f), W);
    const int y1 = image_border(t.wrap_v, (int)igm_floor((float)iy + cubic_h1(fy) + 0.5f), H);
    const Col p00 = image_pixel(sc, t, x0, y0) * (g0x * g0y);
    const Col p10 = image_pixel(sc, t, x1, y0) * (g1x * g0y);
    const Col p01 = image_pixel(sc, t, x0, y1) * (g0x * g1y);
    const Col p11 = image_pixel(sc, t, x1, y1) * (g1x * g1y);
    const Col a{ p00.r + p10.r, p00.g + p10.g, p00.b + p10.b }, c{ p01.r + p11.r, p01.g + p11.g, p01.b + p11.b };
    return Col{ a.r + c.r, a.g + c.g, a.b + c.b };
}

// ---- bump mapping (bsdf/map.art:36-42,64-67; MapBSDF.cpp:44-47)
IG_DEV f3 ensure_valid_reflection(f3 Ng, f3 I, f3 N) // core/sampling.art:118-166
{
    const f3 R            = N * (2 * dot3(N, I)) - I; // vec3_reflect
    const float threshold = igm_min(0.9f * dot3(Ng, I), 0.01f);
    if (dot3(Ng, R) >= threshold)
        return N;
    const float NdotNg = dot3(N, Ng);
    const f3 X         = normalize3(N - Ng * NdotNg);
    const float Ix = dot3(I, X), Iz = dot3(I, Ng);
    const float Ix2 = Ix * Ix, Iz2 = Iz * Iz;
    const float a   = Ix2 + Iz2;
    const float b   = safe_sqrt(Ix2 * (a - threshold * threshold));
    const float c   = Iz * threshold + a;
    const float fac = 0.5f / a;
    const float N1_z2 = fac * (b + c), N2_z2 = fac * (-b + c);
    const bool valid1 = (N1_z2 > 1e-5f) && (N1_z2 <= (1.0f + 1e-5f));
    const bool valid2 = (N2_z2 > 1e-5f) && (N2_z2 <= (1.0f + 1e-5f));
    f2 Nn;
    if (valid1 && valid2) {
        const f2 N1{ safe_sqrt(1 - N1_z2), safe_sqrt(N1_z2) };
        const f2 N2{ safe_sqrt(1 - N2_z2), safe_sqrt(N2_z2) };
        const float R1 = 2 * (N1.x * Ix + N1.y * Iz) * N1.y - Iz;
        const float R2 = 2 * (N2.x * Ix + N2.y * Iz) * N2.y - Iz;
        if (R1 >= 1e-5f && R2 >= 1e-5f)
            Nn = R1 < R2 ? N1 : N2;
        else
            Nn = R1 > R2 ? N1 : N2;
    } else if (valid1 || valid2) {
        const float Nz2 = valid1 ? N1_z2 : N2_z2;
        Nn              = f2{ safe_sqrt(1 - Nz2), safe_sqrt(Nz2) };
    } else {
        Nn = f2{ 0, 1 };
    }
    return X * Nn.x + Ng * Nn.y;
}
IG_DEV m33 align_vectors(f3 a, f3 b) // core/matrix.art:261-284
{
    const f3 axis    = cross3(b, a);
    const float cosA = dot3(a, b);
    m33 m;
    if (cosA <= -1) {
        m.c0 = f3{ -1, 0, 0 }, m.c1 = f3{ 0, -1, 0 }, m.c2 = f3{ 0, 0, -1 };
        return m;
    }
    const float k = igm_rcp(1 + cosA);
    m.c0 = f3{ (axis.x * axis.x * k) + cosA, (axis.y * axis.x * k) - axis.z, (axis.z * axis.x * k) + axis.y };
    m.c1 = f3{ (axis.x * axis.y * k) + axis.z, (axis.y * axis.y * k) + cosA, (axis.z * axis.y * k) - axis.x };
    m.c2 = f3{ (axis.x * axis.z * k) - axis.y, (axis.y * axis.z * k) + axis.x, (axis.z * axis.z * k) + cosA };
    return m;
}
#ifndef IG_EXPR_REGS_LDS
#define IG_EXPR_REGS_LDS 1
#endif
// ---- shading expressions (PExpr strings compiled by the loader into the bytecode of include/ig_expr.h; the reference
// transpiles them to Artic instead, src/runtime/loader/Transpiler.cpp). The variables are those of sInternalVariables
// (Transpiler.cpp:338-363) this backend carries; texture alpha reads as 1 (Col has no alpha).
struct ExprCtx {
    const DevScene* sc;
    const Surf* s;
    f3 view; // "V" / "Rd": -ctx.ray.dir
    IG_DEV static ige_v4 v3(f3 a) { return ige_v4{ { a.x, a.y, a.z, 0.0f } }; }
    IG_DEV ige_v4 var(int id) const
    {
        switch (id) {
        case IGE_VAR_UVW: return ige_v4{ { s->tex.x, s->tex.y, 0.0f, 0.0f } };
        case IGE_VAR_P: return v3(s->point);
        case IGE_VAR_V: return v3(view);
        case IGE_VAR_N: return v3(s->local.c2);
        case IGE_VAR_NG: return v3(s->face_normal);
        case IGE_VAR_NX: return v3(s->local.c0);
        case IGE_VAR_NY: return v3(s->local.c1);
        default: { // IGE_VAR_FRONT
            const float f = s->entering ? 1.0f : 0.0f;
            return ige_v4{ { f, f, f, f } };
        }
        }
    }
    IG_DEV ige_v4 tex(uint32_t id, float u, float v) const
    {
        const Col c = image_lookup(*sc, sc->textures[id], f2{ u, v });
        return ige_v4{ { c.r, c.g, c.b, 1.0f } };
    }
    IG_DEV ige_v4 evr(ige_v4 ng, ige_v4 v, ige_v4 n) const
    {
        return v3(ensure_valid_reflection(f3{ ng.v[0], ng.v[1], ng.v[2] }, f3{ v.v[0], v.v[1], v.v[2] }, f3{ n.v[0], n.v[1], n.v[2] }));
    }
};
// The interpreter's register file in LDS: a program names its registers at run time, so a private array would be scratch memory
// (1.1 KB per lane of the expression kernels in round 2). One column of 12 x 16 B per lane of the largest workgroup (48 KiB; the
// kernels that evaluate expressions run at two waves per SIMD).
#ifndef IG_SHADE_THREADS
#define IG_SHADE_THREADS 256
#endif
constexpr int kExprLanes = IG_SHADE_THREADS > 256 ? IG_SHADE_THREADS : 256;
struct ExprLdsRegs {
    ige_v4* column;
    IG_DEV ige_v4& operator[](uint32_t i) const { return column[i * kExprLanes]; }
};
// Inlined into the expression kernels (IG_EXPR_INLINE=0: a call; a 256-register caller around a 193-register callee then saves and restores
// some 150 registers per evaluation: 1 KB of scratch traffic, cycles-bumpmap 2 900 against 3 640 Mrays/s)
#ifndef IG_EXPR_INLINE
#define IG_EXPR_INLINE 1
#endif
#if IG_EXPR_INLINE
IG_DEV f3 eval_expr(const DevScene& sc, int32_t start, const Surf& s, f3 view)
#else
__attribute__((noinline)) IG_DEV f3 eval_expr(const DevScene& sc, int32_t start, const Surf& s, f3 view)
#endif
{
#if IG_EXPR_REGS_LDS
    __shared__ ige_v4 s_expr_regs[IGE_REGS][kExprLanes];
    const ExprLdsRegs regs{ &s_expr_regs[0][threadIdx.x] };
    for (int i = 0; i < IGE_REGS; ++i)
        regs[i] = ige_v4{ { 0, 0, 0, 0 } };
    const ige_v4 r = ige_run(sc.expr_code + start, ExprCtx{ &sc, &s, view }, regs);
#else
    const ige_v4 r = ige_run(sc.expr_code + start, ExprCtx{ &sc, &s, view });
#endif
    return f3{ r.v[0], r.v[1], r.v[2] };
}

// IG_MAT_EXPR_NUMBERS (ig_tables.h): the material record with its number expressions evaluated at this hit — in a local copy; the
// table's record otherwise. Numbers see the hit's own surface, like every parameter of the material the reference builds per hit.
template <bool EXPR>
IG_DEV const ig_material& resolve_material(const DevScene& sc, const ig_material& mat, const Surf& s, f3 view, ig_material& local)
{
    if constexpr (EXPR) {
        if (mat.flags & IG_MAT_EXPR_NUMBERS) {
            local               = mat;
            const uint32_t* lst = sc.expr_code + igm_bits(mat.r[7]);
            const uint32_t n    = lst[0];
            for (uint32_t i = 0; i < n; ++i)
                ig_material_set_number(&local, lst[1 + 3 * i], igm_float(lst[2 + 3 * i]), eval_expr(sc, (int32_t)lst[3 + 3 * i], s, view).x);
            return local;
        }
    }
    return mat;
}

// the local frame the inner BSDF of a bump-mapped material sees (make_bumpmap -> make_normal_set; camera paths
// are not adjoint, so nothing else of transform_surf_bsdf applies)
template <bool EXPR>
IG_DEV m33 bumped_frame(const DevScene& sc, const ig_material& mat, const Surf& s, f3 ray_dir)
{
    f3 N;
    if (EXPR && (mat.flags & IG_MAT_EXPR_NORMAL)) {
        // "transform" BSDF: make_normal_set(ctx, inner, normal expression) (TransformBSDF.cpp:43-46, bsdf/map.art:36-42)
        N = eval_expr(sc, mat.tex_id, s, -ray_dir);
    } else if (mat.flags & IG_MAT_NORMALMAP) {
        const ig_texture& t = sc.textures[mat.tex_id];
        // make_normalmap (bsdf/map.art:55-61): normal given as [0, 1] RGB; mat3x3_left_mul = (col_i . v)
        const Col c    = image_lookup(sc, t, s.tex);
        const f3 nt    = normalize3(f3{ 2 * c.r - 1, 2 * c.g - 1, 2 * c.b - 1 });
        const f3 oN    = f3{ dot3(s.local.c0, nt), dot3(s.local.c1, nt), dot3(s.local.c2, nt) };
        const float st = mat.p[11];
        N = st != 1 ? normalize3(s.local.c2 + (oN - s.local.c2) * st) : oN;
    } else {
        const ig_texture& t = sc.textures[mat.tex_id];
        const float delta = 0.001f; // texture_dx / texture_dy (texture/common.art:33-43)
        const Col c0      = image_lookup(sc, t, s.tex);
        const Col cx      = image_lookup(sc, t, f2{ s.tex.x + delta, s.tex.y });
        const Col cy      = image_lookup(sc, t, f2{ s.tex.x, s.tex.y + delta });
        const float dx    = (cx.r - c0.r) * (igm_rcp(delta));
        const float dy    = (cy.r - c0.r) * (igm_rcp(delta));
        N = normalize3(s.local.c2 - (s.local.c0 * dx + s.local.c1 * dy) * mat.p[11]);
    }
    const f3 n          = ensure_valid_reflection(s.face_normal, -ray_dir, normalize3(N));
    const m33 trans     = align_vectors(s.local.c2, n);
    m33 out;
    out.c0 = mul33(trans, s.local.c0); // mat3x3_matmul(trans, local), core/matrix.art:124-127
    out.c1 = mul33(trans, s.local.c1);
    out.c2 = mul33(trans, s.local.c2);
    return out;
}

// ---- principled BSDF (bsdf/principled.art); directions of the closure live in the shading frame
IG_DEV float lerpf(float a, float b, float k) { return (1 - k) * a + k * b; }                  // core/common.art:237
IG_DEV float luminance(Col c) { return c.r * 0.2126f + c.g * 0.7152f + c.b * 0.0722f; }       // core/color.art:29,81-83
IG_DEV Col operator+(Col a, Col b) { return Col{ a.r + b.r, a.g + b.g, a.b + b.b }; }
IG_DEV float schlick_approx(float f) // core/fresnel.art:88-91
{
    const float s = clampf(1 - f, 0, 1);
    return (s * s) * (s * s) * s;
}
IG_DEV float schlick_r0(float eta) // core/fresnel.art:101-104
{
    const float factor = clampf((eta - 1) / (eta + 1), -1, 1);
    return factor * factor;
}
IG_DEV float fresnel_dielectric(float eta, float cos_i) // core/math.art:120-123
{
    float cos_t, factor;
    return fresnel(eta, cos_i, cos_t, factor) ? factor : 1.0f;
}
IG_DEV float refl_jacobian(float c) { return safe_div(1, 4 * c); } // core/shading.art:69
IG_DEV float refr_jacobian(float eta, float cos_i, float cos_o)     // core/shading.art:71-74
{
    const float jacob_d = cos_i + cos_o * eta;
    return safe_div(eta * eta * cos_i, jacob_d * jacob_d);
}
IG_DEV f3 halfway_refractive(f3 a, f3 b, float eta) { return normalize3(a + b * eta); }          // core/vector.art:142
IG_DEV bool pos_hemi(f3 v) { return v.z >= 0; }                                                  // core/shading.art:62
IG_DEV bool same_hemi(f3 a, f3 b) { return pos_hemi(a) == pos_hemi(b); }                         // core/shading.art:63
IG_DEV f3 make_same_hemi(f3 a, f3 b) { return same_hemi(a, b) ? b : -b; }                        // core/shading.art:65
IG_DEV f3 make_pos_hemi(f3 v) { return pos_hemi(v) ? v : -v; }                                   // core/shading.art:66

struct Principled {
    m33 local;
    bool entering;
    Col base;
    float refl_ior, refr_ior, diff_trans, spec_trans, spec_tint;
    float ru, rv, flatness, metallic, sheen, sheen_tint, clearcoat, cc_gloss, cc_rough;
    bool thin, cc_top_only;
    float refl_eta, refr_eta;

    static constexpr float kMicroEps   = 1e-5f; // principled.art:243-244
    static constexpr float kGrazingEps = 1e-5f;

    // make_principled_bsdf (principled.art:236-268)
    IG_DEV Principled(const ig_material& m, const m33& frame, bool is_entering, Col base_color)
    {
        local       = frame;
        entering    = is_entering;
        base        = base_color;
        refl_ior    = m.p[3];
        refr_ior    = m.p[4];
        diff_trans  = m.p[5];
        spec_trans  = m.p[6];
        spec_tint   = m.p[7];
        ru          = igm_max(1e-3f, m.p[8]);
        rv          = igm_max(1e-3f, m.p[9]);
        flatness    = m.p[10];
        metallic    = m.r[0];
        sheen       = m.r[1];
        sheen_tint  = m.r[2];
        clearcoat   = m.r[3];
        cc_gloss    = m.r[4];
        cc_rough    = m.r[5];
        thin        = (m.flags & IG_MAT_THIN) != 0;
        cc_top_only = (m.flags & IG_MAT_CLEARCOAT_ALL) == 0;
        refl_eta    = (entering || thin) ? igm_rcp(refl_ior) : refl_ior;
        refr_eta    = (entering || thin) ? igm_rcp(refr_ior) : refr_ior;
    }

    IG_DEV static m33 identity()
    {
        m33 m;
        m.c0 = f3{ 1, 0, 0 }, m.c1 = f3{ 0, 1, 0 }, m.c2 = f3{ 0, 0, 1 };
        return m;
    }
    IG_DEV f3 to_local(f3 v) const { return f3{ dot3(local.c0, v), dot3(local.c1, v), dot3(local.c2, v) }; }
    IG_DEV f3 to_world(f3 v) const { return (local.c0 * v.x + local.c1 * v.y) + local.c2 * v.z; }

    // tint_color (principled.art:52-59)
    IG_DEV static Col tint(Col c)
    {
        const float lum = luminance(c);
        return lum <= kFltEps ? Col{ 1, 1, 1 } : c * safe_div(1, lum);
    }
    // getMicro / getReflectionMicro / getRefractionMicro (principled.art:62-78)
    IG_DEV static Ggx micro(float a, float b) { return Ggx{ identity(), igm_max(1e-3f, a * a), igm_max(1e-3f, b * b) }; }
    IG_DEV Ggx refl_micro() const { return micro(ru, rv); }
    IG_DEV Ggx refr_micro() const
    {
        if (thin)
            return micro(clampf((0.65f * refr_ior - 0.35f) * ru, 0, 1), clampf((0.65f * refr_ior - 0.35f) * rv, 0, 1));
        return micro(ru, rv);
    }

    // evalDisneyFresnelTerm (principled.art:80-95)
    IG_DEV Col fresnel_term(f3 wo, f3 wi, f3 h) const
    {
        const float HdV = abs_cos(wo, h);
        const float HdL = abs_cos(wi, h);
        if (HdV * HdL <= kFltEps)
            return Col{ 0, 0, 0 };
        const float f1v = fresnel_dielectric(refl_eta, HdV);
        const Col f1{ f1v, f1v, f1v };
        const Col color = tint(base);
        const Col a     = lerp_col(Col{ 1, 1, 1 }, color, spec_tint);
        const Col r0    = lerp_col(a * schlick_r0(refl_eta), base, metallic);
        const float s   = schlick_approx(HdL); // schlick(r0, white, HdL), core/fresnel.art:93-97
        const Col f2{ r0.r + (1 - r0.r) * s, r0.g + (1 - r0.g) * s, r0.b + (1 - r0.b) * s };
        return lerp_col(f1, f2, metallic);
    }
    // evalSubsurfaceTerm (principled.art:97-108)
    IG_DEV float subsurface_term(f3 wo, f3 wi, f3 h) const
    {
        const float r2    = ru * rv;
        const float HdotL = dot3(wi, h);
        const float fss90 = HdotL * HdotL * r2;
        const float aNdL  = igm_abs(wi.z);
        const float aNdV  = igm_abs(wo.z);
        const float lk    = schlick_approx(aNdL);
        const float vk    = schlick_approx(aNdV);
        const float fss   = (1 - lk + fss90 * lk) * (1 - vk + fss90 * vk);
        return 1.25f * (fss * (igm_rcp(aNdL + aNdV + 1e-5f) - 0.5f) + 0.5f);
    }
    // evalSheenTerm (principled.art:110-113)
    IG_DEV Col sheen_term(f3 wi) const
    {
        const float lk = schlick_approx(igm_abs(wi.z));
        return lerp_col(Col{ 1, 1, 1 }, tint(base), sheen_tint) * (sheen * lk * igm_abs(wi.z));
    }
    // evalDiffuseTerm (principled.art:115-128)
    IG_DEV float diffuse_term(f3 wo, f3 wi, f3 h) const
    {
        const float lk    = schlick_approx(igm_abs(wi.z));
        const float vk    = schlick_approx(igm_abs(wo.z));
        const float diff  = (1 - 0.5f * lk) * (1 - 0.5f * vk);
        const float VdotL = abs_cos(wi, wo);
        const float rr    = (VdotL + 1) * (ru + rv) / 2;
        const float retro = rr * (lk + vk + lk * vk * (rr - 1));
        const float ss    = thin ? 1 - flatness + subsurface_term(wo, wi, h) * flatness : 1.0f;
        return kInvPi * (diff + retro) * ss * igm_abs(wi.z);
    }
    // evalTranslucentTerm (principled.art:130-137)
    IG_DEV float translucent_term(f3 wo, f3 wi) const
    {
        const float lk   = schlick_approx(igm_abs(wi.z));
        const float vk   = schlick_approx(igm_abs(wo.z));
        const float diff = (1 - 0.5f * lk) * (1 - 0.5f * vk);
        return kInvPi * diff * igm_abs(wi.z);
    }
    // evalReflectionTerm (principled.art:139-148)
    IG_DEV Col reflection_term(f3 wo, f3 wi, f3 h) const
    {
        const Ggx m       = refl_micro();
        const Col F       = fresnel_term(wo, wi, h);
        const float D     = m.D(h);
        const float G     = m.G1(wi) * m.G1(wo);
        const float jacob = refl_jacobian(wo.z);
        return F * igm_abs(D * G * jacob);
    }
    // evalRefractionTerm (principled.art:150-176)
    IG_DEV Col refraction_term(f3 wo, f3 wi, f3 h) const
    {
        if (thin) {
            const float fterm = fresnel_dielectric(refr_eta, igm_abs(wo.z));
            const float F     = fterm + (1 - fterm) * fterm / (fterm + 1);
            return Col{ igm_sqrt(base.r), igm_sqrt(base.g), igm_sqrt(base.b) } * (1 - F);
        }
        const Ggx m       = refr_micro();
        const float HdI   = dot3(wi, h);
        const float HdO   = dot3(wo, h);
        const float F     = fresnel_dielectric(refr_eta, igm_abs(HdO));
        const float D     = m.D(h);
        const float G     = m.G1(wi) * m.G1(wo);
        const float jacob = refr_jacobian(refr_eta, HdI, HdO);
        const float norm  = igm_abs(safe_div(HdO * jacob, wo.z));
        return base * ((1 - F) * D * G * norm);
    }
    // evalClearcoatTerm (principled.art:178-191)
    IG_DEV Col clearcoat_term(f3 wo, f3 wi, f3 h) const
    {
        const float F0   = 0.04f;
        const float R    = 0.25f;
        const float R2   = igm_max(0.001f, cc_rough * (1 - cc_gloss) + 0.01f * cc_gloss);
        const float aHdL = abs_cos(wi, h);
        const float d    = Ggx{ identity(), R2, R2 }.D(h);
        const float f    = F0 + (1 - F0) * schlick_approx(aHdL); // schlick_f, core/fresnel.art:99
        const Ggx gm{ identity(), R, R };
        const float g     = gm.G1(wi) * gm.G1(wo);
        const float jacob = refl_jacobian(wo.z);
        const float v     = igm_abs(R * d * f * g * jacob * wi.z);
        return Col{ v, v, v };
    }

    struct Lobes {
        float diff_refl, diff_trans, spec_refl, spec_trans;
    };
    // calcLobeDistribution (principled.art:200-233)
    IG_DEV Lobes lobes(f3 wo) const
    {
        const float metallic_in   = clampf(metallic, 0, 1);
        const float diff_trans_in = clampf(diff_trans, 0, 1);
        const float spec_trans_in = clampf(spec_trans, 0, 1);
        const float abs_gen       = luminance(base);
        const float abs_spec      = lerpf(1, luminance(tint(base)), spec_tint);
        const float d_refl        = clampf(abs_gen * (1 - metallic_in) * (1 - spec_trans_in), 0, 1);
        const float F             = fresnel_dielectric(refr_eta, igm_abs(wo.z));
        const float s_refl        = clampf(abs_spec * (1 - F) + F, 0, 1);
        const bool has_transmission = diff_trans_in > 0 || spec_trans_in > 0;
        if (!has_transmission) {
            const float norm = d_refl + s_refl;
            if (norm > kFltEps)
                return Lobes{ d_refl / norm, 0, s_refl / norm, 0 };
            return Lobes{ 1, 0, 0, 0 };
        }
        const float d_trans = clampf(abs_gen * diff_trans_in * d_refl, 0, 1);
        const float s_trans = clampf((1 - F) * abs_gen * (1 - metallic_in) * spec_trans_in, 0, 1);
        const float norm    = d_refl + s_refl + d_trans + s_trans;
        if (norm > kFltEps)
            return Lobes{ d_refl / norm, d_trans / norm, s_refl / norm, s_trans / norm };
        return Lobes{ 1, 0, 0, 0 };
    }

    // eval (principled.art:270-334)
    IG_DEV Col eval(f3 in_dir, f3 out_dir) const
    {
        const f3 wo = to_local(out_dir);
        const f3 wi = to_local(in_dir);
        const bool is_transmission = !same_hemi(wi, wo);
        const f3 h = make_same_hemi(wo, is_transmission ? halfway_refractive(wi, wo, refr_eta) : normalize3(wi + wo));
        const bool in_front  = entering == pos_hemi(wi);
        const bool out_front = entering == pos_hemi(wo);
        const bool upper     = in_front && out_front;
        if (igm_abs(wi.z) <= kGrazingEps)
            return Col{ 0, 0, 0 };
        Col contrib{ 0, 0, 0 };
        const float diffuse_weight = (thin ? 1.0f : 1 - clampf(metallic, 0, 1)) * (1 - clampf(spec_trans, 0, 1));
        const float trans_weight   = (1 - clampf(metallic, 0, 1)) * clampf(spec_trans, 0, 1);
        const float spec_weight    = 1;
        if (!is_transmission) {
            if (diffuse_weight > 0)
                contrib = contrib + base * (diffuse_term(wo, wi, h) * diffuse_weight);
            if (sheen > 0)
                contrib = contrib + sheen_term(wi) * diffuse_weight;
            contrib = contrib + reflection_term(wo, wi, h) * spec_weight;
            if ((!cc_top_only || upper) && clearcoat > 0)
                contrib = contrib + clearcoat_term(wo, wi, h) * clearcoat;
        } else {
            if (thin && diff_trans > 0)
                contrib = contrib + base * (translucent_term(wo, wi) * diff_trans);
            if (spec_trans > 0)
                contrib = contrib + refraction_term(wo, wi, h) * trans_weight;
        }
        return contrib;
    }

    // diffPdf_local / specReflPdf_local / specTransPdf_local (principled.art:336-359)
    IG_DEV static float bound_spec_pdf(float v) { return v > kMicroEps ? v : 0.0f; }
    IG_DEV static float diff_pdf_local(f3 wi) { return igm_abs(wi.z) / kPi; }
    IG_DEV float spec_refl_pdf_local(f3 wo, f3 wi) const
    {
        const f3 pwo    = make_pos_hemi(wo);
        const f3 pwi    = make_pos_hemi(wi);
        const Ggx m     = refl_micro();
        const f3 H      = normalize3(pwo + pwi);
        const float cho = dot3(pwo, H);
        return igm_abs(bound_spec_pdf(m.pdf(pwo, H)) * refl_jacobian(cho));
    }
    IG_DEV float spec_trans_pdf_local(f3 wo, f3 wi) const
    {
        const f3 pwo    = make_pos_hemi(wo);
        const f3 pwi    = -make_pos_hemi(wi);
        const Ggx m     = refr_micro();
        const f3 H      = halfway_refractive(pwi, pwo, refr_eta);
        const float chi = dot3(pwi, H);
        const float cho = dot3(pwo, H);
        return igm_abs(bound_spec_pdf(m.pdf(pwo, H)) * refr_jacobian(refr_eta, chi, cho));
    }
    // pdf (principled.art:361-377)
    IG_DEV float pdf(f3 in_dir, f3 out_dir) const
    {
        const f3 wo = to_local(out_dir);
        const f3 wi = to_local(in_dir);
        if (igm_abs(wo.z) <= kGrazingEps || igm_abs(wi.z) <= kGrazingEps)
            return 0;
        const Lobes l        = lobes(wo);
        const float diff_pdf = diff_pdf_local(wi);
        if (same_hemi(wo, wi))
            return l.diff_refl * diff_pdf + l.spec_refl * spec_refl_pdf_local(wo, wi);
        if (thin)
            return l.diff_trans * diff_pdf + l.spec_trans;
        return l.diff_trans * diff_pdf + l.spec_trans * spec_trans_pdf_local(wo, wi);
    }

    // sample_cosine_hemisphere (core/sampling.art:62-70)
    IG_DEV static f3 cosine_hemisphere(Tea& rnd, float& pdf_out)
    {
        const float u   = rnd.f32();
        const float v   = rnd.f32();
        const float c   = safe_sqrt(v);
        const float s   = safe_sqrt(1 - v);
        const float phi = 2 * kPi * u;
        pdf_out         = c / kPi;
        return f3{ s * igm_cos(phi), s * igm_sin(phi), c };
    }

    // sample (principled.art:382-476), adjoint = false; false = reject_bsdf_sample()
    IG_DEV bool sample(Tea& rnd, f3 out_dir, f3& in_dir, float& pdf_out, Col& color, float& s_eta, bool adjoint = false) const
    {
        const f3 wo = to_local(out_dir);
        if (igm_abs(wo.z) <= kGrazingEps)
            return false;
        const Lobes l    = lobes(wo);
        const float pick = rnd.f32();
        f3 dir;
        float spdf;
        if (pick < l.diff_refl) {
            float cpdf;
            const f3 d = cosine_hemisphere(rnd, cpdf);
            dir        = make_same_hemi(wo, d);
            spdf       = cpdf * l.diff_refl + spec_refl_pdf_local(wo, dir) * l.spec_refl;
        } else if (pick < l.diff_refl + l.diff_trans) {
            float cpdf;
            const f3 d = cosine_hemisphere(rnd, cpdf);
            dir        = -make_same_hemi(wo, d);
            spdf       = cpdf * l.diff_trans + spec_trans_pdf_local(wo, dir) * l.spec_trans;
        } else if (pick < l.diff_refl + l.diff_trans + l.spec_trans) {
            if (thin) {
                dir  = -wo;
                spdf = l.spec_trans;
            } else {
                const f3 pwo     = make_pos_hemi(wo);
                const Ggx m      = refr_micro();
                const f3 n       = m.sample(rnd, pwo);
                const float mpdf = m.pdf(pwo, n);
                if (mpdf <= kMicroEps || dot3(n, n) <= kFltEps)
                    return false;
                const f3 oH     = normalize3(n);
                const f3 H      = igm_signbit(dot3(oH, pwo)) ? -oH : oH;
                const float cho = dot3(pwo, H);
                float cos_t, factor;
                if (fresnel(refr_eta, cho, cos_t, factor)) {
                    const f3 pwi = normalize3(H * (refr_eta * cho - cos_t) - pwo * refr_eta); // vec3_refract (core/vector.art:126)
                    if (!same_hemi(pwo, pwi) && cho > kFltEps && -pwi.z > kGrazingEps) {
                        dir  = -make_same_hemi(wo, pwi);
                        spdf = igm_abs(mpdf * refr_jacobian(refr_eta, dot3(pwi, H), cho)) * l.spec_trans + diff_pdf_local(dir) * l.diff_trans;
                    } else {
                        return false;
                    }
                } else { // total reflection
                    const f3 pwi = normalize3(H * (2 * dot3(H, pwo)) - pwo); // vec3_reflect (core/vector.art:123)
                    if (same_hemi(pwo, pwi) && cho > kFltEps && pwi.z > kGrazingEps) {
                        dir  = make_same_hemi(wo, pwi);
                        spdf = mpdf * refl_jacobian(cho) * l.spec_trans + diff_pdf_local(dir) * l.diff_trans;
                    } else {
                        return false;
                    }
                }
            }
        } else {
            const f3 pwo     = make_pos_hemi(wo);
            const Ggx m      = refl_micro();
            const f3 n       = m.sample(rnd, pwo);
            const float mpdf = m.pdf(pwo, n);
            if (mpdf <= kMicroEps || dot3(n, n) <= kFltEps)
                return false;
            const f3 oH     = normalize3(n);
            const f3 H      = igm_signbit(dot3(oH, pwo)) ? -oH : oH;
            const float cho = dot3(pwo, H);
            const f3 pwi    = normalize3(H * (2 * dot3(H, pwo)) - pwo);
            if (same_hemi(pwo, pwi) && cho > kFltEps && pwi.z > kGrazingEps) {
                dir  = make_same_hemi(wo, pwi);
                spdf = igm_abs(mpdf * refl_jacobian(cho)) * l.spec_refl + diff_pdf_local(dir) * l.diff_refl;
            } else {
                return false;
            }
        }
        if (!(spdf > kFltEps && igm_abs(spdf) <= 3.402823466e+38f))
            return false;
        s_eta   = (thin || same_hemi(wo, dir)) ? 1.0f : refr_eta;
        in_dir  = to_world(dir);
        pdf_out = spdf;
        // light paths carry 1 / eta^2 across a refraction (principled.art:471)
        const float spread = (adjoint && !thin && !same_hemi(wo, dir)) ? igm_rcp(refr_eta * refr_eta) : 1.0f;
        color   = eval(in_dir, out_dir) * (spread / spdf);
        return true;
    }
};

// make_rough_dielectric_bsdf (bsdf/dielectric.art:64-191) over the VNDF-GGX distribution of the surface frame
struct RoughDielectric {
    f3 N;
    float eta, pdf_eps;
    Col ks, kt;
    Ggx micro;
    static constexpr float kCosEps = 1e-5f;

    IG_DEV RoughDielectric(const ig_material& m, const m33& frame, bool entering)
    {
        N       = frame.c2;
        eta     = entering ? m.p[0] / m.p[1] : m.p[1] / m.p[0];
        pdf_eps = m.p[8];
        ks      = Col{ m.p[2], m.p[3], m.p[4] };
        kt      = Col{ m.p[5], m.p[6], m.p[7] };
        micro   = Ggx{ frame, m.p[9], m.p[10] };
    }
    IG_DEV Col eval(f3 in_dir, f3 out_dir) const
    {
        const float cos_i = dot3(N, in_dir);
        const float cos_o = dot3(N, out_dir);
        if (igm_abs(cos_i * cos_o) <= kCosEps)
            return Col{ 0, 0, 0 };
        const bool is_transmission = igm_signbit(cos_i * cos_o);
        const f3 H      = is_transmission ? halfway_refractive(in_dir, out_dir, eta) : normalize3(in_dir + out_dir);
        const float chi = dot3(H, in_dir);
        const float cho = dot3(H, out_dir);
        if (igm_abs(chi * cho) <= kCosEps)
            return Col{ 0, 0, 0 };
        const float fterm = fresnel_dielectric(eta, igm_abs(cho));
        const float D     = micro.D(H);
        const float G     = micro.G1(in_dir) * micro.G1(out_dir);
        if (!is_transmission)
            return ks * (fterm * D * G * igm_abs(refl_jacobian(cos_o)));
        const float jacob = refr_jacobian(eta, chi, cho);
        const float norm  = igm_abs(safe_div(cho * jacob, cos_o));
        return kt * ((1 - fterm) * D * G * norm);
    }
    IG_DEV float pdf(f3 in_dir, f3 out_dir) const
    {
        const float cos_i = dot3(N, in_dir);
        const float cos_o = dot3(N, out_dir);
        if (igm_abs(cos_i * cos_o) <= kCosEps)
            return 0;
        const bool is_transmission = igm_signbit(cos_i * cos_o);
        const f3 H      = is_transmission ? halfway_refractive(in_dir, out_dir, eta) : normalize3(in_dir + out_dir);
        const float chi = dot3(H, in_dir);
        const float cho = dot3(H, out_dir);
        if (igm_abs(chi * cho) <= kCosEps)
            return 0;
        const float fterm = fresnel_dielectric(eta, igm_abs(cho));
        const float mpdf  = micro.pdf(out_dir, H);
        if (mpdf <= pdf_eps)
            return 0;
        if (!is_transmission)
            return fterm * mpdf * igm_abs(refl_jacobian(cho));
        return (1 - fterm) * mpdf * igm_abs(refr_jacobian(eta, chi, cho));
    }
    IG_DEV bool sample(Tea& rnd, f3 out_dir, f3& in_dir, float& pdf_out, Col& color, float& s_eta, bool adjoint = false) const
    {
        const float cos_o = dot3(N, out_dir);
        if (igm_abs(cos_o) <= kCosEps)
            return false;
        const f3 m       = micro.sample(rnd, out_dir);
        const float mpdf = micro.pdf(out_dir, m);
        if (dot3(m, m) <= kFltEps || mpdf <= pdf_eps)
            return false;
        const f3 oH     = normalize3(m);
        const f3 H      = igm_signbit(dot3(oH, out_dir)) ? -oH : oH;
        const float cho = dot3(H, out_dir);
        if (igm_abs(cho) <= kCosEps)
            return false;
        float cos_t = 0, factor = 1;
        if (!fresnel(eta, cho, cos_t, factor)) {
            cos_t  = 0;
            factor = 1;
        }
        float sel_pdf;
        if (rnd.f32() > factor) {
            in_dir  = normalize3(H * (eta * cho - cos_t) - out_dir * eta); // vec3_refract (core/vector.art:126)
            sel_pdf = (1 - factor) * igm_abs(refr_jacobian(eta, dot3(H, in_dir), cho));
        } else {
            in_dir  = normalize3(H * (2 * dot3(H, out_dir)) - out_dir); // vec3_reflect (core/vector.art:123)
            sel_pdf = factor * igm_abs(refl_jacobian(cho));
        }
        const float cos_i = dot3(N, in_dir);
        pdf_out           = mpdf * sel_pdf;
        color             = eval(in_dir, out_dir) * safe_div((igm_signbit(cos_i * cos_o) && adjoint) ? igm_rcp(eta * eta) : 1.0f, pdf_out); // dielectric.art:181-185
        s_eta             = !igm_signbit(cos_i * cos_o) ? 1.0f : eta;
        return true;
    }
};

// fresnel_diffuse_factor (core/fresnel.art:42-63)
IG_DEV float fresnel_diffuse_factor(float eta)
{
    if (eta < 1)
        return -1.4399f * (eta * eta) + 0.7099f * eta + 0.6681f + 0.0636f / eta;
    const float ieta1 = igm_rcp(eta);
    const float ieta2 = ieta1 * ieta1;
    const float ieta3 = ieta2 * ieta1;
    const float ieta4 = ieta3 * ieta1;
    const float ieta5 = ieta4 * ieta1;
    return 0.919317f - 3.4793f * ieta1 + 6.75335f * ieta2 - 7.80989f * ieta3 + 4.98554f * ieta4 - 1.36881f * ieta5;
}

// make_plastic_bsdf (bsdf/plastic.art:2-41) = make_join_bsdf (bsdf/mix.art:4-65) of lobe 0, a lambertian with the inner
// scattering factor, and lobe 1, the conductor with eta = black / k = white (PlasticBSDF.cpp:36-39: rough,
// conductor.art:47-116, or the mirror of conductor.art:2-10 without roughness), mixed by the Fresnel term of out_dir
struct Plastic {
    m33 local;
    Col kd, ks;
    float eta, fdr;
    bool smooth;
    Ggx micro;

    IG_DEV Plastic(const ig_material& m, const m33& frame, Col diffuse)
    {
        local  = frame;
        kd     = diffuse;
        ks     = Col{ m.p[6], m.p[7], m.p[8] };
        eta    = m.p[3] / m.p[4];
        fdr    = fresnel_diffuse_factor(eta);
        smooth = (m.flags & IG_MAT_SMOOTH) != 0;
        micro  = Ggx{ frame, m.p[9], m.p[10] };
    }
    IG_DEV float diff_scattering(float cos_i) const
    {
        const float fi = fresnel_dielectric(eta, cos_i);
        return (1 - fi) * eta * eta / (1 - fdr);
    }
    IG_DEV float mix(f3 out_dir) const { return fresnel_dielectric(eta, abs_cos(out_dir, local.c2)); }

    IG_DEV Col lobe_eval(int lobe, f3 in_dir, f3 out_dir) const
    {
        const f3 N = local.c2;
        if (lobe == 0)
            return (kd * (pos_cos(in_dir, N) * kInvPi)) * diff_scattering(abs_cos(in_dir, N));
        if (smooth)
            return Col{ 0, 0, 0 };
        const float cos_o = abs_cos(out_dir, N);
        const float cos_i = abs_cos(in_dir, N);
        if (cos_o <= kFltEps || cos_i <= kFltEps)
            return Col{ 0, 0, 0 };
        const f3 H    = normalize3(in_dir + out_dir);
        const float D = micro.D(H);
        const float G = micro.G1(in_dir) * micro.G1(out_dir);
        const float f = conductor_factor(0, 1, abs_cos(out_dir, H));
        const Col F{ f, f, f }, IF{ 1 - f, 1 - f, 1 - f };
        const Col c{ 0.0f * IF.r + ks.r * F.r, 0.0f * IF.g + ks.g * F.g, 0.0f * IF.b + ks.b * F.b };
        return c * (D * G / (4 * cos_o));
    }
    IG_DEV float lobe_pdf(int lobe, f3 in_dir, f3 out_dir) const
    {
        if (lobe == 0)
            return pos_cos(in_dir, local.c2) / kPi;
        if (smooth)
            return 0;
        const f3 H      = normalize3(in_dir + out_dir);
        const float cho = abs_cos(out_dir, H);
        return micro.pdf(out_dir, H) * safe_div(1, 4 * cho);
    }
    IG_DEV bool lobe_sample(int lobe, Tea& rnd, f3 out_dir, f3& in_dir, float& pdf, Col& color, bool& sdelta) const
    {
        const f3 N = local.c2;
        sdelta     = false;
        if (lobe == 0) {
            const float u   = rnd.f32();
            const float v   = rnd.f32();
            const float c   = safe_sqrt(v);
            const float s   = safe_sqrt(1 - v);
            const float phi = 2 * kPi * u;
            in_dir          = mul33(local, f3{ s * igm_cos(phi), s * igm_sin(phi), c });
            pdf             = c / kPi;
            color           = kd * diff_scattering(abs_cos(in_dir, N));
            return true;
        }
        if (smooth) {
            in_dir = N * (2 * dot3(N, out_dir)) - out_dir;
            pdf    = 1;
            color  = ks;
            sdelta = true;
            return true;
        }
        if (abs_cos(out_dir, N) <= kFltEps)
            return false;
        const f3 m       = micro.sample(rnd, out_dir);
        const float mpdf = micro.pdf(out_dir, m);
        if (dot3(m, m) <= kFltEps)
            return false;
        const f3 oH = normalize3(m);
        const f3 H  = igm_signbit(dot3(oH, out_dir)) ? -oH : oH;
        in_dir      = H * (2 * dot3(H, out_dir)) - out_dir;
        if (abs_cos(in_dir, N) <= kFltEps)
            return false;
        const float jacob = igm_rcp(4 * abs_cos(out_dir, H));
        pdf               = mpdf * jacob;
        color             = lobe_eval(1, in_dir, out_dir) * safe_div(1, pdf);
        return true;
    }

    IG_DEV Col eval(f3 in_dir, f3 out_dir) const { return lerp_col(lobe_eval(0, in_dir, out_dir), lobe_eval(1, in_dir, out_dir), mix(out_dir)); }
    IG_DEV float pdf(f3 in_dir, f3 out_dir) const { return lerpf(lobe_pdf(0, in_dir, out_dir), lobe_pdf(1, in_dir, out_dir), mix(out_dir)); }
    // sample_mat (mix.art:28-38)
    IG_DEV bool sample_lobe(int first, float t, Tea& rnd, f3 out_dir, f3& in_dir, float& pdf_out, Col& color, bool& sdelta) const
    {
        if (!lobe_sample(first, rnd, out_dir, in_dir, pdf_out, color, sdelta))
            return false;
        const int second = 1 - first;
        const float p    = lerpf(pdf_out, lobe_pdf(second, in_dir, out_dir), t);
        const Col c      = lerp_col(color * pdf_out, lobe_eval(second, in_dir, out_dir), t);
        pdf_out          = p;
        color            = c * safe_div(1, p);
        return true;
    }
    IG_DEV bool sample(Tea& rnd, f3 out_dir, f3& in_dir, float& pdf_out, Col& color, bool& sdelta) const // mix.art:40-55
    {
        const float k = mix(out_dir);
        if (rnd.f32() < 1 - k) {
            if (sample_lobe(0, k, rnd, out_dir, in_dir, pdf_out, color, sdelta))
                return true;
            return sample_lobe(1, k, rnd, out_dir, in_dir, pdf_out, color, sdelta);
        }
        if (sample_lobe(1, 1 - k, rnd, out_dir, in_dir, pdf_out, color, sdelta))
            return true;
        return sample_lobe(0, 1 - k, rnd, out_dir, in_dir, pdf_out, color, sdelta);
    }
};

// ---- Radiance's BRTDfunc with constant arguments and the Roos glazing model on top of it (bsdf/rad.art:7-56): make_add_bsdf
// (bsdf/mix.art:68) = make_join_bsdf with colour addition, two levels deep over four leaf BSDFs
struct RadLeaf {
    int kind; // 0 make_lambertian_bsdf (diffuse.art:2-12), 1 make_lambertian_transmission_bsdf (:14-24), 2 make_mirror_bsdf (conductor.art:2-10),
              // 3 make_perfect_refraction_bsdf (dielectric.art:3-11)
    Col c;
    IG_DEV static float neg_cos(f3 a, f3 b)
    {
        const float cs = dot3(a, b);
        return cs <= 0 ? cs : 0.0f; // core/common.art:265-268
    }
    IG_DEV Col eval(const m33& local, f3 in_dir) const
    {
        if (kind == 0)
            return c * (pos_cos(in_dir, local.c2) * kInvPi);
        if (kind == 1)
            return c * (-neg_cos(in_dir, local.c2) * kInvPi);
        return Col{ 0, 0, 0 };
    }
    IG_DEV float pdf(const m33& local, f3 in_dir) const
    {
        if (kind == 0)
            return pos_cos(in_dir, local.c2) / kPi; // cosine_hemisphere_pdf
        if (kind == 1)
            return -neg_cos(in_dir, local.c2) / kPi;
        return 0;
    }
    IG_DEV void sample(const m33& local, Tea& rnd, f3 out_dir, f3& in_dir, float& pdf_out, Col& color, bool& sdelta) const
    {
        color = c;
        if (kind <= 1) {
            const float u   = rnd.f32();
            const float v   = rnd.f32();
            const float cs  = safe_sqrt(v); // sample_cosine_hemisphere (core/sampling.art:65-73)
            const float sn  = safe_sqrt(1 - v);
            const float phi = 2 * kPi * u;
            const f3 gdir   = mul33(local, f3{ sn * igm_cos(phi), sn * igm_sin(phi), cs });
            in_dir  = kind == 0 ? gdir : -gdir;
            pdf_out = cs / kPi;
            sdelta  = false;
        } else {
            in_dir  = kind == 2 ? local.c2 * (2 * dot3(local.c2, out_dir)) - out_dir : -out_dir;
            pdf_out = 1;
            sdelta  = true;
        }
    }
};
template <class A, class B>
struct RadAdd {
    A a;
    B b;
    float k;
    IG_DEV Col eval(const m33& l, f3 in_dir) const { return a.eval(l, in_dir) + b.eval(l, in_dir); }
    IG_DEV float pdf(const m33& l, f3 in_dir) const
    {
        if (k <= 0)
            return a.pdf(l, in_dir);
        if (k >= 1)
            return b.pdf(l, in_dir);
        return lerpf(a.pdf(l, in_dir), b.pdf(l, in_dir), k);
    }
    template <class F, class S>
    IG_DEV static void sample_mat(const F& first, const S& second, float t, const m33& l, Tea& rnd, f3 out_dir, f3& in_dir, float& pdf_out, Col& color, bool& sdelta)
    {
        first.sample(l, rnd, out_dir, in_dir, pdf_out, color, sdelta);
        const float p = lerpf(pdf_out, second.pdf(l, in_dir), t);
        const Col c   = color * pdf_out + second.eval(l, in_dir);
        pdf_out       = p;
        color         = c * safe_div(1, p);
    }
    IG_DEV void sample(const m33& l, Tea& rnd, f3 out_dir, f3& in_dir, float& pdf_out, Col& color, bool& sdelta) const
    {
        if (rnd.f32() < 1 - k)
            sample_mat(a, b, k, l, rnd, out_dir, in_dir, pdf_out, color, sdelta);
        else
            sample_mat(b, a, 1 - k, l, rnd, out_dir, in_dir, pdf_out, color, sdelta);
    }
};
using RadBrtd = RadAdd<RadAdd<RadLeaf, RadLeaf>, RadAdd<RadLeaf, RadLeaf>>;
IG_DEV float col_avg(Col c) { return (c.r + c.g + c.b) / 3; } // color_average (core/color.art:28)
IG_DEV RadBrtd make_rad_brtd(bool entering, Col refl_spec, Col trns_spec, Col refl_f, Col refl_b, Col trns_diff)
{
    const Col refl_diff = entering ? refl_f : refl_b;
    RadBrtd r;
    r.b.a = RadLeaf{ 2, refl_spec };
    r.b.b = RadLeaf{ 3, trns_spec };
    r.b.k = safe_div(col_avg(trns_spec), col_avg(refl_spec) + col_avg(trns_spec));
    r.a.a = RadLeaf{ 0, refl_diff };
    r.a.b = RadLeaf{ 1, trns_diff };
    r.a.k = safe_div(col_avg(trns_diff), col_avg(refl_diff) + col_avg(trns_diff));
    const float sum_spec = col_avg(refl_spec + trns_spec);
    const float sum_diff = col_avg(refl_diff + trns_diff);
    r.k = safe_div(sum_spec, sum_diff + sum_spec);
    return r;
}
// make_rad_roos_bsdf (rad.art:36-56): specular reflection (x) and transmission (y) from the cosine between ray and shading normal
IG_DEV f2 rad_roos_factors(const ig_material& m, float cosN)
{
    const float trns_w = m.p[0], trns_p = m.p[1], trns_q = m.p[2], refl_w = m.p[3], refl_p = m.p[4], refl_q = m.p[5];
    const float a = 8, beta = 2;
    const float bq    = 0.25f / trns_q;
    const float cq    = 1 - a - bq;
    const float alpha = 5.2f + 0.7f * trns_q;
    const float gt    = (5.26f + 0.06f * trns_p) + (0.73f + 0.04f * trns_p) * trns_q;
    const float gr    = (5.26f + 0.06f * refl_p) + (0.73f + 0.04f * refl_p) * refl_q;
    const float z     = igm_acos(igm_abs(clampf(cosN, -1, 1))) * 0.636619772368f;
    const float tau   = trns_w * (1 - a * igm_pow(z, alpha) - bq * igm_pow(z, beta) - cq * igm_pow(z, gt));
    const float rf    = refl_w + (1 - refl_w) * igm_pow(z, gr);
    return f2{ rf, tau };
}

// FULL = false leaves the principled BSDF out of the kernel (scenes without one run the lean variant)
// TOP = false: the context of a BSDF inside a blend (same surface, no further nesting)
struct BlendInner {
    const DevScene* sc; // the inner contexts are rebuilt where they are used, so a blend costs the other materials no registers
};
struct NoBlendInner {
};

// constant, checkerboard or bitmap colour of a material (diffuse reflectance, principled base colour, ...)
// EXPR: the kernel instantiation for scenes with shading expressions (as a run-time branch of the ordinary full kernel the
// interpreter call cost it its occupancy: 226 VGPRs / 1584 B of scratch against 168 / 472)
template <bool EXPR>
IG_DEV Col material_color(const DevScene& sc, const ig_material& m, const Surf& s, f3 view)
{
    if constexpr (EXPR) {
        if (m.flags & IG_MAT_EXPR_COLOR) { // vec4_to_color / vec3_to_color of the expression (Transpiler.cpp:1297-1302), a number as grey
            const f3 c = eval_expr(sc, m.tex_refl, s, view);
            return Col{ c.x, c.y, c.z };
        }
    }
    if (m.flags & IG_MAT_IMAGE)
        return image_lookup(sc, sc.textures[m.tex_refl], s.tex);
    if (m.flags & IG_MAT_CHECKER) {
        const bool px = ((int)wrapf(s.tex.x * m.q[6], 0, 2) % 2) == 0;
        const bool py = ((int)wrapf(s.tex.y * m.q[7], 0, 2) % 2) == 0;
        return (px ^ py) ? Col{ m.q[0], m.q[1], m.q[2] } : Col{ m.q[3], m.q[4], m.q[5] };
    }
    return Col{ m.p[0], m.p[1], m.p[2] };
}

// RARE: the BSDFs only the instantiation for scenes with expressions / Radiance materials carries (k_shade<true, *, true>): in the
// ordinary full kernel the BRTDfunc / Roos code cost diamond_scene_principled 3 % of its shading time
// TYPES: the BSDF models this instantiation carries (bit IG_BSDF_*; the kernels split by material class, shade_kernel.h): a model
// outside it is not compiled in, the caller guarantees no such material arrives. The BSDFs inside a blend are built with all of them.
template <bool FULL, bool TOP = true, bool RARE = false, uint32_t TYPES = ~0u>
struct BsdfCtx {
    template <int T>
    IG_DEV bool is() const { return ((TYPES >> T) & 1u) != 0u && mat->bsdf_type == T; }
    const ig_material* mat;
    Surf surf; // the surface the BSDF is built on (bump-mapped materials: re-oriented local frame)
    Col kd;    // diffuse reflectance (constant or checkerboard)
    std::conditional_t<FULL && TOP, BlendInner, NoBlendInner> blend;
    // make_doublesided_bsdf (bsdf/common.art:28-46) on a surface hit from behind: the BSDF is built as if entered and every
    // direction is negated on the way in, the sampled one on the way out (full variant, outermost wrapper only)
    bool ds_flip = false;

    template <bool EXPR = false>
    IG_DEV BsdfCtx(const DevScene& sc, const ig_material& m, const Surf& s, f3 ray_dir, std::bool_constant<EXPR> = {})
        : mat(&m)
        , surf(s)
    {
        if (m.flags & (IG_MAT_BUMP | IG_MAT_NORMALMAP | (EXPR ? IG_MAT_EXPR_NORMAL : 0)))
            surf.local = bumped_frame<EXPR>(sc, m, s, ray_dir);
        kd = material_color<EXPR>(sc, m, surf, -ray_dir); // an expression sees the surface the BSDF is built on (bsdf_inner(ctx.{surf = surf2}))
        if constexpr (FULL && TOP) {
            // a blend's weight travels in kd.r (= p[0] for a constant one: material_color of a blend is its p[0..2])
            if (EXPR && (m.flags & IG_MAT_EXPR_WEIGHT))
                kd.r = eval_expr(sc, m.tex_id, surf, -ray_dir).x;
        }
        if constexpr (FULL && RARE) {
            if (is<IG_BSDF_RAD_ROOS>()) { // cosN = -dot(ctx.ray.dir, ctx.surf.local.col(2)) (RadRoosBSDF.cpp:28); kd carries (rf, tau)
                const f2 ft = rad_roos_factors(m, -dot3(ray_dir, surf.local.c2));
                kd          = Col{ ft.x, ft.y, 0 };
            }
        }
        if constexpr (FULL && TOP) {
            blend.sc = &sc;
            ds_flip  = (m.flags & IG_MAT_DOUBLESIDED) && !s.entering;
            if (ds_flip)
                surf.entering = true;
        }
    }
    // a BSDF inside a blend: it sees the blend's surface (make_mix_bsdf, bsdf/mix.art:4-68)
    IG_DEV BsdfCtx(const ig_material& m, const Surf& s, Col color)
        : mat(&m)
        , surf(s)
        , kd(color)
    {
    }
    IG_DEV BsdfCtx<FULL, false> inner(int i) const
    {
        const ig_material& m = blend.sc->materials[mat->pad[i]];
        return BsdfCtx<FULL, false>(m, surf, material_color<false>(*blend.sc, m, surf, f3{ 0, 0, 0 })); // the loader keeps expressions out of blends
    }

    IG_DEV bool all_delta() const
    {
        if constexpr (FULL && TOP) {
            if (is<IG_BSDF_BLEND>()) // mat1.is_all_delta & mat2.is_all_delta (mix.art:63)
                return inner(0).all_delta() && inner(1).all_delta();
        }
        return is<IG_BSDF_DIELECTRIC>() || (FULL && is<IG_BSDF_TRANSPARENT>()) || (is<IG_BSDF_CONDUCTOR>() && (mat->flags & IG_MAT_SMOOTH));
    }
    IG_DEV Ggx ggx() const { return Ggx{ surf.local, mat->p[9], mat->p[10] }; }
    // the weight of a blend: p[0], or (instantiation with expressions only) the value the constructor left in kd.r
    IG_DEV float blend_weight() const { return RARE ? kd.r : mat->p[0]; }
    IG_DEV bool is_rad() const { return FULL && RARE && (is<IG_BSDF_RAD_BRTD>() || is<IG_BSDF_RAD_ROOS>()); }
    // make_rad_brtdfunc_bsdf / make_rad_roos_bsdf (bsdf/rad.art) from the material record
    IG_DEV RadBrtd rad() const
    {
        const Col td{ mat->q[0], mat->q[1], mat->q[2] };
        const Col rf{ mat->p[6], mat->p[7], mat->p[8] }, rb{ mat->p[9], mat->p[10], mat->p[11] };
        if (is<IG_BSDF_RAD_ROOS>()) {
            const Col black{ 0, 0, 0 };
            return make_rad_brtd(surf.entering, Col{ kd.r, kd.r, kd.r }, Col{ kd.g, kd.g, kd.g }, rf + black, rb + black, td);
        }
        return make_rad_brtd(surf.entering, Col{ mat->p[0], mat->p[1], mat->p[2] }, Col{ mat->p[3], mat->p[4], mat->p[5] }, rf, rb, td);
    }

    // make_orennayar_bsdf.eval (bsdf/diffuse.art:28-39): p[3] alpha
    IG_DEV Col orennayar_eval(f3 in_dir, f3 out_dir) const
    {
        const f3 N     = surf.local.c2;
        const float a2 = mat->p[3] * mat->p[3];
        const float p1 = pos_cos(in_dir, N);
        const float p2 = pos_cos(out_dir, N);
        const float sv = -p1 * p2 + pos_cos(out_dir, in_dir);
        const float t  = sv <= kFltEps ? 1.0f : igm_max(kFltEps, igm_max(p1, p2));
        const float A  = 1 - 0.5f * a2 / (a2 + 0.33f);
        const float B  = 0.45f * a2 / (a2 + 0.09f);
        const float C  = 0.17f * a2 / (a2 + 0.13f);
        return (kd * ((A + (B * sv / t)) / kPi) + kd * (kd * (C / kPi))) * p1;
    }
    // fastpow = fastpow2(p * fastlog2(x)) (core/common.art:71-90): float and integer arithmetic only
    IG_DEV static float fastpow(float x, float p)
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
    IG_DEV Col phong_eval(f3 in_dir, f3 out_dir) const
    {
        const f3 N     = surf.local.c2;
        const float ns = mat->p[3];
        const f3 refl  = N * (2 * dot3(N, out_dir)) - out_dir; // vec3_reflect (core/vector.art:123)
        return Col{ mat->p[0], mat->p[1], mat->p[2] } * (pos_cos(in_dir, N) * fastpow(pos_cos(in_dir, refl), ns) * (ns + 2) / (2 * kPi));
    }
    IG_DEV float phong_pdf(f3 in_dir, f3 out_dir) const
    {
        const f3 N     = surf.local.c2;
        const float ns = mat->p[3];
        const f3 refl  = N * (2 * dot3(N, out_dir)) - out_dir;
        return fastpow(pos_cos(in_dir, refl), ns) * (ns + 1) * (1 / (2 * kPi)); // cosine_power_hemisphere_pdf (core/sampling.art:79-81)
    }
    IG_DEV void phong_sample(Tea& rnd, f3 out_dir, f3& in_dir, float& pdf_out, Col& color) const
    {
        const f3 N      = surf.local.c2;
        const float ns  = mat->p[3];
        const f3 refl   = N * (2 * dot3(N, out_dir)) - out_dir;
        const float u   = rnd.f32();
        const float v   = rnd.f32();
        const float c   = igm_min(fastpow(v, igm_rcp(ns + 1)), 1.0f); // sample_cosine_power_hemisphere (core/sampling.art:84-96)
        const float sn  = igm_sqrt(1 - c * c);
        const float phi = 2 * kPi * u;
        pdf_out         = (c != 0 ? v / c : 0.0f) * (ns + 1) * (1 / (2 * kPi));
        in_dir          = mul33(orthonormal_basis(refl), f3{ sn * igm_cos(phi), sn * igm_sin(phi), c });
        color           = Col{ mat->p[0], mat->p[1], mat->p[2] } * (pos_cos(in_dir, N) * (ns + 2) / (ns + 1));
    }

    // lambertian (bsdf/diffuse.art:3), rough conductor (bsdf/conductor.art:70-84)
    IG_DEV Principled principled() const { return Principled(*mat, surf.local, surf.entering, kd); }

    // Bsdf::albedo of each model (the "Albedo" AOV of technique/internal/infobuffer.art:13-21)
    IG_DEV Col albedo(f3 out_dir) const
    {
        if constexpr (FULL && TOP)
            out_dir = ds_flip ? -out_dir : out_dir;
        const f3 N = surf.local.c2;
        if (is_rad()) { // make_join_bsdf.albedo (mix.art:56-61): colour lerps of the leaves' colours
            const RadBrtd r = rad();
            return lerp_col(lerp_col(r.a.a.c, r.a.b.c, r.a.k), lerp_col(r.b.a.c, r.b.b.c, r.b.k), r.k);
        }
        switch (mat->bsdf_type) {
        case IG_BSDF_PHONG:       // ks (bsdf/phong.art:20)
        case IG_BSDF_TRANSPARENT: // make_perfect_refraction_bsdf: kt (bsdf/dielectric.art:9)
            return Col{ mat->p[0], mat->p[1], mat->p[2] };
        case IG_BSDF_ROUGH_DIELECTRIC:
        case IG_BSDF_DIELECTRIC: // make_pure_dielectric_bsdf / make_rough_dielectric_bsdf (bsdf/dielectric.art:35,190); thin: ks (:60)
            if (is<IG_BSDF_DIELECTRIC>() && (mat->flags & IG_MAT_THIN))
                return Col{ mat->p[2], mat->p[3], mat->p[4] };
            return lerp_col(Col{ mat->p[2], mat->p[3], mat->p[4] }, Col{ mat->p[5], mat->p[6], mat->p[7] }, 0.5f);
        case IG_BSDF_CONDUCTOR: { // compute_albedo (bsdf/conductor.art:50-56), kd = black
            const float c = abs_cos(out_dir, N);
            const Col F   = Col{ conductor_factor(mat->p[0], mat->p[3], c), conductor_factor(mat->p[1], mat->p[4], c), conductor_factor(mat->p[2], mat->p[5], c) };
            const Col IF  = Col{ 1 - F.r, 1 - F.g, 1 - F.b };
            return Col{ 0.0f * IF.r + mat->p[6] * F.r, 0.0f * IF.g + mat->p[7] * F.g, 0.0f * IF.b + mat->p[8] * F.b };
        }
        case IG_BSDF_PLASTIC: { // make_join_bsdf.albedo (bsdf/mix.art:56-61) over lambertian and mirror / conductor albedo
            const Plastic pl(*mat, surf.local, kd);
            const Col ks{ mat->p[6], mat->p[7], mat->p[8] };
            Col coat = ks; // make_mirror_bsdf (bsdf/conductor.art:8)
            if (!(mat->flags & IG_MAT_SMOOTH)) {
                const float f = conductor_factor(0, 1, abs_cos(out_dir, N));
                coat          = Col{ 0.0f * (1 - f) + ks.r * f, 0.0f * (1 - f) + ks.g * f, 0.0f * (1 - f) + ks.b * f };
            }
            return lerp_col(kd, coat, pl.mix(out_dir));
        }
        case IG_BSDF_BLEND: // mix.art:56-61
            if constexpr (FULL && TOP)
                return lerp_col(inner(0).albedo(out_dir), inner(1).albedo(out_dir), blend_weight());
            return kd;
        default: // lambertian kd (bsdf/diffuse.art:10), principled base colour (bsdf/principled.art:478)
            return kd;
        }
    }

    IG_DEV Col eval(f3 in_dir, f3 out_dir) const
    {
        const f3 N = surf.local.c2;
        if constexpr (FULL && TOP) {
            in_dir  = ds_flip ? -in_dir : in_dir;
            out_dir = ds_flip ? -out_dir : out_dir;
            if (is<IG_BSDF_BLEND>()) // eval_f = color_lerp (mix.art:5-8,68)
                return lerp_col(inner(0).eval(in_dir, out_dir), inner(1).eval(in_dir, out_dir), blend_weight());
        }
        if constexpr (FULL) {
            if (is_rad())
                return rad().eval(surf.local, in_dir);
            if (is<IG_BSDF_PHONG>())
                return phong_eval(in_dir, out_dir);
            if (is<IG_BSDF_PRINCIPLED>())
                return principled().eval(in_dir, out_dir);
            if (is<IG_BSDF_PLASTIC>())
                return Plastic(*mat, surf.local, kd).eval(in_dir, out_dir);
            if (is<IG_BSDF_ROUGH_DIELECTRIC>())
                return RoughDielectric(*mat, surf.local, surf.entering).eval(in_dir, out_dir);
        }
        if (is<IG_BSDF_DIFFUSE>()) {
            if (FULL && mat->p[3] > kFltEps) // make_diffuse_bsdf (bsdf/diffuse.art:52-58): a roughness selects Oren-Nayar
                return orennayar_eval(in_dir, out_dir);
            return kd * (pos_cos(in_dir, N) * kInvPi);
        }
        if (is<IG_BSDF_CONDUCTOR>()) {
            const float cos_o = abs_cos(out_dir, N);
            const float cos_i = abs_cos(in_dir, N);
            if (cos_o <= kFltEps || cos_i <= kFltEps)
                return Col{ 0, 0, 0 };
            const Ggx g   = ggx();
            const f3 H    = normalize3(in_dir + out_dir);
            const float D = g.D(H);
            const float G = g.G1(in_dir) * g.G1(out_dir);
            const float c = abs_cos(out_dir, H);
            const Col F   = Col{ conductor_factor(mat->p[0], mat->p[3], c), conductor_factor(mat->p[1], mat->p[4], c), conductor_factor(mat->p[2], mat->p[5], c) };
            const Col IF  = Col{ 1 - F.r, 1 - F.g, 1 - F.b };
            const Col col = Col{ 0.0f * IF.r + mat->p[6] * F.r, 0.0f * IF.g + mat->p[7] * F.g, 0.0f * IF.b + mat->p[8] * F.b };
            return col * (D * G / (4 * cos_o));
        }
        return Col{ 0, 0, 0 };
    }
    IG_DEV float pdf(f3 in_dir, f3 out_dir) const
    {
        if constexpr (FULL && TOP) {
            in_dir  = ds_flip ? -in_dir : in_dir;
            out_dir = ds_flip ? -out_dir : out_dir;
            if (is<IG_BSDF_BLEND>()) { // mix.art:10-22 with a constant weight
                const float k = blend_weight();
                if (k <= 0)
                    return inner(0).pdf(in_dir, out_dir);
                if (k >= 1)
                    return inner(1).pdf(in_dir, out_dir);
                return lerpf(inner(0).pdf(in_dir, out_dir), inner(1).pdf(in_dir, out_dir), k);
            }
        }
        if constexpr (FULL) {
            if (is_rad())
                return rad().pdf(surf.local, in_dir);
            if (is<IG_BSDF_PHONG>())
                return phong_pdf(in_dir, out_dir);
            if (is<IG_BSDF_PRINCIPLED>())
                return principled().pdf(in_dir, out_dir);
            if (is<IG_BSDF_PLASTIC>())
                return Plastic(*mat, surf.local, kd).pdf(in_dir, out_dir);
            if (is<IG_BSDF_ROUGH_DIELECTRIC>())
                return RoughDielectric(*mat, surf.local, surf.entering).pdf(in_dir, out_dir);
        }
        if (is<IG_BSDF_DIFFUSE>())
            return pos_cos(in_dir, surf.local.c2) / kPi;
        if (is<IG_BSDF_CONDUCTOR>()) {
            const f3 H      = normalize3(in_dir + out_dir);
            const float cho = abs_cos(out_dir, H);
            return ggx().pdf(out_dir, H) * safe_div(1, 4 * cho);
        }
        return 0;
    }
    // returns false when the sample is rejected
    // adjoint: bsdf.sample(rnd, out_dir, true) of light paths (the light tracer)
    IG_DEV bool sample(Tea& rnd, f3 out_dir, f3& in_dir, float& pdf_out, Col& color, float& s_eta, bool& sdelta, bool adjoint = false) const
    {
        if constexpr (FULL && TOP) {
            const bool ok = sample_inner(rnd, ds_flip ? -out_dir : out_dir, in_dir, pdf_out, color, s_eta, sdelta, adjoint);
            in_dir        = ds_flip ? -in_dir : in_dir;
            return ok;
        } else {
            return sample_inner(rnd, out_dir, in_dir, pdf_out, color, s_eta, sdelta, adjoint);
        }
    }
    IG_DEV bool sample_inner(Tea& rnd, f3 out_dir, f3& in_dir, float& pdf_out, Col& color, float& s_eta, bool& sdelta, bool adjoint = false) const
    {
        const f3 N = surf.local.c2;
        if constexpr (FULL && TOP) {
            if (is<IG_BSDF_BLEND>()) {
                // make_join_bsdf.sample (mix.art:27-55); sample_mat(first, second, t)
                const float k    = blend_weight();
                const bool pick1 = rnd.f32() < 1 - k;
                const float t    = pick1 ? k : 1 - k;
                for (int attempt = 0; attempt < 2; ++attempt) {
                    const int first = (pick1 ? 0 : 1) ^ attempt;
                    const BsdfCtx<FULL, false> a = inner(first), b = inner(first ^ 1);
                    if (!a.sample(rnd, out_dir, in_dir, pdf_out, color, s_eta, sdelta, adjoint))
                        continue;
                    const float p = lerpf(pdf_out, b.pdf(in_dir, out_dir), t);
                    const Col c   = lerp_col(color * pdf_out, b.eval(in_dir, out_dir), t);
                    pdf_out       = p;
                    color         = c * safe_div(1, p);
                    return true;
                }
                return false;
            }
        }
        if constexpr (FULL) {
            if (is_rad()) {
                rad().sample(surf.local, rnd, out_dir, in_dir, pdf_out, color, sdelta);
                s_eta = 1;
                return true;
            }
            if (is<IG_BSDF_PHONG>()) {
                phong_sample(rnd, out_dir, in_dir, pdf_out, color);
                s_eta  = 1;
                sdelta = false;
                return true;
            }
            if (is<IG_BSDF_TRANSPARENT>()) { // make_perfect_refraction_bsdf.sample (bsdf/dielectric.art:6-8)
                in_dir  = -out_dir;
                pdf_out = 1;
                color   = Col{ mat->p[0], mat->p[1], mat->p[2] };
                s_eta   = 1;
                sdelta  = true;
                return true;
            }
            if (is<IG_BSDF_PRINCIPLED>()) {
                sdelta = false;
                return principled().sample(rnd, out_dir, in_dir, pdf_out, color, s_eta, adjoint);
            }
            if (is<IG_BSDF_ROUGH_DIELECTRIC>()) {
                sdelta = false;
                return RoughDielectric(*mat, surf.local, surf.entering).sample(rnd, out_dir, in_dir, pdf_out, color, s_eta, adjoint);
            }
            if (is<IG_BSDF_PLASTIC>()) {
                s_eta = 1;
                return Plastic(*mat, surf.local, kd).sample(rnd, out_dir, in_dir, pdf_out, color, sdelta);
            }
        }
        if (is<IG_BSDF_DIFFUSE>()) {
            // make_lambertian_bsdf.sample (bsdf/diffuse.art:5-9), sample_cosine_hemisphere (core/sampling.art:62-70)
            const float u   = rnd.f32();
            const float v   = rnd.f32();
            const float c   = safe_sqrt(v);
            const float s   = safe_sqrt(1 - v);
            const float phi = 2 * kPi * u;
            in_dir          = mul33(surf.local, f3{ s * igm_cos(phi), s * igm_sin(phi), c });
            pdf_out         = c / kPi;
            color           = kd;
            if (FULL && mat->p[3] > kFltEps)
                color = orennayar_eval(in_dir, out_dir) * (igm_rcp(pdf_out)); // bsdf/diffuse.art:46
            s_eta           = 1;
            sdelta          = false;
            return true;
        }
        if (is<IG_BSDF_CONDUCTOR>() && (mat->flags & IG_MAT_SMOOTH)) {
            // delta branch of make_rough_base_conductor_bsdf (bsdf/conductor.art:56-68): compute_albedo(out_dir), kd = black
            const float cos_o = abs_cos(out_dir, N);
            const Col F  = Col{ conductor_factor(mat->p[0], mat->p[3], cos_o), conductor_factor(mat->p[1], mat->p[4], cos_o), conductor_factor(mat->p[2], mat->p[5], cos_o) };
            const Col IF = Col{ 1 - F.r, 1 - F.g, 1 - F.b };
            in_dir  = N * (2 * dot3(N, out_dir)) - out_dir; // vec3_reflect
            pdf_out = 1;
            color   = Col{ 0.0f * IF.r + mat->p[6] * F.r, 0.0f * IF.g + mat->p[7] * F.g, 0.0f * IF.b + mat->p[8] * F.b };
            s_eta   = 1;
            sdelta  = true;
            return true;
        }
        if (is<IG_BSDF_CONDUCTOR>()) {
            // make_rough_base_conductor_bsdf.sample (bsdf/conductor.art:93-114)
            const float cos_o = abs_cos(out_dir, N);
            if (cos_o <= kFltEps)
                return false;
            const Ggx g      = ggx();
            const f3 m       = g.sample(rnd, out_dir);
            const float mpdf = g.pdf(out_dir, m);
            if (dot3(m, m) <= kFltEps)
                return false;
            const f3 oH = normalize3(m);
            const f3 H  = igm_signbit(dot3(oH, out_dir)) ? -oH : oH;
            in_dir      = H * (2 * dot3(H, out_dir)) - out_dir; // vec3_reflect
            if (abs_cos(in_dir, N) <= kFltEps)
                return false;
            const float cho = abs_cos(out_dir, H);
            pdf_out         = mpdf * (igm_rcp(4 * cho));
            color           = eval(in_dir, out_dir) * safe_div(1, pdf_out);
            s_eta           = 1;
            sdelta          = false;
            return true;
        }
        if constexpr (FULL) {
            if (mat->flags & IG_MAT_THIN) {
                // make_thin_dielectric_bsdf (bsdf/dielectric.art:40-61): always from outside to inside
                const float kk    = mat->p[0] / mat->p[1];
                const float fterm = fresnel_dielectric(kk, abs_cos(out_dir, N));
                const float F     = fterm + (1 - fterm) * fterm / (fterm + 1);
                if (rnd.f32() > F) {
                    in_dir = -out_dir;
                    color  = Col{ mat->p[5], mat->p[6], mat->p[7] };
                } else {
                    in_dir = normalize3(N * (2 * dot3(N, out_dir)) - out_dir);
                    color  = Col{ mat->p[2], mat->p[3], mat->p[4] };
                }
                pdf_out = 1;
                s_eta   = 1;
                sdelta  = true;
                return true;
            }
        }
        // make_pure_dielectric_bsdf.sample (bsdf/dielectric.art:18-34); n1 = ext_ior, n2 = int_ior
        const float n1 = mat->p[0], n2 = mat->p[1];
        const float k     = surf.entering ? n1 / n2 : n2 / n1;
        const float cos_o = dot3(out_dir, N);
        float cos_t = 0, F = 1;
        if (!fresnel(k, cos_o, cos_t, F)) {
            cos_t = 0;
            F     = 1;
        }
        if (rnd.f32() > F) {
            in_dir = N * (k * cos_o - cos_t) - out_dir * k; // vec3_refract (core/vector.art:126)
            color  = Col{ mat->p[5], mat->p[6], mat->p[7] } * (adjoint ? k * k : 1.0f); // adjoint_term (dielectric.art:28)
            s_eta  = k;
        } else {
            in_dir = N * (2 * dot3(N, out_dir)) - out_dir; // vec3_reflect (core/vector.art:123)
            color  = Col{ mat->p[2], mat->p[3], mat->p[4] };
            s_eta  = 1;
        }
        pdf_out = 1;
        sdelta  = true;
        return true;
    }
};

// ---------------------------------------------------------------- environment map sampling

// make_cdf_1d over a buffer without the leading 0 (core/cdf.art:43-73), interval::binary_search (core/interval.art:7-23)
struct Cdf1D {
    const float* data;
    int size;

    IG_DEV float get(int i) const { return i == 0 ? 0.0f : data[i - 1]; }
    IG_DEV float pdf_discrete(int x) const { return get(x + 1) - get(x); }
    IG_DEV int sample_discrete(float u, float& pdf) const
    {
        int first = 0, len = size + 1;
        while (len > 0) {
            const int half   = len / 2;
            const int middle = first + half;
            if (get(middle) <= u) {
                first = middle + 1;
                len -= half + 1;
            } else {
                len = half;
            }
        }
        const int found = min(max(first - 1, 0), size); // clamp(first - 1, 0, (size + 1) - 1)
        const int off   = min(found, size - 1);
        pdf             = pdf_discrete(off);
        return off;
    }
    IG_DEV float pdf_continuous(float x, int& off) const
    {
        off = min(max((int)(x * (float)size), 0), size - 1);
        return pdf_discrete(off) * (float)size;
    }
    IG_DEV float sample_continuous(float u, int& off, float& pdf) const
    {
        float dpdf;
        off             = sample_discrete(u, dpdf);
        const float rem = safe_div(u - get(off), dpdf);
        pdf             = dpdf * (float)size;
        return clampf(((float)off + rem) / (float)size, 0, 1);
    }
};

// make_cdf_2d_from_buffer (core/cdf.art:108-159): marginal first, then one conditional per row
struct Cdf2D {
    const float* data;
    int size_x, size_y;

    IG_DEV Cdf1D marginal() const { return Cdf1D{ data, size_y }; }
    IG_DEV Cdf1D conditional(int row) const { return Cdf1D{ data + size_y + (size_t)row * size_x, size_x }; }
    IG_DEV f2 sample_continuous(float ux, float uy, float& pdf) const
    {
        int oy, ox;
        float p1, p2;
        const float py = marginal().sample_continuous(uy, oy, p1);
        const float px = conditional(oy).sample_continuous(ux, ox, p2);
        pdf            = p1 * p2;
        return f2{ px, py };
    }
    IG_DEV float pdf_continuous(f2 pos) const
    {
        int oy, ox;
        const float p1 = marginal().pdf_continuous(pos.y, oy);
        const float p2 = conditional(oy).pdf_continuous(pos.x, ox);
        return p1 * p2;
    }
};

// light/env.art:11-21 (switch_env_up, map_env_uv), core/warp.art:44-48 (spherical_from_dir)
IG_DEV f3 switch_env_up(f3 v) { return f3{ v.x, v.z, v.y }; }
IG_DEV f2 map_env_uv(f3 dir)
{
    const float theta = igm_acos(dir.z);
    float phi         = igm_atan2(dir.y, dir.x);
    if (phi < 0)
        phi = phi + 2 * kPi;
    const float v = theta / kPi;
    const float u = phi / (2 * kPi);
    const float r = u + 0.25f;
    return f2{ r - igm_floor(r), 1 - v };
}

// make_environment_light_textured (light/env.art:109-157)
IG_DEV f3 square_to_sphere(float px, float py);

struct TexturedEnv {
    const DevScene& sc;
    Col scale;
    m33 transform;
    const ig_texture* tex;
    Cdf2D cdf;

    IG_DEV TexturedEnv(const DevScene& scene, const ig_light& L)
        : sc(scene)
    {
        scale        = Col{ L.d[0], L.d[1], L.d[2] };
        transform.c0 = f3{ L.d[3], L.d[4], L.d[5] };
        transform.c1 = f3{ L.d[6], L.d[7], L.d[8] };
        transform.c2 = f3{ L.d[9], L.d[10], L.d[11] };
        tex          = &sc.textures[igm_bits(L.d[12])];
        cdf          = Cdf2D{ sc.cdf_data + igm_bits(L.d[13]), (int)igm_bits(L.d[14]), (int)igm_bits(L.d[15]) };
    }
    // sample_dir (env.art:112-123): the intensity is the bare texture value, without `scale`
    IG_DEV void sample_dir(Tea& rnd, f3& dir, Col& intensity, float& pdf_dir) const
    {
        const float u0 = rnd.f32();
        const float u1 = rnd.f32();
        if (cdf.size_x == 0) {
            // "cdf": "none" -> make_environment_light (env.art:161-164) over make_environment_light_function_spherical
            // (:79-93): a uniform direction, the function value includes `scale`
            dir       = square_to_sphere(u0, u1);
            intensity = emission(dir);
            pdf_dir   = 1 / (4 * kPi);
            return;
        }
        float pdf;
        const f2 pos      = cdf.sample_continuous(u0, u1, pdf);
        intensity         = image_lookup(sc, *tex, pos);
        const float theta = (1 - pos.y) * kPi;
        const float phi   = (pos.x - 0.25f) * 2 * kPi;
        const float st = igm_sin(theta), ct = igm_cos(theta);
        const f3 d           = f3{ st * igm_cos(phi), st * igm_sin(phi), ct }; // dir_from_spherical (core/warp.art:50-57)
        const float sinTheta = safe_sqrt(1 - d.z * d.z);
        pdf_dir              = safe_div(pdf, sinTheta * kPi * kPi * 2);
        const f3 e           = switch_env_up(d);
        dir                  = f3{ dot3(transform.c0, e), dot3(transform.c1, e), dot3(transform.c2, e) }; // mat3x3_left_mul
    }
    IG_DEV f3 local_dir(f3 ray_dir) const { return switch_env_up(mul33(transform, ray_dir)); }
    IG_DEV float pdf(f3 ray_dir) const
    {
        if (cdf.size_x == 0)
            return 1 / (4 * kPi); // equal_area_sphere_pdf (env.art:101)
        const f3 ldir        = local_dir(ray_dir);
        const float sinTheta = safe_sqrt(1 - ldir.z * ldir.z);
        return safe_div(cdf.pdf_continuous(map_env_uv(ldir)), sinTheta * kPi * kPi * 2);
    }
    IG_DEV Col emission(f3 ray_dir) const { return scale * image_lookup(sc, *tex, map_env_uv(local_dir(ray_dir))); }
};

// make_shape_area_emitter (light/area.art:58-103) under make_area_light (area.art:10-43) with a constant colour: a uniformly
// chosen triangle of the emissive entity's mesh, a uniform point on it
struct MeshEmitter {
    m34 global;
    const float* verts;
    const float* inds;
    int num_tris;
    Col radiance;

    IG_DEV MeshEmitter(const DevScene& sc, const ig_light& L)
    {
        const int ent   = L.entity_id;
        const float4* e = reinterpret_cast<const float4*>(sc.entities + (size_t)ent * IG_ENTITY_FLOATS);
        const float4 r3 = e[3], r4 = e[4], r5 = e[5];
        global.c0 = f3{ r3.x, r3.y, r3.z }, global.c1 = f3{ r3.w, r4.x, r4.y }, global.c2 = f3{ r4.z, r4.w, r5.x }, global.c3 = f3{ r5.y, r5.z, r5.w };
        const uint4 ext = sc.entity_ext[ent];
        verts           = reinterpret_cast<const float*>(sc.shape_data + ext.x);
        inds            = reinterpret_cast<const float*>(sc.shape_data + ext.z);
        num_tris        = (int)*reinterpret_cast<const uint32_t*>(sc.shape_data + ext.x - 48); // shape header: faces first
        radiance        = Col{ L.d[0], L.d[1], L.d[2] };
    }
    // what the light needs of shape.surface_element_for_point (shapes/trimesh.art:41-68)
    IG_DEV void surface(int f, float u, float v, f3& point, f3& face_normal, float& area) const
    {
        const int4 tri = *reinterpret_cast<const int4*>(inds + f * 4);
        const f3 v0    = xform_point(global, ld3v(verts + tri.x * 4));
        const f3 v1    = xform_point(global, ld3v(verts + tri.y * 4));
        const f3 v2    = xform_point(global, ld3v(verts + tri.z * 4));
        const f3 n     = stable_normal(v2 - v0, v0 - v1, v1 - v2);
        const float nn = len3(n);
        face_normal    = n * (igm_rcp(nn));
        area           = nn / 2;
        point          = f3{ lerp2(v0.x, v1.x, v2.x, u, v), lerp2(v0.y, v1.y, v2.y, u, v), lerp2(v0.z, v1.z, v2.z, u, v) };
    }
    IG_DEV void address(float uvx, float uvy, int& f, float& u, float& v) const
    {
        const float ux = uvx * (float)num_tris;
        f              = min((int)ux, num_tris - 1);
        const float a = ux - (float)f, b = uvy;
        if (a + b > 1) // sample_triangle (core/sampling.art:34-36)
            u = 1 - a, v = 1 - b;
        else
            u = a, v = b;
    }
    // pdf_direct(surf.prim_coords, .) (area.art:39,74-82): the hit's barycentrics go through the same addressing as a sample's uv
    IG_DEV float pdf_area(float uvx, float uvy) const
    {
        int f;
        float u, v, area;
        f3 p, n;
        address(uvx, uvy, f, u, v);
        surface(f, u, v, p, n, area);
        return safe_div(1, area) / (float)num_tris;
    }
};

IG_DEV f3 square_to_sphere(float px, float py);

// make_sphere_area_emitter (light/area.art:259-317) for IG_LIGHT_SPHERE: d = centre (shape space), radius, radiance, area
struct SphereEmitter {
    m34 global;
    m33 nmat;
    f3 origin;
    float radius, area;
    Col radiance;

    IG_DEV SphereEmitter(const DevScene& sc, const ig_light& L)
    {
        const float4* e = reinterpret_cast<const float4*>(sc.entities + (size_t)L.entity_id * IG_ENTITY_FLOATS);
        const float4 r3 = e[3], r4 = e[4], r5 = e[5], r6 = e[6], r7 = e[7], r8 = e[8];
        global.c0 = f3{ r3.x, r3.y, r3.z }, global.c1 = f3{ r3.w, r4.x, r4.y }, global.c2 = f3{ r4.z, r4.w, r5.x }, global.c3 = f3{ r5.y, r5.z, r5.w };
        nmat.c0 = f3{ r6.x, r6.y, r6.z }, nmat.c1 = f3{ r6.w, r7.x, r7.y }, nmat.c2 = f3{ r7.z, r7.w, r8.x };
        origin   = f3{ L.d[0], L.d[1], L.d[2] };
        radius   = L.d[3];
        radiance = Col{ L.d[4], L.d[5], L.d[6] };
        area     = L.d[7];
    }
    // sphere_compute_surface_element_for_normal (shapes/sphere.art:30-46): point and face normal
    IG_DEV void surface(f3 normal, f3& point, f3& face_normal) const
    {
        face_normal = normalize3(mul33(nmat, normal));
        point       = xform_point(global, origin + normal * radius);
    }
    // sample_direct (area.art:268-294): a uniform point; one on the far side is mirrored through the centre and mapped back
    // with pmset.to_local_normal exactly as written (driver/pointmapper.art:31: not a unit vector)
    IG_DEV void sample(float ux, float uy, f3 from, f3& point, f3& face_normal) const
    {
        const f3 glb_org = xform_point(global, origin);
        surface(square_to_sphere(ux, uy), point, face_normal);
        const f3 os = from - glb_org, ps = from - point;
        if (!(dot3(ps, ps) <= dot3(os, os))) {
            const f3 np   = point + (glb_org - point) * 2;
            const f3 norm = normalize3(np - glb_org);
            const f3 diag = f3{ nmat.c0.x, nmat.c1.y, nmat.c2.z };
            const f3 ln   = f3{ dot3(nmat.c0, norm), dot3(nmat.c1, norm), dot3(nmat.c2, norm) } * (igm_rcp(dot3(diag, diag)));
            surface(ln, point, face_normal);
        }
    }
};

// CIE sky models (light/cie.art:1-41) as function environments (light/env.art:24-105); directions in the light's Y-up frame
struct CieSky {
    int kind;
    bool has_ground;
    Col zenith, ground, scale;
    float ground_brightness, zenith_brightness, c2;
    f3 sun_dir;
    m33 transform;

    IG_DEV explicit CieSky(const ig_light& L)
    {
        kind              = L.pad[0];
        has_ground        = L.pad[1] != 0;
        zenith            = Col{ L.d[0], L.d[1], L.d[2] };
        ground            = Col{ L.d[3], L.d[4], L.d[5] };
        ground_brightness = L.d[6];
        zenith_brightness = L.d[7];
        c2                = L.d[8];
        sun_dir           = f3{ L.d[9], L.d[10], L.d[11] };
        scale             = Col{ L.d[12], L.d[13], L.d[14] };
        transform.c0      = f3{ L.d[15], L.d[16], L.d[17] };
        transform.c1      = f3{ L.d[18], L.d[19], L.d[20] };
        transform.c2      = f3{ L.d[21], L.d[22], L.d[23] };
    }
    // cie_wmean (cie.art:1-7); pow(x, 10) as x^8 * x^2
    IG_DEV static Col wmean(float cos_theta, Col c1, Col c2)
    {
        const float x  = cos_theta + 1.01f;
        const float x2 = x * x;
        const float x4 = x2 * x2;
        const float a  = (x4 * x4) * x2;
        const float f1 = a * a / (a * a + 1);
        const float f2 = igm_rcp(a * a + 1);
        return c1 * f1 + c2 * f2;
    }
    IG_DEV Col radiance(f3 dir) const
    {
        const float cos_theta = dir.y;
        if (kind == IG_CIE_PEREZ) {
            // sky_function of make_perez_light_raw over perez::eval (light/perez.art:235-242,293-298); the explicit parameters
            // (a, b, c, d, e) travel in ground_brightness, zenith_brightness, c2, scale.r, scale.g
            const float cos_sun = clampf(dot3(dir, sun_dir), -1, 1);
            const float sun_a   = igm_acos(cos_sun);
            const float A       = 1 + ground_brightness * igm_exp(zenith_brightness / igm_max(1e-5f, cos_theta));
            const float B       = 1 + c2 * igm_exp(scale.r * sun_a) + scale.g * cos_sun * cos_sun;
            return wmean(cos_theta, zenith * (A * B), ground);
        }
        if (!has_ground && cos_theta < 0)
            return Col{ 0, 0, 0 };
        if (kind == IG_CIE_UNIFORM || kind == IG_CIE_CLOUDY) {
            const bool cloudy = kind == IG_CIE_CLOUDY;
            const float c1    = cloudy ? (1 + 2 * cos_theta) / 3 : 1.0f;
            const float k2    = cloudy ? 0.777777777f : 1.0f;
            return wmean(cos_theta, zenith * c1, ground * (ground_brightness * k2));
        }
        const float cos_gamma = dot3(dir, sun_dir);
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
        return scale * wmean(cos_theta, zenith * (zenith_brightness * c1), ground * (ground_brightness * c2));
    }
    IG_DEV Col emission(f3 ray_dir) const
    {
        const f3 local_dir = mul33(transform, ray_dir);
        if (!has_ground)
            return local_dir.y > kFltEps ? radiance(local_dir) : Col{ 0, 0, 0 };
        return radiance(local_dir);
    }
    IG_DEV float pdf(f3 ray_dir) const
    {
        if (has_ground)
            return 1 / (4 * kPi);
        const f3 local_dir = mul33(transform, ray_dir);
        return local_dir.y > kFltEps ? local_dir.y / kPi : 0.0f;
    }
};

// square_to_concentric_disk (core/warp.art:2-22)
IG_DEV f2 concentric_disk(float px, float py)
{
    const float a = 2 * px - 1;
    const float b = 2 * py - 1;
    if (a == 0 && b == 0)
        return f2{ 0, 0 };
    if (a * a > b * b) {
        const float phi = (kPi / 4) * safe_div(b, a);
        return f2{ igm_cos(phi) * a, igm_sin(phi) * a };
    }
    const float phi = (kPi / 2) - (kPi / 4) * safe_div(a, b);
    return f2{ igm_cos(phi) * b, igm_sin(phi) * b };
}

// ---------------------------------------------------------------- light selection

// equal_area_square_to_sphere (core/warp.art:63-91)
IG_DEV f3 square_to_sphere(float px, float py)
{
    const float u  = 2 * px - 1;
    const float v  = 2 * py - 1;
    const float au = igm_abs(u), av = igm_abs(v);
    const float sd = 1 - (au + av);
    const float d  = igm_abs(sd);
    const float r  = 1 - d;
    const float phi = (r == 0 ? 1.0f : (av - au) / r + 1) * kPi / 4;
    const float ct  = igm_copysign(1 - r * r, sd);
    const float st  = safe_sqrt(2 - r * r) * r;
    const float cp  = igm_copysign(igm_cos(phi), u);
    const float sp  = igm_copysign(igm_sin(phi), v);
    return f3{ cp * st, sp * st, ct };
}

// light/light_hierarchy.art:14-96
struct HierEntry {
    f3 pos, dir;
    float flux;
    int id;
    bool has_dir, is_leaf;
};
IG_DEV HierEntry hier_load(const DevScene& sc, int id)
{
    const float4* e = reinterpret_cast<const float4*>(sc.light_hierarchy + (size_t)id * 8);
    const float4 e1 = e[0], e2 = e[1];
    const int index = (int)igm_bits(e2.w);
    HierEntry h;
    h.pos     = f3{ e1.x, e1.y, e1.z };
    h.dir     = f3{ e2.x, e2.y, e2.z };
    h.flux    = igm_abs(e1.w);
    h.id      = index < 0 ? -index - 1 : index;
    h.has_dir = !igm_signbit(e1.w);
    h.is_leaf = index >= 0;
    return h;
}
IG_DEV float hier_cost(const HierEntry& e, f3 pos)
{
    const f3 cdir     = e.pos - pos;
    const float dist2 = dot3(cdir, cdir);
    const float cos_d = e.has_dir ? igm_abs(dot3(e.dir, normalize3(cdir))) : 1.0f;
    return safe_div(e.flux * cos_d, dist2);
}
IG_DEV float hier_left_prop(const HierEntry& l, const HierEntry& r, f3 pos) { return igm_rcp(1 + hier_cost(r, pos) / hier_cost(l, pos)); }
IG_DEV int hier_sample(const DevScene& sc, Tea& rnd, f3 pos, float& pdf)
{
    pdf           = 1;
    HierEntry ent = hier_load(sc, 0);
    while (!ent.is_leaf) {
        const HierEntry left = hier_load(sc, ent.id), right = hier_load(sc, ent.id + 1);
        const float prop     = hier_left_prop(left, right, pos);
        const bool is_left   = rnd.f32() < prop;
        ent                  = is_left ? left : right;
        pdf *= is_left ? prop : 1 - prop;
    }
    return ent.id;
}
IG_DEV float hier_pdf(const DevScene& sc, int finite_id, f3 pos)
{
    uint32_t code = sc.light_codes[finite_id];
    float pdf     = 1;
    HierEntry ent = hier_load(sc, 0);
    while (!ent.is_leaf) {
        const HierEntry left = hier_load(sc, ent.id), right = hier_load(sc, ent.id + 1);
        const float prop     = hier_left_prop(left, right, pos);
        const bool is_left   = (code & 0x1) == 0;
        ent                  = is_left ? left : right;
        pdf *= is_left ? prop : 1 - prop;
        code >>= 1;
    }
    return pdf;
}

// LightSelector (light/light_selector.art): uniform (:26-46) or hierarchy (:80-110)
IG_DEV int pick_light_id(Tea& rnd, int num) { return num <= 1 ? 0 : rnd.range(0, num - 1); }

// make_cdf_light_selector (light/light_selector.art:48-78): finite lights by the CDF over their flux
template <bool FULL>
IG_DEV int select_light_cdf(const DevScene& sc, Tea& rnd, float& pdf)
{
    const int n_inf = (int)sc.infinite_light_count, n_fin = (int)(sc.light_count - sc.infinite_light_count);
    const Cdf1D cdf{ sc.light_cdf, n_fin };
    if (n_inf == 0)
        return cdf.sample_discrete(rnd.f32(), pdf);
    const float q = rnd.f32();
    if (q < 0.5f) {
        const int id = pick_light_id(rnd, n_inf);
        pdf          = (igm_rcp((float)n_inf)) * 0.5f;
        return id;
    }
    float p;
    const int fid = cdf.sample_discrete(rnd.f32(), p);
    pdf           = p * (1 - 0.5f);
    return n_inf + fid;
}
template <bool FULL>
IG_DEV float select_pdf_cdf(const DevScene& sc, int li)
{
    const int n_inf = (int)sc.infinite_light_count, n_fin = (int)(sc.light_count - sc.infinite_light_count);
    const Cdf1D cdf{ sc.light_cdf, n_fin };
    if (n_inf == 0)
        return cdf.pdf_discrete(li);
    return li < n_inf ? (igm_rcp((float)n_inf)) * 0.5f : cdf.pdf_discrete(li - n_inf) * (1 - 0.5f);
}

template <bool FULL>
IG_DEV int select_light(const DevScene& sc, Tea& rnd, f3 from_pos, float& pdf)
{
    const int n_inf = (int)sc.infinite_light_count, n_fin = (int)(sc.light_count - sc.infinite_light_count);
    if (FULL && sc.light_cdf)
        return select_light_cdf<FULL>(sc, rnd, pdf);
    if (!sc.use_hierarchy) {
        const int num = (int)sc.light_count;
        pdf           = num == 0 ? 1.0f : igm_rcp((float)num);
        return pick_light_id(rnd, num);
    }
    if (n_inf == 0) {
        if (n_fin == 1) {
            pdf = 1;
            return 0;
        }
        return hier_sample(sc, rnd, from_pos, pdf);
    }
    const float pdf_inf = igm_rcp((float)n_inf);
    const float q       = rnd.f32();
    if (q < 0.5f) {
        const int id = pick_light_id(rnd, n_inf);
        pdf          = pdf_inf * 0.5f;
        return id;
    }
    float p = 1;
    int fid = 0;
    if (n_fin != 1)
        fid = hier_sample(sc, rnd, from_pos, p);
    pdf = p * (1 - 0.5f);
    return n_inf + fid;
}

template <bool FULL>
IG_DEV float select_pdf(const DevScene& sc, int li, f3 from_pos)
{
    const int n_inf = (int)sc.infinite_light_count, n_fin = (int)(sc.light_count - sc.infinite_light_count);
    if (FULL && sc.light_cdf)
        return select_pdf_cdf<FULL>(sc, li);
    if (!sc.use_hierarchy)
        return sc.light_count == 0 ? 1.0f : igm_rcp((float)sc.light_count);
    if (n_inf == 0)
        return n_fin == 1 ? 1.0f : hier_pdf(sc, li, from_pos);
    if (li < n_inf)
        return (igm_rcp((float)n_inf)) * 0.5f;
    return (n_fin == 1 ? 1.0f : hier_pdf(sc, li - n_inf, from_pos)) * (1 - 0.5f);
}

// ---------------------------------------------------------------- one path vertex

// Where does a wave of k_shade spend its cycles? A build with -DIG_SHADE_CLOCKS (tools/shade_clocks.py) reads the shader clock at
// phase boundaries after draining the outstanding memory operations, so a phase is charged with the latencies of the loads it issued;
// everywhere else the marks are empty.
struct NoClock {
    IG_DEV void mark(int) {}
};
#ifdef IG_SHADE_CLOCKS
struct PhaseClock {
    unsigned long long last;
    unsigned long long acc[12];
    IG_DEV void start()
    {
        for (int k = 0; k < 12; ++k)
            acc[k] = 0;
        last = __builtin_readcyclecounter();
    }
    IG_DEV void mark(int k)
    {
        __builtin_amdgcn_s_waitcnt(0);
        const unsigned long long now = __builtin_readcyclecounter();
        acc[k] += now - last;
        last = now;
    }
};
#endif

// make_homogeneous_medium / make_vacuum_medium (medium/homogeneous.art:1-58, driver/medium.art) with the Henyey-Greenstein phase
// function (phase/henyeygreenstein.art) for the volumetric path tracer
struct Medium {
    bool vacuum, scattering;
    Col sigma_t;
    int sigma_ind;
    float sigma_t_p, g;

    IG_DEV Medium(const DevScene& sc, int id)
    {
        vacuum     = id < 0 || (uint32_t)id >= sc.media_count || sc.media[id].type == IG_MEDIUM_VACUUM; // unknown ids: vacuum (LoaderMedium.cpp:104)
        scattering = false;
        sigma_t    = Col{ 0, 0, 0 };
        sigma_ind  = 0;
        sigma_t_p  = 0;
        g          = 0;
        if (!vacuum) {
            const ig_medium& m = sc.media[id];
            sigma_t    = Col{ m.sigma_a[0] + m.sigma_s[0], m.sigma_a[1] + m.sigma_s[1], m.sigma_a[2] + m.sigma_s[2] };
            scattering = !(igm_abs(m.sigma_s[0]) <= 1e-4f && igm_abs(m.sigma_s[1]) <= 1e-4f && igm_abs(m.sigma_s[2]) <= 1e-4f);
            sigma_ind  = sigma_t.r < sigma_t.g ? (sigma_t.r < sigma_t.b ? 0 : 2) : (sigma_t.g < sigma_t.b ? 1 : 2); // vec3_min_index
            sigma_t_p  = sigma_ind == 0 ? sigma_t.r : (sigma_ind == 1 ? sigma_t.g : sigma_t.b);
            g          = m.g;
        }
    }
    IG_DEV Col eval_tr(float t) const { return Col{ igm_exp(-sigma_t.r * t), igm_exp(-sigma_t.g * t), igm_exp(-sigma_t.b * t) }; }
    IG_DEV Col eval(f3 a, f3 b) const { return vacuum ? Col{ 1, 1, 1 } : eval_tr(len3(b - a)); }
    IG_DEV Col eval_inf() const
    {
        if (vacuum)
            return Col{ 1, 1, 1 };
        const bool clear = igm_abs(sigma_t.r) <= 1e-4f && igm_abs(sigma_t.g) <= 1e-4f && igm_abs(sigma_t.b) <= 1e-4f;
        return (!scattering && clear) ? Col{ 1, 1, 1 } : Col{ 0, 0, 0 };
    }
    // sample (homogeneous.art:38-52): a distance along the segment, or nothing (no random number drawn without scattering)
    IG_DEV bool sample(Tea& rnd, f3 p_start, f3 p_end, f3& pos, Col& weight) const
    {
        if (vacuum || !scattering)
            return false;
        const f3 dir_u    = p_end - p_start;
        const float dist  = len3(dir_u);
        const float ndist = igm_min(dist, -igm_log(1 - rnd.f32() * 0.99999f) / sigma_t_p);
        if (igm_abs(dist - ndist) <= 1e-3f)
            return false;
        pos             = p_start + (dir_u * safe_div(1, dist)) * ndist;
        const Col tr    = eval_tr(ndist);
        const float pdf = (sigma_ind == 0 ? tr.r : (sigma_ind == 1 ? tr.g : tr.b)) * sigma_t_p;
        weight          = Col{ tr.r / pdf, tr.g / pdf, tr.b / pdf };
        return true;
    }
    // make_henyeygreenstein_phase(g).sample (henyeygreenstein.art:20-38), weight 1; the anisotropic direction is returned in the
    // sampling frame, as written there
    IG_DEV f3 sample_phase(Tea& rnd) const
    {
        if (igm_abs(g) <= 1e-3f) {
            const float u   = rnd.f32();
            const float v   = rnd.f32();
            const float c   = 2 * v - 1;
            const float sn  = safe_sqrt(1 - c * c);
            const float phi = 2 * kPi * u;
            return f3{ sn * igm_cos(phi), sn * igm_sin(phi), c };
        }
        const float sqr_term  = (1 - g * g) / (1 + g - 2 * g * rnd.f32());
        const float cos_theta = -(1 + g * g - sqr_term * sqr_term) / (2 * g);
        const float sin_theta = igm_sqrt(igm_max(0.0f, 1 - cos_theta * cos_theta));
        const float phi       = 2 * kPi * rnd.f32();
        return f3{ sin_theta * igm_cos(phi), sin_theta * igm_sin(phi), cos_theta };
    }
};

struct PathVertexIn {
    int ray_id;
    f3 org, dir;
    uint32_t rnd;
    float inv_pdf;
    Col contrib;
    int depth; // low 16 bits: path depth; high 16 bits: current medium + 1 (volumetric path tracer: VPTRayPayload.medium)
    float eta;
    // hit (ent < 0: miss)
    int ent, prim;
    float t, u, v;
    // the path's generator seed, when the stream carries it (kStreamShaded: a bounce ray's rayB.w; the tail keeps it in a register);
    // 0 = not known: make_seed from the ray id (what a camera ray's first vertex does). A seed that IS 0 is recomputed, to the same value.
    uint32_t seed;
};

struct PathVertexOut {
    bool has_radiance; // emission / environment contribution to splat
    Col radiance;
    bool shadow;       // NEE shadow ray + its pre-weighted colour
    f3 s_org, s_dir;
    float s_tmax;
    Col s_col;
    bool bounce;       // continued path
    f3 b_org, b_dir;
    float b_tmin;
    uint32_t b_rnd, b_seed; // b_seed: the path's generator seed (the path tracer's callbacks; 0 from the others)
    float b_inv_pdf, b_eta;
    Col b_contrib;
    int b_depth;
};

// colormap::palette (core/colormap.art:68-92); in constant memory: as a local array it would live in scratch
__device__ __constant__ const float kDebugPalette[23][3] = {
        { 0.450000f, 0.376630f, 0.112500f }, { 0.112500f, 0.450000f, 0.405978f }, { 0.112500f, 0.450000f, 0.229891f }, { 0.450000f, 0.112500f, 0.376630f },
        { 0.435326f, 0.450000f, 0.112500f }, { 0.112500f, 0.141848f, 0.450000f }, { 0.435326f, 0.112500f, 0.450000f }, { 0.112500f, 0.450000f, 0.141848f },
        { 0.347283f, 0.450000f, 0.112500f }, { 0.450000f, 0.112500f, 0.200543f }, { 0.112500f, 0.229891f, 0.450000f }, { 0.450000f, 0.288587f, 0.112500f },
        { 0.347283f, 0.112500f, 0.450000f }, { 0.450000f, 0.112500f, 0.288587f }, { 0.450000f, 0.112500f, 0.112500f }, { 0.450000f, 0.200543f, 0.112500f },
        { 0.171196f, 0.450000f, 0.112500f }, { 0.112500f, 0.450000f, 0.317935f }, { 0.259239f, 0.450000f, 0.112500f }, { 0.259239f, 0.112500f, 0.450000f },
        { 0.112500f, 0.405978f, 0.450000f }, { 0.171196f, 0.112500f, 0.450000f }, { 0.112500f, 0.317935f, 0.450000f }
};
IG_DEV Col debug_palette(int i)
{
    const int k = i % 23;
    return Col{ kDebugPalette[k][0], kDebugPalette[k][1], kDebugPalette[k][2] };
}

// on_hit of make_debug_renderer (technique/debugtracer.art:3-140) over the point mappers of driver/pointmapper.art:28-36
template <bool FULL, class Ctx>
IG_DEV Col debug_color(const DevScene& sc, int mode, const PathVertexIn& in, const Surf& surf, const Ctx& bsdf, const ig_material& mat, int mat_id)
{
    const float4* e = reinterpret_cast<const float4*>(sc.entities + (size_t)in.ent * IG_ENTITY_FLOATS);
    auto absv       = [](f3 n) { return Col{ igm_abs(n.x), igm_abs(n.y), igm_abs(n.z) }; };
    auto local_n    = [&](f3 n) { // to_local_normal: mat3x3_left_mul(normal_mat, n) / |diag(normal_mat)|^2
        const float4 r6 = e[6], r7 = e[7], r8 = e[8];
        const f3 c0{ r6.x, r6.y, r6.z }, c1{ r6.w, r7.x, r7.y }, c2{ r7.z, r7.w, r8.x };
        const f3 d{ c0.x, c1.y, c2.z };
        return normalize3(f3{ dot3(c0, n), dot3(c1, n), dot3(c2, n) } * (igm_rcp(dot3(d, d))));
    };
    auto local_p = [&](f3 p) { // to_local_point: entity.local_mat
        const float4 r0 = e[0], r1 = e[1], r2 = e[2];
        m34 local;
        local.c0 = f3{ r0.x, r0.y, r0.z }, local.c1 = f3{ r0.w, r1.x, r1.y }, local.c2 = f3{ r1.z, r1.w, r2.x }, local.c3 = f3{ r2.y, r2.z, r2.w };
        return xform_point(local, p);
    };
    const Col yes{ 0, 0, 1 }, no{ 1, 0, 0 }; // true_color = blue, false_color = red (core/color.art:74-75)
    const uint4 ext   = sc.entity_ext[in.ent];
    const bool sphere = ext.y == 0xFFFFFFFFu;
    const int inner = (mat.pad[2] & 0xFFFF) - 1, outer = ((mat.pad[2] >> 16) & 0xFFFF) - 1;
    switch (mode) {
    case 1: return absv(surf.local.c0);
    case 2: return absv(surf.local.c1);
    case 3: return absv(surf.face_normal);
    case 4: return absv(local_n(surf.local.c2));
    case 5: return absv(local_n(surf.local.c0));
    case 6: return absv(local_n(surf.local.c1));
    case 7: return absv(local_n(surf.face_normal));
    case 8: return Col{ igm_abs(surf.tex.x), igm_abs(surf.tex.y), 0 };
    case 9: return Col{ igm_abs(in.u), igm_abs(in.v), 0 };
    case 10: return Col{ surf.point.x, surf.point.y, surf.point.z };
    case 11: {
        const f3 p = local_p(surf.point);
        return Col{ p.x, p.y, p.z };
    }
    case 12: { // to_normalized_point: the shape's bounding box (floats 4..6 / 8..10 of a mesh header; a sphere's centre +- radius)
        const f3 lp = local_p(surf.point);
        f3 lo, hi;
        if (sphere) {
            const float4 sp = *reinterpret_cast<const float4*>(sc.shape_data + ext.x);
            lo = f3{ sp.x - sp.w, sp.y - sp.w, sp.z - sp.w };
            hi = f3{ sp.x + sp.w, sp.y + sp.w, sp.z + sp.w };
        } else {
            const float* hdr = reinterpret_cast<const float*>(sc.shape_data + ext.x) - 12; // the vertices follow the 12-float header
            lo = f3{ hdr[4], hdr[5], hdr[6] };
            hi = f3{ hdr[8], hdr[9], hdr[10] };
        }
        return Col{ safe_div(lp.x - lo.x, hi.x - lo.x), safe_div(lp.y - lo.y, hi.y - lo.y), safe_div(lp.z - lo.z, hi.z - lo.z) };
    }
    case 13: return Col{ in.t, in.t, in.t };
    case 14: { // surf.area: the triangle in scene space (core/triangle.art:12-29), or compute_ellipsoid_area (shapes/sphere.art:21-28)
        const float4 r3 = e[3], r4 = e[4], r5 = e[5];
        m34 global;
        global.c0 = f3{ r3.x, r3.y, r3.z }, global.c1 = f3{ r3.w, r4.x, r4.y }, global.c2 = f3{ r4.z, r4.w, r5.x }, global.c3 = f3{ r5.y, r5.z, r5.w };
        float area;
        if (sphere) {
            const float r  = reinterpret_cast<const float4*>(sc.shape_data + ext.x)->w;
            const f3 a = global.c0 * r, b = global.c1 * r, c = global.c2 * r;
            const float l1 = dot3(a, a), l2 = dot3(b, b), l3 = dot3(c, c);
            const float P  = 1.6f;
            area = 4 * kPi * igm_pow((igm_pow(l1 * l2, P / 2) + igm_pow(l1 * l3, P / 2) + igm_pow(l2 * l3, P / 2)) / 3, 1 / P);
        } else {
            const float* verts = reinterpret_cast<const float*>(sc.shape_data + ext.x);
            const int4 tri     = *reinterpret_cast<const int4*>(reinterpret_cast<const float*>(sc.shape_data + ext.z) + in.prim * 4);
            const f3 v0 = xform_point(global, ld3v(verts + tri.x * 4)), v1 = xform_point(global, ld3v(verts + tri.y * 4)), v2 = xform_point(global, ld3v(verts + tri.z * 4));
            area = len3(stable_normal(v2 - v0, v0 - v1, v1 - v2)) / 2;
        }
        return Col{ area, area, area };
    }
    case 15: return Col{ (float)in.prim, (float)in.prim, (float)in.prim };
    case 16: return debug_palette(in.prim);
    case 17: return Col{ (float)in.ent, (float)in.ent, (float)in.ent };
    case 18: return debug_palette(in.ent);
    case 19: return Col{ (float)mat_id, (float)mat_id, (float)mat_id };
    case 20: return debug_palette(mat_id);
    case 21: return mat.light_id >= 0 ? yes : no;
    case 22: return bsdf.all_delta() ? yes : no;
    case 23: return surf.entering ? yes : no;
    case 24: { // DEBUG_CHECK_BSDF: red / orange / yellow / blue = neither / only the pdf / only the weight / both agree; pink = no sample
        auto verdict = [](int index) { // [red, orange, yellow, blue](index)
            return index == 0 ? Col{ 1, 0, 0 } : (index == 1 ? Col{ 1, 0.5f, 0 } : (index == 2 ? Col{ 1, 1, 0 } : Col{ 0, 0, 1 }));
        };
        const f3 N = surf.local.c2, out_dir = -in.dir;
        if (bsdf.all_delta()) {
            const f3 r       = N * (2 * dot3(N, out_dir)) - out_dir;
            const Col evl    = bsdf.eval(r, out_dir);
            const float pdf  = bsdf.pdf(r, out_dir);
            const int pdf_ok = igm_abs(0 - pdf) <= kFltEps ? 1 : 0;
            const int w_ok   = igm_abs(0 - evl.r) + igm_abs(0 - evl.g) + igm_abs(0 - evl.b) <= kFltEps ? 1 : 0;
            return verdict((w_ok << 1) | pdf_ok);
        }
        Tea tmp{ fnv_step(fnv_step(fnv_step(0x811C9DC5u, igm_bits(in.t)), igm_bits(in.u)), igm_bits(in.v)), 1 };
        f3 in_dir;
        float spdf, s_eta;
        Col scol;
        bool sdelta;
        if (!bsdf.sample(tmp, out_dir, in_dir, spdf, scol, s_eta, sdelta))
            return Col{ 1, 0, 1 };
        const float pdf  = bsdf.pdf(in_dir, out_dir);
        const Col evl    = bsdf.eval(in_dir, out_dir) * safe_div(1, pdf);
        const int pdf_ok = igm_abs(spdf - pdf) <= 0.001f ? 1 : 0;
        const int w_ok   = igm_abs(scol.r - evl.r) + igm_abs(scol.g - evl.g) + igm_abs(scol.b - evl.b) <= 0.001f ? 1 : 0;
        return verdict((w_ok << 1) | pdf_ok);
    }
    case 25: return bsdf.albedo(-in.dir);
    case 26: return inner < 0 ? Col{ 0, 0, 0 } : debug_palette(inner);
    case 27: return outer < 0 ? Col{ 0, 0, 0 } : debug_palette(outer);
    default: return absv(surf.local.c2);
    }
}

constexpr float kRayOffset = 0.001f; // technique/pathtracer.art:41

IG_DEV Col clamp_color(const ig_technique& tech, Col c) // handle_color, technique/pathtracer.art:46-50
{
    if (tech.clamp > 0)
        return Col{ igm_min(c.r, tech.clamp), igm_min(c.g, tech.clamp), igm_min(c.b, tech.clamp) };
    return c;
}

// gpu_hit_shade / gpu_miss_shade body (driver/mapping_gpu.art:123-274) with the path tracer
// callbacks on_hit / on_shadow / on_bounce / on_miss (technique/pathtracer.art:52-210).
// DEBUG_VIEWS: the instantiation for the debug technique (its 28 views, with a second copy of the BSDF code for the BSDF check,
// cost the ordinary full kernel ten times its spills when they were a run-time branch of it)
// EXPR: the instantiation for scenes whose materials carry shading expressions (include/ig_expr.h)
template <bool FULL, bool DEBUG_VIEWS = false, bool EXPR = false, uint32_t TYPES = ~0u, class CLK = NoClock>
IG_DEV void shade_vertex(const DevScene& sc, const ShadeFrame& fr, const PathVertexIn& in, PathVertexOut& out, CLK&& clk = CLK{})
{
    out.has_radiance = false;
    out.shadow       = false;
    out.bounce       = false;
    out.radiance     = Col{ 0, 0, 0 };

    const ig_technique tech = sc.tech;
    const bool nee          = tech.nee != 0;
    // make_volume_path_renderer (technique/volpathtracer.art:37-260): the same callbacks with a current medium in the payload
    const bool volumetric = FULL && tech.type == IG_TECHNIQUE_VOLPATH;
    const int depth       = FULL ? (in.depth & 0xFFFF) : in.depth;
    const int medium_id   = FULL ? (int)((uint32_t)in.depth >> 16) - 1 : -1; // (unsigned: medium ids up to 65 534 fill the upper half)
    const float mis_inv_pdf = volumetric ? igm_max(0.0f, in.inv_pdf) : in.inv_pdf; // "ignore medium interactions" (volpathtracer.art:101,134)

    if (in.ent < 0) {
        // a sample for which the camera had no ray (masked fishlens) carries the zero ray of
        // gpu_generate_rays (mapping_gpu.art:655-658): it can only miss, and it is dropped here
        if (in.dir.x == 0 && in.dir.y == 0 && in.dir.z == 0)
            return;
        if (FULL && (tech.type == IG_TECHNIQUE_AO || tech.type == IG_TECHNIQUE_DEBUG || tech.type == IG_TECHNIQUE_WIREFRAME))
            return; // make_ao_renderer, make_debug_renderer and make_wireframe_renderer have no on_miss
        // ---- miss: on_miss (technique/pathtracer.art:141-168) over infinite, non-delta lights
        Col sum{ 0, 0, 0 };
        for (uint32_t li = 0; li < sc.infinite_light_count; ++li) {
            const ig_light& L = sc.lights[li];
            Col emit;
            float pdf_s;
            if (L.type == IG_LIGHT_ENV) {
                emit  = Col{ L.d[0], L.d[1], L.d[2] };
                pdf_s = 1 / (4 * kPi); // equal_area_sphere_pdf (light/env.art:101)
            } else if (FULL && L.type == IG_LIGHT_ENV_TEXTURED) {
                const TexturedEnv env(sc, L);
                emit  = env.emission(in.dir);
                pdf_s = env.pdf(in.dir);
            } else if (FULL && L.type == IG_LIGHT_CIE) {
                const CieSky sky(L);
                emit  = sky.emission(in.dir);
                pdf_s = sky.pdf(in.dir);
            } else if (FULL && L.type == IG_LIGHT_SUN) {
                // make_sun_light.emission / pdf_direct (light/sun.art:31-45)
                const bool hit = dot3(f3{ L.d[0], L.d[1], L.d[2] }, in.dir) >= L.d[3];
                emit           = hit ? Col{ L.d[4], L.d[5], L.d[6] } : Col{ 0, 0, 0 };
                pdf_s          = hit ? safe_div(1, 2 * kPi * (1 - L.d[3])) : 0.0f;
            } else if (FULL && L.type == IG_LIGHT_PEREZ) {
                // make_perez_light_raw with a sun (light/perez.art:301-317): the sun's emission plus the sky function, the sun's pdf
                const CieSky sky(L);
                const bool hit = dot3(f3{ L.d[27], L.d[28], L.d[29] }, in.dir) >= L.d[14];
                const f3 d     = f3{ dot3(sky.transform.c0, in.dir), dot3(sky.transform.c1, in.dir), dot3(sky.transform.c2, in.dir) };
                emit           = (hit ? Col{ L.d[24], L.d[25], L.d[26] } : Col{ 0, 0, 0 }) + sky.radiance(d);
                pdf_s          = hit ? safe_div(1, 2 * kPi * (1 - L.d[14])) : 0.0f;
            } else {
                continue; // delta lights
            }
            const float mis   = nee ? igm_rcp(1 + mis_inv_pdf * select_pdf<FULL>(sc, (int)li, in.org) * pdf_s) : 1.0f;
            if (volumetric)
                emit = emit * Medium(sc, medium_id).eval_inf(); // volpathtracer.art:135
            const Col c       = clamp_color(tech, (in.contrib * emit) * mis);
            sum               = Col{ sum.r + c.r, sum.g + c.g, sum.b + c.b };
        }
        out.has_radiance = true;
        out.radiance     = sum;
        return;
    }

    // the path tracer itself (emission, NEE geometry, ray offsets) keeps the unperturbed surface
    const Surf surf = surface_element<FULL>(sc, in.ent, in.prim, in.org, in.dir, in.t, in.u, in.v);
    clk.mark(2);
    ig_material mat_local;
    const ig_material& mat = resolve_material<EXPR>(sc, sc.materials[sc.entity_material[in.ent]], surf, -in.dir, mat_local);
    const BsdfCtx<FULL, true, EXPR, TYPES> bsdf(sc, mat, surf, in.dir, std::bool_constant<EXPR>{});
    clk.mark(3);
    const f3 N       = surf.local.c2;
    const f3 out_dir = -in.dir;

    // RNG resumes where the path left off (mapping_gpu.art:171)
    // (the seed is a hash of the sample's coordinates: six FNV steps behind three divisions; a path computes it once and carries it)
    uint32_t seed = in.seed;
    if (seed == 0u) {
        int it_l, sample, px, row; // multi-iteration call: which of its iterations; the pixel's sample; local pixel
        fr.decompose(in.ray_id, it_l, sample, px, row);
        const int py = fr.row_offset + row * fr.row_stride;
        seed         = make_seed(sample, fr.iteration + it_l, fr.frame, px, py, fr.seed);
    }
    out.b_seed = seed;
#ifdef IG_EXP_CHEAP_SEED // experiment: what do the seed hash and the three integer divisions in front of it cost? (wrong images)
    Tea rnd{ (uint32_t)in.ray_id * 747796405u + 12345u, in.rnd };
#else
    Tea rnd{ seed, in.rnd };
#endif

    if constexpr (FULL && DEBUG_VIEWS) {
        if (tech.type == IG_TECHNIQUE_WIREFRAME) {
            // make_wireframe_renderer (technique/wireframe.art:21-73); the payload's distance travels in the inv_pdf slot.
            // is_edge_hit (:24-31): clampf(0, 1, 1 - u - v) as written = min(1 - u - v, 1)
            const float w      = clampf(0, 1, 1 - in.u - in.v);
            const float edge_t = igm_min(in.u, igm_min(in.v, w));
            const float cond   = 0.01f * ((in.t + in.inv_pdf) * fr.wire_footprint) * igm_sqrt(triangle_inv_area(sc, in.ent, in.prim));
            if (edge_t <= cond) { // on_hit: color_lerp(white, black, t); no bounce
                const float c    = (1 - edge_t) * 1.0f + edge_t * 0.0f;
                out.has_radiance = true;
                out.radiance     = Col{ c, c, c };
            } else { // on_bounce: straight on
                out.bounce    = true;
                out.b_org     = surf.point;
                out.b_dir     = in.dir;
                out.b_tmin    = kRayOffset;
                out.b_rnd     = in.rnd;
                out.b_inv_pdf = in.inv_pdf + in.t;
                out.b_contrib = in.contrib;
                out.b_depth   = depth + 1;
                out.b_eta     = in.eta;
            }
            return;
        }
        // on_hit of make_debug_renderer (technique/debugtracer.art:3-140): one of 28 properties of the first hit; nothing else
        out.has_radiance = true;
        out.radiance     = debug_color<FULL>(sc, tech.debug_mode, in, surf, bsdf, mat, sc.entity_material[in.ent]);
        return;
    }
    if (FULL && tech.type == IG_TECHNIQUE_AO) {
        // make_ao_renderer.on_shadow (technique/aotracer.art:7-19): a cosine-distributed direction around the surface frame
        // (make_lambertian_bsdf(ctx.surf, white).sample), traced as a "shadow" ray with the bounce visibility flag and no far
        // end; its colour (kd = white) is splatted where it escapes. Nothing else: no emission, no bounce.
        float cpdf;
        const f3 d   = Principled::cosine_hemisphere(rnd, cpdf);
        out.shadow   = true;
        out.s_org    = surf.point;
        out.s_dir    = mul33(surf.local, d);
        out.s_tmax   = kFltMax;
        out.s_col    = Col{ 1, 1, 1 };
        return;
    }

    // ---- on_hit (technique/pathtracer.art:119-139): emission with MIS
    if (mat.light_id >= 0 && surf.entering) {
        const float dcos = -dot3(in.dir, N);
        if (dcos > kFltEps) {
            const ig_light& EL = sc.lights[mat.light_id];
            Col emit;
            float pdf_s;
            if (FULL && EL.type == IG_LIGHT_MESH_AREA) {
                const MeshEmitter me(sc, EL);
                emit  = me.radiance;
                pdf_s = me.pdf_area(in.u, in.v) * (in.t * in.t) / dcos; // Pdf::as_solid (driver/pdf.art:19-38)
            } else if (FULL && EL.type == IG_LIGHT_SPHERE) {
                emit  = Col{ EL.d[4], EL.d[5], EL.d[6] };
                pdf_s = safe_div(1, EL.d[7]) * (in.t * in.t) / dcos; // make_area_pdf(inv_area).as_solid
            } else {
                const PlaneLight pl(EL);
                emit  = pl.radiance;
                pdf_s = pl.pdf(in.org);
            }
            const float mis   = nee ? igm_rcp(1 + mis_inv_pdf * select_pdf<FULL>(sc, mat.light_id, in.org) * pdf_s) : 1.0f;
            if (volumetric)
                emit = emit * Medium(sc, medium_id).eval(in.org, surf.point); // volpathtracer.art:104-105
            out.has_radiance  = true;
            out.radiance      = clamp_color(tech, (in.contrib * emit) * mis);
        }
    }

    clk.mark(4);
    // ---- on_shadow (technique/pathtracer.art:52-117): next event estimation
    if (nee && !bsdf.all_delta() && sc.light_count != 0 && depth + 1 <= tech.max_depth) {
        float sel_pdf;
        const int lid     = select_light<FULL>(sc, rnd, surf.point, sel_pdf);
        const ig_light& L = sc.lights[lid];
        f3 lpos{}, ldir{};
        Col lint{ 0, 0, 0 };
        float pdf_value = 0, lcos = 0, ldist = 0;
        bool pdf_area = false, delta = false, infinite = false;
        if (L.type == IG_LIGHT_PLANE) {
            // make_area_light.sample_direct (light/area.art:10-26)
            const PlaneLight pl(L);
            const float ux = rnd.f32();
            const float uy = rnd.f32();
            float weight;
            pl.sample(ux, uy, surf.point, lpos, pdf_value, weight);
            const f3 d_ = lpos - surf.point;
            ldist       = len3(d_);
            ldir        = d_ * safe_div(1, ldist);
            lcos        = dot3(ldir, pl.normal) * (surf.entering ? -1.0f : 1.0f);
            lint        = pl.radiance * weight;
        } else if (L.type == IG_LIGHT_POINT) {
            // make_point_light.sample_direct (light/point.art:3-8)
            lpos        = f3{ L.d[0], L.d[1], L.d[2] };
            const f3 d_ = lpos - surf.point;
            ldist       = len3(d_);
            ldir        = d_ * safe_div(1, ldist);
            lint        = Col{ L.d[4], L.d[5], L.d[6] };
            pdf_value   = 1;
            pdf_area    = true;
            lcos        = 1;
            delta       = true;
        } else if (L.type == IG_LIGHT_SPOT) {
            // make_spot_light.sample_direct (light/spot.art:8-42); cosines of cutoff / falloff come from the table
            lpos                = f3{ L.d[0], L.d[1], L.d[2] };
            const f3 sdir       = f3{ L.d[4], L.d[5], L.d[6] };
            const float cos_cut = L.d[3], blend = L.d[7] - L.d[3];
            const f3 d_         = lpos - surf.point;
            ldist               = len3(d_);
            ldir                = d_ * safe_div(1, ldist);
            const float cos_a   = dot3(-ldir, sdir);
            float factor;
            if (blend <= kFltEps) {
                factor = cos_a <= cos_cut ? 0.0f : 1.0f;
            } else {
                const float x = clampf((cos_a - cos_cut) / blend, 0, 1);
                factor        = x * x * (3 - 2 * x);
            }
            lint      = Col{ L.d[8] * factor, L.d[9] * factor, L.d[10] * factor };
            pdf_value = dot3(-ldir, sdir) > cos_cut ? 1.0f : 0.0f;
            pdf_area  = true;
            lcos      = -dot3(ldir, sdir);
            delta     = true;
        } else if (L.type == IG_LIGHT_DIRECTIONAL) {
            // make_directional_light.sample_direct (light/directional.art:6)
            const f3 ddir = f3{ L.d[0], L.d[1], L.d[2] };
            lpos          = surf.point + ddir * (-sc.scene_radius);
            ldir          = -ddir;
            lint          = Col{ L.d[4], L.d[5], L.d[6] };
            pdf_value     = 1;
            lcos          = 1;
            ldist         = sc.scene_radius;
            delta         = true;
            infinite      = true;
        } else if (FULL && L.type == IG_LIGHT_ENV_TEXTURED) {
            // make_environment_light_textured.sample_direct (light/env.art:135-138)
            const TexturedEnv env(sc, L);
            Col intensity;
            env.sample_dir(rnd, ldir, intensity, pdf_value);
            lint     = intensity * (igm_rcp(pdf_value));
            lpos     = surf.point + ldir * sc.scene_radius;
            lcos     = 1.0f;
            ldist    = sc.scene_radius;
            infinite = true;
        } else if (FULL && L.type == IG_LIGHT_MESH_AREA) {
            // make_area_light.sample_direct over make_shape_area_emitter (light/area.art:12-26,62-71)
            const MeshEmitter me(sc, L);
            const float ux = rnd.f32();
            const float uy = rnd.f32();
            int f;
            float u, v, area;
            f3 fnorm;
            me.address(ux, uy, f, u, v);
            me.surface(f, u, v, lpos, fnorm, area);
            pdf_value   = safe_div(1, area) / (float)me.num_tris;
            pdf_area    = true;
            const f3 d_ = lpos - surf.point;
            ldist       = len3(d_);
            ldir        = d_ * safe_div(1, ldist);
            lcos        = dot3(ldir, fnorm) * (surf.entering ? -1.0f : 1.0f);
            lint        = me.radiance * (area * (float)me.num_tris);
        } else if (FULL && L.type == IG_LIGHT_SPHERE) {
            // make_area_light.sample_direct over make_sphere_area_emitter (light/area.art:12-26,268-294)
            const SphereEmitter se(sc, L);
            const float ux = rnd.f32();
            const float uy = rnd.f32();
            f3 fnorm;
            se.sample(ux, uy, surf.point, lpos, fnorm);
            pdf_value   = safe_div(1, se.area);
            pdf_area    = true;
            const f3 d_ = lpos - surf.point;
            ldist       = len3(d_);
            ldir        = d_ * safe_div(1, ldist);
            lcos        = dot3(ldir, fnorm) * (surf.entering ? -1.0f : 1.0f);
            lint        = se.radiance * se.area;
        } else if (FULL && L.type == IG_LIGHT_CIE) {
            // make_environment_light_function_{hemi,spherical}.sample_direct (light/env.art:31-37,79-93)
            const CieSky sky(L);
            const float ux = rnd.f32();
            const float uy = rnd.f32();
            if (!sky.has_ground) {
                const float c   = safe_sqrt(uy); // sample_cosine_hemisphere (core/sampling.art:62-70)
                const float sn  = safe_sqrt(1 - uy);
                const float phi = 2 * kPi * ux;
                const f3 dir    = switch_env_up(f3{ sn * igm_cos(phi), sn * igm_sin(phi), c });
                pdf_value       = c / kPi;
                lint            = sky.radiance(dir) * (igm_rcp(pdf_value));
                ldir            = f3{ dot3(sky.transform.c0, dir), dot3(sky.transform.c1, dir), dot3(sky.transform.c2, dir) };
            } else {
                ldir      = square_to_sphere(ux, uy);
                pdf_value = 1 / (4 * kPi);
                lint      = sky.radiance(mul33(sky.transform, ldir)) * (igm_rcp(pdf_value));
            }
            lpos     = surf.point + ldir * sc.scene_radius;
            lcos     = 1.0f;
            ldist    = sc.scene_radius;
            infinite = true;
        } else if (FULL && (L.type == IG_LIGHT_SUN || L.type == IG_LIGHT_PEREZ)) {
            // make_sun_light.sample_direct (light/sun.art:21-25), sample_uniform_cone (core/sampling.art:109-116); the sun of a
            // Perez sky keeps its terms further back in the record
            const bool perez    = L.type == IG_LIGHT_PEREZ;
            const f3 sun_dir    = perez ? f3{ L.d[27], L.d[28], L.d[29] } : f3{ L.d[0], L.d[1], L.d[2] };
            const float cos_a   = perez ? L.d[14] : L.d[3];
            const float ux      = rnd.f32();
            const float uy      = rnd.f32();
            const float c1      = 1 - cos_a;
            const f2 p          = concentric_disk(ux, uy);
            const float n2      = p.x * p.x + p.y * p.y;
            const float z       = cos_a + c1 * (1 - n2);
            const float k       = safe_sqrt(c1 * (2 - c1 * n2));
            const f3 ndir       = mul33(orthonormal_basis(-sun_dir), f3{ p.x * k, p.y * k, z });
            const float inv_pdf = 2 * kPi * (1 - cos_a);
            ldir                = -ndir;
            lint                = (perez ? Col{ L.d[24], L.d[25], L.d[26] } : Col{ L.d[4], L.d[5], L.d[6] }) * inv_pdf;
            pdf_value           = safe_div(1, 2 * kPi * (1 - cos_a)); // uniform_cone_pdf
            if (perez) {
                // make_perez_light_raw.sample_direct (light/perez.art:304-308): plus the sky seen in the sampled direction
                const CieSky sky(L);
                const f3 d = f3{ dot3(sky.transform.c0, ldir), dot3(sky.transform.c1, ldir), dot3(sky.transform.c2, ldir) };
                lint       = lint + sky.radiance(d) * (igm_rcp(pdf_value));
            }
            lcos                = z;
            ldist               = __builtin_inff();
            infinite            = true;
        } else {
            // constant environment: make_environment_light_function_spherical.sample_direct (light/env.art:89-93)
            const float ux = rnd.f32();
            const float uy = rnd.f32();
            ldir           = square_to_sphere(ux, uy);
            pdf_value      = 1 / (4 * kPi);
            lint           = Col{ L.d[0], L.d[1], L.d[2] } * (igm_rcp(pdf_value));
            lpos           = surf.point + ldir * sc.scene_radius;
            lcos           = 1.0f;
            ldist          = sc.scene_radius;
            infinite       = true;
        }
        const float dist2   = ldist * ldist;
        const float pdf_l_s = (pdf_area ? pdf_value * dist2 / lcos : pdf_value) * sel_pdf; // driver/pdf.art:19-38
        if (pdf_l_s > kFltEps && lcos > kFltEps) {
            float mis = 1;
            if (!delta && !(volumetric && igm_signbit(in.inv_pdf))) // was_medium_interaction (volpathtracer.art:39,53)
                mis = igm_rcp(1 + bsdf.pdf(ldir, out_dir) / pdf_l_s);
            const float factor = pdf_value / pdf_l_s;
            const Col c        = clamp_color(tech, (lint * (in.contrib * bsdf.eval(ldir, out_dir))) * (mis * factor));
            if ((c.r + c.g + c.b) / 3 > kFltEps) {
                out.shadow = true;
                out.s_org  = surf.point;
                out.s_dir  = infinite ? ldir : lpos - surf.point;
                out.s_tmax = infinite ? kFltMax : 1 - kRayOffset;
                out.s_col  = c;
                if (volumetric) {
                    // volpathtracer.art:41,71-83: the current medium from the ray origin to the hit and on towards the light
                    const Medium medium(sc, medium_id);
                    out.s_col = c * (medium.eval(in.org, surf.point) * (infinite ? medium.eval_inf() : medium.eval(surf.point, lpos)));
                }
            }
        }
    }

    clk.mark(5);
    // ---- on_bounce (technique/pathtracer.art:170-210; technique/volpathtracer.art:155-247)
    if (depth + 1 <= tech.max_depth) {
        out.b_tmin = kRayOffset;
        Col path_contrib = in.contrib;
        if (volumetric) {
            // try the medium first: a scattering event between the ray origin and the hit replaces the surface bounce
            const Medium medium(sc, medium_id);
            f3 mpos;
            Col mweight;
            if (medium.sample(rnd, in.org, surf.point, mpos, mweight)) {
                const f3 pdir       = medium.sample_phase(rnd);
                const Col nc        = in.contrib * mweight;
                const float e2      = in.eta * in.eta;
                const float rr_prob = (depth + 1 > tech.min_depth) ? clampf(igm_max(nc.r * e2, igm_max(nc.g * e2, nc.b * e2)), 0.05f, 0.95f) : 1.0f;
                if (!(rnd.f32() >= rr_prob)) {
                    out.bounce    = true;
                    out.b_org     = mpos;
                    out.b_dir     = pdir;
                    out.b_tmin    = 0;
                    out.b_rnd     = rnd.counter;
                    out.b_inv_pdf = -1; // "the last interaction was a medium"
                    out.b_contrib = nc * (igm_rcp(rr_prob));
                    out.b_depth   = (depth + 1) | ((medium_id + 1) << 16);
                    out.b_eta     = in.eta;
                }
                return;
            }
            path_contrib = medium.eval(in.org, surf.point) * in.contrib;
        }
        f3 in_dir;
        float pdf, s_eta;
        Col color;
        bool sdelta;
        if (bsdf.sample(rnd, out_dir, in_dir, pdf, color, s_eta, sdelta) && pdf > kFltEps) {
            const Col nc        = path_contrib * color;
            const float e2      = in.eta * in.eta;
            const float rr_prob = (depth + 1 > tech.min_depth) ? clampf(igm_max(nc.r * e2, igm_max(nc.g * e2, nc.b * e2)), 0.05f, 0.95f) : 1.0f;
            if (!(rnd.f32() >= rr_prob)) {
                out.bounce    = true;
                out.b_org     = surf.point;
                out.b_dir     = in_dir;
                out.b_rnd     = rnd.counter;
                out.b_inv_pdf = sdelta ? 0 : igm_rcp(pdf);
                out.b_contrib = nc * (igm_rcp(rr_prob));
                out.b_depth   = depth + 1;
                out.b_eta     = in.eta * s_eta;
                if (volumetric) {
                    // a transmission continues in the medium on the other side (make_medium_interface.pick, driver/medium.art:34-38)
                    const bool transmission = igm_signbit(dot3(N, in_dir));
                    const int inner = (mat.pad[2] & 0xFFFF) - 1, outer = ((mat.pad[2] >> 16) & 0xFFFF) - 1;
                    const int next  = transmission ? (surf.entering ? inner : outer) : medium_id;
                    out.b_depth     = (depth + 1) | ((next + 1) << 16);
                }
            }
        }
    }
}

} // namespace igdev
