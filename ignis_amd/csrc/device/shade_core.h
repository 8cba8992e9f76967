// shade_core.h — per-ray shading for the gfx950 kernels: RNG, surface reconstruction, BSDFs,
// lights and the path-tracer callbacks, shared by the wavefront kernel (shade.hip) and the tail
// kernel (tail.hip). Arithmetic follows src/artic/{core,bsdf,light,technique} expression by
// expression (citations inline); transcendental functions come from include/ig_detmath.h.
#pragma once

#include "dev_math.h"
#include "kernels.h"

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
#pragma unroll
    for (int i = 0; i < 4; ++i) {
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
IG_DEV Surf surface_element(const DevScene& sc, int ent_id, int prim_id, f3 org, f3 dir, float t, float u, float v)
{
    // entity record = 9 x 16 bytes; words 12..35 hold toGlobal 3x4, the normal 3x3, shape id, material id
    const float4* e = reinterpret_cast<const float4*>(sc.entities + (size_t)ent_id * IG_ENTITY_FLOATS);
    const float4 r3 = e[3], r4 = e[4], r5 = e[5], r6 = e[6], r7 = e[7], r8 = e[8];
    m34 global;
    global.c0 = f3{ r3.x, r3.y, r3.z }, global.c1 = f3{ r3.w, r4.x, r4.y }, global.c2 = f3{ r4.z, r4.w, r5.x }, global.c3 = f3{ r5.y, r5.z, r5.w };
    m33 nmat;
    nmat.c0 = f3{ r6.x, r6.y, r6.z }, nmat.c1 = f3{ r6.w, r7.x, r7.y }, nmat.c2 = f3{ r7.z, r7.w, r8.x };
    const int shape_id = (int)igm_bits(r8.y);

    const uint8_t* base = sc.shape_data + sc.shape_offsets[shape_id];
    const int4 hdr      = *reinterpret_cast<const int4*>(base); // faces, vertices, normals, texcoords
    const float* f      = reinterpret_cast<const float*>(base);
    const float* verts  = f + 12;
    const float* norms  = verts + hdr.y * 4;
    const int4 tri      = *reinterpret_cast<const int4*>(norms + hdr.z * 4 + prim_id * 4);

    const f3 v0 = xform_point(global, ld3v(verts + tri.x * 4));
    const f3 v1 = xform_point(global, ld3v(verts + tri.y * 4));
    const f3 v2 = xform_point(global, ld3v(verts + tri.z * 4));
    const f3 e1 = v2 - v0, e2 = v0 - v1, e3 = v1 - v2;
    const f3 n  = stable_normal(e1, e2, e3); // make_triangle, core/triangle.art:12-29
    const float nn = len3(n);
    const f3 fn    = n * (1 / nn);

    const f3 n0 = ld3v(norms + tri.x * 4), n1 = ld3v(norms + tri.y * 4), n2 = ld3v(norms + tri.z * 4);
    const f3 ln = f3{ lerp2(n0.x, n1.x, n2.x, u, v), lerp2(n0.y, n1.y, n2.y, u, v), lerp2(n0.z, n1.z, n2.z, u, v) };
    const f3 sn = normalize3(mul33(nmat, ln));

    Surf s;
    s.entering    = dot3(dir, fn) <= 0;
    s.point       = org + dir * t;
    s.face_normal = s.entering ? fn : -fn;
    s.local       = orthonormal_basis(s.entering ? sn : -sn);
    return s;
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
        ex             = xa * (1 / width);
        ey             = ya * (1 / height);
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
    const float eta2   = cos_i < 0 ? 1 / eta : eta;
    const float cos2_t = 1 - (1 - cos_i * cos_i) * eta2 * eta2; // snell
    if (cos2_t <= 0.0f)
        return false;
    const float ct = igm_sqrt(cos2_t);
    cos_t          = cos_i < 0 ? -ct : ct;
    factor         = fresnel_factor(eta2, igm_abs(cos_i), ct);
    return true;
}

// ---------------------------------------------------------------- one path vertex

struct PathVertexIn {
    int ray_id;
    f3 org, dir;
    uint32_t rnd;
    float inv_pdf;
    Col contrib;
    int depth;
    float eta;
    // hit (ent < 0: miss)
    int ent, prim;
    float t, u, v;
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
    uint32_t b_rnd;
    float b_inv_pdf, b_eta;
    Col b_contrib;
    int b_depth;
};

constexpr float kRayOffset = 0.001f; // technique/pathtracer.art:41

IG_DEV Col clamp_color(const ig_technique& tech, Col c) // handle_color, technique/pathtracer.art:46-50
{
    if (tech.clamp > 0)
        return Col{ igm_min(c.r, tech.clamp), igm_min(c.g, tech.clamp), igm_min(c.b, tech.clamp) };
    return c;
}

// gpu_hit_shade / gpu_miss_shade body (driver/mapping_gpu.art:123-274) with the path tracer
// callbacks on_hit / on_shadow / on_bounce / on_miss (technique/pathtracer.art:52-210).
IG_DEV void shade_vertex(const DevScene& sc, const ShadeFrame& fr, const PathVertexIn& in, PathVertexOut& out)
{
    out.has_radiance = false;
    out.shadow       = false;
    out.bounce       = false;
    out.radiance     = Col{ 0, 0, 0 };

    const ig_technique tech = sc.tech;
    // make_uniform_light_selector (light/light_selector.art:26-46)
    const float sel_pdf = sc.light_count == 0 ? 1.0f : 1 / (float)sc.light_count;
    const bool nee      = tech.nee != 0;

    if (in.ent < 0) {
        // ---- miss: on_miss (technique/pathtracer.art:141-168) over infinite, non-delta lights
        Col sum{ 0, 0, 0 };
        for (uint32_t li = 0; li < sc.infinite_light_count; ++li) {
            const ig_light& L = sc.lights[li];
            if (L.type != IG_LIGHT_ENV)
                continue;
            const float pdf_s = 1 / (4 * kPi);
            const float mis   = nee ? 1 / (1 + in.inv_pdf * sel_pdf * pdf_s) : 1.0f;
            const Col c       = clamp_color(tech, (in.contrib * Col{ L.d[0], L.d[1], L.d[2] }) * mis);
            sum               = Col{ sum.r + c.r, sum.g + c.g, sum.b + c.b };
        }
        out.has_radiance = true;
        out.radiance     = sum;
        return;
    }

    const Surf surf        = surface_element(sc, in.ent, in.prim, in.org, in.dir, in.t, in.u, in.v);
    const ig_material& mat = sc.materials[sc.entity_material[in.ent]];
    const f3 N             = surf.local.c2;
    const f3 out_dir       = -in.dir;

    // RNG resumes where the path left off (mapping_gpu.art:171)
    const int sample = in.ray_id % fr.spi;
    const int lpix   = in.ray_id / fr.spi;
    const int px     = lpix % fr.width;
    const int py     = fr.row_offset + (lpix / fr.width) * fr.row_stride;
    Tea rnd{ make_seed(sample, fr.iteration, fr.frame, px, py, fr.seed), in.rnd };

    // ---- on_hit (technique/pathtracer.art:119-139): emission with MIS
    if (mat.light_id >= 0 && surf.entering) {
        const float dcos = -dot3(in.dir, N);
        if (dcos > kFltEps) {
            const PlaneLight pl(sc.lights[mat.light_id]);
            const float pdf_s = pl.pdf(in.org);
            const float mis   = nee ? 1 / (1 + in.inv_pdf * sel_pdf * pdf_s) : 1.0f;
            out.has_radiance  = true;
            out.radiance      = clamp_color(tech, (in.contrib * pl.radiance) * mis);
        }
    }

    const bool is_delta = mat.bsdf_type == IG_BSDF_DIELECTRIC;
    const Col kd        = Col{ mat.p[0], mat.p[1], mat.p[2] };

    // ---- on_shadow (technique/pathtracer.art:52-117): next event estimation
    if (nee && !is_delta && sc.light_count != 0 && in.depth + 1 <= tech.max_depth) {
        const int lid     = sc.light_count <= 1 ? 0 : rnd.range(0, (int)sc.light_count - 1); // pick_light_id
        const ig_light& L = sc.lights[lid];
        f3 lpos{}, ldir{};
        Col lint{ 0, 0, 0 };
        float pdf_value = 0, lcos = 0, ldist = 0;
        bool pdf_area = false, delta = false, usable = true;
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
        } else {
            usable = false;
        }
        if (usable) {
            const float dist2   = ldist * ldist;
            const float pdf_l_s = (pdf_area ? pdf_value * dist2 / lcos : pdf_value) * sel_pdf; // driver/pdf.art:19-38
            if (pdf_l_s > kFltEps && lcos > kFltEps) {
                float mis = 1;
                if (!delta) {
                    const float pdf_e_s = pos_cos(ldir, N) / kPi; // lambertian pdf (bsdf/diffuse.art:4)
                    mis                 = 1 / (1 + pdf_e_s / pdf_l_s);
                }
                const float factor = pdf_value / pdf_l_s;
                const Col ev       = kd * (pos_cos(ldir, N) * kInvPi); // lambertian eval (bsdf/diffuse.art:3)
                const Col c        = clamp_color(tech, (lint * (in.contrib * ev)) * (mis * factor));
                if ((c.r + c.g + c.b) / 3 > kFltEps) {
                    out.shadow = true;
                    out.s_org  = surf.point;
                    out.s_dir  = lpos - surf.point;
                    out.s_tmax = 1 - kRayOffset;
                    out.s_col  = c;
                }
            }
        }
    }

    // ---- on_bounce (technique/pathtracer.art:170-210)
    if (in.depth + 1 <= tech.max_depth) {
        f3 in_dir;
        float pdf, s_eta;
        Col color;
        bool sdelta;
        if (!is_delta) {
            // make_lambertian_bsdf.sample (bsdf/diffuse.art:5-9), sample_cosine_hemisphere (core/sampling.art:62-70)
            const float u   = rnd.f32();
            const float v   = rnd.f32();
            const float c   = safe_sqrt(v);
            const float s   = safe_sqrt(1 - v);
            const float phi = 2 * kPi * u;
            in_dir          = mul33(surf.local, f3{ s * igm_cos(phi), s * igm_sin(phi), c });
            pdf             = c / kPi;
            color           = kd;
            s_eta           = 1;
            sdelta          = false;
        } else {
            // make_pure_dielectric_bsdf.sample (bsdf/dielectric.art:18-34); n1 = ext_ior, n2 = int_ior
            const float n1 = mat.p[0], n2 = mat.p[1];
            const float k     = surf.entering ? n1 / n2 : n2 / n1;
            const float cos_o = dot3(out_dir, N);
            float cos_t = 0, F = 1;
            if (!fresnel(k, cos_o, cos_t, F)) {
                cos_t = 0;
                F     = 1;
            }
            if (rnd.f32() > F) {
                in_dir = N * (k * cos_o - cos_t) - out_dir * k; // vec3_refract (core/vector.art:126)
                color  = Col{ mat.p[5], mat.p[6], mat.p[7] } * 1.0f;
                s_eta  = k;
            } else {
                in_dir = N * (2 * dot3(N, out_dir)) - out_dir; // vec3_reflect (core/vector.art:123)
                color  = Col{ mat.p[2], mat.p[3], mat.p[4] };
                s_eta  = 1;
            }
            pdf    = 1;
            sdelta = true;
        }
        if (pdf > kFltEps) {
            const Col nc        = in.contrib * color;
            const float e2      = in.eta * in.eta;
            const float rr_prob = (in.depth + 1 > tech.min_depth) ? clampf(igm_max(nc.r * e2, igm_max(nc.g * e2, nc.b * e2)), 0.05f, 0.95f) : 1.0f;
            if (!(rnd.f32() >= rr_prob)) {
                out.bounce    = true;
                out.b_org     = surf.point;
                out.b_dir     = in_dir;
                out.b_rnd     = rnd.counter;
                out.b_inv_pdf = sdelta ? 0 : 1 / pdf;
                out.b_contrib = nc * (1 / rr_prob);
                out.b_depth   = in.depth + 1;
                out.b_eta     = in.eta * s_eta;
            }
        }
    }
}

} // namespace igdev
