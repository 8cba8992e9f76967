// Light tracer (src/artic/technique/lighttracer.art) on the wavefront pipeline: k_generate_light starts one path per (pixel,
// sample) index on a light (make_lt_emitter, :35-62); the shading kernel connects every non-delta vertex to the pinhole camera
// (on_shadow, :75-113) and continues the path with an adjoint BSDF sample (on_bounce, :123-163); the any-hit traversal adds the
// unoccluded connections into the pixel the vertex projects to (on_advanced_shadow_miss, :116-120). The reference needs its
// "advanced shadow" kernels (driver/mapping_gpu.art:293-333) for that, because only they hand a callback the secondary payload
// with the pixel; here the shadow ray simply carries the accumulator slot of that pixel instead of its own.
#pragma once

#include "shade_core.h"

namespace igdev {

struct LtCamera {
    f3 eye;
    m33 view;
    float sx, sy;
    int width, height;
    IG_DEV explicit LtCamera(const LtCameraArgs& a)
        : eye(f3{ a.eye[0], a.eye[1], a.eye[2] })
        , sx(a.sx)
        , sy(a.sy)
        , width(a.width)
        , height(a.height)
    {
        view.c0 = f3{ a.view[0], a.view[1], a.view[2] }, view.c1 = f3{ a.view[3], a.view[4], a.view[5] }, view.c2 = f3{ a.view[6], a.view[7], a.view[8] };
    }
};

struct EmissionSample { // make_emission_sample: what the light tracer reads of it
    f3 pos, dir;
    Col intensity;
    float cos;
};

// env_sample_pos (light/env.art:2-6)
IG_DEV f3 env_sample_pos(const DevScene& sc, Tea& rnd, f3 dir, float& pdf)
{
    const float r = sc.scene_radius;
    const float u = rnd.f32();
    const float v = rnd.f32();
    const f2 d    = concentric_disk(u, v); // sample_uniform_disk (core/sampling.art:101-103)
    pdf           = 1 / (kPi * r * r);
    return f3{ sc.scene_center[0], sc.scene_center[1], sc.scene_center[2] } + (dir * r + mul33(orthonormal_basis(dir), f3{ d.x * r, d.y * r, 0 }));
}

// Light::sample_emission of every light type the loader lowers
IG_DEV bool sample_emission(const DevScene& sc, const ig_light& L, Tea& rnd, EmissionSample& e)
{
    switch (L.type) {
    case IG_LIGHT_POINT: { // light/point.art:9-12, sample_uniform_sphere (core/sampling.art:42-47)
        const float u = rnd.f32();
        const float v = rnd.f32();
        const float c = 2 * v - 1, sn = safe_sqrt(1 - c * c), phi = 2 * kPi * u;
        const float pdf = 1 / (4 * kPi);
        e.pos           = f3{ L.d[0], L.d[1], L.d[2] };
        e.dir           = f3{ sn * igm_cos(phi), sn * igm_sin(phi), c };
        e.intensity     = Col{ L.d[4], L.d[5], L.d[6] } * (1 / pdf);
        e.cos           = 1;
        return true;
    }
    case IG_LIGHT_SPOT: { // light/spot.art:8-47, sample_uniform_cone (core/sampling.art:109-116)
        const f3 dir               = f3{ L.d[4], L.d[5], L.d[6] };
        const float cosCutoffAngle = L.d[3], cosFalloffAngle = L.d[7];
        const float blendRange  = cosFalloffAngle - cosCutoffAngle;
        const float spot_radius = igm_sqrt(1 - cosCutoffAngle * cosCutoffAngle) / cosCutoffAngle;
        const float spot_area   = kPi * spot_radius * spot_radius;
        const float u  = rnd.f32();
        const float v  = rnd.f32();
        const float c1 = 1 - cosCutoffAngle;
        const f2 p     = concentric_disk(u, v);
        const float n2 = p.x * p.x + p.y * p.y;
        const float z  = cosCutoffAngle + c1 * (1 - n2);
        const float k  = safe_sqrt(c1 * (2 - c1 * n2));
        const float pdf = safe_div(1, 2 * kPi * (1 - cosCutoffAngle));
        const f3 out_dir      = mul33(orthonormal_basis(dir), f3{ p.x * k, p.y * k, z });
        const float cos_angle = dot3(out_dir, dir);
        float factor;
        if (blendRange <= kFltEps) {
            factor = cos_angle <= cosCutoffAngle ? 0.0f : 1.0f;
        } else {
            const float x = clampf((cos_angle - cosCutoffAngle) / blendRange, 0, 1);
            factor        = x * x * (3 - 2 * x);
        }
        e.pos       = f3{ L.d[0], L.d[1], L.d[2] };
        e.dir       = out_dir;
        e.intensity = (Col{ L.d[8], L.d[9], L.d[10] } * factor) * (1 / (spot_area * pdf));
        e.cos       = z;
        return true;
    }
    case IG_LIGHT_PLANE:
    case IG_LIGHT_SPHERE:
    case IG_LIGHT_MESH_AREA: { // make_area_light.sample_emission (light/area.art:26-37)
        const float u0 = rnd.f32();
        const float u1 = rnd.f32();
        f3 point, normal;
        float area_pdf;
        Col radiance;
        if (L.type == IG_LIGHT_SPHERE) { // make_sphere_area_emitter.sample_emission (area.art:296-299)
            const SphereEmitter se(sc, L);
            se.surface(square_to_sphere(u0, u1), point, normal);
            area_pdf = safe_div(1, se.area);
            radiance = se.radiance;
        } else if (L.type == IG_LIGHT_PLANE) { // make_plane_area_emitter.sample (area.art:230-250)
            const float* d = L.d;
            point    = (f3{ d[4], d[5], d[6] } * u0 + f3{ d[8], d[9], d[10] } * u1) + f3{ d[0], d[1], d[2] };
            normal   = f3{ d[3], d[7], d[11] };
            area_pdf = safe_div(1, d[23]);
            radiance = Col{ d[20], d[21], d[22] };
        } else { // make_shape_area_emitter.sample (area.art:62-72) over shape.surface_element_for_point (shapes/trimesh.art:41-68)
            const MeshEmitter me(sc, L);
            int f;
            float bu, bv, area;
            f3 fn;
            me.address(u0, u1, f, bu, bv);
            me.surface(f, bu, bv, point, fn, area);
            const float4* e4 = reinterpret_cast<const float4*>(sc.entities + (size_t)L.entity_id * IG_ENTITY_FLOATS);
            const float4 r6 = e4[6], r7 = e4[7], r8 = e4[8];
            m33 nmat;
            nmat.c0 = f3{ r6.x, r6.y, r6.z }, nmat.c1 = f3{ r6.w, r7.x, r7.y }, nmat.c2 = f3{ r7.z, r7.w, r8.x };
            const float* norms = reinterpret_cast<const float*>(sc.shape_data + sc.entity_ext[L.entity_id].y);
            const int4 tri     = *reinterpret_cast<const int4*>(me.inds + f * 4);
            const f3 n0 = ld3v(norms + tri.x * 4), n1 = ld3v(norms + tri.y * 4), n2 = ld3v(norms + tri.z * 4);
            normal   = normalize3(mul33(nmat, f3{ lerp2(n0.x, n1.x, n2.x, bu, bv), lerp2(n0.y, n1.y, n2.y, bu, bv), lerp2(n0.z, n1.z, n2.z, bu, bv) }));
            area_pdf = safe_div(1, area) / (float)me.num_tris;
            radiance = me.radiance;
        }
        float cpdf;
        const f3 d         = Principled::cosine_hemisphere(rnd, cpdf); // sample_cosine_hemisphere (core/sampling.art:62-70)
        const float weight = safe_div(1, area_pdf * cpdf);
        e.pos       = point;
        e.dir       = mul33(orthonormal_basis(normal), d);
        e.intensity = radiance * weight;
        e.cos       = d.z;
        return true;
    }
    case IG_LIGHT_DIRECTIONAL: { // light/directional.art:7-10
        const f3 dir = f3{ L.d[0], L.d[1], L.d[2] };
        float pos_pdf;
        e.pos       = env_sample_pos(sc, rnd, -dir, pos_pdf);
        e.dir       = dir;
        e.intensity = Col{ L.d[4], L.d[5], L.d[6] } * safe_div(1, pos_pdf);
        e.cos       = 1;
        return true;
    }
    case IG_LIGHT_SUN: { // make_sun_light.sample_emission (light/sun.art:24-29)
        const f3 sun_dir  = f3{ L.d[0], L.d[1], L.d[2] };
        const float cos_a = L.d[3];
        const float u  = rnd.f32();
        const float v  = rnd.f32();
        const float c1 = 1 - cos_a;
        const f2 p     = concentric_disk(u, v);
        const float n2 = p.x * p.x + p.y * p.y;
        const float z  = cos_a + c1 * (1 - n2);
        const float k  = safe_sqrt(c1 * (2 - c1 * n2));
        const f3 ndir  = mul33(orthonormal_basis(-sun_dir), f3{ p.x * k, p.y * k, z });
        const float inv_pdf = 2 * kPi * (1 - cos_a);
        float pos_pdf;
        e.pos       = env_sample_pos(sc, rnd, -ndir, pos_pdf);
        e.dir       = ndir;
        e.intensity = Col{ L.d[4], L.d[5], L.d[6] } * safe_div(inv_pdf, pos_pdf);
        e.cos       = z;
        return true;
    }
    case IG_LIGHT_ENV: { // make_environment_light_function_spherical.sample_emission (light/env.art:87-93), constant colour
        const float u   = rnd.f32();
        const float v   = rnd.f32();
        const f3 dir    = square_to_sphere(u, v);
        const float pdf = 1 / (4 * kPi);
        float pos_pdf;
        e.pos       = env_sample_pos(sc, rnd, dir, pos_pdf);
        e.dir       = -dir;
        e.intensity = Col{ L.d[0], L.d[1], L.d[2] } * safe_div(1, pos_pdf * pdf);
        e.cos       = 1.0f;
        return true;
    }
    case IG_LIGHT_ENV_TEXTURED: { // make_environment_light_textured.sample_emission (light/env.art:141-145); "cdf": "none" = the spherical function environment (:87-93)
        const TexturedEnv env(sc, L);
        f3 dir;
        Col intensity;
        float pdf_dir, pos_pdf;
        env.sample_dir(rnd, dir, intensity, pdf_dir);
        e.pos       = env_sample_pos(sc, rnd, dir, pos_pdf);
        e.dir       = -dir;
        e.intensity = intensity * safe_div(1, pos_pdf * pdf_dir);
        e.cos       = 1.0f;
        return true;
    }
    case IG_LIGHT_CIE: { // make_environment_light_function_{hemi, spherical}.sample_emission (light/env.art:38-46,87-93) over the sky function
        const CieSky sky(L);
        const float ux = rnd.f32();
        const float uy = rnd.f32();
        f3 gdir;
        Col intensity;
        float pdf;
        if (!sky.has_ground) {
            const float c   = safe_sqrt(uy); // sample_cosine_hemisphere (core/sampling.art:62-70)
            const float sn  = safe_sqrt(1 - uy);
            const float phi = 2 * kPi * ux;
            const f3 dir    = switch_env_up(f3{ sn * igm_cos(phi), sn * igm_sin(phi), c });
            pdf             = c / kPi;
            intensity       = sky.radiance(dir);
            gdir            = f3{ dot3(sky.transform.c0, dir), dot3(sky.transform.c1, dir), dot3(sky.transform.c2, dir) }; // mat3x3_left_mul
        } else {
            gdir      = square_to_sphere(ux, uy);
            pdf       = 1 / (4 * kPi);
            intensity = sky.radiance(mul33(sky.transform, gdir));
        }
        float pos_pdf;
        e.pos       = env_sample_pos(sc, rnd, gdir, pos_pdf);
        e.dir       = -gdir;
        e.intensity = intensity * safe_div(1, pdf * pos_pdf);
        e.cos       = 1.0f;
        return true;
    }
    case IG_LIGHT_PEREZ: { // make_perez_light_raw.sample_emission (light/perez.art:309-313): the sun's sample (sun.art:24-29) plus the sky seen against it
        const f3 sun_dir    = f3{ L.d[27], L.d[28], L.d[29] };
        const float cos_a   = L.d[14];
        const float u       = rnd.f32();
        const float v       = rnd.f32();
        const float c1      = 1 - cos_a;
        const f2 p          = concentric_disk(u, v);
        const float n2      = p.x * p.x + p.y * p.y;
        const float z       = cos_a + c1 * (1 - n2);
        const float k       = safe_sqrt(c1 * (2 - c1 * n2));
        const f3 ndir       = mul33(orthonormal_basis(-sun_dir), f3{ p.x * k, p.y * k, z });
        const float inv_pdf = 2 * kPi * (1 - cos_a);
        const float dir_pdf = safe_div(1, 2 * kPi * (1 - cos_a)); // uniform_cone_pdf
        float pos_pdf;
        e.pos = env_sample_pos(sc, rnd, -ndir, pos_pdf);
        e.dir = ndir;
        const CieSky sky(L);
        const f3 to_sky = -ndir;
        const f3 d      = f3{ dot3(sky.transform.c0, to_sky), dot3(sky.transform.c1, to_sky), dot3(sky.transform.c2, to_sky) };
        e.intensity     = Col{ L.d[24], L.d[25], L.d[26] } * safe_div(inv_pdf, pos_pdf) + sky.radiance(d) * (1 / (pos_pdf * dir_pdf));
        e.cos           = z;
        return true;
    }
    default:
        return false;
    }
}

// shading_normal_adjoint (bsdf/map.art:1-7)
IG_DEV float shading_normal_adjoint(f3 in_dir, f3 out_dir, f3 ns, f3 ng)
{
    const float ons = pos_cos(out_dir, ns), ins = pos_cos(in_dir, ns);
    const float ong = pos_cos(out_dir, ng), ing = pos_cos(in_dir, ng);
    return (ins <= kFltEps || ong <= kFltEps) ? 0.0f : (ons / ins) * (ing / ong);
}

// on_shadow / on_bounce of make_lt_renderer; s_slot: the accumulator slot (a ray id) of the pixel the connection lands in
IG_DEV void shade_vertex_lt(const DevScene& sc, const ShadeFrame& fr, const LtCamera& cam, const PathVertexIn& in, PathVertexOut& out, int& s_slot)
{
    out.has_radiance = false; // TechniqueNoHitFunction / TechniqueNoMissFunction
    out.shadow       = false;
    out.bounce       = false;
    out.radiance     = Col{ 0, 0, 0 };
    if (in.ent < 0)
        return;
    const ig_technique tech = sc.tech;
    const int depth         = in.depth & 0xFFFF;

    const Surf surf = surface_element<true>(sc, in.ent, in.prim, in.org, in.dir, in.t, in.u, in.v);
    ig_material mat_local;
    const ig_material& mat = resolve_material<true>(sc, sc.materials[sc.entity_material[in.ent]], surf, -in.dir, mat_local);
    const BsdfCtx<true, true, true> bsdf(sc, mat, surf, in.dir, std::true_type{});
    const f3 N       = surf.local.c2;
    const f3 out_dir = -in.dir;

    int it_l, sample, px, py;
    fr.decompose(in.ray_id, it_l, sample, px, py);
    Tea rnd{ make_seed(sample, fr.iteration + it_l, fr.frame, px, py, fr.seed), in.rnd };

    // ---- on_shadow (lighttracer.art:75-113): camera.sample_pixel of make_perspective_camera (camera/perspective.art:16-26,43-57)
    if (!bsdf.all_delta() && depth + 1 <= tech.max_depth) {
        const f3 d     = surf.point - cam.eye;
        const f3 un    = f3{ dot3(cam.view.c0, d), dot3(cam.view.c1, d), dot3(cam.view.c2, d) }; // mat3x3_left_mul
        const float nx = un.x / (un.z * cam.sx);
        const float ny = un.y / (un.z * cam.sy);
        if (nx >= -1 && nx <= 1 && ny >= -1 && ny <= 1) {
            const f3 cam_dir  = cam.eye - surf.point;
            const f3 in_dir   = normalize3(cam_dir);
            const float cos_o = dot3(out_dir, N);
            const float cos_i = dot3(in_dir, N);
            if (cos_o * cos_i > kFltEps) {
                const float d2     = dot3(cam_dir, cam_dir);
                const float factor = safe_div(cos_i, cos_o * d2);
                out.shadow = true;
                out.s_org  = surf.point;
                out.s_dir  = cam_dir;
                out.s_tmax = 1 - kRayOffset;
                out.s_col  = clamp_color(tech, (in.contrib * bsdf.eval(out_dir, in_dir)) * factor); // camera_sample.weight = 1 (image_area = 1)
                // make_pixelcoord_from_normalized (driver/camera.art:45-57)
                const int x = min((int)igm_floor((float)cam.width * (nx + 1) / 2), cam.width - 1);
                const int y = min((int)igm_floor((float)cam.height * (1 - ny) / 2), cam.height - 1);
                s_slot      = it_l * fr.rays_per_iteration + (y * cam.width + x) * fr.spi + sample;
            }
        }
    }

    // ---- on_bounce (lighttracer.art:123-163): the path tracer's with adjoint = true
    if (depth + 1 <= tech.max_depth) {
        f3 in_dir;
        float pdf, s_eta;
        Col color;
        bool sdelta;
        if (bsdf.sample(rnd, out_dir, in_dir, pdf, color, s_eta, sdelta, true) && pdf > kFltEps) {
            if (mat.flags & (IG_MAT_BUMP | IG_MAT_NORMALMAP | IG_MAT_EXPR_NORMAL)) {
                // transform_surf_bsdf.sample with adjoint (bsdf/map.art:19-30), inside the two-sided wrapper if there is one
                const f3 li = bsdf.ds_flip ? -in_dir : in_dir, lo = bsdf.ds_flip ? -out_dir : out_dir;
                color       = color * shading_normal_adjoint(li, lo, bsdf.surf.local.c2, N);
            }
            const Col nc        = in.contrib * color;
            const float e2      = in.eta * in.eta;
            const float rr_prob = (depth + 1 > tech.min_depth) ? clampf(igm_max(nc.r * e2, igm_max(nc.g * e2, nc.b * e2)), 0.05f, 0.95f) : 1.0f;
            if (!(rnd.f32() >= rr_prob)) {
                out.bounce    = true;
                out.b_org     = surf.point;
                out.b_dir     = in_dir;
                out.b_tmin    = kRayOffset;
                out.b_rnd     = rnd.counter;
                out.b_inv_pdf = sdelta ? 0.0f : 1 / pdf; // (not part of LTRayPayload; the kernel bins continuation rays by it)
                out.b_contrib = nc * (1 / rr_prob);
                out.b_depth   = depth + 1;
                out.b_eta     = in.eta * s_eta;
            }
        }
    }
}

} // namespace igdev
