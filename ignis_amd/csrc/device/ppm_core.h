// Photon mapper (src/artic/technique/photonmapper.art, PhotonMappingTechnique.cpp) on the wavefront pipeline. Two passes per
// iteration, each a wavefront of its own:
//   light pass  (make_ppm_light_emitter :141-165, make_ppm_light_renderer :169-245): one path per photon index from a light chosen by
//               the light selector; it walks through all-delta surfaces (adjoint BSDF samples) and leaves ONE photon at the first
//               other surface it meets — "our PPM implementation only handles direct (LDE) and caustic (LS*DE) paths";
//   camera pass (make_ppm_path_renderer :262-388): a path tracer without next-event estimation that, at every non-delta vertex,
//               gathers the photons within the merge radius (Simpson kernel, voxel grid :60-110) instead of sampling lights.
// The reference stores photons at atomically reserved slots and sorts them into the grid with atomics, so its photon order — and with
// it the float sums of a gather — depends on scheduling. Here a photon's slot is its light path's index and the grid holds them in
// (cell, index) order (photon.hip), so the image is a function of the seed like every other technique's.
#pragma once

#include "ig_photon.h"
#include "lt_core.h"

namespace igdev {

// payload (PPMRayPayload :113-139): contrib -> pay.yzw, depth -> low half of the depth word, path_type -> its high half, eta -> eta,
// radius_or_light -> the inv_pdf slot
IG_DEV float ppm_radius_of(const PpmArgs& pp, const PathVertexIn& in, int depth) // get_radius (:275-283)
{
    return depth > 1 ? in.inv_pdf : igm_min(pp.radius, in.t * 0.017455064f);
}

// light_cache.query (:67-108) with the body of on_hit's final gather (:314-330)
template <class Bsdf>
IG_DEV Col ppm_gather(const DevScene& sc, const PpmArgs& pp, const Bsdf& bsdf, f3 point, f3 N, f3 out_dir, float cos_o, float radius, int depth)
{
    Col total{ 0, 0, 0 };
    if (radius <= kFltEps || pp.valid_count == 0)
        return total;
    const float r2     = radius * radius;
    const float lo[3]  = { point.x - radius, point.y - radius, point.z - radius };
    const float hi[3]  = { point.x + radius, point.y + radius, point.z + radius };
    int32_t cmin[3], cmax[3];
    igp_grid_pos(lo, pp.bbox_min, pp.bbox_max, cmin);
    igp_grid_pos(hi, pp.bbox_min, pp.bbox_max, cmax);
    for (int iz = cmin[2]; iz <= cmax[2]; ++iz)
        for (int iy = cmin[1]; iy <= cmax[1]; ++iy)
            for (int ix = cmin[0]; ix <= cmax[0]; ++ix) {
                const int32_t cell = igp_morton_3d(ix, iy, iz);
                const uint32_t i0 = pp.cell_offset[cell], i1 = pp.cell_offset[cell + 1];
                Col cc{ 0, 0, 0 };
                for (uint32_t i = i0; i < i1; ++i) {
                    const int4 a   = reinterpret_cast<const int4*>(pp.photons)[2 * i];
                    const float4 b = reinterpret_cast<const float4*>(pp.photons)[2 * i + 1];
                    const f3 d     = point - f3{ b.x, b.y, b.z };
                    const float dist2 = dot3(d, d);
                    if (dist2 <= r2) {
                        float dir[3], pw[3];
                        igp_decode_normal_32(a.x, dir);
                        const f3 in_dir{ dir[0], dir[1], dir[2] };
                        const float cos_i = dot3(in_dir, N);
                        if (depth + a.w <= sc.tech.max_depth && cos_o * cos_i > kFltEps) {
                            igp_decode_rgbe(a.z, pw);
                            const float kf = igp_kernel(r2, dist2);
                            // the cosine of eval is divided out again: the projection is already in the photon's power
                            const Col c = (Col{ pw[0], pw[1], pw[2] } * bsdf.eval(in_dir, out_dir)) * safe_div(kf, igm_abs(cos_i));
                            cc          = cc + c;
                        }
                    }
                }
                total = total + cc;
            }
    return total;
}

// emission of an infinite, non-delta light towards -dir (Light::emission, the cases of shade_vertex's miss branch); false: delta light
IG_DEV bool ppm_infinite_emission(const DevScene& sc, const ig_light& L, f3 dir, Col& emit)
{
    if (L.type == IG_LIGHT_ENV) {
        emit = Col{ L.d[0], L.d[1], L.d[2] };
    } else if (L.type == IG_LIGHT_ENV_TEXTURED) {
        emit = TexturedEnv(sc, L).emission(dir);
    } else if (L.type == IG_LIGHT_CIE) {
        emit = CieSky(L).emission(dir);
    } else if (L.type == IG_LIGHT_SUN) {
        emit = dot3(f3{ L.d[0], L.d[1], L.d[2] }, dir) >= L.d[3] ? Col{ L.d[4], L.d[5], L.d[6] } : Col{ 0, 0, 0 };
    } else if (L.type == IG_LIGHT_PEREZ) {
        const CieSky sky(L);
        const bool hit = dot3(f3{ L.d[27], L.d[28], L.d[29] }, dir) >= L.d[14];
        const f3 d     = f3{ dot3(sky.transform.c0, dir), dot3(sky.transform.c1, dir), dot3(sky.transform.c2, dir) };
        emit           = (hit ? Col{ L.d[24], L.d[25], L.d[26] } : Col{ 0, 0, 0 }) + sky.radiance(d);
    } else {
        return false;
    }
    return true;
}

template <bool LIGHT_PASS>
IG_DEV void shade_vertex_ppm(const DevScene& sc, const ShadeFrame& fr, const PpmArgs& pp, const PathVertexIn& in, PathVertexOut& out)
{
    out.has_radiance = false;
    out.shadow       = false; // TechniqueNoShadowFunction in both passes
    out.bounce       = false;
    out.radiance     = Col{ 0, 0, 0 };
    const ig_technique tech = sc.tech;
    const int depth         = in.depth & 0xFFFF;
    const int path_type     = in.depth >> 16;

    if (in.ent < 0) {
        if (LIGHT_PASS || (in.dir.x == 0 && in.dir.y == 0 && in.dir.z == 0))
            return; // TechniqueNoMissFunction; the zero ray of a sample without a camera ray
        // on_miss (:333-360): to prevent double counting only paths without a diffuse bounce see the environment
        if (path_type == 1)
            return;
        int inflights = 0;
        Col color{ 0, 0, 0 };
        for (uint32_t li = 0; li < sc.infinite_light_count; ++li) {
            Col emit;
            if (!ppm_infinite_emission(sc, sc.lights[li], in.dir, emit))
                continue;
            ++inflights;
            color = color + in.contrib * emit;
        }
        if (inflights > 0) {
            out.has_radiance = true;
            out.radiance     = clamp_color(tech, color);
        }
        return;
    }

    const Surf surf = surface_element<true>(sc, in.ent, in.prim, in.org, in.dir, in.t, in.u, in.v);
    ig_material mat_local;
    const ig_material& mat = resolve_material<true>(sc, sc.materials[sc.entity_material[in.ent]], surf, -in.dir, mat_local);
    const BsdfCtx<true, true, true> bsdf(sc, mat, surf, in.dir, std::true_type{});
    const f3 N           = surf.local.c2;
    const f3 out_dir     = -in.dir;
    const bool emissive  = mat.light_id >= 0;
    const bool all_delta = bsdf.all_delta();

    int it_l, sample, px, row;
    fr.decompose(in.ray_id, it_l, sample, px, row);
    const int py     = fr.row_offset + row * fr.row_stride;
    const int within = in.ray_id - it_l * fr.rays_per_iteration; // the light path's index inside its iteration: its photon slot
    Tea rnd{ make_seed(sample, fr.iteration + it_l, fr.frame, px, py, fr.seed), in.rnd };

    if constexpr (LIGHT_PASS) {
        // ---- on_hit (:172-195): a photon on the first surface that is neither emissive nor all-delta
        if (!emissive && !all_delta) {
            const float cos_o = dot3(out_dir, N);
            if (cos_o > kFltEps) {
                int4* slot = reinterpret_cast<int4*>(pp.photons) + 2 * (size_t)within;
                slot[0]    = make_int4(igp_encode_normal_32(out_dir.x, out_dir.y, out_dir.z), (int)in.inv_pdf, igp_encode_rgbe(in.contrib.r, in.contrib.g, in.contrib.b), depth);
                reinterpret_cast<float4*>(slot)[1] = make_float4(surf.point.x, surf.point.y, surf.point.z, in.eta);
            }
        }
        // ---- on_bounce (:197-227): on through delta surfaces only
        if (all_delta && depth + 2 <= tech.max_light_depth) {
            f3 in_dir;
            float pdf, s_eta;
            Col color;
            bool sdelta;
            if (bsdf.sample(rnd, out_dir, in_dir, pdf, color, s_eta, sdelta, true)) {
                if (mat.flags & (IG_MAT_BUMP | IG_MAT_NORMALMAP | IG_MAT_EXPR_NORMAL)) {
                    const f3 li = bsdf.ds_flip ? -in_dir : in_dir, lo = bsdf.ds_flip ? -out_dir : out_dir;
                    color       = color * shading_normal_adjoint(li, lo, bsdf.surf.local.c2, N);
                }
                const Col nc = in.contrib * color;
                if (col_avg(nc) > kFltEps) {
                    out.bounce    = true;
                    out.b_org     = surf.point;
                    out.b_dir     = in_dir;
                    out.b_tmin    = kRayOffset;
                    out.b_rnd     = rnd.counter;
                    out.b_inv_pdf = in.inv_pdf; // radius_or_light: the light's id
                    out.b_contrib = nc;
                    out.b_depth   = (depth + 1) | (path_type << 16);
                    out.b_eta     = in.eta * s_eta;
                }
            }
        }
    } else {
        // ---- on_hit (:285-331)
        bool answered = false;
        if (path_type == 0 && emissive && surf.entering) { // light sources count on LS*E paths only
            const float dcos = dot3(out_dir, N);
            if (dcos > kFltEps) {
                const ig_light& EL = sc.lights[mat.light_id];
                Col emit;
                if (EL.type == IG_LIGHT_MESH_AREA)
                    emit = MeshEmitter(sc, EL).radiance;
                else if (EL.type == IG_LIGHT_SPHERE)
                    emit = Col{ EL.d[4], EL.d[5], EL.d[6] };
                else
                    emit = PlaneLight(EL).radiance;
                out.has_radiance = true;
                out.radiance     = clamp_color(tech, in.contrib * emit);
                answered         = true;
            }
        }
        const float radius = ppm_radius_of(pp, in, depth);
        if (!answered && depth + 1 <= tech.max_depth && !emissive && !all_delta) {
            const float cos_o = dot3(out_dir, N);
            if (igm_abs(cos_o) > kFltEps) {
                const Col g = ppm_gather(sc, pp, bsdf, surf.point, N, out_dir, cos_o, radius, depth);
                const float n = (float)pp.photon_count;
                out.has_radiance = true;
                out.radiance     = clamp_color(tech, in.contrib * Col{ g.r / n, g.g / n, g.b / n });
            }
        }
        // ---- on_bounce (:362-399): no next-event estimation, Russian roulette as in the path tracer
        if (depth + 1 <= tech.max_depth) {
            f3 in_dir;
            float pdf, s_eta;
            Col color;
            bool sdelta;
            if (bsdf.sample(rnd, out_dir, in_dir, pdf, color, s_eta, sdelta, false) && pdf > kFltEps) {
                const Col nc        = in.contrib * color;
                const float e2      = in.eta * in.eta;
                const float rr_prob = (depth + 1 > tech.min_depth) ? clampf(igm_max(nc.r * e2, igm_max(nc.g * e2, nc.b * e2)), 0.05f, 0.95f) : 1.0f;
                if (!(rnd.f32() >= rr_prob)) {
                    out.bounce    = true;
                    out.b_org     = surf.point;
                    out.b_dir     = in_dir;
                    out.b_tmin    = kRayOffset;
                    out.b_rnd     = rnd.counter;
                    out.b_inv_pdf = radius; // radius_or_light = get_radius(ctx.hit, pt)
                    out.b_contrib = nc * (1 / rr_prob);
                    out.b_depth   = (depth + 1) | ((sdelta ? path_type : 1) << 16);
                    out.b_eta     = in.eta * s_eta;
                }
            }
        }
    }
}

} // namespace igdev
