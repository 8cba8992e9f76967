/* ig_photon.h — the photon record and the voxel grid of the photon mapper (src/artic/technique/photonmapper.art), shared by the
 * HIP kernels and the test oracle like ig_detmath.h: bit-level encodings and integer arithmetic, no rendering algorithm.
 *
 *   Photon (photonmapper.art:1-39): 32 bytes = int4 {octahedron-projected direction, light id, RGBE power, depth} +
 *   float4 {position, eta}. The encodings are lossy and part of the result: encode_normal_32 / decode_normal_32
 *   (core/common.art:156-204), encode_rgbe / decode_rgbe (core/color.art:341-368) over frexp (core/common.art:96-112).
 *   As written, encode_signed_norm_16 scales by (1 << 16) - 1 = 65535 and narrows to i16, so a component beyond +-0.5 leaves
 *   the range of the type; the conversion is restated as every code generator here performs it (f32 -> i32, low 16 bits).
 *
 *   Grid (photonmapper.art:395-431): 128^3 cells over the scene's bounding box, linearised by a 30-bit Morton code.
 */
#ifndef IG_PHOTON_H
#define IG_PHOTON_H

#include "ig_detmath.h"

#define IGP_GRID_SIZE 128
#define IGP_GRID_CELLS (IGP_GRID_SIZE * IGP_GRID_SIZE * IGP_GRID_SIZE)

typedef struct igp_photon {
    int32_t dir;   /* encode_normal_32(in_dir): the direction the photon came from */
    int32_t light; /* light id; -1: this light path left no photon */
    int32_t power; /* encode_rgbe(radiance) */
    int32_t depth; /* vertices on the light path so far, 1 = first surface hit */
    float pos[3];
    float eta;
} igp_photon;

IGM_FN float igp_round(float v) { return __builtin_roundf(v); } /* math_builtins::round: half away from zero */
IGM_FN float igp_clampf(float v, float l, float u) { return v < l ? l : (v > u ? u : v); }        /* core/common.art:261 (select form) */
IGM_FN float igp_safe_div(float a, float b) { return igm_abs(b) <= IGM_FLT_EPS ? 0.0f : a / b; } /* core/common.art:263 */

/* encode_signed_norm_16 / decode_signed_norm_16 (core/common.art:186-190) */
IGM_FN int32_t igp_encode_snorm16(float f) { return (int32_t)(int16_t)(uint16_t)((uint32_t)(int32_t)igp_round(f * 65535.0f) & 0xFFFFu); }
IGM_FN float igp_decode_snorm16(int32_t v16) { return igp_clampf((float)v16 / 65535.0f, -1.0f, 1.0f); }

/* encode_normal_32 (core/common.art:192-197) over encode_oct_proj (:155-169) */
IGM_FN int32_t igp_encode_normal_32(float x, float y, float z)
{
    const float a  = igm_abs(x) + igm_abs(y) + igm_abs(z);
    const float ox = x / a, oy = y / a;
    float px = ox, py = oy;
    if (z < 0) {
        px = (1 - igm_abs(oy)) * (ox >= 0 ? 1.0f : -1.0f);
        py = (1 - igm_abs(ox)) * (oy >= 0 ? 1.0f : -1.0f);
    }
    const int32_t ex = igp_encode_snorm16(px), ey = igp_encode_snorm16(py);
    return (int32_t)(((uint32_t)ex << 16) | (uint32_t)ey); /* (ex as i32 << 16) | (ey as i32): a negative ey sets the upper half too */
}

/* decode_normal_32 (:199-203) over decode_oct_proj (:171-182; the reflected branch as written) */
IGM_FN void igp_decode_normal_32(int32_t val, float out[3])
{
    const float dx = igp_decode_snorm16((int32_t)(int16_t)(val >> 16));
    const float dy = igp_decode_snorm16((int32_t)(int16_t)val);
    const float oz = 1 - igm_abs(dx) - igm_abs(dy);
    float ox = dx, oy = dy;
    if (oz < 0) {
        ox = 1 - igm_abs(dy) * (dx >= 0 ? 1.0f : -1.0f);
        oy = 1 - igm_abs(dx) * (dy >= 0 ? 1.0f : -1.0f);
    }
    /* vec3_normalize (core/vector.art:138): v * (1 / sqrt(dot)), dot as the fma chain of vec3_dot */
    const float inv = 1 / igm_sqrt(igm_fma(ox, ox, igm_fma(oy, oy, oz * oz)));
    out[0] = ox * inv, out[1] = oy * inv, out[2] = oz * inv;
}

/* encode_rgbe (core/color.art:341-353); frexp of a positive normal or denormal float (core/common.art:96-112) */
IGM_FN int32_t igp_encode_rgbe(float r, float g, float b)
{
    const float mx = r > g ? (r > b ? r : b) : (g > b ? g : b); /* color_max_component */
    if (mx <= IGM_FLT_EPS)
        return 0;
    const uint32_t bits = igm_bits(mx) & 0x7FFFFFFFu;
    float val;
    int32_t ex;
    if (bits >= 0x7F800000u) {
        val = mx, ex = 0;
    } else {
        val = igm_float((igm_bits(mx) & 0x807FFFFFu) | 0x3F000000u);
        ex  = (int32_t)(bits >> 23) - 126;
    }
    const float val2 = val * 256 / mx;
    const int32_t ri = (int32_t)(r * val2) & 0xFF, gi = (int32_t)(g * val2) & 0xFF, bi = (int32_t)(b * val2) & 0xFF;
    return (int32_t)(((uint32_t)ri << 24) | ((uint32_t)gi << 16) | ((uint32_t)bi << 8)) | (ex + 128);
}

/* decode_rgbe (core/color.art:355-367): ldexp(1, e - 136) as an exponent field (e - 136 lies within the normal range for every
 * power a light path can carry; below it the value is a denormal of the same magnitude) */
IGM_FN void igp_decode_rgbe(int32_t c, float out[3])
{
    if (c == 0) {
        out[0] = out[1] = out[2] = 0;
        return;
    }
    const float r = (float)((c >> 24) & 0xFF), g = (float)((c >> 16) & 0xFF), b = (float)((c >> 8) & 0xFF);
    const int32_t e = (c & 0xFF) - 128 - 8;
    const float f   = e >= -126 ? igm_float((uint32_t)(e + 127) << 23) : (e >= -149 ? igm_float(1u << (uint32_t)(e + 149)) : 0.0f);
    out[0] = r * f, out[1] = g * f, out[2] = b * f;
}

/* ppm_expand_bits / morton_3d (photonmapper.art:395-412) */
IGM_FN uint32_t igp_expand_bits(uint32_t v)
{
    uint32_t x = v & 0x000003ffu;
    x = (x ^ (x << 16)) & 0xff0000ffu;
    x = (x ^ (x << 8)) & 0x0300f00fu;
    x = (x ^ (x << 4)) & 0x030c30c3u;
    x = (x ^ (x << 2)) & 0x09249249u;
    return x;
}
IGM_FN int32_t igp_morton_3d(int32_t x, int32_t y, int32_t z) { return (int32_t)(igp_expand_bits((uint32_t)x) + (igp_expand_bits((uint32_t)y) << 1) + (igp_expand_bits((uint32_t)z) << 2)); }

/* grid_scene_pos (photonmapper.art:414-422) */
IGM_FN void igp_grid_pos(const float pos[3], const float bmin[3], const float bmax[3], int32_t cell[3])
{
    for (int k = 0; k < 3; ++k) {
        const float n = igp_clampf(igp_safe_div(pos[k] - bmin[k], bmax[k] - bmin[k]) * 0.99f, 0.0f, 1.0f);
        const int32_t i = (int32_t)(n * (float)IGP_GRID_SIZE);
        cell[k]         = i < IGP_GRID_SIZE - 1 ? i : IGP_GRID_SIZE - 1;
    }
}
IGM_FN int32_t igp_grid_cell(const float pos[3], const float bmin[3], const float bmax[3])
{
    int32_t c[3];
    igp_grid_pos(pos, bmin, bmax, c);
    return igp_morton_3d(c[0], c[1], c[2]);
}

/* ppm_kernel, the Simpson kernel (photonmapper.art:44-48) */
IGM_FN float igp_kernel(float r2, float d2)
{
    const float ir2  = igp_safe_div(1.0f, r2);
    const float term = 1 - d2 * ir2;
    return term * term * 3 * ir2 * IGM_INV_PI;
}

/* ppm_compute_radius (photonmapper.art:251-259): the merge radius of iteration `iter` */
IGM_FN float igp_compute_radius(float max_radius, int32_t iter)
{
    float radius = max_radius;
    for (int32_t i = 0; i < iter; ++i)
        radius *= ((float)i + 1 + 0.8f) / (float)(i + 2);
    return radius > 1e-5f ? radius : 1e-5f;
}

#endif /* IG_PHOTON_H */
