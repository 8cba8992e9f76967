/* ig_expr.h — register bytecode for the shading expressions of a scene (PExpr strings) and its interpreter.
 *
 * The reference transpiles every PExpr string of a scene ("reflectance": "mix(tex(uv), color(1,0,0), 0.5)") into Artic
 * source and JIT-compiles it into the shading kernel (src/runtime/loader/Transpiler.cpp:960-1230,
 * src/runtime/loader/ShadingTree.cpp). This backend has no run-time compiler: the host loader compiles the same strings
 * (ignis_amd/csrc/host/pexpr.h) into the bytecode below, the table travels as igd_scene.expr_code, and the shading kernel
 * interprets it where a material names a program (IG_MAT_EXPR_COLOR / IG_MAT_EXPR_NORMAL).
 *
 * Like ig_detmath.h this header is arithmetic shared by the product (HIP kernels, host constant folding) and the test
 * oracle; the functions are the ones Transpiler.cpp:602-922 maps the PExpr names to, cited per opcode.
 *
 * Value model: every value is four floats. bool / int / num live splatted in all four lanes (bool as 0 / 1, int as the
 * float of its value: exact to 2^24), vec2 / vec3 in the leading lanes. A scalar * vector product is therefore the plain
 * lane-wise product (vec3_mulf = vec3_mul with the expanded scalar, core/vector.art:84-85).
 *
 * Instruction word: op | dst << 8 | a << 12 | b << 16 | c << 20 | imm << 24; IGE_CONST is followed by four float words,
 * IGE_TEX by the texture index, IGE_BUMP by a word holding three more registers (d | e << 4 | f << 8).
 */
#ifndef IG_EXPR_H
#define IG_EXPR_H

#include "ig_detmath.h"

#define IGE_REGS 12

enum ige_op {
    IGE_END = 0,  /* result = r[a] */
    IGE_CONST,    /* r[dst] = the next four words */
    IGE_VAR,      /* r[dst] = variable imm (enum ige_var) */
    IGE_ADD,      /* vecN_add; onAddSub / onMulDiv / onScale: lane-wise (Transpiler.cpp:1027-1070) */
    IGE_SUB,
    IGE_MUL,
    IGE_DIV,
    IGE_NEG,
    IGE_SWZ,      /* r[dst].i = r[a].(imm >> 2 i & 3) (onAccess, Transpiler.cpp:1152-1196) */
    IGE_LT,       /* onRelOp: scalars */
    IGE_GT,
    IGE_LE,
    IGE_GE,
    IGE_EQ,       /* onEqual: all of the leading imm lanes equal */
    IGE_NOT,
    IGE_AND,
    IGE_OR,
    IGE_SELECT,   /* r[a] ? r[b] : r[c] */
    IGE_MIX,      /* lerp / vecN_lerp (core/common.art:237, core/vector.art:145-146): (1 - k) a + k b, k = r[c] */
    IGE_MIN,      /* math_builtins::fmin / fmax lane-wise */
    IGE_MAX,
    IGE_CLAMP,    /* clampf(v, l, u) = fmin(u, fmax(l, v)) (core/common.art:285) */
    IGE_F1,       /* lane-wise function imm (enum ige_f1) */
    IGE_POW,      /* math_builtins::pow lane-wise (a ^ f and pow(a, b)) */
    IGE_DOT,      /* over the leading imm lanes */
    IGE_LENGTH,
    IGE_NORM,     /* vecN_normalize = v * (1 / len) (core/vector.art:137-139) */
    IGE_CROSS,    /* core/vector.art:104-107 */
    IGE_SUM,
    IGE_AVG,
    IGE_LUMINANCE, /* color_luminance (core/color.art:29,81-83) */
    IGE_REFLECT,  /* vec3_reflect(v, n) = n * (2 n.v) - v (core/vector.art:124) */
    IGE_TEX,      /* bitmap texture (next word) looked up at r[a].xy */
    IGE_CHECKER,  /* node_checkerboard2 / 3 (texture/checkerboard.art:1-2), imm = 2 / 3 */
    IGE_BUMP,     /* node_bump(r[a], r[b], r[c], r[d].x, r[e].x, r[f].x) (texture/bump.art:3-11) */
    IGE_EVR,      /* ensure_valid_reflection(r[a], r[b], r[c]) (core/sampling.art:118-166) */
    IGE_IMOD,     /* int % int */
    IGE_IDIV,     /* int / int */
    IGE_ATAN2,
    IGE_FMOD,     /* math::fmod (core/math.art:79) */
    IGE_WRAP,     /* math::wrap(v, min, max) (core/math.art:88-91) */
    IGE_DIST,
    IGE_PACK,     /* make_vecN: (r[a].x, r[b].x, r[c].x, r[imm].x) */
    IGE_NOISE,    /* the noises of texture/noise.art: f(r[a] leading lanes, seed r[b].x), imm = enum ige_noise | 4 colour form | 8 signed | dims << 4 */
    IGE_VORONOI,  /* voronoiN / cvoronoiN / fbmN / cfbmN (texture/voronoi.art:46-63,100-119,158-181,221-274) and gabor2 (texture/noise.art:131-150):
                   * f(r[a] leading lanes, seed r[b].x), imm bit 0: fbm, bit 1: gabor, bit 2: colour, dims << 4 */
    IGE_OP_COUNT
};

enum ige_var {
    IGE_VAR_UVW = 0, /* ctx.uvw = (tex_coords, 0) (driver/shading_context.art:38); "uv" is its .xy */
    IGE_VAR_P,       /* ctx.surf.point */
    IGE_VAR_V,       /* -ctx.ray.dir ("V", "Rd") */
    IGE_VAR_N,       /* ctx.surf.local.col(2) */
    IGE_VAR_NG,      /* ctx.surf.face_normal */
    IGE_VAR_NX,      /* ctx.surf.local.col(0) */
    IGE_VAR_NY,      /* ctx.surf.local.col(1) */
    IGE_VAR_FRONT,   /* ctx.surf.is_entering */
    IGE_VAR_COUNT
};

enum ige_noise {
    IGE_NOISE_WHITE = 0, /* noise2_v / cnoise2: one value per distinct coordinate */
    IGE_NOISE_CELL  = 1, /* cellnoise2 / ccellnoise2: per integer cell */
    IGE_NOISE_VALUE = 2, /* pnoise2 / cpnoise2: smoothstep-interpolated values of the cell corners */
    IGE_NOISE_PERLIN = 3, /* perlin2 / cperlin2 (gradient noise); imm bit 3: sperlin2, the signed form */
};

enum ige_f1 {
    IGE_F_SIN = 0, IGE_F_COS, IGE_F_TAN, IGE_F_ASIN, IGE_F_ACOS, IGE_F_ATAN, IGE_F_EXP, IGE_F_EXP2, IGE_F_LOG, IGE_F_LOG2, IGE_F_LOG10,
    IGE_F_FLOOR, IGE_F_CEIL, IGE_F_ROUND, IGE_F_FRACT, IGE_F_TRUNC, IGE_F_SQRT, IGE_F_ABS, IGE_F_SIGN, IGE_F_RAD, IGE_F_DEG,
    IGE_F_SMOOTHSTEP, IGE_F_SMOOTHERSTEP, IGE_F_COUNT
};

#define IGE_INS(op, dst, a, b, c, imm) ((uint32_t)(op) | (uint32_t)(dst) << 8 | (uint32_t)(a) << 12 | (uint32_t)(b) << 16 | (uint32_t)(c) << 20 | (uint32_t)(imm) << 24)

#ifdef __cplusplus

struct ige_v4 {
    float v[4];
};

/* float -> int as v_cvt_i32_f32 does it (saturating, NaN -> 0), so that the oracle on x86 agrees for any input */
IGM_FN int ige_ftoi(float x)
{
    if (x != x)
        return 0;
    if (x >= 2147483648.0f)
        return 2147483647;
    if (x <= -2147483648.0f)
        return -2147483647 - 1;
    return (int)x;
}

/* hash_combine (FNV over the four bytes, core/random.art:7-13), sample_tea_u32 (:15-24) and the first next_f32 of
 * create_random_generator(seed) (:65-70,82-87: counter starts at 1; a float in [1, 2) minus 1) */
IGM_FN uint32_t ige_hash_combine(uint32_t h, uint32_t d)
{
    h = (h * 16777619u) ^ (d & 0xFFu);
    h = (h * 16777619u) ^ ((d >> 8) & 0xFFu);
    h = (h * 16777619u) ^ ((d >> 16) & 0xFFu);
    h = (h * 16777619u) ^ ((d >> 24) & 0xFFu);
    return h;
}
IGM_FN uint32_t ige_tea(uint32_t v0, uint32_t v1)
{
    uint32_t sum = 0;
    for (int i = 0; i < 4; ++i) {
        sum += 0x9e3779b9u;
        v0 += ((v1 << 4) + 0xa341316cu) ^ (v1 + sum) ^ ((v1 >> 5) + 0xc8013ea4u);
        v1 += ((v0 << 4) + 0xad90777du) ^ (v0 + sum) ^ ((v0 >> 5) + 0x7e95761eu);
    }
    return v1;
}
/* noiseN[T](coordinates, seed) (texture/noise.art:2-4,35-37,152-154) on the BITS of its coordinates: float bits for noiseN_v / pnoiseN, integer
 * bits for cellnoiseN */
IGM_FN float ige_noise_bits(int dims, const uint32_t* cb, float seed)
{
    uint32_t h = ige_hash_combine(0x811C9DC5u, igm_bits(seed));
    for (int i = 0; i < dims; ++i)
        h = ige_hash_combine(h, cb[i]);
    return igm_float((ige_tea(h, 1u) & 0x7FFFFFu) | 0x3F800000u) - 1.0f;
}
/* sperlin2 (texture/noise.art:79-127, "classic Perlin noise" after the cited gist), every operation as written there: vec4_divf is a
 * product with 1 / t, vec2_dot / vec2_len2 an fma, norm.y belongs to g01 and norm.z to g10 although n10 takes norm.y and n01 norm.z */
IGM_FN float ige_mod289(float x) { return x - igm_floor(x / 289.0f) * 289.0f; }
IGM_FN float ige_permute289(float v) { return ige_mod289((v * 34.0f) * v + v); }
IGM_FN float ige_sperlin2(float u, float v, float seed)
{
    /* noise1(1234, seed) (:2-4, the coordinate an integer): the offset of the lattice */
    const uint32_t h  = ige_hash_combine(ige_hash_combine(0x811C9DC5u, igm_bits(seed)), 1234u);
    const float shift = igm_float((ige_tea(h, 1u) & 0x7FFFFFu) | 0x3F800000u) - 1.0f;
    const float px = u + shift, py = v + shift;
    const float pix_ = igm_floor(px), piy_ = igm_floor(py), piz_ = pix_ + 1, piw_ = piy_ + 1;
    const float pfx = px - pix_, pfy = py - piy_, pfz = pfx - 1, pfw = pfy - 1;
    const float pix = ige_mod289(pix_), piy = ige_mod289(piy_), piz = ige_mod289(piz_), piw = ige_mod289(piw_);
    const float vix[4] = { pix, piz, pix, piz }, viy[4] = { piy, piy, piw, piw }, vfx[4] = { pfx, pfz, pfx, pfz }, vfy[4] = { pfy, pfy, pfw, pfw };
    float gx2[4], gy[4];
    const float inv41 = 1 / 41.0f;
    for (int i = 0; i < 4; ++i) {
        const float vi = ige_permute289(ige_permute289(vix[i]) + viy[i]);
        const float q  = vi * inv41;
        const float gx = (q - igm_floor(q)) * 2.0f - 1.0f;
        gy[i]          = igm_abs(gx) - 0.5f;
        gx2[i]         = gx - igm_floor(gx + 0.5f);
    }
    /* g00 = lane 0, g10 = lane 1, g01 = lane 2, g11 = lane 3; norm = (len2 g00, len2 g01, len2 g10, len2 g11) */
    const float len2[4] = { igm_fma(gx2[0], gx2[0], gy[0] * gy[0]), igm_fma(gx2[1], gx2[1], gy[1] * gy[1]), igm_fma(gx2[2], gx2[2], gy[2] * gy[2]), igm_fma(gx2[3], gx2[3], gy[3] * gy[3]) };
    const float norm[4] = { 1.79284291400159f - len2[0] * 0.85373472095314f, 1.79284291400159f - len2[2] * 0.85373472095314f, 1.79284291400159f - len2[1] * 0.85373472095314f,
                            1.79284291400159f - len2[3] * 0.85373472095314f };
    const float n00 = igm_fma(gx2[0], vfx[0], gy[0] * vfy[0]) * norm[0];
    const float n10 = igm_fma(gx2[1], vfx[1], gy[1] * vfy[1]) * norm[1];
    const float n01 = igm_fma(gx2[2], vfx[2], gy[2] * vfy[2]) * norm[2];
    const float n11 = igm_fma(gx2[3], vfx[3], gy[3] * vfy[3]) * norm[3];
    const float fx = pfx * pfx * pfx * (pfx * (pfx * 6 - 15) + 10), fy = pfy * pfy * pfy * (pfy * (pfy * 6 - 15) + 10); /* smootherstep (core/common.art:242) */
    const float nx0 = (1 - fx) * n00 + fx * n10, nx1 = (1 - fx) * n01 + fx * n11; /* vec2_lerp((n00, n01), (n10, n11), fade_x) */
    return 2.3f * ((1 - fy) * nx0 + fy * nx1);
}

IGM_FN float ige_noise(int kind, int dims, const float* x, float seed)
{
    uint32_t cb[3] = { 0, 0, 0 };
    if (kind == IGE_NOISE_PERLIN) /* perlin2 = (sperlin2 + 1) / 2 (:128); two coordinates only */
        return (ige_sperlin2(x[0], x[1], seed) + 1) / 2;
    if (kind == IGE_NOISE_CELL) { /* cellnoiseN: noiseN(x as i32, ..., seed) (:10,44,161) */
        for (int i = 0; i < dims; ++i)
            cb[i] = (uint32_t)ige_ftoi(x[i]);
        return ige_noise_bits(dims, cb, seed);
    }
    if (kind == IGE_NOISE_VALUE) {
        /* pnoiseN (:14-22,47-59,164-184): math::trunc, |smoothstep| of the fractions, the corners' values interpolated along x, then y, then z with
         * lerp(a, b, k) = (1 - k) a + k b */
        float ip[3], k[3], p[8];
        for (int i = 0; i < dims; ++i) {
            ip[i]         = (float)ige_ftoi(x[i]);
            const float f = x[i] - ip[i];
            k[i]          = igm_abs(f * f * (3 - 2 * f));
        }
        for (int c = 0; c < (1 << dims); ++c) {
            for (int i = 0; i < dims; ++i)
                cb[i] = igm_bits((c >> i) & 1 ? ip[i] + 1 : ip[i]);
            p[c] = ige_noise_bits(dims, cb, seed);
        }
        for (int i = 0; i < dims; ++i)
            for (int c = 0; c < (1 << (dims - 1 - i)); ++c)
                p[c] = (1 - k[i]) * p[2 * c] + k[i] * p[2 * c + 1];
        return p[0];
    }
    for (int i = 0; i < dims; ++i) /* noiseN_v (:2-4,39,156) */
        cb[i] = igm_bits(x[i]);
    return ige_noise_bits(dims, cb, seed);
}

/* voronoiN_f1_gen with the Euclidean distance and randomness 1 (texture/voronoi.art:46-63,100-119,158-181: what voronoiN / cvoronoiN and the octaves of
 * fbmN use): the nearest feature point of the 3^N cells around x, the first coordinate in the innermost loop; the point of cell k sits at
 * noiseN_v(k, seed [+ DEFAULT_CNOISE_SEED_SHIFT0 = 175391, + SHIFT1 = 822167 for the second / third coordinate]); distances |d|, vec2_len, vec3_len;
 * returns the distance, `col` = cnoiseN(nearest cell, seed) */
IGM_FN float ige_voronoi(int dims, const float* x, float seed, float* col)
{
    float ip[3] = { 0, 0, 0 }, fp[3] = { 0, 0, 0 }, t[3] = { 0, 0, 0 };
    for (int i = 0; i < dims; ++i)
        ip[i] = igm_floor(x[i]), fp[i] = x[i] - ip[i];
    const float shifts[3] = { 0.0f, 175391.0f, 822167.0f };
    float dist = 8.0f;
    const int cells = dims == 1 ? 3 : (dims == 2 ? 9 : 27);
    for (int c = 0; c < cells; ++c) {
        const float g[3] = { (float)(c % 3 - 1), (float)((c / 3) % 3 - 1), (float)(c / 9 - 1) };
        float k[3], dd[3] = { 0, 0, 0 };
        for (int i = 0; i < dims; ++i)
            k[i] = ip[i] + g[i];
        for (int i = 0; i < dims; ++i)
            dd[i] = (g[i] + ige_noise(IGE_NOISE_WHITE, dims, k, i == 0 ? seed : seed + shifts[i]) * 1.0f) - fp[i];
        const float d = dims == 1 ? igm_abs(dd[0]) : igm_sqrt(dims == 2 ? igm_fma(dd[0], dd[0], dd[1] * dd[1]) : igm_fma(dd[0], dd[0], igm_fma(dd[1], dd[1], dd[2] * dd[2])));
        if (d < dist) {
            for (int i = 0; i < dims; ++i)
                t[i] = k[i];
            dist = d;
        }
    }
    col[0] = ige_noise(IGE_NOISE_WHITE, dims, t, seed), col[1] = ige_noise(IGE_NOISE_WHITE, dims, t, seed + 1234.0f), col[2] = ige_noise(IGE_NOISE_WHITE, dims, t, seed + 5678.0f);
    col[3] = 1.0f;
    return dist;
}
/* fbmN / cfbmN = fbmN_gen(x, seed, 6, 2, 0.5, F1, Euclidean) (:221-257,267-274): the colour sum takes the amplitude AFTER its update */
IGM_FN float ige_fbm(int dims, const float* x0, float seed, float* col)
{
    float s = 0.0f, m = 0.0f, a = 0.5f, b[4] = { 0.0f, 0.0f, 0.0f, 1.0f }; /* color_builtins::black = (0, 0, 0, 1) */
    float x[3] = { x0[0], x0[1], x0[2] };
    for (int o = 0; o < 6; ++o) {
        float c[4];
        const float f = ige_voronoi(dims, x, seed, c);
        s += a * f;
        m += a;
        a *= 0.5f;
        for (int i = 0; i < dims; ++i)
            x[i] = x[i] * 2.0f;
        for (int i = 0; i < 3; ++i)
            b[i] = b[i] + c[i] * a;
        b[3] = igm_min(1.0f, b[3] + c[3] * a); /* color_add clamps the alpha (core/color.art:11) */
    }
    for (int i = 0; i < 4; ++i)
        col[i] = b[i] / m;
    return s / m;
}

/* gabor2 = gabor2_gen(uv, seed, 100, 20, 5, 0.01) (texture/noise.art:131-150): a hundred Gabor kernels whose positions and orientations are
 * noise2 of INTEGER coordinates (i, 0 .. 3); exp / cos / sin / atan2 are this backend's deterministic ones (ig_detmath.h) */
IGM_FN float ige_gabor2(float u, float v, float seed)
{
    const float pi = 3.14159265359f, phase = 20.0f, frequency = 5.0f, bandwidth = 0.01f;
    float acc = 0.0f;
    for (int i = 0; i < 100; ++i) {
        uint32_t cb[2] = { (uint32_t)i, 0u };
        float n[4];
        for (uint32_t k = 0; k < 4; ++k) {
            cb[1] = k;
            n[k]  = ige_noise_bits(2, cb, seed);
        }
        const float omega_d = igm_atan2(n[2], n[3]) * phase;
        const float ox = igm_cos(omega_d), oy = igm_sin(omega_d);
        const float len = igm_sqrt(igm_fma(n[0], n[0], n[1] * n[1]));
        const float kk  = igm_exp(-len * bandwidth * pi);
        const float dx = u - n[0], dy = v - n[1];
        acc += kk * igm_cos(2 * pi * frequency * igm_fma(dx, ox, dy * oy));
    }
    return acc / igm_sqrt(100.0f);
}

IGM_FN float ige_f1_apply(int f, float x)
{
    switch (f) {
    case IGE_F_SIN: return igm_sin(x);
    case IGE_F_COS: return igm_cos(x);
    case IGE_F_TAN: return igm_sin(x) / igm_cos(x);
    case IGE_F_ASIN: return igm_asin(x);
    case IGE_F_ACOS: return igm_acos(x);
    case IGE_F_ATAN: return igm_atan2(x, 1.0f);
    case IGE_F_EXP: return igm_exp(x);
    case IGE_F_EXP2: return igm_exp(x * 0.6931471805599453f);
    case IGE_F_LOG: return igm_log(x);
    case IGE_F_LOG2: return igm_log(x) * 1.4426950408889634f;
    case IGE_F_LOG10: return igm_log(x) * 0.4342944819032518f;
    case IGE_F_FLOOR: return igm_floor(x);
    case IGE_F_CEIL: return -igm_floor(-x);
    case IGE_F_ROUND: return igm_copysign(igm_floor(igm_abs(x) + 0.5f), x); /* roundf: halfway cases away from zero */
    case IGE_F_FRACT: return x - igm_floor(x);                               /* math::fract (core/math.art:74) */
    case IGE_F_TRUNC: return (float)ige_ftoi(x);                                 /* math::trunc (core/math.art:73) */
    case IGE_F_SQRT: return igm_sqrt(x);
    case IGE_F_ABS: return igm_abs(x);
    case IGE_F_SIGN: return x == 0 ? 0.0f : (igm_signbit(x) ? -1.0f : 1.0f); /* math::signf (core/math.art:76) */
    case IGE_F_RAD: return x * (IGM_PI / 180.0f);                             /* core/common.art rad / deg */
    case IGE_F_DEG: return x * (180.0f / IGM_PI);
    case IGE_F_SMOOTHSTEP: return x * x * (3 - 2 * x);                        /* core/common.art:241-242 */
    case IGE_F_SMOOTHERSTEP: return x * x * x * (x * (x * 6 - 15) + 10);
    default: return 0.0f;
    }
}

/* powf for the cases a shading expression meets: negative bases only with integral exponents */
IGM_FN float ige_pow(float x, float p)
{
    if (p == 0.0f)
        return 1.0f;
    if (x > 0.0f)
        return igm_pow(x, p);
    if (x == 0.0f)
        return p > 0.0f ? 0.0f : __builtin_inff();
    const float ip = igm_floor(p);
    if (ip != p)
        return __builtin_nanf("");
    const float m   = igm_pow(-x, p);
    const bool  odd = (ip - 2 * igm_floor(ip * 0.5f)) != 0.0f;
    return odd ? -m : m;
}

IGM_FN float ige_wrap(float v, float lo, float hi) /* math::wrap (core/math.art:88-91) */
{
    const float range = hi - lo;
    return range <= IGM_FLT_EPS ? lo : v - (range * igm_floor((v - lo) / range));
}

IGM_FN int ige_parity(float v) { return ige_ftoi(ige_wrap(v, 0.0f, 2.0f)) % 2; }

/* component k of v, chosen by comparisons: an index computed at run time would put v into scratch memory on the GPU */
IGM_FN float ige_pick(const ige_v4& v, uint32_t k) { return k == 0 ? v.v[0] : (k == 1 ? v.v[1] : (k == 2 ? v.v[2] : v.v[3])); }

/* The register file of a run. A plain array wherever private memory is cheap (host, oracle); the kernels hand in one that lives
 * in LDS (`r[i]` = element i of a lane's column), because registers named by the program are indexed dynamically and a private
 * array would be scratch memory. */
#if defined(__HIPCC__)
#define IGE_MEMBER __host__ __device__
#else
#define IGE_MEMBER
#endif
struct ige_private_regs {
    ige_v4 r[IGE_REGS];
    IGE_MEMBER ige_private_regs()
    {
        for (int i = 0; i < IGE_REGS; ++i)
            for (int k = 0; k < 4; ++k)
                r[i].v[k] = 0.0f;
    }
    IGE_MEMBER ige_v4& operator[](uint32_t i) { return r[i]; }
};

/* Ctx supplies: ige_v4 var(int id), ige_v4 tex(uint32_t id, float u, float v), ige_v4 evr(ige_v4 ng, ige_v4 v, ige_v4 n) */
template <class Ctx, class Regs>
IGM_FN ige_v4 ige_run(const uint32_t* code, const Ctx& ctx, Regs&& r)
{
    for (;;) {
        const uint32_t w   = *code++;
        const uint32_t op  = w & 0xFFu;
        const uint32_t dst = (w >> 8) & 0xFu;
        const uint32_t ia = (w >> 12) & 0xFu, ib = (w >> 16) & 0xFu, ic = (w >> 20) & 0xFu;
        const uint32_t imm = w >> 24;
        const ige_v4 a = r[ia], b = r[ib], c = r[ic];
        ige_v4 o = a;
        switch (op) {
        case IGE_END:
            return a;
        case IGE_CONST:
            for (int i = 0; i < 4; ++i)
                o.v[i] = igm_float(*code++);
            break;
        case IGE_VAR:
            o = ctx.var((int)imm);
            break;
        case IGE_ADD:
            for (int i = 0; i < 4; ++i)
                o.v[i] = a.v[i] + b.v[i];
            break;
        case IGE_SUB:
            for (int i = 0; i < 4; ++i)
                o.v[i] = a.v[i] - b.v[i];
            break;
        case IGE_MUL:
            for (int i = 0; i < 4; ++i)
                o.v[i] = a.v[i] * b.v[i];
            break;
        case IGE_DIV:
            for (int i = 0; i < 4; ++i)
                o.v[i] = a.v[i] / b.v[i];
            break;
        case IGE_NEG:
            for (int i = 0; i < 4; ++i)
                o.v[i] = -a.v[i];
            break;
        case IGE_SWZ:
            for (int i = 0; i < 4; ++i)
                o.v[i] = ige_pick(a, (imm >> (2 * i)) & 3u);
            break;
        case IGE_LT:
        case IGE_GT:
        case IGE_LE:
        case IGE_GE: {
            const bool t = op == IGE_LT ? a.v[0] < b.v[0] : (op == IGE_GT ? a.v[0] > b.v[0] : (op == IGE_LE ? a.v[0] <= b.v[0] : a.v[0] >= b.v[0]));
            for (int i = 0; i < 4; ++i)
                o.v[i] = t ? 1.0f : 0.0f;
            break;
        }
        case IGE_EQ: {
            bool t = true;
            for (uint32_t i = 0; i < 4; ++i) /* the first imm components */
                t = t && (i >= imm || a.v[i] == b.v[i]);
            for (int i = 0; i < 4; ++i)
                o.v[i] = t ? 1.0f : 0.0f;
            break;
        }
        case IGE_NOT:
            for (int i = 0; i < 4; ++i)
                o.v[i] = a.v[0] != 0 ? 0.0f : 1.0f;
            break;
        case IGE_AND:
            for (int i = 0; i < 4; ++i)
                o.v[i] = (a.v[0] != 0 && b.v[0] != 0) ? 1.0f : 0.0f;
            break;
        case IGE_OR:
            for (int i = 0; i < 4; ++i)
                o.v[i] = (a.v[0] != 0 || b.v[0] != 0) ? 1.0f : 0.0f;
            break;
        case IGE_SELECT:
            o = a.v[0] != 0 ? b : c;
            break;
        case IGE_MIX:
            for (int i = 0; i < 4; ++i)
                o.v[i] = (1 - c.v[0]) * a.v[i] + c.v[0] * b.v[i];
            break;
        case IGE_MIN:
            for (int i = 0; i < 4; ++i)
                o.v[i] = igm_min(a.v[i], b.v[i]);
            break;
        case IGE_MAX:
            for (int i = 0; i < 4; ++i)
                o.v[i] = igm_max(a.v[i], b.v[i]);
            break;
        case IGE_CLAMP:
            for (int i = 0; i < 4; ++i)
                o.v[i] = igm_clamp(a.v[i], b.v[i], c.v[i]);
            break;
        case IGE_F1:
            for (int i = 0; i < 4; ++i)
                o.v[i] = ige_f1_apply((int)imm, a.v[i]);
            break;
        case IGE_POW:
            for (int i = 0; i < 4; ++i)
                o.v[i] = ige_pow(a.v[i], b.v[i]);
            break;
        case IGE_ATAN2:
            for (int i = 0; i < 4; ++i)
                o.v[i] = igm_atan2(a.v[i], b.v[i]);
            break;
        case IGE_FMOD: /* x - trunc(x / n) * n */
            for (int i = 0; i < 4; ++i)
                o.v[i] = a.v[i] - (float)ige_ftoi(a.v[i] / b.v[i]) * b.v[i];
            break;
        case IGE_WRAP:
            for (int i = 0; i < 4; ++i)
                o.v[i] = ige_wrap(a.v[i], b.v[i], c.v[i]);
            break;
        case IGE_IMOD: {
            const int d = ige_ftoi(b.v[0]);
            const float m = (d == 0 || d == -1) ? 0.0f : (float)(ige_ftoi(a.v[0]) % d);
            for (int i = 0; i < 4; ++i)
                o.v[i] = m;
            break;
        }
        case IGE_IDIV: {
            const int d = ige_ftoi(b.v[0]);
            const float m = d == 0 ? 0.0f : (d == -1 ? -a.v[0] : (float)(ige_ftoi(a.v[0]) / d));
            for (int i = 0; i < 4; ++i)
                o.v[i] = m;
            break;
        }
        case IGE_DOT:
        case IGE_LENGTH:
        case IGE_NORM:
        case IGE_DIST:
        case IGE_SUM:
        case IGE_AVG: {
            /* vecN_dot = fmaf(x, x', fmaf(y, y', z z')) and vecN_reduce = f(x, f(y, z)): both fold from the last lane
             * (core/vector.art:37-39,95-97) */
            ige_v4 p = a, q = b;
            if (op == IGE_LENGTH || op == IGE_NORM)
                q = a;
            if (op == IGE_DIST) {
                for (int i = 0; i < 4; ++i)
                    p.v[i] = b.v[i] - a.v[i];
                q = p;
            }
            float s = 0;
            for (int i = (int)imm - 1; i >= 0; --i) {
                if (op == IGE_SUM || op == IGE_AVG)
                    s = i == (int)imm - 1 ? p.v[i] : p.v[i] + s;
                else
                    s = i == (int)imm - 1 ? p.v[i] * q.v[i] : igm_fma(p.v[i], q.v[i], s);
            }
            if (op == IGE_LENGTH || op == IGE_NORM || op == IGE_DIST)
                s = igm_sqrt(s);
            if (op == IGE_AVG)
                s = s / (float)imm;
            if (op == IGE_NORM) {
                const float inv = 1 / s;
                for (int i = 0; i < 4; ++i)
                    o.v[i] = a.v[i] * inv;
            } else {
                for (int i = 0; i < 4; ++i)
                    o.v[i] = s;
            }
            break;
        }
        case IGE_CROSS:
            o.v[0] = a.v[1] * b.v[2] - a.v[2] * b.v[1];
            o.v[1] = a.v[2] * b.v[0] - a.v[0] * b.v[2];
            o.v[2] = a.v[0] * b.v[1] - a.v[1] * b.v[0];
            o.v[3] = 0;
            break;
        case IGE_LUMINANCE: {
            const float l = a.v[0] * 0.2126f + a.v[1] * 0.7152f + a.v[2] * 0.0722f;
            for (int i = 0; i < 4; ++i)
                o.v[i] = l;
            break;
        }
        case IGE_REFLECT: { /* a = v, b = n */
            const float d = 2 * igm_fma(b.v[0], a.v[0], igm_fma(b.v[1], a.v[1], b.v[2] * a.v[2]));
            for (int i = 0; i < 3; ++i)
                o.v[i] = b.v[i] * d - a.v[i];
            o.v[3] = 0;
            break;
        }
        case IGE_TEX:
            o = ctx.tex(*code++, a.v[0], a.v[1]);
            break;
        case IGE_CHECKER: {
            const bool xy = ige_parity(a.v[0]) == ige_parity(a.v[1]);
            const bool t  = imm == 2 ? xy : (xy == (ige_parity(a.v[2]) == 1));
            for (int i = 0; i < 4; ++i)
                o.v[i] = t ? 1.0f : 0.0f;
            break;
        }
        case IGE_BUMP: {
            const uint32_t x = *code++;
            const float dist = r[x & 0xFu].v[0], sdx = r[(x >> 4) & 0xFu].v[0], sdy = r[(x >> 8) & 0xFu].v[0];
            /* a = input, b = Nx, c = Ny */
            const float rx[3] = { c.v[1] * a.v[2] - c.v[2] * a.v[1], c.v[2] * a.v[0] - c.v[0] * a.v[2], c.v[0] * a.v[1] - c.v[1] * a.v[0] };
            const float ry[3] = { a.v[1] * b.v[2] - a.v[2] * b.v[1], a.v[2] * b.v[0] - a.v[0] * b.v[2], a.v[0] * b.v[1] - a.v[1] * b.v[0] };
            const float det   = igm_fma(b.v[0], rx[0], igm_fma(b.v[1], rx[1], b.v[2] * rx[2]));
            const float sg    = (det == 0 ? 0.0f : (igm_signbit(det) ? -1.0f : 1.0f)) * dist;
            float n[3];
            for (int i = 0; i < 3; ++i) {
                const float grad = rx[i] * sdx + ry[i] * sdy;
                n[i]             = a.v[i] * igm_abs(det) - grad * sg;
            }
            const float inv = 1 / igm_sqrt(igm_fma(n[0], n[0], igm_fma(n[1], n[1], n[2] * n[2])));
            for (int i = 0; i < 3; ++i)
                o.v[i] = n[i] * inv;
            o.v[3] = 0;
            break;
        }
        case IGE_EVR:
            o = ctx.evr(a, b, c);
            break;
        case IGE_PACK:
            o.v[0] = a.v[0], o.v[1] = b.v[0], o.v[2] = c.v[0], o.v[3] = r[imm & 0xFu].v[0];
            break;
        case IGE_NOISE: {
            const int kind = (int)(imm & 3u), dims = (int)((imm >> 4) & 3u);
            float x[3] = { a.v[0], a.v[1], a.v[2] };
            if (!(imm & 4u)) {
                const float n = (imm & 8u) ? ige_sperlin2(x[0], x[1], b.v[0]) : ige_noise(kind, dims, x, b.v[0]);
                for (int i = 0; i < 4; ++i)
                    o.v[i] = n;
                break;
            }
            /* cnoiseN = (noise(seed), noise(seed + 1234), noise(seed + 5678), 1) (:8,42,159); ccellnoiseN hashes the FLOATS of the truncated
             * coordinates (:11,45,162); cpnoiseN interpolates the corners' colours, alpha 1 throughout (:24-33,61-75,186-206: the same lerps per
             * channel); cperlin2 = cpnoise2 * perlin2, alpha too (color_mulf, :211-214) */
            int k = kind;
            if (kind == IGE_NOISE_CELL) {
                for (int i = 0; i < dims; ++i)
                    x[i] = (float)ige_ftoi(x[i]);
                k = IGE_NOISE_WHITE;
            }
            if (kind == IGE_NOISE_PERLIN)
                k = IGE_NOISE_VALUE;
            o.v[0] = ige_noise(k, dims, x, b.v[0]);
            o.v[1] = ige_noise(k, dims, x, b.v[0] + 1234.0f);
            o.v[2] = ige_noise(k, dims, x, b.v[0] + 5678.0f);
            o.v[3] = 1.0f;
            if (kind == IGE_NOISE_PERLIN) {
                const float f = ige_noise(IGE_NOISE_PERLIN, dims, x, b.v[0]);
                for (int i = 0; i < 4; ++i)
                    o.v[i] *= f;
            }
            break;
        }
        case IGE_VORONOI: {
            float col[4];
            const int dims = (int)((imm >> 4) & 3u);
            const float f  = (imm & 2u) ? ige_gabor2(a.v[0], a.v[1], b.v[0]) : ((imm & 1u) ? ige_fbm(dims, a.v, b.v[0], col) : ige_voronoi(dims, a.v, b.v[0], col));
            for (int i = 0; i < 4; ++i)
                o.v[i] = (imm & 4u) ? col[i] : f;
            break;
        }
        default:
            break;
        }
        r[dst] = o;
    }
}

template <class Ctx>
IGM_FN ige_v4 ige_run(const uint32_t* code, const Ctx& ctx)
{
    return ige_run(code, ctx, ige_private_regs());
}

/* Walks the program that starts at word `start`: every opcode known, every register below IGE_REGS, every texture
 * below texture_count, and an IGE_END before the table ends. The device checks each program a material names. */
IGM_FN bool ige_validate(const uint32_t* code, uint32_t count, uint32_t start, uint32_t texture_count)
{
    uint32_t pc = start;
    while (pc < count) {
        const uint32_t w  = code[pc++];
        const uint32_t op = w & 0xFFu;
        if (op >= IGE_OP_COUNT)
            return false;
        if (((w >> 8) & 0xFu) >= IGE_REGS || ((w >> 12) & 0xFu) >= IGE_REGS || ((w >> 16) & 0xFu) >= IGE_REGS || ((w >> 20) & 0xFu) >= IGE_REGS)
            return false;
        const uint32_t imm = w >> 24;
        switch (op) {
        case IGE_END:
            return true;
        case IGE_CONST:
            pc += 4;
            break;
        case IGE_VAR:
            if (imm >= IGE_VAR_COUNT)
                return false;
            break;
        case IGE_F1:
            if (imm >= IGE_F_COUNT)
                return false;
            break;
        case IGE_EQ:
        case IGE_DOT:
        case IGE_LENGTH:
        case IGE_NORM:
        case IGE_DIST:
        case IGE_SUM:
        case IGE_AVG:
            if (imm < 1 || imm > 4)
                return false;
            break;
        case IGE_PACK:
            if ((imm & 0xFu) >= IGE_REGS)
                return false;
            break;
        case IGE_TEX:
            if (pc >= count || code[pc++] >= texture_count)
                return false;
            break;
        case IGE_BUMP: {
            if (pc >= count)
                return false;
            const uint32_t x = code[pc++];
            if ((x & 0xFu) >= IGE_REGS || ((x >> 4) & 0xFu) >= IGE_REGS || ((x >> 8) & 0xFu) >= IGE_REGS)
                return false;
            break;
        }
        default:
            break;
        }
    }
    return false;
}

#endif /* __cplusplus */
#endif /* IG_EXPR_H */
