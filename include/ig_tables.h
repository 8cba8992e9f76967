/* ig_tables.h — POD byte layouts shared by the host loader, the HIP device and
 * the CPU oracle.
 *
 * The BVH / table structs below are byte-for-byte the reference's own upload
 * formats for its <N=8, M=4> configuration (CPU target with vector width >= 8),
 * so a SceneDatabase produced by the reference runtime can be handed to the HIP
 * device unchanged:
 *
 *   ig_node8         src/artic/traversal/bvh.art:85-89      (Node8, 256 B)
 *   ig_tri4          src/artic/shapes/trimesh.art:116-122   (Tri4, 208 B)
 *   ig_entity_leaf1  src/artic/traversal/bvh.art:64-73      (EntityLeaf1, 96 B)
 *   ig_lookup_entry  src/runtime/table/DynTable.h:6-10
 *   entity record    src/runtime/loader/LoaderEntity.cpp:150-162 (36 floats)
 *   shape record     src/runtime/shape/TriMeshProvider.cpp:575-596
 *   prim-BVH record  src/runtime/shape/TriMeshProvider.cpp:305-323
 *
 * Materials, lights, camera and technique reach the reference device only as
 * generated Artic source; the ig_material / ig_light /
 * ig_camera / ig_technique PODs below are this backend's lowering of the same
 * parameters (cited per field).
 */
#ifndef IG_TABLES_H
#define IG_TABLES_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- BVH -------------------------------------------------------------- */

/* child > 0: inner node id + 1; child < 0: ~first_leaf; child == 0: empty slot
 * (bounds = +inf / -inf), src/runtime/bvh/BvhNAdapter.h:37-93. */
typedef struct ig_node8 {
    float bounds[6][8]; /* min_x, max_x, min_y, max_y, min_z, max_z  x 8 children */
    int32_t child[8];
    /* Unused by the reference. This backend's host builder may leave the node's quantisation grid here (csrc/host/bvh.cpp,
     * quantise_node8): pad[0..2] = origin.xyz (float bits), pad[3] = IG_NODE8_QUANT_MARK | biased exponents of the three scales
     * (x in bits 0-7, y 8-15, z 16-23); every plane of a used slot is then fmaf(q, 2^(e - 127), origin) for an integer q in 0..255. */
    int32_t pad[8];
} ig_node8;
#define IG_NODE8_QUANT_MARK 0x51000000u

/* Four triangles in Moeller-Trumbore form: v0, e1 = v2 - v0, e2 = v0 - v1,
 * n = stable(e1 x e2); prim_id == -1 marks an unused slot, bit 31 of
 * prim_id[3] marks the last packet of a leaf
 * (src/runtime/bvh/TriBVHAdapter.h:94-151). */
typedef struct ig_tri4 {
    float v0[3][4];
    float e1[3][4];
    float e2[3][4];
    float n[3][4];
    int32_t prim_id[4];
} ig_tri4;

/* Scene-BVH leaf; bit 31 of entity_id marks the last leaf of a run
 * (src/runtime/bvh/SceneBVHAdapter.h:68-100). `local` is the 3x4 to-local
 * matrix, column major. `user` is the prim-BVH offset (in floats) into the
 * "trimesh_primbvh" fix table, split into two u32. */
typedef struct ig_entity_leaf1 {
    float min[3];
    int32_t entity_id;
    float max[3];
    int32_t shape_id;
    float local[12];
    uint32_t flags;
    int32_t mat_id;
    int32_t user[2];
} ig_entity_leaf1;

typedef struct ig_lookup_entry {
    uint32_t type_id;
    uint32_t flags;
    uint64_t offset; /* bytes into the dyn-table data blob */
} ig_lookup_entry;

#define IG_ENTITY_FLOATS 36 /* toLocal 3x4 | toGlobal 3x4 | normal 3x3 | shape | mat | pad */

/* ig_lookup_entry.type_id of the "shapes" dyn table: which provider wrote the record (ShapeProvider::id in the reference).
 * A sphere record is 4 floats {centre.xyz, radius} (src/runtime/shape/SphereProvider.cpp:38-47, src/artic/shapes/sphere.art:96-103). */
#define IG_SHAPE_TRIMESH 0u
#define IG_SHAPE_SPHERE 1u

/* Ray visibility flags, src/artic/traversal/ray.art:21-25 */
#define IG_RAY_FLAG_CAMERA 0x1u
#define IG_RAY_FLAG_LIGHT 0x2u
#define IG_RAY_FLAG_BOUNCE 0x4u
#define IG_RAY_FLAG_SHADOW 0x8u
#define IG_RAY_FLAG_TYPE_MASK 0xFu

/* ---- Materials -------------------------------------------------------- */

enum ig_bsdf_type {
    IG_BSDF_DIFFUSE    = 0, /* src/artic/bsdf/diffuse.art:2-61,   runtime/bsdf/DiffuseBSDF.cpp:13-27 */
    IG_BSDF_DIELECTRIC = 1, /* src/artic/bsdf/dielectric.art:15-37, runtime/bsdf/DielectricBSDF.cpp:13-41 */
    IG_BSDF_CONDUCTOR  = 2, /* src/artic/bsdf/conductor.art:47-141, runtime/bsdf/ConductorBSDF.cpp:13-34 */
    IG_BSDF_PRINCIPLED = 3, /* src/artic/bsdf/principled.art:236-481, runtime/bsdf/PrincipledBSDF.cpp:14-98 */
    IG_BSDF_ROUGH_DIELECTRIC = 5, /* src/artic/bsdf/dielectric.art:64-191 (make_dielectric_bsdf with a rough interface) */
    IG_BSDF_PLASTIC    = 4,
    IG_BSDF_TRANSPARENT = 7, /* make_perfect_refraction_bsdf (src/artic/bsdf/dielectric.art:1-11; runtime/bsdf/TransparentBSDF.cpp "transparent", PassthroughBSDF.cpp "passthrough" = white): p[0..2] colour */
    /* make_rad_brtdfunc_bsdf (src/artic/bsdf/rad.art:7-29, RadBRTDFuncBSDF.cpp "rad_brtdfunc"): Radiance's BRTDfunc with constant
     * arguments = add(add(lambertian, lambertian transmission), add(mirror, perfect refraction)) over make_add_bsdf (bsdf/mix.art:68).
     * p[0..2] reflection_specular, p[3..5] transmission_specular, p[6..8] reflection_front_diffuse + direct_diffuse,
     * p[9..11] reflection_back_diffuse + direct_diffuse, q[0..2] transmission_diffuse */
    IG_BSDF_RAD_BRTD = 9,
    /* make_rad_roos_bsdf (rad.art:36-56, RadRoosBSDF.cpp "rad_roos"): the Roos model for coated glazing on top of it; specular
     * reflection and transmission follow from the angle of incidence. p[0..2] (w, p, q) of the transmission, p[3..5] (w, p, q) of the
     * reflection AS THE ARTIC FUNCTION RECEIVES THEM (the C++ side passes the "refl_*" properties into the trns_* arguments and vice
     * versa; restated as written), p[6..8] reflection_front_diffuse, p[9..11] reflection_back_diffuse, q[0..2] transmission_diffuse */
    IG_BSDF_RAD_ROOS = 10,
    IG_BSDF_PHONG      = 8, /* make_phong_bsdf (src/artic/bsdf/phong.art:1-22, runtime/bsdf/PhongBSDF.cpp): p[0..2] specular_reflectance, p[3] exponent;
                             * powers through the reference's own fastpow (core/common.art:71-90), which is plain float / integer arithmetic */
    IG_BSDF_BLEND      = 6, /* make_mix_bsdf (src/artic/bsdf/mix.art:4-68), runtime/bsdf/BlendBSDF.cpp:14-56 ("blend" / "mix") */ /* src/artic/bsdf/plastic.art:2-41 over mix.art:4-65, runtime/bsdf/PlasticBSDF.cpp:13-44 */
};

enum ig_material_flags {
    IG_MAT_THIN       = 1u << 0, /* dielectric "thin" */
    IG_MAT_BUMP       = 1u << 1, /* wrapped in a bumpmap (src/artic/bsdf/map.art:36-42,64-67, MapBSDF.cpp:44-47): tex_id, p[11] */
    IG_MAT_CHECKER    = 1u << 2, /* reflectance is a checkerboard texture */
    IG_MAT_NORMALMAP  = 1u << 3, /* wrapped in a normalmap (src/artic/bsdf/map.art:36-42,55-61): tex_id, p[11] */
    IG_MAT_SMOOTH     = 1u << 5, /* conductor without roughness ("mirror", or roughness <= 1e-4): the delta branch of
                                  * make_rough_base_conductor_bsdf, src/artic/bsdf/conductor.art:56-68 */
    IG_MAT_IMAGE      = 1u << 4, /* diffuse reflectance / principled base colour is the bitmap texture tex_refl
                                  * (DiffuseBSDF.cpp:18, texture/image.art) */
    IG_MAT_CLEARCOAT_ALL = 1u << 6, /* principled: clearcoat_top_only = false (PrincipledBSDF.cpp:56) */
    IG_MAT_DOUBLESIDED = 1u << 7, /* wrapped in make_doublesided_bsdf (src/artic/bsdf/common.art:28-46; DoubleSidedBSDF.cpp "twosided" /
                                    * "doublesided"): hit from behind, the BSDF is built as if entered and used with both directions negated */
    IG_MAT_EXPR_COLOR  = 1u << 8, /* the colour material_color resolves (diffuse reflectance, plastic diffuse reflectance, principled base
                                   * colour) is the shading expression whose program starts at igd_scene.expr_code[tex_refl]
                                   * (include/ig_expr.h; ShadingTree::addColor with a PExpr string, src/runtime/loader/ShadingTree.cpp) */
    IG_MAT_EXPR_WEIGHT = 1u << 10, /* blend / mask: the weight is the number expression at igd_scene.expr_code[tex_id] (BlendBSDF.cpp:40,
                                    * MaskBSDF.cpp:30-55 with ShadingTree::addNumber), p[0] is unused */
    IG_MAT_EXPR_NUMBERS = 1u << 11, /* some NUMBER properties (roughness, metallic, ...) are shading expressions evaluated per hit
                                     * (ShadingTree::addNumber with a PExpr string or texture name, src/runtime/loader/ShadingTree.cpp:
                                     * 211-251; a colour-valued one counts as its average). bits(r[7]) = offset in igd_scene.expr_code of
                                     * the list: word 0 = count, then per entry three words {kind | slot_a << 8 | slot_b << 16,
                                     * bits(aspect), program offset} (enum ig_number_kind). The record keeps each property's default as
                                     * the value where no hit exists (the "Albedo" AOV). Not inside blends. */
    IG_MAT_EXPR_NORMAL = 1u << 9, /* wrapped in a "transform" BSDF (TransformBSDF.cpp:17-49, make_normal_set src/artic/bsdf/map.art:36-42)
                                   * whose normal is the program at igd_scene.expr_code[tex_id] */
};

/* Entries of a material's number list (IG_MAT_EXPR_NUMBERS). Slots index the record's floats: 0-11 p[], 12-19 q[], 20-27 r[]. */
enum ig_number_kind {
    IG_NUM_PLAIN = 0,       /* slot_a = value */
    /* roughness with a constant anisotropy (microfacet::compute_explicit, src/artic/core/microfacet.art:427-432):
     * slot_a = value / aspect, slot_b = value * aspect */
    IG_NUM_ROUGHNESS = 1,
    /* ... of a conductor / plastic coating: also IG_MAT_SMOOTH <=> either alpha <= 1e-4 (check_if_delta_distribution, :298) */
    IG_NUM_ROUGHNESS_DELTA = 2,
    /* ... of a dielectric interface: also bsdf_type rough <=> both alphas > 1e-4, p[8] = the pdf epsilon (dielectric.art:67-82) */
    IG_NUM_ROUGHNESS_DIELECTRIC = 3,
};

/* One record per material (= unique bsdf / area-light entity,
 * src/runtime/loader/LoaderEntity.cpp:82-96). 144 bytes. */
typedef struct ig_material {
    int32_t bsdf_type;
    int32_t light_id; /* >= 0: emissive, index into lights (area light on this entity) */
    uint32_t flags;
    int32_t tex_id;   /* bitmap texture index of the bump / normal map, -1 = none */
    /* diffuse:    p[0..2] reflectance, p[3] alpha (roughness)
     * dielectric: p[0] ext_ior (n1), p[1] int_ior (n2), p[2..4] specular_reflectance,
     *             p[5..7] specular_transmittance
     * bump / normal map: p[11] strength
     * conductor:  p[0..2] eta, p[3..5] k, p[6..8] specular_reflectance,
     *             p[9] alpha_u, p[10] alpha_v
     * bump:       p[11] strength
     * checker:    q[0..2] color0, q[3..5] color1, q[6] scale_x, q[7] scale_y
     * rough dielectric: as dielectric, plus p[8] pdf epsilon (dielectric.art:67-82), p[9] alpha_u, p[10] alpha_v
     * plastic:    p[0..2] diffuse_reflectance (or checker / image), p[3] ext_ior, p[4] int_ior,
     *             p[6..8] specular_reflectance, p[9] alpha_u, p[10] alpha_v (IG_MAT_SMOOTH: mirror coating)
     * blend:      p[0] weight; pad[0], pad[1] = indices of the two inner materials, which follow the entity-bound ones in the
     *             table (entity_per_material 0) and are not blends themselves
     * principled: p[0..2] base_color (or checker / image like the diffuse reflectance), p[3] reflective_ior,
     *             p[4] refractive_ior, p[5] diffuse_transmission, p[6] specular_transmission, p[7] specular_tint,
     *             p[8] roughness_u, p[9] roughness_v, p[10] flatness; r[0] metallic, r[1] sheen, r[2] sheen_tint,
     *             r[3] clearcoat, r[4] clearcoat_gloss, r[5] clearcoat_roughness; IG_MAT_THIN, IG_MAT_CLEARCOAT_ALL */
    float p[12];
    float q[8];
    int32_t tex_refl; /* bitmap texture index of the diffuse reflectance (IG_MAT_IMAGE), -1 = none */
    int32_t pad[3];   /* pad[0], pad[1]: blend; pad[2]: the medium interface of the entities that use this material
                       * (make_medium_interface, src/artic/driver/medium.art:27-44; LoaderEntity.cpp:57-80):
                       * (inner + 1) | (outer + 1) << 16, so 0 = no_medium_interface */
    float r[8];
} ig_material;

/* Writes one evaluated number into a (local copy of a) material record; shared by the HIP kernels and the oracle. */
#if defined(__HIPCC__)
__host__ __device__
#endif
static inline void ig_material_set_number(ig_material* m, uint32_t head, float aspect, float value)
{
    const uint32_t kind = head & 0xFFu, a = (head >> 8) & 0xFFu, b = (head >> 16) & 0xFFu;
    float* slots[3]     = { m->p, m->q, m->r };
    if (kind == IG_NUM_PLAIN) {
        slots[a < 12 ? 0 : (a < 20 ? 1 : 2)][a < 12 ? a : (a < 20 ? a - 12 : a - 20)] = value;
        return;
    }
    const float au = value / aspect, av = value * aspect;
    slots[a < 12 ? 0 : (a < 20 ? 1 : 2)][a < 12 ? a : (a < 20 ? a - 12 : a - 20)] = au;
    slots[b < 12 ? 0 : (b < 20 ? 1 : 2)][b < 12 ? b : (b < 20 ? b - 12 : b - 20)] = av;
    if (kind == IG_NUM_ROUGHNESS_DELTA) {
        m->flags = (au <= 1e-4f || av <= 1e-4f) ? (m->flags | IG_MAT_SMOOTH) : (m->flags & ~(uint32_t)IG_MAT_SMOOTH);
    } else if (kind == IG_NUM_ROUGHNESS_DIELECTRIC) {
        const int rough   = au > 1e-4f && av > 1e-4f;
        const float alpha = au < av ? au : av;
        m->bsdf_type      = rough ? IG_BSDF_ROUGH_DIELECTRIC : IG_BSDF_DIELECTRIC;
        m->p[8]           = alpha <= 0.01f ? 1e-3f : (alpha <= 0.1f ? 1e-4f : 1e-5f);
    }
}

/* ---- Bitmap textures --------------------------------------------------- */

/* A "packed" image as the reference uploads it for 8-bit files (src/runtime/Image.cpp:714-808,
 * src/artic/driver/image.art:9-16): texels are bytes, rows bottom-to-top (stb's vertical flip), sRGB files
 * already mapped to linear and re-quantised by the loader; 4 channels = one little-endian u32 RGBA per texel,
 * 1 channel = one byte. The device divides by 255. */
enum ig_tex_filter { IG_TEX_NEAREST = 0, IG_TEX_BILINEAR = 1, IG_TEX_BICUBIC = 2 }; /* src/artic/texture/image.art:85-156 */
enum ig_tex_wrap { IG_WRAP_REPEAT = 0, IG_WRAP_MIRROR = 1, IG_WRAP_CLAMP = 2 };      /* image.art:9-40 */

#define IG_TEX_FLOAT_BIT 0x100u

typedef struct ig_texture {
    uint32_t width, height;
    uint32_t channels; /* 1 or 4 packed 8-bit values per texel (LDR files, Image::loadAsPacked); with IG_TEX_FLOAT_BIT set, 1 or 4
                        * 32-bit floats per texel (OpenEXR / Radiance HDR files, Image::load: src/runtime/Image.cpp:497-712) */
    uint32_t filter;   /* enum ig_tex_filter */
    uint32_t wrap_u, wrap_v;
    uint64_t offset;   /* bytes into texture_data, 16-byte aligned */
} ig_texture;

/* ---- Lights ----------------------------------------------------------- */

enum ig_light_type {
    IG_LIGHT_PLANE = 0, /* "SimplePlaneLight", src/artic/light/area.art:416-440, 24 floats */
    IG_LIGHT_POINT = 1, /* "SimplePointLight", src/artic/light/point.art:20-35, 8 floats */
    IG_LIGHT_ENV   = 2, /* constant environment radiance, src/artic/light/env.art */
    IG_LIGHT_SPOT  = 3, /* "SimpleSpotLight", src/artic/light/spot.art:8-53, SpotLight.cpp:83-97 (finite, delta) */
    IG_LIGHT_DIRECTIONAL = 4, /* src/artic/light/directional.art:1-17, DirectionalLight.cpp (infinite, delta) */
    /* environment map sampled through a marginal / conditional CDF (make_environment_light_textured, src/artic/light/
     * env.art:109-157; EnvironmentLight.cpp:40-98): d[0..2] scale, d[3..11] the 3x3 "_transform" column by column,
     * then as integer bits d[12] texture index, d[13] offset of the CDF in igd_scene.cdf_data (floats),
     * d[14] CDF width, d[15] CDF height. Width = height = 0 ("cdf": "none"): no table, directions are sampled uniformly over the
     * sphere and the sample carries scale * texture (make_environment_light, env.art:161-164) */
    IG_LIGHT_ENV_TEXTURED = 5,
    /* make_sun_light (src/artic/light/sun.art:8-48, SunLight.cpp:32-57; infinite, not delta): d[0..2] direction
     * (scene to light, normalised), d[3] cos of the half angle, d[4..6] radiance */
    IG_LIGHT_SUN = 6,
    /* CIE sky models as function environments (src/artic/light/cie.art:1-41 over env.art:24-105, CIELight.cpp:38-107):
     * pad[0] = enum ig_cie_kind, pad[1] = has_ground; d[0..2] zenith, d[3..5] ground, d[6] ground_brightness,
     * d[7] zenith brightness / factor (sunny kinds), d[8] c2 (sunny kinds), d[9..11] sun direction, d[12..14] scale,
     * d[15..23] the 3x3 "_transform" column by column. Sampled over the sphere, or the cosine-weighted upper
     * hemisphere when there is no ground. */
    IG_LIGHT_CIE = 7,
    /* area light over an arbitrary triangle mesh (make_shape_area_emitter, src/artic/light/area.art:44-103; AreaLight.cpp
     * representation "None": non-planar meshes or "optimize": false): a uniformly chosen triangle, a uniform point on it.
     * entity_id = the emissive entity, d[0..2] radiance. Finite, not delta. */
    IG_LIGHT_MESH_AREA = 8,
    /* area light on a sphere (make_sphere_area_emitter, src/artic/light/area.art:259-317; AreaLight.cpp representation
     * "Sphere": an analytic sphere shape, or a mesh TriMesh::getAsSphere recognises as one): entity_id = the emissive entity,
     * d[0..2] centre in shape space, d[3] radius, d[4..6] radiance, d[7] area of the ellipsoid the entity transform makes of
     * it (compute_ellipsoid_area, src/artic/shapes/sphere.art:21-28, evaluated by the loader). Finite, not delta. */
    IG_LIGHT_SPHERE = 9,
    /* the Perez sky WITH its sun (make_perez_light_raw, src/artic/light/perez.art:301-317: make_sun_light whose samples and
     * emission also carry the sky function): the IG_CIE_PEREZ record above, plus d[14] cos of the sun's half angle, d[24..26] sun
     * radiance, d[27..29] sun direction in scene space ("_transform" * d[9..11]). Infinite, not delta. */
    IG_LIGHT_PEREZ = 10,
};

/* d[] for PLANE: origin.xyz, normal.x | x_axis.xyz, normal.y | y_axis.xyz, normal.z |
 *                t0.xy, t1.xy | t2.xy, t3.xy | radiance.rgb, area
 * d[] for POINT: position.xyz, 0 | intensity.rgb, 0
 * d[] for ENV:   radiance.rgb (= scale * radiance), 0   -- constant environment, sampled uniformly over
 *                the sphere (make_environment_light -> make_environment_light_function_spherical,
 *                src/artic/light/env.art:83-108,161-164)
 * d[] for SPOT:  position.xyz, cos(cutoff) | direction.xyz, cos(falloff) | intensity.rgb, 0   (the reference stores the
 *                angles and takes the cosines on the device; the loader does it here)
 * d[] for DIRECTIONAL: direction.xyz (the way the light travels, normalised), 0 | irradiance.rgb, 0 */
typedef struct ig_light {
    int32_t type;
    int32_t entity_id; /* emissive entity for area lights, -1 otherwise */
    int32_t pad[2];
    float d[32];
} ig_light;

enum ig_cie_kind {
    IG_CIE_UNIFORM = 0, IG_CIE_CLOUDY = 1, IG_CIE_CLEAR = 2, IG_CIE_INTERMEDIATE = 3,
    /* the Perez all-weather sky as a function environment (sky_function of make_perez_light_raw, src/artic/light/perez.art:292-299;
     * "has_sun": false): d[0..2] sky colour (tint * diffuse normalisation), d[3..5] ground radiance, the explicit parameters
     * (a, b, c) in d[6..8] and (d, e) in d[12..13], d[9..11] sun direction in the light's frame, d[15..23] "_transform".
     * The model behind these numbers (perez.art:94-290,321-375) is evaluated by the loader. */
    IG_CIE_PEREZ = 4,
};

enum ig_light_selector {
    IG_SELECTOR_UNIFORM   = 0, /* src/artic/light/light_selector.art:26-46 */
    IG_SELECTOR_HIERARCHY = 1, /* src/artic/light/light_selector.art:80-110 */
    IG_SELECTOR_SIMPLE    = 2, /* make_cdf_light_selector (:48-78): finite lights by a CDF over their flux (igd_scene.light_cdf) */
};

/* ---- Camera / technique ----------------------------------------------- */

enum ig_camera_type {
    IG_CAMERA_PERSPECTIVE = 0, /* src/artic/camera/perspective.art:29-42; with aperture_radius > eps: :69-84 (depth of field) */
    IG_CAMERA_ORTHOGONAL  = 1, /* src/artic/camera/orthogonal.art:14-26 */
    IG_CAMERA_FISHLENS    = 2, /* src/artic/camera/fishlens.art:8-79 ("fishlens" / "fisheye") */
};

enum ig_fisheye_mode { /* FisheyeAspectMode, src/artic/camera/fishlens.art:1-5 */
    IG_FISHEYE_CIRCULAR = 0,
    IG_FISHEYE_CROPPED  = 1,
    IG_FISHEYE_FULL     = 2,
};

/* Where in its pixel a camera sample lies (RayGenerationShader::generatePixelSampler, src/runtime/shader/RayGenerationShader.cpp:36-50) */
enum ig_pixel_sampler {
    IG_PIXEL_SAMPLER_UNIFORM = 0, /* make_uniform_pixel_sampler, src/artic/sampler/pixel_sampler.art:4-10 ("independent" and anything else) */
    IG_PIXEL_SAMPLER_MJITT   = 1, /* make_mjitt_pixel_sampler(4, 4), :13-34: correlated multi-jittered 4 x 4 */
    IG_PIXEL_SAMPLER_HALTON  = 2, /* setup_/make_halton_pixel_sampler, :97-167: bases 2 and 3, per-pixel index offset */
};

typedef struct ig_camera {
    float eye[3];  /* T * 0,             src/runtime/camera/PerspectiveCamera.cpp:69-76 */
    float dir[3];  /* T.linear.col(2) */
    float up[3];   /* T.linear.col(1) */
    float fov;     /* radians */
    int32_t fov_is_vertical;
    float near_clip, far_clip;
    float aspect_ratio; /* <= 0: use width / height */
    int32_t type;         /* enum ig_camera_type */
    int32_t fisheye_mode; /* enum ig_fisheye_mode, FishLensCamera.cpp:22-28 */
    int32_t fisheye_mask; /* FishLensCamera.cpp:17: samples with r > 1 carry no ray */
    float scale;          /* orthogonal: OrthogonalCamera.cpp:16 (registry "__camera_scale") */
    float aperture_radius, focal_length; /* PerspectiveCamera.cpp:19-20 */
    int32_t pixel_sampler; /* enum ig_pixel_sampler: "film": {"sampler": ...} (src/runtime/Runtime.cpp:51-53) */
} ig_camera;

enum ig_technique_type {
    IG_TECHNIQUE_PATH = 0, /* make_path_renderer, src/artic/technique/pathtracer.art:40-228 */
    /* ambient occlusion (make_ao_renderer, src/artic/technique/aotracer.art:1-24, AOTechnique.cpp): at every camera-ray hit one
     * cosine-distributed ray with the visibility flag of a bounce ray and no far end; white where it escapes. No bounces. */
    IG_TECHNIQUE_AO = 1,
    /* volumetric path tracing (make_volume_path_renderer, src/artic/technique/volpathtracer.art:37-260, VolumePathTechnique.cpp):
     * the path tracer with a current medium in the payload — transmittance on every segment, distance sampling and phase-function
     * scattering in on_bounce, the medium changing at transmissions through entities with a medium interface. */
    IG_TECHNIQUE_VOLPATH = 2,
    /* the debug views (make_debug_renderer, src/artic/technique/debugtracer.art:1-151, DebugTechnique.cpp): the first hit of a
     * camera ray shown as one of 28 properties (ig_technique.debug_mode = enum DebugMode, src/runtime/technique/DebugMode.h:6-35;
     * registry parameter "__debug_mode"); no shadow rays, no bounces, nothing on a miss. */
    IG_TECHNIQUE_DEBUG = 3,
    /* the light tracer (make_lt_emitter / make_lt_renderer, src/artic/technique/lighttracer.art:35-167, LightTracerTechnique.cpp): every
     * (pixel, sample) index starts one path on a light chosen by the light selector; each non-delta vertex is connected to the
     * pinhole camera and, unoccluded, splatted into the pixel it projects to; bounces sample the BSDF with adjoint = true.
     * max_depth, min_depth, clamp, light_selector as for the path tracer. Perspective cameras without depth of field; point, spot,
     * plane / mesh / sphere area, directional, sun and constant environment lights. */
    IG_TECHNIQUE_LIGHTTRACER = 4,
    /* make_wireframe_renderer (src/artic/technique/wireframe.art:21-73, WireframeTechnique.cpp): a hit closer to a triangle edge than
     * the pixel's footprint (camera.differential) shows white fading to black, any other hit lets the ray continue straight on; no
     * parameters. Perspective and orthogonal cameras, triangle meshes. */
    IG_TECHNIQUE_WIREFRAME = 5,
    /* the photon mapper (src/artic/technique/photonmapper.art, PhotonMappingTechnique.cpp): per iteration a light pass of
     * `photon_count` paths that leave one photon each at the first non-delta surface (LDE and LS*DE paths), then a camera pass
     * that gathers them within the merge radius at every non-delta vertex instead of sampling lights. max_depth / min_depth =
     * "max_depth" | "max_camera_depth", "min_depth" | "min_camera_depth"; lights as for the light tracer. */
    IG_TECHNIQUE_PPM = 6,
};

/* One record per medium, in the order entities acquire them (LoaderMedium::acquire, src/runtime/loader/LoaderMedium.cpp:113-121;
 * HomogeneousMedium.cpp: "homogeneous" / "constant"; VacuumMedium.cpp). The device builds make_homogeneous_medium
 * (src/artic/medium/homogeneous.art:1-58) with a Henyey-Greenstein phase function (src/artic/phase/henyeygreenstein.art). */
enum ig_medium_type { IG_MEDIUM_HOMOGENEOUS = 0, IG_MEDIUM_VACUUM = 1 };
typedef struct ig_medium {
    float sigma_a[3];
    float sigma_s[3];
    float g;
    int32_t type; /* enum ig_medium_type */
} ig_medium;

typedef struct ig_technique {
    int32_t max_depth;      /* src/runtime/technique/PathTechnique.cpp:11 (default 64) */
    int32_t min_depth;      /* default 2 */
    float clamp;            /* 0 = off */
    int32_t nee;            /* default 1 */
    int32_t light_selector; /* enum ig_light_selector */
    int32_t type;           /* enum ig_technique_type */
    int32_t aov_mis;        /* path tracer only (PathTechnique.cpp:16-27,56-61): also accumulate the AOVs "Direct Weights" (the MIS-weighted
                             * emission of surfaces a path hits, pathtracer.art:119-139) and "NEE Weights" (the next-event
                             * contributions of unoccluded shadow rays, on_shadow_miss :212-218) */
    int32_t debug_mode;     /* IG_TECHNIQUE_DEBUG: 0 normal, 1 tangent, ... 27 medium outer (DebugMode.h:6-35) */
    /* IG_TECHNIQUE_PPM (PhotonMappingTechnique.cpp:14-22,96-101) */
    int32_t photon_count;    /* "photons", default 1 000 000, at least 100 */
    int32_t max_light_depth; /* "max_light_depth", default 8 */
    float merge_radius;      /* "radius" (default 0.01) x the scene diameter: the registry's __tech_radius */
    int32_t reserved;
} ig_technique;

/* ---- Scene ------------------------------------------------------------ */

/* Everything IRenderDevice::assignScene receives through SceneSettings
 * (src/runtime/device/IRenderDevice.h:20-28: database, entity_per_material)
 * plus the POD lowering of what the reference passes as generated Artic.
 * All pointers are borrowed and must outlive the renders, as in the reference. */
typedef struct igd_scene {
    /* FixTable "entities" */
    const float* entities;
    uint32_t entity_count;
    /* DynTable "shapes" */
    const ig_lookup_entry* shape_lookups;
    uint32_t shape_count;
    const uint8_t* shape_data;
    uint64_t shape_data_size;
    /* FixTable "trimesh_primbvh": per shape {u32 node_count, tri_count, pad, pad} ig_node8[] ig_tri4[] */
    const uint8_t* primbvh;
    uint64_t primbvh_size;
    /* SceneBVHs["trimesh"] */
    const ig_node8* scene_nodes;
    uint32_t scene_node_count;
    const ig_entity_leaf1* scene_leaves;
    uint32_t scene_leaf_count;
    /* materials (entities are stored material-ordered, LoaderEntity.cpp:99-103) */
    const ig_material* materials;
    uint32_t material_count;
    const int32_t* entity_per_material; /* material_count entries */
    /* lights: infinite lights first, then finite */
    const ig_light* lights;
    uint32_t light_count;
    uint32_t infinite_light_count;
    /* light hierarchy for IG_SELECTOR_HIERARCHY over the finite lights (src/runtime/light/LightHierarchy.cpp:
     * 47-125, src/artic/light/light_hierarchy.art:14-96): 8 floats per node {pos.xyz, +-flux, dir.xyz, id};
     * id >= 0: leaf = finite light id, id < 0: inner, children at -id-1 and -id. light_codes[finite id] =
     * left(0)/right(1) decisions from the root, LSB first (used for the selection pdf). */
    const float* light_hierarchy;
    uint32_t light_hierarchy_nodes;
    const uint32_t* light_codes;
    ig_camera camera;
    ig_technique technique;
    float bbox_min[3];
    float bbox_max[3];
    int32_t film_width, film_height;
    float scene_radius; /* bbox_radius(scene_bbox) * 1.01, src/artic/light/env.art:88 */
    /* bitmap textures referenced by ig_material.tex_id */
    const ig_texture* textures;
    uint32_t texture_count;
    const uint8_t* texture_data;
    uint64_t texture_data_size;
    /* sampling tables of textured environment lights: per light the marginal CDF (height floats) followed by the
     * conditional CDFs (width floats per row), without the leading zeros (src/runtime/CDF.cpp:71-150,
     * src/artic/core/cdf.art:70-73,155-159) */
    const float* cdf_data;
    uint64_t cdf_data_count;
    /* SceneBVHs["sphere"]: the entities whose shape is an analytic sphere have a scene BVH of their own, traversed after
     * the triangle one with its result as the initial hit (one SceneGeometry per shape provider, TraversalShader.cpp:73-95,
     * src/artic/driver/mapping_cpu.art:385-403). Leaves: EntityLeaf1 with user = 0. Counts 0 when the scene has no spheres. */
    const ig_node8* sphere_nodes;
    uint32_t sphere_node_count;
    const ig_entity_leaf1* sphere_leaves;
    uint32_t sphere_leaf_count;
    /* IG_SELECTOR_SIMPLE: the CDF over the finite lights' flux as CDF::computeForArray writes it (src/runtime/CDF.cpp:14-44,
     * LoaderLight.cpp:455-478): one float per finite light, without the leading zero, last entry 1 */
    const float* light_cdf;
    uint32_t light_cdf_count;
    /* participating media (IG_TECHNIQUE_VOLPATH); ig_material.pad[2] names the two sides of an entity's surface */
    const ig_medium* media;
    uint32_t media_count;
    /* programs of the scene's shading expressions (include/ig_expr.h), one after the other, each ending in IGE_END;
     * IG_MAT_EXPR_COLOR / IG_MAT_EXPR_NORMAL materials name the word a program starts at */
    const uint32_t* expr_code;
    uint32_t expr_code_count;
} igd_scene;

#ifdef __cplusplus
}
#endif
#endif /* IG_TABLES_H */
