// kernels.h — argument blocks shared by the HIP kernels and the host-side device code.
#pragma once

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "ig_tables.h"

namespace igdev {

// x / d for 0 <= x < 2^31 and a divisor known on the host: one 64-bit multiply and a shift instead of the ~35 instructions of an
// integer division (Granlund & Montgomery: m = ceil(2^(31 + s) / d), s = ceil(log2 d); the error m d - 2^(31 + s) < 2^s, so the floor is
// exact for every x below 2^31). Ray ids are decomposed into (iteration, pixel, sample) with three of them per shaded vertex.
struct FastDiv {
    uint32_t m, shift, d;
#ifdef __HIPCC__
    __device__ __forceinline__ uint32_t div(uint32_t x) const { return (uint32_t)(((unsigned long long)x * m) >> shift); }
#endif
    static FastDiv make(uint32_t d)
    {
        FastDiv f{};
        f.d        = d ? d : 1u;
        uint32_t s = 0;
        while (((unsigned long long)1 << s) < f.d)
            ++s;
        f.shift = 31 + s;
        f.m     = (uint32_t)((((unsigned long long)1 << f.shift) + f.d - 1) / f.d);
        return f;
    }
};

// Geometry resident in HBM. `geom` = ["trimesh_primbvh" fix table | scene Node8 array], so every
// BVH node / Tri4 packet is addressed as geom + 32-bit byte offset (SGPR base + VGPR offset).
constexpr int kDeepStack = 104; // traversal stack entries per lane beyond the LDS part, in DevScene::deep_stack

constexpr int kDevLeafRows = 8; // float4 rows per packed scene-BVH leaf (DevScene::leaves)

struct DevScene {
    const uint8_t* geom;
    uint32_t scene_nodes_off;      // byte offset of SceneBVH nodes inside geom
    uint32_t scene_node_count;     // 0: empty scene
    // SceneBVH leaves as the entity-leaf section of k_traverse reads them (kDevLeafRows rows of 16 bytes per leaf, packed by
    // igd_assign_scene from the EntityLeaf1 table: the L1 / texture path is priced per load instruction and lane, so a leaf is
    // scanned with two loads instead of three and a one-leaf shape's box comes with two instead of seven):
    //   0: (min.xyz, entity_id)  1: (max.xyz, flags)            the scan of a run
    //   2-4: the 3x4 to-local matrix                            entering
    //   5: (byte offset of the shape's Node8[] | bit 0: the shape's BVH is one node with one triangle leaf in slot 0,
    //       byte offset of its Tri4 packets, that leaf's child word, 0); spheres: (byte offset of {centre, radius} in shape_data, 0, 0, 0)
    //   6, 7: (lo.xyz, 0), (hi.xyz, 0) of that one child box
    const float4* leaves;
    const float4* leaf_scan; // rows 0 and 1 of every record once more, two rows per leaf: what the scan of a run reads (four leaves per 128-byte line)
    // shading tables
    const float* entities;         // 36 floats each
    const uint8_t* shape_data;     // dyn table "shapes" blob
    const uint64_t* shape_offsets; // byte offset per shape
    const ig_material* materials;
    const int32_t* entity_material; // material id per entity
    const ig_light* lights;
    uint32_t entity_count, material_count, light_count, infinite_light_count;
    ig_technique tech;
    // light selection (light/light_selector.art): hierarchy table + per-light codes, or uniform
    const float* light_hierarchy;
    const uint32_t* light_codes;
    uint32_t use_hierarchy;
    const float* light_cdf; // IG_SELECTOR_SIMPLE: CDF over the finite lights' flux (null otherwise)
    const ig_medium* media; // IG_TECHNIQUE_VOLPATH: participating media, ig_material.pad[2] names an entity's two sides
    uint32_t media_count;
    const uint32_t* expr_code; // programs of IG_MAT_EXPR_COLOR / IG_MAT_EXPR_NORMAL materials (include/ig_expr.h)
    float scene_radius;
    float scene_center[3]; // bbox_center(scene_bbox): env_sample_pos of the light tracer (light/env.art:2-6)
    // per entity: byte offsets of its shape's vertex / normal / index / texcoord arrays inside shape_data, so that the
    // shading chain is entity -> indices -> attributes (the reference walks entity -> shape table -> shape header first)
    const uint4* entity_ext;
    // per entity: the first row of its shape's triangle records in prim_records (six 16-byte rows per triangle: v0 v1 v2 n0 n1 n2 with
    // the three texture coordinates in the .w lanes), what surface_element() reads for a hit
    const uint32_t* entity_rec;
    const float4* prim_records;
    // bitmap textures (ig_material.tex_id)
    const ig_texture* textures;
    const uint8_t* texture_data;
    const float* cdf_data; // sampling tables of textured environment lights (igd_scene.cdf_data)
    // deep traversal stacks: entry e of resident lane l at deep_stack[e * deep_stride + l]; the persistent traversal
    // grid uses lanes [0, deep_tail_base), the tail kernel's grid the lanes from deep_tail_base on (they overlap in time)
    // analytic spheres (igd_scene.sphere_*): their scene BVH's nodes inside geom, leaves, and per leaf {byte offset of the
    // sphere's {centre, radius} record in shape_data, 0}. sphere_node_count == 0: no spheres, no sphere pass
    uint32_t sphere_nodes_off;
    uint32_t sphere_node_count;
    const float4* sphere_leaves; // same record layout as `leaves`
    const float4* sphere_leaf_scan;
    uint2* deep_stack;
    uint32_t deep_stride;
    uint32_t deep_tail_base; // bbox_radius(scene) * 1.01 for the environment light (light/env.art:88)
    uint32_t node_repeat;    // traverse_core.h step(): extra executions of the inner-node section per pass (0 unless the BVH outgrows the L2s)
};

// Ray queues in HBM. The reference's streams are one float per column (src/artic/driver/streams.art:
// 1-32); on CDNA a lane should move 16 bytes per memory instruction, so the same fields are kept as
// structure-of-arrays of 16-byte groups (5 vector columns + 2 scalar columns = 88 B per ray, the
// reference's live size):
//   rayA = (org.xyz, tmin)   rayB = (dir.xyz, tmax)   meta = (id, flags, rnd counter, depth)
//   pay  = (inv_pdf, contrib.rgb)   eta   |   hit = (ent_id, prim_id, t, u)   hit_v
struct PrimaryCols {
    float4* rayA;
    float4* rayB;
    int4* meta;
    float4* pay;
    float* eta;
    float4* hit; // ent_id / prim_id stored as bit patterns
    float* hit_v;
};

// Which columns of a primary stream carry data depends on its writer (round 5: what is constant is not moved):
//   kStreamShaded (k_shade, k_tail's spill): meta.y = the bits of eta (every such ray is a bounce ray: its flags are IG_RAY_FLAG_BOUNCE,
//                  which the traversal takes as uniform_flags), no eta column traffic; rayB.w = the path's generator seed (a bounce ray's
//                  tmax is FLT_MAX, which the traversal takes as uniform_tmax): the six FNV steps and three divisions of make_seed run
//                  once per path, not once per vertex
//   kStreamCamera (k_generate): meta.y = the ray's flags; the payload is init_pt_raypayload's constant (inv_pdf 0, contrib white, eta 1,
//                  technique/pathtracer.art:33-38): neither pay nor eta is written or read. With CameraStream::compact (a camera whose
//                  rays all leave one point: perspective without a lens, unmasked fishlens) ONLY rayB is stored: rayA = (eye, near clip),
//                  the flags, depth 1 and the generator's counter are the same for every ray and the id is first_id + the stream index
//   kStreamLight  (k_generate_light): meta.y = flags, pay = the light path's payload, eta = 1 (not written)
constexpr int kStreamShaded = 0, kStreamCamera = 1, kStreamLight = 2;
struct CameraStream {
    int32_t compact;      // != 0: the stream holds rayB only
    uint32_t rnd_counter; // Tea::counter after the pixel sampler's draws
    int64_t first_id;     // ray id of stream index 0
    float4 rayA;          // (eye, near clip)
};

// Shadow-ray queue: rayA = (org.xyz, tmin), rayB = (dir.xyz, tmax), col = (rgb, ray id bits)
struct SecondaryCols {
    float4* rayA;
    float4* rayB;
    float4* col;
    uint32_t* path_id; // light tracer only (null otherwise): the id of the light path a connection comes from (launch_lt_splat's sort key)
};

// Device-resident queue state: no host round trip per bounce (the reference reads counters back
// 3x per bounce, mapping_gpu.art:457-465,686-711).
// The ray indices of a traversal launch are handed out by kWorkShards counters, each on a cache line of its own and each over its own
// contiguous share of the stream: one counter word sustains ~88 atomics / us, which at 64 rays per reservation is the very rate the
// headline scene's closest-hit launches run at (5.5 G rays / s) — a launch of a few million rays spent most of its time in that queue
// (profiles/r05_experiment_ab.txt section 26). The workgroups of one XCD (blockIdx mod 8) start on the same share(s), so that an XCD's L2 sees one
// front of the stream, not all of them; a wave moves on to the next share when its own is used up.
#ifndef IG_WORK_SHARDS
#define IG_WORK_SHARDS 8
#endif
constexpr int kWorkShards     = IG_WORK_SHARDS; // a multiple of the 8 XCDs
constexpr int kWorkShardWords = 32; // 128 bytes between two counters
struct WorkCounters {
    uint32_t w[kWorkShards][kWorkShardWords];
};

struct QueueState {
    // q[s].primary: size of primary stream s; q[s].secondary: shadow rays generated together with it. The pair is one
    // aligned 64-bit word so that k_shade reserves space in both queues with ONE atomic (a counter word sustains only
    // ~88 atomics/us, and there are 2^16 workgroup windows per 16 M hits).
    struct Counts {
        uint32_t primary, secondary;
    };
    alignas(8) Counts q[2];
    uint32_t deep_count;       // rays of the traversal launch in flight whose stack outgrew LDS (re-traversed by the DEEP launch)
    // ---- from here on: cleared once per igd_render, not per chunk
    uint32_t error_flags;      // bit 0: traversal stack overflow
    uint32_t tail_rays;        // paths handed to the tail kernel (tail.hip)
    uint32_t deep_total;       // sum of deep_count over the chunk's launches: the host switches to DEEP-as-primary launches on it
    // statistics (Statistics.h:57-64)
    unsigned long long camera_rays, bounce_rays, shadow_rays, unoccluded;
    unsigned long long nodes[2], tris[2], leaves[2]; // [0] closest-hit launches, [1] any-hit launches
    unsigned long long section_passes[6], section_lanes[6]; // igd_stats: wave-level section executions and the lanes with work in them
    uint32_t tail_pass_in[24]; // paths each pass of the chunk's tail started with (tail.hip): the host sizes the next chunk's passes by them
    // ---- not part of what the host mirrors (kQueueStateHead bytes)
    // dynamic ray fetch: [0] traverse primary, [1] its DEEP launch, [2] traverse secondary, [3] its DEEP launch, [4] / [5] the sphere passes
    alignas(128) WorkCounters work[6];
};
constexpr size_t kQueueStateHead = offsetof(QueueState, work);

// The hit of a closest-hit launch in ONE 16-byte row (round 5): x = (entity << bits) | prim (all ones: a miss), y = t, z = u, w = v —
// when the scene's entity count and largest mesh fit 32 bits together (igd_assign_scene decides; `bits` = 0: the two-column form
// hit = (entity, prim, t, u), hit_v = v). The 4-byte hit_v store of every finished ray costs the traversal kernel 3 - 4 % (a lane's
// store is a cache line of its own), the load a little of the shading kernel's.
__device__ __forceinline__ float4 pack_hit(uint32_t bits, int ent, int prim, float t, float u, float v)
{
    const uint32_t w = prim < 0 ? 0xFFFFFFFFu : (((uint32_t)ent << bits) | (uint32_t)prim);
    return make_float4(__uint_as_float(w), t, u, v);
}
__device__ __forceinline__ void unpack_hit_ids(uint32_t bits, uint32_t w, int& ent, int& prim)
{
    const bool miss = w == 0xFFFFFFFFu;
    ent  = miss ? -1 : (int)(w >> bits);
    prim = miss ? -1 : (int)(w & ((1u << bits) - 1u));
}

struct TraverseArgs {
    DevScene scene;
    // input rays: rayA = (org, tmin), rayB = (dir, tmax); meta == nullptr -> uniform_flags
    const float4* rayA;
    const float4* rayB;
    const int4* meta;
    uint32_t uniform_flags;
    const uint32_t* count;  // device pointer to the number of rays
    // first launch: rays whose stack outgrows LDS are appended here (indices) instead of being finished;
    // DEEP launch: the rays to traverse are index_list[0 .. *count)
    uint32_t* index_list;
    uint32_t* index_count;
    uint32_t* work_counter; // a WorkCounters (kWorkShards counters, kWorkShardWords apart), zero before launch
    QueueState* qs;
    // outputs: hit = (ent_id, prim_id, t, u), hit_v = v. Any-hit launches may leave them null.
    float4* hit;
    float* hit_v;
    float4 uniform_rayA; // rayA == nullptr: every ray's (org, tmin) (CameraStream::compact)
    float uniform_tmax;  // use_uniform_tmax != 0: every ray's tmax (a kStreamShaded stream keeps the path's seed in rayB.w)
    int32_t use_uniform_tmax;
    uint32_t hit_pack; // closest hit: > 0: the packed one-row form with that many prim bits (pack_hit), hit_v is not written
    // any-hit epilogue (shadow rays): unoccluded rays add col.rgb into accum[id - id_base], id = bits(col.w)
    const float4* col;
    float4* accum;
    float4* accum_nee; // "NEE Weights" (ig_technique.aov_mis): the same splat once more, or null
    int64_t id_base;
    float inv_spi;
    // scenes with analytic spheres: the launch over the triangle BVH is followed by one over the sphere BVH that starts from its
    // hits. 0: single pass; 1: first of two (any-hit: the hit must be stored and the splat is left to the second); 2: the sphere pass
    int32_t sphere_pass;
    uint32_t* sphere_work_counter; // (a WorkCounters too) zero before launch
    uint32_t work_shards;          // > 1: the stream is handed out in kWorkShards shares (1: by the first counter alone)
    // the launch takes its rays in this order (raysort.hip: stream position -> ray index), or null: in stream order. Everything a ray
    // reads and leaves behind stays at its own index
    const uint32_t* sort_idx;
};

// raysort.hip: the rays of a stream ordered by (octant of the direction, Morton code of the origin in the scene's box)
struct RaySortArgs {
    const float4* rayA;
    const float4* rayB;
    const uint32_t* count;
    float box_min[3];
    float box_scale[3];  // 2^cell_bits / extent of the box
    uint32_t cell_bits;  // bits per axis of the origin's cell (<= 9: the key is 3 + 3 * cell_bits bits)
    uint32_t octant_low; // != 0: the octant in the key's low bits (cell-major order) instead of its high ones
    uint32_t dir_bits;   // > 0: instead of the octant, the direction's cell on the octahedral map, dir_bits bits per axis (2 * dir_bits + 3 * cell_bits <= 32)
    uint32_t* keys[2];   // ping-pong, one word per ray
    uint32_t* idx[2];
    uint32_t* wg_hist;   // 256 x grid words
    uint32_t* state;     // 256 words
};

// k_generate_light: make_lt_emitter (technique/lighttracer.art:35-62), one light path per ray id
struct GenerateLightArgs {
    DevScene scene;
    PrimaryCols out;
    uint32_t* out_count;
    QueueState* qs;
    int32_t width, spi;
    int32_t iteration, frame, seed;
    int64_t first_local_id;
    int32_t rays_per_iteration;
    uint32_t n;
    int32_t ppm; // make_ppm_light_emitter instead of make_lt_emitter: the payload carries the light's id (radius_or_light)
};

struct GenerateArgs {
    PrimaryCols out;
    uint32_t* out_count;
    QueueState* qs;
    ig_camera cam;
    float sx, sy; // compute_scale_from_hfov / _vfov (camera/perspective.art:2-13), evaluated on the host
    int32_t width, height, spi;
    int32_t iteration, frame, seed;
    int32_t row_offset, row_stride; // tile sharding: local row r -> film row row_offset + r * row_stride
    int64_t first_local_id;         // first local ray id of this chunk
    int32_t rays_per_iteration;     // local pixels * spi (multi-iteration calls: iteration += id / rays_per_iteration)
    uint32_t n;                     // rays to generate
    FastDiv by_rays_per_iteration, by_spi, by_width; // (launch_generate fills them)
    const float* list_rays;         // list emitter (emitter.art:18-30): 8 floats per ray, or nullptr
    int32_t compact;                // CameraStream::compact: only rayB is written
    float4* accum_clear;            // the chunk's per-sample accumulators, cleared here (sample i <-> slot i) instead of by a memset of their own; or null
    // Halton pixel sampler (sampler/pixel_sampler.art:101-150): what setup_halton_pixel_sampler derives from the film
    // size, filled by launch_generate; the per-pixel offset it keeps in "__halton_offset" is recomputed per sample
    uint32_t halton_scale_x, halton_scale_y, halton_exp_x, halton_exp_y;
    int32_t halton_inv_x, halton_inv_y;
};

struct ShadeFrame { // per-iteration constants (src/artic/driver/settings.art:2-11)
    int32_t width, spi;
    int32_t iteration, frame, seed;
    int32_t row_offset, row_stride;
    int32_t rays_per_iteration; // local pixels * spi: ray ids of a multi-iteration call continue across iterations
    float wire_footprint;       // IG_TECHNIQUE_WIREFRAME: |dx x dy| of camera.differential (technique/wireframe.art:25-26)
    FastDiv by_rays_per_iteration, by_spi, by_width; // (set by ShadeFrame::finish)
    void finish()
    {
        by_rays_per_iteration = FastDiv::make((uint32_t)rays_per_iteration);
        by_spi                = FastDiv::make((uint32_t)spi);
        by_width              = FastDiv::make((uint32_t)width);
    }
    // a ray id -> (iteration of the call, sample of the pixel, local pixel x, local row)
    __device__ __forceinline__ void decompose(int ray_id, int& it_local, int& sample, int& px, int& row) const
    {
        const uint32_t id     = (uint32_t)ray_id;
        const uint32_t it     = by_rays_per_iteration.div(id);
        const uint32_t within = id - it * (uint32_t)rays_per_iteration;
        const uint32_t lpix   = by_spi.div(within);
        const uint32_t r      = by_width.div(lpix);
        it_local = (int)it, sample = (int)(within - lpix * (uint32_t)spi), px = (int)(lpix - r * (uint32_t)width), row = (int)r;
    }
};

// the pinhole camera of the light tracer's connections (Camera::sample_pixel, camera/perspective.art:16-57)
struct LtCameraArgs {
    float eye[3];
    float view[9]; // right, up, dir
    float sx, sy;
    int32_t width, height;
};

// IG_TECHNIQUE_PPM (ppm_core.h): the photon array — light pass: slot = light path index, written; camera pass: sorted by
// (grid cell, index), read through cell_offset[cell] .. cell_offset[cell + 1]
struct PpmArgs {
    int32_t pass; // 0 not the photon mapper, 1 light pass, 2 camera pass
    void* photons; // igp_photon[] (include/ig_photon.h), 32 bytes each
    const uint32_t* cell_offset; // IGP_GRID_CELLS + 1 entries
    int32_t photon_count;        // light paths per iteration: the normalisation of a gather (light_cache.max_count)
    int32_t valid_count;         // photons actually stored (light_cache.count)
    float radius;                // ppm_compute_radius(merge radius, iteration)
    float bbox_min[3], bbox_max[3];
};

struct ShadeArgs {
    DevScene scene;
    PrimaryCols in;
    PrimaryCols out;
    SecondaryCols sec;
    const uint32_t* in_count;
    uint32_t* out_count;
    QueueState* qs;
    float4* accum;   // per-sample radiance accumulators, (r, g, b, unused)
    float4* accum_direct; // "Direct Weights" (ig_technique.aov_mis): the emission of surfaces hit, or null
    int64_t id_base; // local ray id of accum[0]
    ShadeFrame frame;
    float inv_spi;
    LtCameraArgs lt_cam; // IG_TECHNIQUE_LIGHTTRACER
    PpmArgs ppm;         // IG_TECHNIQUE_PPM
    // the by-class launches (shade_kernel.h): the round's hits sorted by material and this launch's run of them {first, count}
    const uint32_t* sort_idx;
    const uint32_t* cls_range;
    uint32_t hit_pack; // the `in` stream's hits are packed rows (pack_hit) with that many prim bits; 0: hit + hit_v
    int32_t in_kind;   // who wrote the `in` stream (kStreamShaded / kStreamCamera / kStreamLight below)
    CameraStream cam_stream; // in_kind == kStreamCamera
    int32_t skip_misses; // the scene has no infinite light: a miss needs no shading (the kernels without the sort look at the hit first)
};

// Global counting sort of a round's hits by material (K3a-e, gpu_sort_primary, mapping_gpu.art:409-502), keys in class-major order
// so that each material class of the shading kernels is one run of the sorted index list. `state` (uint32 words):
//   [0, 256) size of a bin (bin = material id, material_count = miss)   [256, 512) first slot of a bin   [512, 768) unused
//   [768, 776) per class {first, count}
// `wg_hist`: [bin][workgroup] the bin's rays inside each workgroup's chunk of the stream, then their exclusive prefix over the workgroups
constexpr int kSortBins       = 256;
constexpr int kSortStateWords = 3 * kSortBins + 8;
constexpr int kSortClasses    = 4;
constexpr int kSortDeadBin    = 255; // bin_class of a bin whose rays need no shading at all
struct BinSortArgs {
    const float4* hit;
    uint32_t hit_pack;
    const uint32_t* count;
    const int32_t* entity_material;
    uint32_t material_count;
    uint8_t* keys;           // one per ray
    uint32_t* sort_idx;      // out: ray indices, sorted
    uint32_t* state;
    uint32_t* wg_hist;       // kSortBins x grid words
    const uint8_t* bin_order; // [material_count + 1]: the bins in class-major order
    const uint8_t* bin_class; // [material_count + 1]: class of a bin (0 basic + misses, 1 principled, 2 coated, 3 blend, kSortDeadBin)
};

struct TailArgs {
    DevScene scene;
    PrimaryCols in;
    const uint32_t* in_count;
    uint32_t* work_counter; // zero before launch
    QueueState* qs;
    float4* accum;
    int64_t id_base;
    ShadeFrame frame;
    float inv_spi;
    // max_bounces > 0: a path still alive after that many bounces inside this launch is appended to `out`
    // (same columns as `in`) instead of being followed further; a second launch over `out` finishes it.
    int32_t max_bounces;
    PrimaryCols out;
    uint32_t* out_count;
    int32_t count_paths; // add the input size to qs->tail_rays
    int32_t pass;        // index of this pass (qs->tail_pass_in)
    uint32_t deep_lane_base; // first deep-stack column of this launch's lanes
    // a wave that follows no more than this many paths traverses their closest-hit rays one at a time with all its lanes
    // (wide_core.h); 0: never
    uint32_t wide_lanes;
    // a wave that follows more than wide_lanes and no more than this many paths traverses their closest-hit rays eight at a time, eight lanes
    // per ray (group_core.h); 0: never
    uint32_t wide8_lanes;
    int32_t in_kind; // who wrote `in` (kStreamShaded / kStreamCamera / kStreamLight); `out` is always kStreamShaded
};

struct ResolveArgs {
    const float4* accum;
    float* fb;
    int32_t width, spi;
    int32_t row_offset, row_stride;
    int64_t first_local_pixel; // local (virtual) pixel index of accum[0]
    uint32_t pixels;           // virtual pixels in accum (a pixel of iteration k of the call is virtual pixel k * local_pixels + p)
    uint32_t local_pixels;     // pixels of one iteration
    uint32_t iterations;       // > 1: accum holds that many whole iterations, one thread sums a pixel's iterations in order
};

// k_info: the "Normals" / "Albedo" AOVs of the info-buffer wrapper (technique/internal/infobuffer.art): per-sample values of the
// camera rays' first hits, summed per pixel by k_resolve
struct InfoArgs {
    DevScene scene;
    PrimaryCols in; // camera rays of one chunk of iteration 0, traversed
    const uint32_t* count;
    float4* normals; // [chunk samples]
    float4* albedo;
    int64_t id_base; // local ray id of sample 0 of the chunk
    float inv_spi;
};

} // namespace igdev
