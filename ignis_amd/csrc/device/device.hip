// device.hip — host side of the MI355X render device and its C ABI (include/igd_device.h).
//
// Counterpart of src/device/Device.cpp: owns HBM (scene tables, SoA ray streams, per-sample
// accumulators, framebuffer), uploads the scene once, and drives the wavefront loop of
// `gpu_trace` (src/artic/driver/mapping_gpu.art:727-867). Differences that are the point of the
// MI355X design: queue sizes stay in HBM (no D2H/H2D per stage), each bounce round is
// traverse -> shade -> shadow-traverse (3 persistent kernels + 2 one-thread bookkeeping kernels
// instead of >= 8 + M launches with >= 5 host syncs), and the streams are sized for a whole
// iteration in flight (288 GB HBM) instead of 1 M rays.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <functional>
#include <limits>
#include <memory>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "igd_device.h"
#include "comm.h"
#include "kernels.h"
#include "ig_expr.h"
#include "ig_photon.h"

namespace igdev {
void launch_traverse(const TraverseArgs& args, bool any_hit, bool stats, int grid_blocks, uint32_t* deep_work_counter, hipStream_t stream, int deep_grid_blocks = 1 << 20, bool deep_primary = false);
void launch_traverse_q8(const TraverseArgs& args, bool any_hit, bool stats, int grid_blocks, uint32_t* deep_work_counter, hipStream_t stream, int deep_grid_blocks = 1 << 20, bool deep_primary = false);
void launch_tail_q8(const TailArgs& args, bool stats, bool full_bsdfs, int grid_blocks, hipStream_t stream);
int traverse_workgroups_per_cu();
void launch_generate(const GenerateArgs& args, hipStream_t stream);
void launch_generate_light(const GenerateLightArgs& args, hipStream_t stream);
void launch_shade(const ShadeArgs& args, int grid_blocks, bool full_bsdfs, hipStream_t stream, uint32_t classes);
void launch_bin_sort(const BinSortArgs& args, int grid, hipStream_t stream);
const uint32_t* launch_ray_sort(const RaySortArgs& args, int grid, hipStream_t stream);
void launch_round_end(QueueState* qs, int in_slot, hipStream_t stream);
void launch_secondary_end(QueueState* qs, int slot, QueueState* mirror, hipStream_t stream);
void launch_resolve(const ResolveArgs& args, hipStream_t stream);
void launch_info(const InfoArgs& args, int grid_blocks, hipStream_t stream);
void launch_tail(const TailArgs& args, bool stats, bool full_bsdfs, int grid_blocks, hipStream_t stream);
void launch_copy_paths(const PrimaryCols& src, const PrimaryCols& dst, const uint32_t* count, uint32_t max_count, const CameraStream& cam, hipStream_t stream);
// photon.hip (IG_TECHNIQUE_PPM)
void launch_shade_ppm(const ShadeArgs& args, int grid_blocks, hipStream_t stream);
size_t photon_grid_temp_bytes(uint32_t n);
size_t lt_splat_temp_bytes(uint32_t bound);
hipError_t launch_lt_splat(const float4* col, const float4* verdict, const uint32_t* path_id, const uint32_t* count, uint32_t bound, unsigned long long* keys, uint32_t* vals,
                           void* temp, size_t temp_bytes, float4* accum, int64_t id_base, float inv_spi, hipStream_t stream);
hipError_t build_photon_grid(const igp_photon* photons, uint32_t n, const PpmArgs& grid, igp_photon* sorted, uint32_t* cell_count, uint32_t* cell_offset, unsigned long long* keys, uint32_t* valid,
                       void* temp, size_t temp_bytes, hipStream_t stream);
} // namespace igdev

using namespace igdev;

static thread_local std::string g_error;

namespace {

struct HipError {
    int code;
    std::string msg;
};

#define HIP_CHECK(expr)                                                                                          \
    do {                                                                                                         \
        hipError_t _e = (expr);                                                                                  \
        if (_e != hipSuccess)                                                                                    \
            throw HipError{ _e == hipErrorOutOfMemory ? IGD_ERR_OUT_OF_MEMORY : IGD_ERR_DEVICE,                  \
                            std::string(#expr) + ": " + hipGetErrorString(_e) };                                 \
    } while (0)

template <typename T>
struct DevBuf {
    T* ptr       = nullptr;
    size_t count = 0;
    void alloc(size_t n)
    {
        if (n <= count && ptr)
            return;
        release();
        if (n == 0)
            return;
        HIP_CHECK(hipMalloc(&ptr, n * sizeof(T)));
        count = n;
    }
    void upload(const T* src, size_t n)
    {
        alloc(n);
        if (n)
            HIP_CHECK(hipMemcpy(ptr, src, n * sizeof(T), hipMemcpyHostToDevice));
    }
    void release()
    {
        if (ptr)
            (void)hipFree(ptr);
        ptr   = nullptr;
        count = 0;
    }
    ~DevBuf() { release(); }
};

constexpr int kPrimaryCols   = 22; // 5 vector columns (20 floats) + eta + hit_v
constexpr int kSecondaryCols = 12; // 3 vector columns
constexpr int kMaxTailPasses = 24;

} // namespace

struct igd_device {
    igd_setup setup{};
    int num_cus = 0;
    hipStream_t stream = nullptr; // wavefront rounds
    static constexpr int kMaxFlights = 8;
    hipStream_t side[kMaxFlights] = {}; // tail + resolve of a chunk, overlapping the rounds of the next chunks
    int n_flights = 4;                  // IGD_FLIGHTS (2 .. kMaxFlights)

    // scene
    bool has_scene = false;
    DevBuf<uint8_t> geom, shape_data;
    size_t primbvh_bytes = 0; // the "trimesh_primbvh" fix table at the start of geom
    DevBuf<ig_entity_leaf1> leaves, sphere_leaves;
    DevBuf<float4> dev_leaves, dev_sphere_leaves; // DevScene::leaves / sphere_leaves (packed records)
    DevBuf<float4> dev_leaf_scan, dev_sphere_leaf_scan; // DevScene::leaf_scan / sphere_leaf_scan
    std::vector<std::pair<uint64_t, uint32_t>> tri_spans; // where igd_assign_scene re-ordered triangle packets inside geom: (byte offset, packets)
    // photon mapper (IG_TECHNIQUE_PPM): photons by light path index, the same in grid order, sort keys, cell counts / offsets
    DevBuf<igp_photon> ppm_photons, ppm_sorted;
    DevBuf<unsigned long long> ppm_keys;
    DevBuf<uint32_t> ppm_cell_count, ppm_cell_offset, ppm_valid;
    DevBuf<uint8_t> ppm_temp;
    DevBuf<QueueState> ppm_qs;
    uint32_t ppm_valid_host = 0;
    float scene_bbox[6] = {}; // igd_scene.bbox_min / bbox_max (the photon grid's extent)
    DevBuf<uint8_t> geom_ref; // "trimesh_primbvh" in the reference's Tri4 layout, rebuilt from geom when the named buffer is asked for
    DevBuf<float> secondary_hit; // scenes with spheres: occlusion verdict of the triangle pass for the sphere pass (float4 per shadow ray)
    DevBuf<float> entities;
    DevBuf<uint64_t> shape_offsets;
    DevBuf<ig_material> materials;
    DevBuf<int32_t> entity_material;
    DevBuf<uint4> entity_ext;
    DevBuf<uint32_t> entity_rec;
    DevBuf<float4> prim_records;
    DevBuf<ig_light> lights;
    DevBuf<float> light_hierarchy, light_cdf;
    DevBuf<ig_medium> media;
    DevBuf<uint32_t> expr_code;
    DevBuf<ig_texture> textures;
    DevBuf<uint8_t> texture_data;
    DevBuf<float> cdf_data;
    // info-buffer AOVs (igd_setup.info_aovs): [0] "Normals", [1] "Albedo", film-sized like the colour buffer
    // [2] "Direct Weights", [3] "NEE Weights": path tracer with ig_technique.aov_mis (allocated when such a scene is assigned)
    // [4] "Denoised": never written by the device; the runtime's denoiser (extra/OIDN.cpp:103-104,135-138) fetches it by name, fills
    // it and calls syncFramebufferHostToDevice("Denoised"). Allocated at the first access by name (denoised_wanted).
    DevBuf<float> aov[5];
    std::vector<float> aov_host[5];
    bool aov_host_dirty[5] = { true, true, true, true, true };
    bool denoised_wanted  = false;
    DevBuf<float> info_tmp[2]; // per-sample values of one chunk (float4 each)
    DevBuf<QueueState> info_qs;
    uint32_t tail_lanes = 0; // lanes of one tail grid (its share of the deep-stack columns)
    DevBuf<uint2> deep_stack; // kDeepStack entries for every lane that can be resident (traversal grid + tail grid)
    DevBuf<uint32_t> light_codes;
    DevScene dscene{};
    ig_camera camera{};

    // streams
    size_t capacity = 0;
    DevBuf<float> primary[2], secondary;
    DevBuf<uint32_t> deep_rays; // indices of the rays a traversal launch hands to its DEEP launch
    DevBuf<float> list_rays;
    // light tracer: per shadow ray the id of its light path, and the sort of a round's unoccluded connections by (slot, path)
    // (launch_lt_splat, photon.hip); allocated by the first light-tracer render
    DevBuf<uint32_t> lt_path_id, lt_vals;
    DevBuf<unsigned long long> lt_keys;
    DevBuf<uint8_t> lt_temp;
    size_t lt_capacity = 0;
    // the by-class shading kernels' global sort of a round's hits by material (BinSortArgs, kernels.h): the sorted ray indices and the
    // key column (allocated with the first render that needs them), the sort's state words, the bins' order and classes (per scene)
    DevBuf<uint32_t> sort_idx, sort_state;
    DevBuf<uint8_t> sort_keys, sort_tables;
    size_t sort_capacity = 0;
    // raysort.hip: bounce and shadow rays traversed in (direction octant, origin cell) order when the BVH outgrows the L2s
    DevBuf<uint32_t> rs_keys[2], rs_idx[2], rs_state;
    size_t rs_capacity = 0;
    int ray_sort_env   = -1; // IGD_RAY_SORT=1: on (default off: it does not pay, see assignScene)
    bool ray_sort      = false;
    int ray_sort_bits  = 7;  // IGD_RAY_SORT_BITS: bits per axis of the origin's cell
    int ray_sort_octant_low = 0; // IGD_RAY_SORT_ORDER=cell: cell-major keys
    int ray_sort_dir_bits   = 0; // IGD_RAY_SORT_DIR_BITS: > 0: the direction's cell on the octahedral map instead of its octant
    uint32_t ray_sort_min = 1u << 18; // IGD_RAY_SORT_MIN: streams known to be shorter are traversed as they are
    int ray_sort_which = 3; // IGD_RAY_SORT_STREAMS: bit 0 bounce rays, bit 1 shadow rays

    // Several chunks can be in flight: while side streams finish chunks k - 3 .. k (tail passes, resolve, counter
    // read-back: a latency chain of ~max_depth dependent bounces, little work) the main stream already runs the
    // rounds of chunk k + 1. Everything a chunk's second half touches therefore exists once per flight slot.
    struct Flight {
        DevBuf<float> accum;     // per-sample radiance accumulators of the chunk
        DevBuf<float> accum_mis[2]; // the same for "Direct Weights" / "NEE Weights" (aov_mis), allocated on demand
        DevBuf<float> tail_in;   // the paths handed to the tail kernel (same columns as a primary stream)
        DevBuf<float> tail_long; // those still alive after a pass (the two buffers alternate)
        DevBuf<uint32_t> tail_ctr; // per pass: [2 * j] output count, [2 * j + 1] fetch counter
        size_t tail_capacity = 0;
        QueueState* qs       = nullptr; // device
        QueueState* host     = nullptr; // pinned read-back of qs once the chunk is complete
        hipEvent_t rounds_done = nullptr, resolved = nullptr, done = nullptr;
        bool pending = false, used = false;
        std::vector<std::pair<int, std::pair<hipEvent_t, hipEvent_t>>> spans; // timers recorded on either stream
    } flight[kMaxFlights];
    DevBuf<QueueState> qs_store;
    QueueState* host_store = nullptr; // pinned: one per flight, [kMaxFlights], [kMaxFlights + 1] per-round polling
    hipEvent_t poll_event[2] = {};
    QueueState* host_store_dev = nullptr; // device-side address of host_store (mapped pinned memory)
    // Consecutive iterations are executed as one wavefront (igd_render_settings.iterations): igd_render only records a
    // request that continues the pending one and the batch runs when it is large enough, when a different request
    // arrives, or when anything reads results. A 1080p iteration (16.6 M camera rays) alone leaves the launches of
    // the later bounce rounds too small: 8 iterations per wavefront give +26 %. IGD_BATCH_RAYS (0 = execute every call
    // immediately); interactive setups never defer.
    struct Pending {
        bool active = false;
        igd_render_settings rs{};
        int count = 0;
    } pending;
    uint64_t batch_rays = (uint64_t)1 << 29; // 32 iterations of 1080p x spi 8 (157 GB of streams): +2.5 % over 2^28 at 192 steps, which was +4 % over 2^27 (2^27: +26 % over single iterations)
    uint64_t chunk_seq     = 0;
    bool async_tail        = true; // IGD_ASYNC_TAIL=0: drain the side stream at the end of every igd_render

    // framebuffer
    int fb_w = 0, fb_h = 0;
    DevBuf<float> fb;
    std::vector<float> fb_host;
    bool fb_host_dirty = true;

    // Once at most this many paths are alive the remaining bounces are followed per lane (tail.hip) on the
    // side stream. IGD_TAIL_THRESHOLD overrides (0 disables).
    uint32_t tail_threshold = 1048576;
    int tail_waves_per_cu   = 12; // IGD_TAIL_WAVES (12 = what fits a CU at three waves per SIMD; 8 until late in round 3: +0.8 %)
    // The tail runs as a sequence of launches: each follows its paths for at most this many bounces and hands
    // the survivors, compacted, to the next one. Path lengths are geometric (a path inside a dielectric survives
    // a bounce with p ~ 0.85), so without this every wave idles behind its longest lane and pins registers and
    // LDS that the overlapping traversal launches of the next chunk need. IGD_TAIL_SPLIT overrides (0: one launch).
    uint32_t shade_classes = 1; // material classes of the scene (launch_shade)
    bool shade_by_class    = true; // IGD_SHADE_CLASSES=0: the one full instantiation for every material
    bool sort_single_class = true; // IGD_SORT_SINGLE_CLASS=0: a scene whose materials are all of the basic class skips the sort by material
    // > 0: the closest-hit launches of a render round write their hits as one packed 16-byte row with that many prim bits (kernels.h
    // pack_hit; igd_assign_scene: the entity count and the largest mesh fit 32 bits together, no analytic spheres). IGD_HIT_PACK=0: never
    uint32_t hit_pack_bits = 0;
    bool hit_pack_allowed  = true;
    int work_shards_env = -1;      // IGD_WORK_SHARDS: 1 = the rays of a traversal launch handed out by one counter, 8 = in shares (kernels.h WorkCounters); default by the size of the BVH
    int work_shards = kWorkShards;
    bool clear_in_generate = true; // IGD_CLEAR_IN_GENERATE=0: a memset of the chunk's accumulators in front of k_generate instead of inside it
    bool camera_compact    = true; // IGD_CAMERA_COMPACT=0: camera streams with every column although the rays all leave one point (kernels.h CameraStream)
    bool skip_misses       = true; // IGD_SKIP_MISSES=0: k_shade reads a miss's columns although the scene has no environment light (ShadeArgs::skip_misses)
    int node_repeat = -1; // IGD_NODE_REPEAT: DevScene::node_repeat (-1: by the size of the BVH)
    // IGD_NODE_FORMAT: -1 auto (the 128-byte quantised node records when the builder left every node on its 8-bit grid — it does from
    // 64 MB of nodes on — and the scene has no analytic spheres), 0 always Node8, 1 quantised whenever the tables allow it
    int node_format_mode = -1;
    uint32_t scene_node8_off = 0; // the scene BVH's Node8 records inside geom (the "scene_bvh" named buffer; the kernels may read the quantised copy)
    bool q8_nodes        = false; // this scene's kernels are the _q8 instantiations (traverse.hip, tail.hip with -DIG_QNODE=1)
    int tail_split = 6;
    int tail_wide  = 4; // IGD_TAIL_WIDE: TailArgs::wide_lanes
    int tail_wide8 = 8; // IGD_TAIL_WIDE8: TailArgs::wide8_lanes (one batch of eight rays: + 0.3 - 1 %; two batches and more lose, profiles/r06_experiment_ab.txt section 4)
    // A wave of the tail kernel costs 62 ns whatever it finds to do (3 072 of them: 0.19 ms per pass, measured on empty passes
    // with 256 / 1 024 / 3 072 waves, workgroups of one or four waves alike; a bare launch of that shape costs 1 ns per wave,
    // tools/launch_cost.hip, so the kernel spends it -- where is open). Pass j of a chunk gets as many waves as twice the
    // paths that pass j of the last collected chunk started with, scaled by the sizes of the two tails (one path per wave there,
    // tail.hip). On diamond_scene a pass keeps 55 % of its paths, so only the passes after the last bounce (depth 64: two of eleven)
    // shrink; scenes whose paths end early save more. Only the launch size depends on the guess: a pass that gets too few waves
    // refills them from its counter. IGD_TAIL_ADAPT=0: full grids; IGD_TAIL_DEBUG=1 prints the sizes.
    bool tail_adapt           = true;
    double tail_density       = 0.5; // expected paths per launched wave of a pass after the first (IGD_TAIL_DENSITY)
    double tail_share[24]     = {}; // paths at the start of pass j / paths at the start of pass 0, last collected chunk; [0] == 0: unknown

    igdev::Comm* comm = nullptr; // igd_comm_init: the RCCL communicator of a tile-sharded render (comm.hip)

    // statistics
    igd_stats stats{};
    std::vector<hipEvent_t> events; // pool for the stage timers
    size_t events_used = 0;

    ~igd_device()
    {
        if (stream)
            (void)hipStreamSynchronize(stream);
        for (auto sd : side)
            if (sd)
                (void)hipStreamSynchronize(sd);
        igdev::comm_destroy(comm);
        for (auto e : events)
            (void)hipEventDestroy(e);
        for (auto e : poll_event)
            if (e)
                (void)hipEventDestroy(e);
        for (auto& f : flight) {
            if (f.rounds_done)
                (void)hipEventDestroy(f.rounds_done);
            if (f.done)
                (void)hipEventDestroy(f.done);
            if (f.resolved)
                (void)hipEventDestroy(f.resolved);
        }
        if (host_store)
            (void)hipHostFree(host_store);
        if (stream)
            (void)hipStreamDestroy(stream);
        for (auto sd : side)
            if (sd)
                (void)hipStreamDestroy(sd);
    }

    static PrimaryCols colsAt(float* b, size_t c)
    {
        PrimaryCols p;
        p.rayA  = reinterpret_cast<float4*>(b + 0 * c);
        p.rayB  = reinterpret_cast<float4*>(b + 4 * c);
        p.meta  = reinterpret_cast<int4*>(b + 8 * c);
        p.pay   = reinterpret_cast<float4*>(b + 12 * c);
        p.hit   = reinterpret_cast<float4*>(b + 16 * c);
        p.eta   = b + 20 * c;
        p.hit_v = b + 21 * c;
        return p;
    }

    // column c of a stream starts at base + c * capacity floats; vector columns are 4 floats wide
    PrimaryCols primaryCols(int slot) const { return colsAt(primary[slot].ptr, capacity); }

    SecondaryCols secondaryCols() const
    {
        float* b       = secondary.ptr;
        const size_t c = capacity;
        SecondaryCols q;
        q.rayA = reinterpret_cast<float4*>(b + 0 * c);
        q.rayB = reinterpret_cast<float4*>(b + 4 * c);
        q.col  = reinterpret_cast<float4*>(b + 8 * c);
        q.path_id = lt_path_id.ptr;
        return q;
    }

    size_t wantedCapacity(size_t needed) const
    {
        // an explicit igd_setup.stream_capacity is allocated as given, once (no reallocation when larger batches arrive);
        // otherwise the streams grow with the largest request, up to what one batch can hold
        size_t cap = setup.stream_capacity ? (size_t)setup.stream_capacity
                                           : std::min(std::max<size_t>((size_t)1 << 24, (size_t)batch_rays), needed);
        return (cap + 255) & ~(size_t)255;
    }
    bool mem_capped = false; // the streams are as large as free memory allowed (ensureStreams)
    bool streamsTooSmall(size_t needed) const { return !((wantedCapacity(needed) <= capacity || mem_capped) && primary[0].ptr); }

    void releaseStreams()
    {
        for (int s = 0; s < 2; ++s)
            primary[s].release();
        secondary.release();
        deep_rays.release();
        sort_idx.release();
        sort_keys.release();
        sort_capacity = 0;
        for (int k = 0; k < 2; ++k)
            rs_keys[k].release(), rs_idx[k].release();
        rs_capacity = 0;
        lt_path_id.release();
        lt_vals.release();
        lt_keys.release();
        lt_temp.release();
        lt_capacity = 0;
        for (auto& f : flight) {
            f.accum.release();
            f.accum_mis[0].release();
            f.accum_mis[1].release();
        }
        capacity   = 0;
        mem_capped = false;
    }

    // All-or-nothing: `capacity` names the new size only once every buffer of that size exists. If an allocation fails
    // (the default batch asks for ~157 GB) everything is released and capacity is 0, so that the next call allocates
    // again instead of launching on buffers that are not there. The size is also capped by what hipMemGetInfo reports as free.
    void ensureStreams(size_t needed)
    {
        size_t cap = wantedCapacity(needed);
        if (!streamsTooSmall(needed))
            return;
        releaseStreams();
        bool capped = false;
        const size_t bytes_per_ray = (size_t)(2 * kPrimaryCols + kSecondaryCols + 1 + 4 * n_flights) * sizeof(float) + 5; // (+ sort_idx, sort_keys)
        {
            // (an explicit igd_setup.stream_capacity is an upper bound too: what does not fit is processed in chunks, render())
            size_t free_b = 0, total_b = 0;
            if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
                const size_t fit = (size_t)((double)free_b * 0.9 / (double)bytes_per_ray) & ~(size_t)255;
                if (fit >= 256 && cap > fit) {
                    cap    = fit;
                    capped = true;
                }
            }
        }
        try {
            for (int s = 0; s < 2; ++s)
                primary[s].alloc(cap * kPrimaryCols);
            secondary.alloc(cap * kSecondaryCols);
            deep_rays.alloc(cap);
            for (int k = 0; k < n_flights; ++k)
                flight[k].accum.alloc(cap * 4);
        } catch (...) {
            releaseStreams();
            throw;
        }
        capacity   = cap;
        mem_capped = capped;
    }

    static SecondaryCols secAt(float* b, size_t c)
    {
        SecondaryCols q;
        q.rayA = reinterpret_cast<float4*>(b + 0 * c);
        q.rayB = reinterpret_cast<float4*>(b + 4 * c);
        q.col  = reinterpret_cast<float4*>(b + 8 * c);
        q.path_id = nullptr;
        return q;
    }

    void ensureTailInput(Flight& f, size_t paths)
    {
        paths = ((paths + 4095) & ~(size_t)4095) + 4096;
        if (paths <= f.tail_capacity && f.tail_in.ptr)
            return;
        f.tail_in.release();
        f.tail_in.alloc(paths * kPrimaryCols);
        f.tail_long.release();
        f.tail_long.alloc(paths * kPrimaryCols);
        f.tail_capacity = paths;
    }

    int traverseGrid() const { return num_cus * igdev::traverse_workgroups_per_cu(); } // as many persistent workgroups as fit a CU (VGPRs and the LDS stacks, traverse_core.h)
    bool full_bsdfs = false; // the scene has a principled BSDF, a textured environment or a sun light: k_shade<true> / k_tail<*, true>
    int shade_mult = 64; // workgroups per CU in the k_shade grid (each loops over windows); IGD_SHADE_GRID
    // workgroups of a DEEP traversal launch (IGD_DEEP_GRID): the whole traversal grid. Eight were enough for the odd ray of a small
    // scene, but the 16 M-triangle stand-in sends enough rays there that the full grid is +9 % on it, and an empty full-grid launch
    // costs nothing measurable on diamond_scene (profiles/r02_experiment_ab.txt)
    int deep_grid = 1 << 20;
    // Rays of this scene outgrow the LDS stack often enough (> 1 % of the rays of a chunk, QueueState::deep_total) that the DEEP
    // instantiation runs as the primary traversal kernel: its lanes spill to their HBM columns instead of being listed and
    // re-traversed from the root by a second launch. Results do not depend on it. IGD_DEEP_PRIMARY=auto decides from the counts the
    // device reads back anyway.
    bool deep_primary = false;
    // -1 adaptive, 0 never, 1 always. Default: never — on the 16 M-triangle stand-in (16 - 24 entries needed, 14 in LDS) listing the
    // overflowing rays and re-traversing them with the whole grid measured 1 409 Mrays/s against 1 375 with the DEEP kernel as primary
    // (profiles/r03_experiment_standin.txt): the spill code in every push and pop costs all rays more than the re-traversal costs a few.
    int deep_primary_mode = 0;
    void noteDeep(unsigned long long deep_total, unsigned long long rays)
    {
        if (deep_primary_mode < 0 && !deep_primary && deep_total * 100ull > rays)
            deep_primary = true;
    }
    int shadeGrid() const { return num_cus * shade_mult; }
    int sortGrid() const { return num_cus * 8; } // workgroups (= chunks of the stream) of the sort by material (shade.hip k_bin_*)

    hipEvent_t event(size_t i)
    {
        while (events.size() <= i) {
            hipEvent_t e;
            HIP_CHECK(hipEventCreate(&e));
            events.push_back(e);
        }
        return events[i];
    }
    hipEvent_t nextEvent() { return event(events_used++); }
};

namespace {

int guarded(const char* what, const std::function<void()>& fn);

// the traversal kernels of the scene's node format (traverse.hip is compiled once per format)
void launchTraverse(const igd_device* d, const TraverseArgs& args_in, bool any_hit, bool stats, int grid_blocks, uint32_t* deep_work_counter, hipStream_t stream, int deep_grid_blocks, bool deep_primary)
{
    TraverseArgs args = args_in;
    args.work_shards  = (uint32_t)d->work_shards;
    (d->q8_nodes ? launch_traverse_q8 : launch_traverse)(args, any_hit, stats, grid_blocks, deep_work_counter, stream, deep_grid_blocks, deep_primary);
}

// Node8 records whose builder left them on a per-node 8-bit grid (ig_node8::pad, csrc/host/bvh.cpp quantise_node8) as 128-byte records:
//   row 0 (origin.xyz, the three biased scale exponents)   rows 1 - 3, one per axis: lo[0..7], hi[0..7], a byte per plane
//   rows 4 - 5 child ids   rows 6 - 7 unused (a record is one cache line)
// Lossless or not at all: false as soon as one plane of a used slot is not fmaf(q, 2^e, origin) for a byte q.
bool packQuantisedNodes(const ig_node8* nodes, size_t count, uint8_t* out)
{
    for (size_t n = 0; n < count; ++n) {
        const ig_node8& nd = nodes[n];
        const uint32_t hd  = (uint32_t)nd.pad[3];
        if ((hd & 0xFF000000u) != IG_NODE8_QUANT_MARK)
            return false;
        uint32_t rec[32] = {};
        uint8_t* q       = reinterpret_cast<uint8_t*>(rec + 4);
        for (int a = 0; a < 3; ++a) {
            float origin;
            std::memcpy(&origin, &nd.pad[a], 4);
            const uint32_t eb = (hd >> (8 * a)) & 0xFFu;
            if (eb == 0 || eb == 255 || !std::isfinite(origin))
                return false;
            const float sc = std::ldexp(1.0f, (int)eb - 127);
            rec[a]         = (uint32_t)nd.pad[a];
            for (int i = 0; i < 8; ++i) {
                if (nd.child[i] == 0)
                    continue;
                for (int k = 0; k < 2; ++k) {
                    const float plane = nd.bounds[2 * a + k][i];
                    const double qd   = std::nearbyint(((double)plane - (double)origin) / (double)sc);
                    if (!(qd >= 0 && qd <= 255))
                        return false;
                    // (several q can decode to the same float when the grid is finer than the floats around the origin: any of them will do)
                    bool found = false;
                    for (int dq = 0; dq <= 2 && !found; ++dq)
                        for (int sg = -1; sg <= 1 && !found; sg += 2) {
                            const double c = qd + sg * dq;
                            if (c < 0 || c > 255)
                                continue;
                            const float dec = std::fmaf((float)c, sc, origin);
                            if (std::memcmp(&dec, &plane, 4) == 0) {
                                q[16 * a + 8 * k + i] = (uint8_t)c;
                                found                 = true;
                            }
                        }
                    if (!found)
                        return false;
                }
            }
        }
        rec[3] = hd & 0x00FFFFFFu;
        for (int i = 0; i < 8; ++i)
            rec[16 + i] = (uint32_t)nd.child[i];
        std::memcpy(out + n * 128, rec, 128);
    }
    return true;
}

void assignScene(igd_device* d, const igd_scene* s)
{
    if (s->entity_count > 0 && (!s->entities || !s->shape_data || (s->scene_node_count > 0 && (!s->scene_nodes || !s->scene_leaves || !s->primbvh))
                                || (s->sphere_node_count > 0 && (!s->sphere_nodes || !s->sphere_leaves))))
        throw HipError{ IGD_ERR_INVALID_ARG, "igd_assign_scene: scene tables are incomplete" };
    for (uint32_t m = 0; m < s->material_count; ++m) {
        const ig_material& mat = s->materials[m];
        if (mat.bsdf_type != IG_BSDF_DIFFUSE && mat.bsdf_type != IG_BSDF_DIELECTRIC && mat.bsdf_type != IG_BSDF_CONDUCTOR && mat.bsdf_type != IG_BSDF_PRINCIPLED && mat.bsdf_type != IG_BSDF_PLASTIC && mat.bsdf_type != IG_BSDF_ROUGH_DIELECTRIC && mat.bsdf_type != IG_BSDF_BLEND && mat.bsdf_type != IG_BSDF_TRANSPARENT && mat.bsdf_type != IG_BSDF_PHONG && mat.bsdf_type != IG_BSDF_RAD_BRTD && mat.bsdf_type != IG_BSDF_RAD_ROOS)
            throw HipError{ IGD_ERR_UNSUPPORTED, "igd_assign_scene: material " + std::to_string(m) + " uses a BSDF the HIP backend cannot shade yet" };
        const uint32_t principled_flags = mat.bsdf_type == IG_BSDF_PRINCIPLED ? (uint32_t)(IG_MAT_THIN | IG_MAT_CLEARCOAT_ALL)
                                                                              : (mat.bsdf_type == IG_BSDF_DIELECTRIC ? (uint32_t)IG_MAT_THIN : 0u);
        if (mat.flags & ~(uint32_t)(IG_MAT_CHECKER | IG_MAT_BUMP | IG_MAT_NORMALMAP | IG_MAT_IMAGE | IG_MAT_SMOOTH | IG_MAT_DOUBLESIDED | IG_MAT_EXPR_COLOR | IG_MAT_EXPR_NORMAL | IG_MAT_EXPR_NUMBERS | (mat.bsdf_type == IG_BSDF_BLEND ? (uint32_t)IG_MAT_EXPR_WEIGHT : 0u) | principled_flags))
            throw HipError{ IGD_ERR_UNSUPPORTED, "igd_assign_scene: material " + std::to_string(m) + " carries flags its BSDF type does not define" };
        if (mat.flags & IG_MAT_EXPR_NUMBERS) {
            // the number list: inside the expression table, every slot a float of the record, every program valid
            uint32_t at;
            std::memcpy(&at, &mat.r[7], 4);
            const uint32_t ntex = s->textures && s->texture_data ? s->texture_count : 0u;
            bool ok = s->expr_code && at < s->expr_code_count && mat.bsdf_type != IG_BSDF_BLEND;
            const uint32_t n = ok ? s->expr_code[at] : 0u;
            ok = ok && n >= 1 && n <= 32 && (uint64_t)at + 1 + 3ull * n <= s->expr_code_count;
            for (uint32_t i = 0; ok && i < n; ++i) {
                const uint32_t head = s->expr_code[at + 1 + 3 * i];
                float aspect;
                std::memcpy(&aspect, &s->expr_code[at + 2 + 3 * i], 4);
                ok = (head & 0xFFu) <= IG_NUM_ROUGHNESS_DIELECTRIC && ((head >> 8) & 0xFFu) < 28 && ((head >> 16) & 0xFFu) < 28 && (head >> 24) == 0 && aspect > 0 && aspect <= 1
                     && ige_validate(s->expr_code, s->expr_code_count, s->expr_code[at + 3 + 3 * i], ntex);
            }
            if (!ok)
                throw HipError{ IGD_ERR_INVALID_ARG, "igd_assign_scene: material " + std::to_string(m) + " names no valid number-expression list" };
        }
        // tex_id is a texture index for bump / normal maps and a program offset for an expression weight or normal: one use per record
        if ((mat.flags & IG_MAT_EXPR_WEIGHT) && (mat.flags & (IG_MAT_BUMP | IG_MAT_NORMALMAP | IG_MAT_EXPR_NORMAL)))
            throw HipError{ IGD_ERR_UNSUPPORTED, "igd_assign_scene: material " + std::to_string(m) + " combines an expression weight with a bump map, normal map or expression normal (they share tex_id)" };
        if ((mat.flags & (IG_MAT_BUMP | IG_MAT_NORMALMAP)) && (mat.tex_id < 0 || mat.tex_id >= (int32_t)s->texture_count || !s->textures || !s->texture_data))
            throw HipError{ IGD_ERR_INVALID_ARG, "igd_assign_scene: bump / normal-mapped material " + std::to_string(m) + " has no valid texture" };
        if (mat.bsdf_type == IG_BSDF_BLEND)
            for (int k = 0; k < 2; ++k)
                if (mat.pad[k] < 0 || mat.pad[k] >= (int32_t)s->material_count || s->materials[mat.pad[k]].bsdf_type == IG_BSDF_BLEND
                    || s->materials[mat.pad[k]].bsdf_type == IG_BSDF_RAD_BRTD || s->materials[mat.pad[k]].bsdf_type == IG_BSDF_RAD_ROOS
                    || (s->materials[mat.pad[k]].flags & (IG_MAT_BUMP | IG_MAT_NORMALMAP | IG_MAT_EXPR_NORMAL | IG_MAT_EXPR_COLOR)))
                    throw HipError{ IGD_ERR_INVALID_ARG, "igd_assign_scene: blend material " + std::to_string(m) + " has no valid inner materials" };
        const bool has_albedo = mat.bsdf_type == IG_BSDF_DIFFUSE || mat.bsdf_type == IG_BSDF_PRINCIPLED || mat.bsdf_type == IG_BSDF_PLASTIC; // p[0..2] reflectance / base colour
        if ((mat.flags & IG_MAT_IMAGE) && (!has_albedo || mat.tex_refl < 0 || mat.tex_refl >= (int32_t)s->texture_count || !s->textures || !s->texture_data))
            throw HipError{ IGD_ERR_INVALID_ARG, "igd_assign_scene: image-textured material " + std::to_string(m) + " has no valid texture or is neither diffuse nor principled" };
        if ((mat.flags & IG_MAT_EXPR_COLOR) && (!has_albedo || (mat.flags & (IG_MAT_IMAGE | IG_MAT_CHECKER)) || mat.tex_refl < 0 || !s->expr_code
                                                || !ige_validate(s->expr_code, s->expr_code_count, (uint32_t)mat.tex_refl, s->textures && s->texture_data ? s->texture_count : 0u)))
            throw HipError{ IGD_ERR_INVALID_ARG, "igd_assign_scene: material " + std::to_string(m) + " names no valid colour expression program" };
        if ((mat.flags & IG_MAT_EXPR_NORMAL) && ((mat.flags & (IG_MAT_BUMP | IG_MAT_NORMALMAP)) || mat.tex_id < 0 || !s->expr_code
                                                 || !ige_validate(s->expr_code, s->expr_code_count, (uint32_t)mat.tex_id, s->textures && s->texture_data ? s->texture_count : 0u)))
            throw HipError{ IGD_ERR_INVALID_ARG, "igd_assign_scene: material " + std::to_string(m) + " names no valid normal expression program" };
        if ((mat.flags & IG_MAT_EXPR_WEIGHT) && (mat.tex_id < 0 || !s->expr_code
                                                 || !ige_validate(s->expr_code, s->expr_code_count, (uint32_t)mat.tex_id, s->textures && s->texture_data ? s->texture_count : 0u)))
            throw HipError{ IGD_ERR_INVALID_ARG, "igd_assign_scene: blend material " + std::to_string(m) + " names no valid weight expression program" };
        if ((mat.flags & IG_MAT_CHECKER) && !has_albedo)
            throw HipError{ IGD_ERR_UNSUPPORTED, "igd_assign_scene: checkerboard colours are only lowered for diffuse and principled BSDFs" };
        if (mat.light_id >= (int32_t)s->light_count)
            throw HipError{ IGD_ERR_INVALID_ARG, "igd_assign_scene: material light id out of range" };
        if (mat.light_id >= 0 && s->lights[mat.light_id].type != IG_LIGHT_PLANE && s->lights[mat.light_id].type != IG_LIGHT_MESH_AREA && s->lights[mat.light_id].type != IG_LIGHT_SPHERE)
            throw HipError{ IGD_ERR_UNSUPPORTED, "igd_assign_scene: only area lights can be emissive entities" };
    }
    const uint32_t n_finite = s->light_count - s->infinite_light_count;
    const bool hierarchy    = s->technique.light_selector == IG_SELECTOR_HIERARCHY && n_finite > 0;
    if (hierarchy && n_finite > 1 && (!s->light_hierarchy || !s->light_codes || s->light_hierarchy_nodes == 0))
        throw HipError{ IGD_ERR_INVALID_ARG, "igd_assign_scene: hierarchy light selector without a hierarchy table" };
    for (uint32_t l = 0; l < s->light_count; ++l) {
        const bool inf = l < s->infinite_light_count;
        const int lt = s->lights[l].type;
        if (lt < IG_LIGHT_PLANE || lt > IG_LIGHT_PEREZ)
            throw HipError{ IGD_ERR_UNSUPPORTED, "igd_assign_scene: unknown light type" };
        if (inf != (lt == IG_LIGHT_ENV || lt == IG_LIGHT_DIRECTIONAL || lt == IG_LIGHT_ENV_TEXTURED || lt == IG_LIGHT_SUN || lt == IG_LIGHT_CIE || lt == IG_LIGHT_PEREZ))
            throw HipError{ IGD_ERR_INVALID_ARG, "igd_assign_scene: infinite lights must come first and be environment, directional or sun lights" };
        if (lt == IG_LIGHT_ENV_TEXTURED) {
            uint32_t v[4];
            std::memcpy(v, &s->lights[l].d[12], sizeof(v));
            const bool no_table = v[2] == 0 && v[3] == 0; // "cdf": "none"
            if (v[0] >= s->texture_count || (!no_table && (!s->cdf_data || v[2] == 0 || v[3] == 0 || (uint64_t)v[1] + v[3] + (uint64_t)v[2] * v[3] > s->cdf_data_count)))
                throw HipError{ IGD_ERR_INVALID_ARG, "igd_assign_scene: textured environment light " + std::to_string(l) + " has no valid texture / CDF table" };
        }
    }

    // geometry blob: prim BVH fix table, then the scene BVH nodes
    std::vector<uint8_t> blob(s->primbvh, s->primbvh + s->primbvh_size);
    // Triangle packets in the order the triangle section reads them: the two halves of a packet (triangles 0-1, 2-3) are 96
    // contiguous bytes each — six 16-byte loads per half instead of twelve 8-byte ones — followed by the prim ids:
    //   half h, float 2 k + j = row k (v0.xyz, e1.xyz, e2.xyz, n.xyz) of triangle 2 h + j.
    auto repackTris = [&](uint64_t off, uint32_t node_count, uint32_t tri_count) {
        uint8_t* base = blob.data() + off + 16 + (size_t)node_count * sizeof(ig_node8);
        for (uint32_t t = 0; t < tri_count; ++t) {
            float src[48], dst[48];
            std::memcpy(src, base + (size_t)t * sizeof(ig_tri4), sizeof(src));
            for (int h = 0; h < 2; ++h)
                for (int k = 0; k < 12; ++k)
                    for (int j = 0; j < 2; ++j)
                        dst[h * 24 + k * 2 + j] = src[k * 4 + 2 * h + j];
            std::memcpy(base + (size_t)t * sizeof(ig_tri4), dst, sizeof(dst));
        }
    };
    std::unordered_set<uint64_t> packed_shapes;
    d->tri_spans.clear();
    blob.resize((blob.size() + 255) & ~(size_t)255);
    const uint32_t scene_nodes_off = (uint32_t)blob.size();
    const uint8_t* sn              = reinterpret_cast<const uint8_t*>(s->scene_nodes);
    blob.insert(blob.end(), sn, sn + (size_t)s->scene_node_count * sizeof(ig_node8));
    // the scene BVH of the analytic spheres (igd_scene.sphere_*) follows
    const uint32_t sphere_nodes_off = (uint32_t)blob.size();
    if (s->sphere_node_count) {
        const uint8_t* pn = reinterpret_cast<const uint8_t*>(s->sphere_nodes);
        blob.insert(blob.end(), pn, pn + (size_t)s->sphere_node_count * sizeof(ig_node8));
    }
    // The inner nodes once more as 128-byte quantised records, behind everything else, when every Node8 of the scene allows it
    // (packQuantisedNodes): the traversal kernels of that format (launch_traverse_q8) read nodes there and nowhere else. The Node8
    // records stay in the blob for the named buffers.
    std::unordered_map<uint64_t, uint32_t> q8_at; // prim-BVH offset of a shape -> byte offset of its quantised nodes inside geom
    uint32_t q8_scene_off = 0;
    bool q8 = d->node_format_mode != 0 && s->sphere_node_count == 0 && s->scene_node_count > 0;
    if (q8) {
        std::vector<std::pair<uint64_t, uint32_t>> shapes;
        size_t total = s->scene_node_count;
        for (uint32_t i = 0; i < s->scene_leaf_count && q8; ++i) {
            const ig_entity_leaf1& l = s->scene_leaves[i];
            const uint64_t off       = (((uint64_t)(uint32_t)l.user[1] << 32) | (uint64_t)(uint32_t)l.user[0]) * 4;
            if (off + 16 > s->primbvh_size || q8_at.count(off))
                continue; // (out of range: reported by the loop over the leaves below)
            uint32_t nc;
            std::memcpy(&nc, s->primbvh + off, 4);
            if (off + 16 + (uint64_t)nc * sizeof(ig_node8) > s->primbvh_size)
                continue;
            q8_at[off] = 0;
            shapes.emplace_back(off, nc);
            total += nc;
        }
        std::vector<uint8_t> qn(total * 128);
        q8 = packQuantisedNodes(s->scene_nodes, s->scene_node_count, qn.data());
        size_t at = (size_t)s->scene_node_count * 128;
        const size_t base = blob.size(); // (a multiple of 256)
        for (size_t k = 0; k < shapes.size() && q8; ++k) {
            q8 = packQuantisedNodes(reinterpret_cast<const ig_node8*>(s->primbvh + shapes[k].first + 16), shapes[k].second, qn.data() + at);
            q8_at[shapes[k].first] = (uint32_t)(base + at);
            at += (size_t)shapes[k].second * 128;
        }
        if (q8 && base + qn.size() + 256 < ((size_t)1 << 32)) {
            q8_scene_off = (uint32_t)base;
            blob.insert(blob.end(), qn.begin(), qn.end());
        } else {
            q8 = false;
        }
    }
    d->q8_nodes = q8;
    if (blob.size() >= ((size_t)1 << 32))
        throw HipError{ IGD_ERR_UNSUPPORTED, "igd_assign_scene: BVH blob exceeds 4 GiB (32-bit node offsets)" };
    blob.resize(blob.size() + 256); // tail padding: vector loads never run past the allocation
    d->primbvh_bytes = (size_t)s->primbvh_size;
    d->geom_ref.release();

    // per scene leaf: its packed record (DevScene::leaves). Where its shape's Node8[] / Tri4[] start comes from EntityLeaf1.user
    // (offset in floats, shapes/trimesh.art:201-219: header {node_count, tri_count, pad, pad}, nodes, tris).
    auto packLeaf = [](const ig_entity_leaf1& l, float4* r) {
        r[0] = make_float4(l.min[0], l.min[1], l.min[2], igm_float((uint32_t)l.entity_id));
        r[1] = make_float4(l.max[0], l.max[1], l.max[2], igm_float(l.flags));
        r[2] = make_float4(l.local[0], l.local[1], l.local[2], l.local[3]);
        r[3] = make_float4(l.local[4], l.local[5], l.local[6], l.local[7]);
        r[4] = make_float4(l.local[8], l.local[9], l.local[10], l.local[11]);
        r[5] = r[6] = r[7] = make_float4(0, 0, 0, 0);
    };
    std::vector<float4> dl((size_t)s->scene_leaf_count * kDevLeafRows + 4 * kDevLeafRows); // (+ four records of padding: a scan fetches up to four leaves ahead)
    for (uint32_t i = 0; i < s->scene_leaf_count; ++i) {
        const ig_entity_leaf1& l = s->scene_leaves[i];
        const uint64_t off       = (((uint64_t)(uint32_t)l.user[1] << 32) | (uint64_t)(uint32_t)l.user[0]) * 4;
        if (off + 16 > s->primbvh_size)
            throw HipError{ IGD_ERR_INVALID_ARG, "igd_assign_scene: prim BVH offset out of range" };
        uint32_t hdr[2]; // node_count, tri_count
        std::memcpy(hdr, s->primbvh + off, 8);
        const uint64_t tris_at = off + 16 + (uint64_t)hdr[0] * sizeof(ig_node8);
        if ((off & 3u) || tris_at + (uint64_t)hdr[1] * sizeof(ig_tri4) > s->primbvh_size)
            throw HipError{ IGD_ERR_INVALID_ARG, "igd_assign_scene: prim BVH table is misaligned or truncated" };
        if (packed_shapes.insert(off).second) {
            repackTris(off, hdr[0], hdr[1]);
            d->tri_spans.emplace_back(tris_at, hdr[1]);
        }
        float4* r = dl.data() + (size_t)i * kDevLeafRows;
        packLeaf(l, r);
        // bit 0 of the node offset: the shape's BVH is one node whose only child (slot 0) is a triangle leaf; the entity-leaf
        // section of k_traverse then does that node's visit itself (traverse_core.h) from rows 6 and 7
        uint32_t one_leaf = 0;
        int32_t child0    = 0;
        if (hdr[0] == 1) {
            ig_node8 root;
            std::memcpy(&root, s->primbvh + off + 16, sizeof(root));
            one_leaf = root.child[0] < 0;
            for (int c = 1; c < 8; ++c)
                one_leaf &= root.child[c] == 0 ? 1u : 0u;
            child0 = root.child[0];
            r[6]   = make_float4(root.bounds[0][0], root.bounds[2][0], root.bounds[4][0], 0);
            r[7]   = make_float4(root.bounds[1][0], root.bounds[3][0], root.bounds[5][0], 0);
        }
        r[5] = make_float4(igm_float((q8 ? q8_at[off] : (uint32_t)(off + 16)) | one_leaf), igm_float((uint32_t)tris_at), igm_float((uint32_t)child0), 0);
    }
    d->geom.upload(blob.data(), blob.size());
    d->dev_leaves.upload(dl.data(), dl.size());
    auto scanRows = [](const std::vector<float4>& records) {
        std::vector<float4> rows(records.size() / kDevLeafRows * 2);
        for (size_t i = 0; i < rows.size() / 2; ++i)
            rows[2 * i] = records[i * kDevLeafRows], rows[2 * i + 1] = records[i * kDevLeafRows + 1];
        return rows;
    };
    {
        const std::vector<float4> rows = scanRows(dl);
        d->dev_leaf_scan.upload(rows.data(), rows.size());
    }
    d->leaves.upload(s->scene_leaves, s->scene_leaf_count); // (reference layout: the "scene_bvh_leaves" named buffer)
    {
        // sphere leaves: row 5 = where the {centre, radius} record sits in the "shapes" blob
        std::vector<float4> sl((size_t)s->sphere_leaf_count * kDevLeafRows + 4 * kDevLeafRows);
        for (uint32_t i = 0; i < s->sphere_leaf_count; ++i) {
            const int32_t shape_id = s->sphere_leaves[i].shape_id;
            if (shape_id < 0 || (uint32_t)shape_id >= s->shape_count || s->shape_lookups[shape_id].type_id != IG_SHAPE_SPHERE
                || s->shape_lookups[shape_id].offset + 16 > s->shape_data_size)
                throw HipError{ IGD_ERR_INVALID_ARG, "igd_assign_scene: sphere leaf without a valid sphere shape" };
            float4* r = sl.data() + (size_t)i * kDevLeafRows;
            packLeaf(s->sphere_leaves[i], r);
            r[5] = make_float4(igm_float((uint32_t)s->shape_lookups[shape_id].offset), 0, 0, 0);
        }
        d->dev_sphere_leaves.upload(sl.data(), sl.size());
        const std::vector<float4> rows = scanRows(sl);
        d->dev_sphere_leaf_scan.upload(rows.data(), rows.size());
        d->sphere_leaves.upload(s->sphere_leaves, s->sphere_leaf_count);
    }

    d->entities.upload(s->entities, (size_t)s->entity_count * IG_ENTITY_FLOATS);
    std::vector<uint8_t> sd(s->shape_data, s->shape_data + s->shape_data_size);
    sd.resize(sd.size() + 64);
    d->shape_data.upload(sd.data(), sd.size());
    std::vector<uint64_t> so(s->shape_count);
    for (uint32_t i = 0; i < s->shape_count; ++i)
        so[i] = s->shape_lookups[i].offset;
    d->shape_offsets.upload(so.data(), so.size());
    d->materials.upload(s->materials, s->material_count);
    d->lights.upload(s->lights, s->light_count);
    d->light_hierarchy.upload(s->light_hierarchy, (size_t)s->light_hierarchy_nodes * 8);
    d->light_codes.upload(s->light_codes, s->light_codes ? n_finite : 0);
    const bool simple_selector = s->technique.light_selector == IG_SELECTOR_SIMPLE && n_finite > 0;
    if (simple_selector && (!s->light_cdf || s->light_cdf_count != n_finite))
        throw HipError{ IGD_ERR_INVALID_ARG, "igd_assign_scene: 'simple' light selector without a flux CDF over the finite lights" };
    d->light_cdf.upload(s->light_cdf, simple_selector ? n_finite : 0);
    if (s->media_count && !s->media)
        throw HipError{ IGD_ERR_INVALID_ARG, "igd_assign_scene: media_count without media" };
    // the volumetric path tracer keeps depth and current medium in one word of the stream (16 bits each)
    if (s->technique.type == IG_TECHNIQUE_VOLPATH && (s->media_count > 0xFFFEu || s->technique.max_depth > 0xFFFF))
        throw HipError{ IGD_ERR_UNSUPPORTED, "igd_assign_scene: the volumetric path tracer supports at most 65534 media and a max_depth of 65535" };
    d->media.upload(s->media, s->media_count);
    if (s->expr_code && s->expr_code_count) {
        d->expr_code.upload(s->expr_code, s->expr_code_count);
    } else {
        const uint32_t end_only = 0; // IGE_END: DevScene.expr_code doubles as the switch to the instantiation with the rare shading code
        d->expr_code.upload(&end_only, 1);
    }
    for (uint32_t i = 0; i < s->texture_count; ++i) {
        const ig_texture& t = s->textures[i];
        const uint32_t nc    = t.channels & ~IG_TEX_FLOAT_BIT;
        const uint64_t bytes = (uint64_t)t.width * t.height * nc * ((t.channels & IG_TEX_FLOAT_BIT) ? sizeof(float) : 1);
        if (t.width == 0 || t.height == 0 || (nc != 1 && nc != 4) || (t.offset & 15) || t.offset + bytes > s->texture_data_size)
            throw HipError{ IGD_ERR_INVALID_ARG, "igd_assign_scene: texture " + std::to_string(i) + " is malformed" };
    }
    d->textures.upload(s->textures, s->texture_count);
    {
        std::vector<uint8_t> td(s->texture_data, s->texture_data + (s->texture_count ? s->texture_data_size : 0));
        td.resize(td.size() + 16);
        d->texture_data.upload(td.data(), td.size());
    }
    d->cdf_data.upload(s->cdf_data, s->cdf_data ? s->cdf_data_count : 0);

    // material id per entity (entity table word 34, LoaderEntity.cpp:159)
    std::vector<int32_t> em(s->entity_count);
    for (uint32_t e = 0; e < s->entity_count; ++e) {
        std::memcpy(&em[e], s->entities + (size_t)e * IG_ENTITY_FLOATS + 34, 4);
        if (em[e] < 0 || em[e] >= (int32_t)s->material_count)
            throw HipError{ IGD_ERR_INVALID_ARG, "igd_assign_scene: entity material id out of range" };
    }
    d->entity_material.upload(em.data(), em.size());

    // where each entity's shape keeps its arrays (see DevScene::entity_ext)
    std::vector<uint4> ext4(s->entity_count);
    for (uint32_t e = 0; e < s->entity_count; ++e) {
        uint32_t shape_id;
        std::memcpy(&shape_id, s->entities + (size_t)e * IG_ENTITY_FLOATS + 33, 4);
        if (shape_id >= s->shape_count)
            throw HipError{ IGD_ERR_INVALID_ARG, "igd_assign_scene: entity shape id out of range" };
        const uint64_t base = s->shape_lookups[shape_id].offset;
        if (s->shape_lookups[shape_id].type_id == IG_SHAPE_SPHERE) {
            if (base + 16 > s->shape_data_size || base >= ((uint64_t)1 << 32))
                throw HipError{ IGD_ERR_INVALID_ARG, "igd_assign_scene: shape offset out of range" };
            ext4[e] = make_uint4((uint32_t)base, 0xFFFFFFFFu, 0u, 0u); // surface_element(): .y marks the analytic sphere
            continue;
        }
        if (base + 48 > s->shape_data_size)
            throw HipError{ IGD_ERR_INVALID_ARG, "igd_assign_scene: shape offset out of range" };
        int32_t hdr[4]; // faces, vertices, normals, texcoords
        std::memcpy(hdr, s->shape_data + base, 16);
        const uint64_t verts = base + 48, norms = verts + (uint64_t)hdr[1] * 16, inds = norms + (uint64_t)hdr[2] * 16, texs = inds + (uint64_t)hdr[0] * 16;
        if (texs + (uint64_t)hdr[3] * 8 > s->shape_data_size || texs >= ((uint64_t)1 << 32))
            throw HipError{ IGD_ERR_INVALID_ARG, "igd_assign_scene: shape table is truncated or exceeds 4 GiB" };
        ext4[e] = make_uint4((uint32_t)verts, (uint32_t)norms, (uint32_t)inds, (uint32_t)texs);
    }
    d->entity_ext.upload(ext4.data(), ext4.size());

    // Per triangle of every mesh, what a hit on it reads (DevScene::prim_records): the three vertices, normals and texture
    // coordinates behind its index record, gathered into six consecutive 16-byte rows. A hit then goes entity -> record (one or two
    // cache lines) instead of entity -> indices -> ten lines of three attribute arrays: one dependent round trip and four load
    // instructions fewer per shaded vertex, the same values through the same arithmetic. 96 bytes per triangle of HBM.
    {
        std::vector<uint32_t> shape_rec(s->shape_count, 0xFFFFFFFFu);
        uint64_t rows = 0;
        for (uint32_t i = 0; i < s->shape_count; ++i) {
            if (s->shape_lookups[i].type_id == IG_SHAPE_SPHERE)
                continue;
            const uint64_t base = s->shape_lookups[i].offset;
            if (base + 48 > s->shape_data_size)
                throw HipError{ IGD_ERR_INVALID_ARG, "igd_assign_scene: shape offset out of range" };
            int32_t hdr[4];
            std::memcpy(hdr, s->shape_data + base, 16);
            if (hdr[0] < 0 || hdr[1] < 0 || hdr[2] < 0 || hdr[3] < 0)
                throw HipError{ IGD_ERR_INVALID_ARG, "igd_assign_scene: negative count in a shape header" };
            shape_rec[i] = (uint32_t)rows;
            rows += (uint64_t)hdr[0] * 6;
            if (rows >= ((uint64_t)1 << 32))
                throw HipError{ IGD_ERR_UNSUPPORTED, "igd_assign_scene: more than 715 million triangles in the scene's meshes" };
        }
        std::vector<float4> rec((size_t)rows);
        for (uint32_t i = 0; i < s->shape_count; ++i) {
            if (shape_rec[i] == 0xFFFFFFFFu)
                continue;
            const uint64_t base = s->shape_lookups[i].offset;
            int32_t hdr[4]; // faces, vertices, normals, texcoords
            std::memcpy(hdr, s->shape_data + base, 16);
            const uint64_t verts = base + 48, norms = verts + (uint64_t)hdr[1] * 16, inds = norms + (uint64_t)hdr[2] * 16, texs = inds + (uint64_t)hdr[0] * 16;
            if (texs + (uint64_t)hdr[3] * 8 > s->shape_data_size)
                throw HipError{ IGD_ERR_INVALID_ARG, "igd_assign_scene: shape table is truncated" };
            const uint8_t* sd = s->shape_data;
            float4* out       = rec.data() + shape_rec[i];
            for (int32_t f = 0; f < hdr[0]; ++f, out += 6) {
                int32_t tri[4];
                std::memcpy(tri, sd + inds + (uint64_t)f * 16, 16);
                float tex[3][2];
                for (int k = 0; k < 3; ++k) {
                    if (tri[k] < 0 || tri[k] >= hdr[1] || tri[k] >= hdr[2] || tri[k] >= hdr[3])
                        throw HipError{ IGD_ERR_INVALID_ARG, "igd_assign_scene: a triangle's vertex index is outside of its shape's vertex, normal or texture coordinate array" };
                    std::memcpy(&out[k], sd + verts + (uint64_t)tri[k] * 16, 12);
                    std::memcpy(&out[3 + k], sd + norms + (uint64_t)tri[k] * 16, 12);
                    std::memcpy(tex[k], sd + texs + (uint64_t)tri[k] * 8, 8);
                }
                out[0].w = tex[0][0], out[1].w = tex[0][1], out[2].w = tex[1][0], out[3].w = tex[1][1], out[4].w = tex[2][0], out[5].w = tex[2][1];
            }
        }
        d->prim_records.upload(rec.data(), rec.size());
        std::vector<uint32_t> er(s->entity_count);
        for (uint32_t e = 0; e < s->entity_count; ++e) {
            uint32_t shape_id;
            std::memcpy(&shape_id, s->entities + (size_t)e * IG_ENTITY_FLOATS + 33, 4);
            er[e] = shape_rec[shape_id];
        }
        d->entity_rec.upload(er.data(), er.size());
    }

    DevScene& ds            = d->dscene;
    ds.geom                 = d->geom.ptr;
    ds.scene_nodes_off      = q8 ? q8_scene_off : scene_nodes_off;
    d->scene_node8_off      = scene_nodes_off;
    ds.scene_node_count     = s->scene_node_count;
    ds.leaves               = d->dev_leaves.ptr;
    ds.leaf_scan            = d->dev_leaf_scan.ptr;
    ds.sphere_nodes_off     = sphere_nodes_off;
    ds.sphere_node_count    = s->sphere_node_count;
    ds.sphere_leaves        = d->dev_sphere_leaves.ptr;
    ds.sphere_leaf_scan     = d->dev_sphere_leaf_scan.ptr;
    ds.entities             = d->entities.ptr;
    ds.shape_data           = d->shape_data.ptr;
    ds.shape_offsets        = d->shape_offsets.ptr;
    ds.materials            = d->materials.ptr;
    ds.entity_material      = d->entity_material.ptr;
    ds.entity_ext           = d->entity_ext.ptr;
    ds.entity_rec           = d->entity_rec.ptr;
    ds.prim_records         = d->prim_records.ptr;
    ds.lights               = d->lights.ptr;
    ds.entity_count         = s->entity_count;
    ds.material_count       = s->material_count;
    ds.light_count          = s->light_count;
    ds.infinite_light_count = s->infinite_light_count;
    ds.tech                 = s->technique;
    ds.light_hierarchy      = d->light_hierarchy.ptr;
    ds.light_codes          = d->light_codes.ptr;
    ds.use_hierarchy        = hierarchy ? 1u : 0u;
    ds.light_cdf            = simple_selector ? d->light_cdf.ptr : nullptr;
    ds.media                = d->media.ptr;
    ds.media_count          = s->media_count;
    {
        // the shading kernel with the expression interpreter runs only where a material names a program
        bool any_expr = false;
        for (uint32_t i = 0; i < s->material_count; ++i)
            any_expr |= (s->materials[i].flags & (IG_MAT_EXPR_COLOR | IG_MAT_EXPR_NORMAL | IG_MAT_EXPR_WEIGHT | IG_MAT_EXPR_NUMBERS)) != 0 || s->materials[i].bsdf_type == IG_BSDF_RAD_BRTD
                        || s->materials[i].bsdf_type == IG_BSDF_RAD_ROOS; // the Radiance BSDFs live in that instantiation too
        ds.expr_code = any_expr ? d->expr_code.ptr : nullptr;
    }
    // (the 8 L2s hold 32 MB together; the switch sits at twice that, 64 MB, where the stand-in sweep put the break-even: below it most visits
    // still hit an L2 and the repeats run for too few lanes)
    // a BVH beyond the caches is better walked by ONE front of rays (what the XCDs' L2s and the Infinity Cache hold is then the same part of it);
    // such rays are slow enough for one counter (profiles/r05_experiment_ab.txt section 26)
    d->work_shards          = d->work_shards_env > 0 ? d->work_shards_env : (blob.size() > ((size_t)64 << 20) ? 1 : kWorkShards);
    // (off unless asked for: on the stand-ins the whole span between a random permutation of a bounce stream and camera-ray coherence is
    // 1.9 -> 2.6 G rays / s, the stream order k_shade leaves already sits at 2.26, the best key reaches 2.35 and the sort costs more than
    // that: profiles/r06_ray_order.txt, profiles/r06_experiment_ab.txt section 1)
    d->ray_sort             = d->ray_sort_env > 0;
    ds.node_repeat          = d->node_repeat >= 0 ? (uint32_t)d->node_repeat : (blob.size() > ((size_t)64 << 20) ? 3u : 0u);
    ds.scene_radius         = s->scene_radius;
    for (int k = 0; k < 3; ++k) {
        ds.scene_center[k] = s->bbox_min[k] + (s->bbox_max[k] - s->bbox_min[k]) * 0.5f; // bbox_center (core/bbox.art:22)
        d->scene_bbox[k] = s->bbox_min[k], d->scene_bbox[3 + k] = s->bbox_max[k];
    }
    ds.textures             = d->textures.ptr;
    ds.texture_data         = d->texture_data.ptr;
    ds.cdf_data             = d->cdf_data.ptr;
    {
        const uint32_t trav_lanes = (uint32_t)d->traverseGrid() * 256u;
        const uint32_t tail_lanes = (uint32_t)d->num_cus * (uint32_t)d->tail_waves_per_cu * 64u;
        d->deep_stack.alloc((size_t)(trav_lanes + tail_lanes * (uint32_t)d->n_flights) * (size_t)kDeepStack);
        ds.deep_stack     = d->deep_stack.ptr;
        ds.deep_stride    = trav_lanes + tail_lanes * (uint32_t)d->n_flights;
        d->tail_lanes     = tail_lanes;
        ds.deep_tail_base = trav_lanes;
    }
    d->camera               = s->camera;
    // scenes without a principled BSDF, textured environment or sun light run the lean shading kernels
    d->full_bsdfs = false;
    d->deep_primary = d->deep_primary_mode == 1; // a new scene starts with the LDS-stack kernels again
    for (uint32_t i = 0; i < s->material_count; ++i)
        d->full_bsdfs |= (s->materials[i].flags & (IG_MAT_DOUBLESIDED | IG_MAT_EXPR_COLOR | IG_MAT_EXPR_NORMAL | IG_MAT_EXPR_WEIGHT | IG_MAT_EXPR_NUMBERS)) != 0 || (s->materials[i].bsdf_type == IG_BSDF_DIFFUSE && s->materials[i].p[3] > 1.1920928955e-07f) || s->materials[i].bsdf_type == IG_BSDF_TRANSPARENT || s->materials[i].bsdf_type == IG_BSDF_PHONG || s->materials[i].bsdf_type == IG_BSDF_RAD_BRTD || s->materials[i].bsdf_type == IG_BSDF_RAD_ROOS || s->materials[i].bsdf_type == IG_BSDF_PRINCIPLED || s->materials[i].bsdf_type == IG_BSDF_PLASTIC || s->materials[i].bsdf_type == IG_BSDF_ROUGH_DIELECTRIC || s->materials[i].bsdf_type == IG_BSDF_BLEND
                         || (s->materials[i].bsdf_type == IG_BSDF_DIELECTRIC && (s->materials[i].flags & IG_MAT_THIN));
    {
        // one-row hits (pack_hit): entity ids in the upper bits (the all-ones word is the miss), prim ids below
        uint32_t eb = 1;
        while (eb < 32 && ((uint64_t)1 << eb) - 1 < (uint64_t)s->entity_count)
            ++eb;
        uint64_t max_faces = 0;
        for (uint32_t i = 0; i < s->shape_count; ++i)
            if (s->shape_lookups[i].type_id == IG_SHAPE_TRIMESH && s->shape_lookups[i].offset + 4 <= s->shape_data_size) {
                uint32_t faces;
                std::memcpy(&faces, s->shape_data + s->shape_lookups[i].offset, 4);
                max_faces = std::max<uint64_t>(max_faces, faces);
            }
        const uint32_t pb = 32 - eb;
        d->hit_pack_bits  = (d->hit_pack_allowed && eb < 31 && s->sphere_node_count == 0 && max_faces <= ((uint64_t)1 << pb)) ? pb : 0u;
    }
    d->shade_classes = 1u;
    {
        // the bins of the by-class kernels' sort (one per material + the misses') in class-major order, models of a class together
        auto cls = [&](uint32_t bin) {
            if (bin >= s->material_count)
                return 0; // misses: the basic class
            const int t = s->materials[bin].bsdf_type;
            return t == IG_BSDF_PRINCIPLED ? 1 : (t == IG_BSDF_PLASTIC || t == IG_BSDF_ROUGH_DIELECTRIC) ? 2 : t == IG_BSDF_BLEND ? 3 : 0;
        };
        for (uint32_t i = 0; i < s->material_count; ++i)
            d->shade_classes |= 1u << cls(i);
        std::vector<uint8_t> tables(2 * kSortBins, 0);
        if (s->material_count + 1 <= (uint32_t)kSortBins) {
            std::vector<uint32_t> order(s->material_count + 1);
            for (uint32_t i = 0; i <= s->material_count; ++i)
                order[i] = i;
            auto model = [&](uint32_t bin) { return bin >= s->material_count ? -1 : (int)s->materials[bin].bsdf_type; };
            std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return cls(a) != cls(b) ? cls(a) < cls(b) : model(a) < model(b); });
            for (uint32_t i = 0; i <= s->material_count; ++i) {
                tables[i]             = (uint8_t)order[i];
                tables[kSortBins + i] = (uint8_t)cls(i);
            }
            // on_miss (technique/pathtracer.art:141-168) sums over the infinite lights: without one a miss adds nothing and ends the
            // path — the sort leaves such rays out of every run (the path tracer and its volumetric variant; the AO renderer has no on_miss)
            if (s->infinite_light_count == 0)
                tables[kSortBins + s->material_count] = (uint8_t)kSortDeadBin;
        }
        d->sort_tables.upload(tables.data(), tables.size());
        const size_t sort_words = (size_t)kSortStateWords + (size_t)kSortBins * (size_t)d->sortGrid();
        d->sort_state.alloc(sort_words);
        HIP_CHECK(hipMemset(d->sort_state.ptr, 0, sort_words * sizeof(uint32_t)));
    }
    d->full_bsdfs |= s->sphere_node_count != 0; // surface elements of analytic spheres
    d->full_bsdfs |= s->technique.type == IG_TECHNIQUE_AO || s->technique.type == IG_TECHNIQUE_VOLPATH || s->technique.type == IG_TECHNIQUE_DEBUG;
    d->full_bsdfs |= simple_selector;
    if (s->technique.type != IG_TECHNIQUE_PATH && s->technique.type != IG_TECHNIQUE_AO && s->technique.type != IG_TECHNIQUE_VOLPATH && s->technique.type != IG_TECHNIQUE_DEBUG
        && s->technique.type != IG_TECHNIQUE_LIGHTTRACER && s->technique.type != IG_TECHNIQUE_WIREFRAME && s->technique.type != IG_TECHNIQUE_PPM)
        throw HipError{ IGD_ERR_UNSUPPORTED, "igd_assign_scene: unknown technique type" };
    if (s->technique.type == IG_TECHNIQUE_PPM) {
        // the light pass samples emission like the light tracer (lt_core.h: every light type); photons per iteration bounded by what a launch can index
        for (uint32_t i = 0; i < s->light_count; ++i)
            if (s->lights[i].type < IG_LIGHT_PLANE || s->lights[i].type > IG_LIGHT_PEREZ)
                throw HipError{ IGD_ERR_UNSUPPORTED, "igd_assign_scene: unknown light type" };
        if (s->technique.photon_count < 1 || s->technique.photon_count > (1 << 28) || s->technique.max_light_depth < 0 || !(s->technique.merge_radius >= 0))
            throw HipError{ IGD_ERR_INVALID_ARG, "igd_assign_scene: photon mapper parameters out of range" };
        if (s->sphere_node_count)
            throw HipError{ IGD_ERR_UNSUPPORTED, "igd_assign_scene: the photon mapper is not lowered for scenes with analytic spheres" };
        d->full_bsdfs = true;
    }
    if (s->technique.type == IG_TECHNIQUE_WIREFRAME) {
        if ((s->camera.type != IG_CAMERA_PERSPECTIVE && s->camera.type != IG_CAMERA_ORTHOGONAL) || s->sphere_node_count)
            throw HipError{ IGD_ERR_UNSUPPORTED, "igd_assign_scene: the wireframe technique is lowered for perspective / orthogonal cameras and triangle meshes" };
        d->full_bsdfs = true;
    }
    if (s->technique.type == IG_TECHNIQUE_LIGHTTRACER) {
        // Light::sample_emission exists for every light type, Camera::sample_pixel for the pinhole camera (lt_core.h)
        for (uint32_t i = 0; i < s->light_count; ++i)
            if (s->lights[i].type < IG_LIGHT_PLANE || s->lights[i].type > IG_LIGHT_PEREZ)
                throw HipError{ IGD_ERR_UNSUPPORTED, "igd_assign_scene: unknown light type" };
        if (s->camera.type != IG_CAMERA_PERSPECTIVE || s->camera.aperture_radius > 1.1920928955e-07f)
            throw HipError{ IGD_ERR_UNSUPPORTED, "igd_assign_scene: the light tracer connects to the perspective camera without depth of field only" };
        if (s->sphere_node_count)
            throw HipError{ IGD_ERR_UNSUPPORTED, "igd_assign_scene: the light tracer is not lowered for scenes with analytic spheres" };
        d->full_bsdfs = true;
    }
    for (uint32_t i = s->infinite_light_count; i < s->light_count; ++i) {
        if (s->lights[i].type != IG_LIGHT_MESH_AREA && s->lights[i].type != IG_LIGHT_SPHERE)
            continue;
        d->full_bsdfs = true;
        if (s->lights[i].entity_id < 0 || s->lights[i].entity_id >= (int32_t)s->entity_count)
            throw HipError{ IGD_ERR_INVALID_ARG, "igd_assign_scene: mesh / sphere area light without a valid entity" };
    }
    for (uint32_t i = 0; i < s->infinite_light_count; ++i)
        d->full_bsdfs |= s->lights[i].type == IG_LIGHT_ENV_TEXTURED || s->lights[i].type == IG_LIGHT_SUN || s->lights[i].type == IG_LIGHT_CIE || s->lights[i].type == IG_LIGHT_PEREZ;
    d->has_scene            = true;
    d->tail_share[0]        = 0; // (another scene: its tails decay at their own rate)
}

// Device-memory clears go through the render stream and are waited for: the render streams are non-blocking, i.e. not ordered
// against the null stream, and a null-stream hipMemset of device memory need not be finished when it returns.
void clearOnStream(hipStream_t st, void* ptr, int value, size_t bytes)
{
    HIP_CHECK(hipMemsetAsync(ptr, value, bytes, st));
    HIP_CHECK(hipStreamSynchronize(st));
}

void resizeFb(igd_device* d, int w, int h)
{
    const bool same = w == d->fb_w && h == d->fb_h && d->fb.ptr;
    if (!same) {
        d->fb.release();
        d->fb.alloc((size_t)w * h * 3);
        clearOnStream(d->stream, d->fb.ptr, 0, (size_t)w * h * 3 * sizeof(float));
        d->fb_w = w;
        d->fb_h = h;
        d->fb_host.assign((size_t)w * h * 3, 0.0f);
        d->fb_host_dirty = true;
    }
    for (int k = 0; k < 5; ++k) {
        const bool wanted = k == 4 ? (d->setup.info_aovs != 0 && d->denoised_wanted) : k < 2 ? d->setup.info_aovs != 0 : (d->has_scene && d->dscene.tech.aov_mis != 0);
        if (same && (wanted == (d->aov[k].ptr != nullptr)))
            continue; // (the MIS AOVs come and go with the scene's technique)
        d->aov[k].release();
        if (!wanted)
            continue;
        d->aov[k].alloc((size_t)w * h * 3);
        clearOnStream(d->stream, d->aov[k].ptr, 0, (size_t)w * h * 3 * sizeof(float));
        d->aov_host[k].assign((size_t)w * h * 3, 0.0f);
        d->aov_host_dirty[k] = true;
    }
}

// Polls the queue sizes of the chunk in flight on the main stream (64 bytes, pinned).
void readQueueState(igd_device* d, const QueueState* dev_qs, QueueState& out)
{
    QueueState* slot = d->host_store + igd_device::kMaxFlights + 2;
    HIP_CHECK(hipMemcpyAsync(slot, dev_qs, kQueueStateHead, hipMemcpyDeviceToHost, d->stream));
    HIP_CHECK(hipStreamSynchronize(d->stream));
    out = *slot;
}

void addSpan(igd_device* d, int kind, float ms)
{
    switch (kind) {
    case 0: d->stats.ms_generate += ms; break;
    case 1: d->stats.ms_traverse_primary += ms; break;
    case 2: d->stats.ms_shade += ms; break;
    case 3: d->stats.ms_traverse_secondary += ms; break;
    case 4: d->stats.ms_resolve += ms; break;
    case 6: d->stats.ms_ray_sort += ms; break;
    default: d->stats.ms_tail += ms; break;
    }
}

// Waits for the chunk that last used this flight slot (its tail + resolve run on the side stream), folds its
// counters and timers into the statistics and reports its errors. Errors of an overlapped chunk therefore
// surface at the next call that touches the device, not inside the igd_render that submitted it.
void collect(igd_device* d, igd_device::Flight& f)
{
    if (!f.pending)
        return;
    f.pending = false;
    HIP_CHECK(hipEventSynchronize(f.done));
    const QueueState& q = *f.host;
    d->stats.camera_rays += q.camera_rays;
    d->stats.bounce_rays += q.bounce_rays;
    d->stats.shadow_rays += q.shadow_rays;
    d->stats.unoccluded += q.unoccluded;
    d->stats.tail_rays += q.tail_rays;
    if (q.tail_pass_in[0]) { // (a chunk without a tail leaves the table as it is)
        for (int j = 0; j < 24; ++j)
            d->tail_share[j] = (double)q.tail_pass_in[j] / (double)q.tail_pass_in[0];
        static const bool tail_debug = std::getenv("IGD_TAIL_DEBUG") != nullptr;
        if (tail_debug) {
            std::fprintf(stderr, "[tail] paths at the start of the passes:");
            for (int j = 0; j < 24 && q.tail_pass_in[j]; ++j)
                std::fprintf(stderr, " %u", q.tail_pass_in[j]);
            std::fprintf(stderr, "\n");
        }
    }
    d->noteDeep(q.deep_total, q.camera_rays + q.bounce_rays + q.shadow_rays);
    d->stats.nodes_primary += q.nodes[0], d->stats.nodes_secondary += q.nodes[1];
    d->stats.tris_primary += q.tris[0], d->stats.tris_secondary += q.tris[1];
    d->stats.leaves_primary += q.leaves[0], d->stats.leaves_secondary += q.leaves[1];
    for (int k = 0; k < 6; ++k)
        d->stats.section_passes[k] += q.section_passes[k], d->stats.section_lanes[k] += q.section_lanes[k];
    for (const auto& sp : f.spans) {
        float ms = 0;
        HIP_CHECK(hipEventElapsedTime(&ms, sp.second.first, sp.second.second));
        addSpan(d, sp.first, ms);
    }
    f.spans.clear();
    if (q.error_flags & 1u)
        throw HipError{ IGD_ERR_DEVICE, "traversal stack overflow (BVH deeper than the LDS stack)" };
}

// Drains both streams; afterwards framebuffer, statistics and every buffer are quiescent.
void render(igd_device* d, const igd_render_settings* rs);

// Executes the recorded batch of iterations, if any.
void flushPending(igd_device* d)
{
    if (!d->pending.active)
        return;
    d->pending.active      = false; // first: render() may call finish(), which calls this function
    igd_render_settings rs = d->pending.rs;
    rs.iterations          = d->pending.count;
    render(d, &rs);
}

void finish(igd_device* d)
{
    flushPending(d);
    HipError first{ IGD_OK, "" };
    for (int k = 0; k < d->n_flights; ++k) {
        // oldest chunk first
        igd_device::Flight& f = d->flight[(d->chunk_seq + (uint64_t)k) % (uint64_t)d->n_flights];
        try {
            collect(d, f);
        } catch (const HipError& e) {
            if (first.code == IGD_OK)
                first = e;
        }
    }
    HIP_CHECK(hipStreamSynchronize(d->stream));
    for (int k = 0; k < d->n_flights; ++k)
        HIP_CHECK(hipStreamSynchronize(d->side[k]));
    d->events_used = 0;
    if (first.code != IGD_OK)
        throw first;
}

// What igd_render checks before it accepts (and possibly defers) a request.
void validateSettings(const igd_device* d, const igd_render_settings* rs)
{
    if (!d->has_scene)
        throw HipError{ IGD_ERR_NO_SCENE, "igd_render: no scene assigned" };
    if (rs->spi <= 0 || rs->width <= 0 || rs->height <= 0)
        throw HipError{ IGD_ERR_INVALID_ARG, "igd_render: spi, width and height must be positive" };
    const int row_stride = rs->row_stride > 0 ? rs->row_stride : 1;
    if (rs->row_offset < 0 || rs->row_offset >= row_stride)
        throw HipError{ IGD_ERR_INVALID_ARG, "igd_render: row_offset must be in [0, row_stride)" };
    if (rs->rays != nullptr && rs->height != 1)
        throw HipError{ IGD_ERR_INVALID_ARG, "igd_render: ray-list mode expects width = #rays, height = 1" };
    const int64_t rows = (rs->height - rs->row_offset + row_stride - 1) / row_stride;
    if (rows * rs->width * rs->spi * std::max(1, rs->iterations) >= ((int64_t)1 << 31))
        throw HipError{ IGD_ERR_INVALID_ARG, "igd_render: width * height * spi * iterations must stay below 2^31" };
}

// Accepts a request: runs it now, or records it as the continuation of the pending batch.
void submit(igd_device* d, const igd_render_settings* rs)
{
    validateSettings(d, rs);
    const int cnt = std::max(1, rs->iterations);
    if (d->setup.blocking_render) { // the reference's contract: complete, errors included, when the call returns
        flushPending(d);
        render(d, rs);
        finish(d);
        return;
    }
    if (rs->rays != nullptr || d->setup.is_interactive || d->batch_rays == 0) {
        flushPending(d);
        render(d, rs);
        return;
    }
    const int row_stride = rs->row_stride > 0 ? rs->row_stride : 1;
    const int64_t per_it = (int64_t)((rs->height - rs->row_offset + row_stride - 1) / row_stride) * rs->width * rs->spi;
    igd_device::Pending& p = d->pending;
    if (p.active) {
        const igd_render_settings& q = p.rs;
        const bool continues = q.spi == rs->spi && q.width == rs->width && q.height == rs->height && q.frame == rs->frame && q.user_seed == rs->user_seed
                               && q.row_offset == rs->row_offset && (q.row_stride > 0 ? q.row_stride : 1) == row_stride && rs->iteration == q.iteration + p.count;
        const int64_t merged = per_it * (p.count + cnt);
        if (continues && (uint64_t)merged <= d->batch_rays && merged < ((int64_t)1 << 31)) {
            p.count += cnt;
            if ((uint64_t)(per_it * (p.count + 1)) > d->batch_rays)
                flushPending(d); // the next iteration would not fit: run the batch now rather than at the next call
            return;
        }
        flushPending(d);
    }
    p.active = true;
    p.rs     = *rs;
    p.count  = cnt;
    if ((uint64_t)(per_it * (cnt + 1)) > d->batch_rays)
        flushPending(d);
}

void render(igd_device* d, const igd_render_settings* rs)
{
    validateSettings(d, rs);
    const int row_stride = rs->row_stride > 0 ? rs->row_stride : 1;
    const int row_offset = rs->row_offset;
    const bool list_mode = rs->rays != nullptr;

    const auto t_start = std::chrono::steady_clock::now();
    if (rs->width != d->fb_w || rs->height != d->fb_h || !d->fb.ptr)
        finish(d); // the side stream may still be resolving into the old framebuffer
    resizeFb(d, rs->width, rs->height);

    const int local_rows  = (rs->height - row_offset + row_stride - 1) / row_stride;
    const int iterations  = list_mode ? 1 : std::max(1, rs->iterations);
    const int64_t per_it  = (int64_t)local_rows * rs->width * rs->spi; // rays of one iteration
    const int64_t total   = per_it * iterations;
    if (total >= ((int64_t)1 << 31))
        throw HipError{ IGD_ERR_INVALID_ARG, "igd_render: width * height * spi * iterations must stay below 2^31" };
    if (d->streamsTooSmall((size_t)std::max<int64_t>(total, 256)))
        finish(d);
    d->ensureStreams((size_t)std::max<int64_t>(total, 256));

    if (d->dscene.sphere_node_count && d->secondary_hit.count < d->capacity * 4) {
        finish(d);
        d->secondary_hit.release();
        d->secondary_hit.alloc(d->capacity * 4);
    }
    if (list_mode)
        d->list_rays.upload(rs->rays, (size_t)rs->width * 8);
    // the by-class shading kernels (shade_kernel.h) and the global sort that feeds them: the path tracer's full variant, when the
    // bins fit the sort's key byte (launch_shade takes the other instantiations first: light tracer, debug views, expressions)
    const bool by_class = d->shade_by_class && d->full_bsdfs && d->dscene.material_count + 1 <= (uint32_t)kSortBins && !d->dscene.expr_code
                          && (d->dscene.tech.type == IG_TECHNIQUE_PATH || d->dscene.tech.type == IG_TECHNIQUE_AO || d->dscene.tech.type == IG_TECHNIQUE_VOLPATH);
    if (by_class && d->sort_capacity < d->capacity) {
        finish(d);
        d->sort_idx.release();
        d->sort_keys.release();
        d->sort_idx.alloc(d->capacity);
        d->sort_keys.alloc(d->capacity);
        d->sort_capacity = d->capacity;
    }

    const bool ray_sort = d->ray_sort && !list_mode && d->dscene.tech.type != IG_TECHNIQUE_LIGHTTRACER && d->dscene.tech.type != IG_TECHNIQUE_PPM;
    if (ray_sort && d->rs_capacity < d->capacity) {
        finish(d);
        for (int k = 0; k < 2; ++k) {
            d->rs_keys[k].release(), d->rs_idx[k].release();
            d->rs_keys[k].alloc(d->capacity), d->rs_idx[k].alloc(d->capacity);
        }
        d->rs_state.alloc(256 + (size_t)256 * (size_t)d->sortGrid());
        d->rs_capacity = d->capacity;
    }

    // the per-film constants of the camera, evaluated on the host
    float sx, sy;
    {
        const ig_camera& c = d->camera;
        const float asp    = (float)rs->width / (float)rs->height;
        // aspect = width / height unless the scene fixes it (PerspectiveCamera.cpp:41-45, OrthogonalCamera.cpp:33-35)
        const float aspect = c.aspect_ratio > 0 ? c.aspect_ratio : asp;
        if (c.type == IG_CAMERA_FISHLENS) {
            // (xasp, yasp) of make_fishlens_camera (camera/fishlens.art:12-37)
            if (c.fisheye_mode == IG_FISHEYE_CROPPED) {
                sx = asp < 1 ? 1 / asp : 1;
                sy = asp > 1 ? 1 / asp : 1;
            } else if (c.fisheye_mode == IG_FISHEYE_FULL) {
                const float diameter = std::sqrt(asp * asp + 1) * (float)rs->height;
                const float f        = diameter / (float)std::min(rs->width, rs->height);
                sx                   = asp < 1 ? f : f / asp;
                sy                   = asp > 1 ? f : f * asp;
            } else {
                sx = asp < 1 ? 1 : asp;
                sy = asp > 1 ? 1 : asp;
            }
        } else if (c.type == IG_CAMERA_ORTHOGONAL) {
            // make_vec2(camera_scale, camera_scale / aspect) (OrthogonalCamera.cpp:46)
            sx = c.scale;
            sy = c.scale / aspect;
        } else if (c.fov_is_vertical) {
            // compute_scale_from_vfov / _hfov (camera/perspective.art:2-13)
            sy = std::tan(c.fov / 2);
            sx = sy * aspect;
        } else {
            sx = std::tan(c.fov / 2);
            sy = sx / aspect;
        }
    }

    const bool stats    = d->setup.acquire_stats != 0;
    const bool counters = d->setup.acquire_stats >= 2;
    hipStream_t st      = d->stream;
    const float inv     = 1 / (float)rs->spi;
    if (d->events_used > 8192)
        finish(d); // keeps the timer event pool bounded when nobody asks for the statistics

    // Chunk the call's ray ids so that every pixel's samples stay together and a chunk is either a number of whole
    // iterations (resolved per pixel, iteration after iteration) or at most one iteration's worth of pixels (all distinct).
    int64_t chunk_rays = ((int64_t)d->capacity / rs->spi) * rs->spi;
    if (chunk_rays <= 0)
        throw HipError{ IGD_ERR_INVALID_ARG, "igd_render: stream capacity is smaller than spi" };
    if (per_it > 0 && chunk_rays >= per_it)
        chunk_rays = (chunk_rays / per_it) * per_it;
    const bool light_tracer = d->dscene.tech.type == IG_TECHNIQUE_LIGHTTRACER;
    const bool ppm          = d->dscene.tech.type == IG_TECHNIQUE_PPM;
    if (ppm) {
        if (list_mode)
            throw HipError{ IGD_ERR_UNSUPPORTED, "igd_render: the photon mapper renders camera films only (no ray lists)" };
        chunk_rays = std::min(chunk_rays, per_it); // a chunk never spans two iterations: each has its own photon map
    }
    PpmArgs ppm_args{};
    // The light pass of iteration `iteration` and its query structure (photonmapper.art: variant 0, then ppm_handle_before_iteration_camera),
    // stream-ordered in front of that iteration's camera wavefront. Plain rounds: closest-hit traversal + k_shade<light pass>, no shadow
    // rays, the live count read back after every round (a light path ends at its first non-delta surface: a handful of rounds).
    auto photonPass = [&](int iteration) {
        const ig_technique& tech = d->dscene.tech;
        const uint32_t P         = (uint32_t)tech.photon_count;
        d->ppm_photons.alloc(P);
        d->ppm_sorted.alloc(P);
        d->ppm_keys.alloc((size_t)P * 2);
        d->ppm_cell_count.alloc(IGP_GRID_CELLS + 1);
        d->ppm_cell_offset.alloc(IGP_GRID_CELLS + 1);
        d->ppm_valid.alloc(1);
        d->ppm_qs.alloc(1);
        const size_t temp_bytes = photon_grid_temp_bytes(P);
        if (temp_bytes == 0)
            throw HipError{ IGD_ERR_DEVICE, "igd_render: the photon grid's sort / scan could not be sized" };
        d->ppm_temp.alloc(temp_bytes);
        QueueState* lq = d->ppm_qs.ptr;
        HIP_CHECK(hipMemsetAsync(d->ppm_photons.ptr, 0xFF, (size_t)P * sizeof(igp_photon), st)); // light = -1: no photon
        ppm_args.pass         = 1;
        ppm_args.photons      = d->ppm_photons.ptr;
        ppm_args.cell_offset  = d->ppm_cell_offset.ptr;
        ppm_args.photon_count = (int32_t)P;
        ppm_args.valid_count  = 0;
        ppm_args.radius       = igp_compute_radius(tech.merge_radius, iteration);
        for (int k = 0; k < 3; ++k)
            ppm_args.bbox_min[k] = d->scene_bbox[k], ppm_args.bbox_max[k] = d->scene_bbox[3 + k];
        ShadeFrame lframe{ (int32_t)P, 1, iteration, rs->frame, rs->user_seed, 0, 1, (int32_t)P, 0.0f };
        lframe.finish();
        const uint32_t lchunk = (uint32_t)std::min<size_t>(d->capacity, P);
        for (uint32_t lfirst = 0; lfirst < P; lfirst += lchunk) {
            const uint32_t ln = std::min(lchunk, P - lfirst);
            HIP_CHECK(hipMemsetAsync(lq, 0, sizeof(QueueState), st));
            GenerateLightArgs gl{};
            gl.scene     = d->dscene;
            gl.out       = d->primaryCols(0);
            gl.out_count = &lq->q[0].primary;
            gl.qs        = lq;
            gl.width     = (int32_t)P;
            gl.spi       = 1;
            gl.iteration = iteration;
            gl.frame     = rs->frame;
            gl.seed      = rs->user_seed;
            gl.first_local_id     = lfirst;
            gl.rays_per_iteration = (int32_t)P;
            gl.n                  = ln;
            gl.ppm                = 1;
            launch_generate_light(gl, st);
            int lslot = 0;
            for (int round = 0; round < tech.max_light_depth + 2; ++round) {
                const PrimaryCols in = d->primaryCols(lslot);
                TraverseArgs ta{};
                ta.scene = d->dscene;
                ta.rayA = in.rayA, ta.rayB = in.rayB, ta.meta = in.meta;
                if (round > 0) { // a stream k_shade wrote: bounce rays all (its meta.y carries eta, its rayB.w the seed: kernels.h kStream*)
                    ta.meta             = nullptr;
                    ta.uniform_flags    = IG_RAY_FLAG_BOUNCE;
                    ta.uniform_tmax     = std::numeric_limits<float>::max();
                    ta.use_uniform_tmax = 1;
                }
                ta.count        = &lq->q[lslot].primary;
                ta.work_counter = &lq->work[0].w[0][0];
                ta.index_list   = d->deep_rays.ptr;
                ta.index_count  = &lq->deep_count;
                ta.qs           = lq;
                ta.hit = in.hit, ta.hit_v = in.hit_v;
                ta.sphere_work_counter = &lq->work[4].w[0][0];
                launchTraverse(d, ta, false, counters, d->traverseGrid(), &lq->work[1].w[0][0], st, d->deep_grid, d->deep_primary);
                ShadeArgs sa{};
                sa.scene     = d->dscene;
                sa.in        = in;
                sa.out       = d->primaryCols(lslot ^ 1);
                sa.sec       = d->secondaryCols();
                sa.in_count  = &lq->q[lslot].primary;
                sa.out_count = &lq->q[lslot ^ 1].primary;
                sa.qs        = lq;
                sa.accum     = reinterpret_cast<float4*>(d->flight[0].accum.ptr); // (the light pass never splats)
                sa.id_base   = 0;
                sa.frame     = lframe;
                sa.inv_spi   = 1;
                sa.ppm       = ppm_args;
                sa.in_kind   = round > 0 ? kStreamShaded : kStreamLight;
                launch_shade_ppm(sa, d->shadeGrid(), st);
                launch_round_end(lq, lslot, st);
                lslot ^= 1;
                uint32_t alive = 0;
                HIP_CHECK(hipMemcpyAsync(&alive, &lq->q[lslot].primary, sizeof(alive), hipMemcpyDeviceToHost, st));
                HIP_CHECK(hipStreamSynchronize(st));
                if (alive == 0)
                    break;
            }
            QueueState host{};
            HIP_CHECK(hipMemcpyAsync(&host, lq, kQueueStateHead, hipMemcpyDeviceToHost, st));
            HIP_CHECK(hipStreamSynchronize(st));
            if (host.error_flags & 1u)
                throw HipError{ IGD_ERR_DEVICE, "traversal stack overflow (BVH deeper than the LDS stack)" };
            d->stats.camera_rays += host.camera_rays; // the emitter's rays count as the variant's camera rays (mapping_gpu.art:763)
            d->stats.bounce_rays += host.bounce_rays;
            d->stats.nodes_primary += host.nodes[0], d->stats.tris_primary += host.tris[0], d->stats.leaves_primary += host.leaves[0];
        }
        HIP_CHECK(build_photon_grid(d->ppm_photons.ptr, P, ppm_args, d->ppm_sorted.ptr, d->ppm_cell_count.ptr, d->ppm_cell_offset.ptr, d->ppm_keys.ptr, d->ppm_valid.ptr,
                                    d->ppm_temp.ptr, temp_bytes, st));
        HIP_CHECK(hipMemcpyAsync(&d->ppm_valid_host, d->ppm_valid.ptr, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipStreamSynchronize(st));
        HIP_CHECK(hipGetLastError());
        ppm_args.pass        = 2;
        ppm_args.photons     = d->ppm_sorted.ptr;
        ppm_args.valid_count = (int32_t)d->ppm_valid_host;
    };
    if (light_tracer && (row_stride != 1 || row_offset != 0 || list_mode || chunk_rays < per_it))
        // a connection lands in any pixel of the film: its accumulator slot has to exist in the chunk that traces the path
        throw HipError{ IGD_ERR_UNSUPPORTED, "igd_render: the light tracer needs the whole film in one wavefront (no row sharding, no ray lists, stream capacity >= width * height * spi)" };

    if (light_tracer && d->lt_capacity < d->capacity) {
        // the connections' deterministic splat (launch_lt_splat): path ids, verdicts, sort buffers for one round's shadow rays
        finish(d);
        d->lt_path_id.release(), d->lt_vals.release(), d->lt_keys.release(), d->lt_temp.release();
        d->lt_path_id.alloc(d->capacity);
        d->lt_vals.alloc(2 * d->capacity);
        d->lt_keys.alloc(2 * d->capacity);
        const size_t tb = lt_splat_temp_bytes((uint32_t)std::min<size_t>(d->capacity, 0x7FFFFFFFu));
        if (tb == 0)
            throw HipError{ IGD_ERR_DEVICE, "igd_render: the light tracer's connection sort could not be sized" };
        d->lt_temp.alloc(tb);
        d->lt_capacity = d->capacity;
        if (d->secondary_hit.count < d->capacity * 4) {
            d->secondary_hit.release();
            d->secondary_hit.alloc(d->capacity * 4);
        }
    }
    // (the light tracer has no camera-flagged rays, so the wrapper below never splats for it: "Normals" / "Albedo" stay zero)
    if (d->setup.info_aovs && !list_mode && rs->iteration == 0 && !light_tracer) {
        // wrap_infobuffer_renderer (technique/internal/infobuffer.art:4-30): normals and albedo of the camera rays' first hits of
        // iteration 0. Run as a pass of its own in front of the wavefront — the same camera rays (same RNG), closest-hit
        // traversal, k_info, per-pixel sums in sample order — so the shading kernels do not carry it.
        const int64_t info_chunk = ((int64_t)d->capacity / rs->spi) * rs->spi;
        d->info_qs.alloc(1);
        d->info_tmp[0].alloc((size_t)std::min<int64_t>(info_chunk, per_it) * 4);
        d->info_tmp[1].alloc((size_t)std::min<int64_t>(info_chunk, per_it) * 4);
        QueueState* iq = d->info_qs.ptr;
        for (int64_t first = 0; first < per_it; first += info_chunk) {
            const uint32_t n = (uint32_t)std::min<int64_t>(info_chunk, per_it - first);
            HIP_CHECK(hipMemsetAsync(iq, 0, sizeof(QueueState), st));
            GenerateArgs ga{};
            ga.out            = d->primaryCols(0);
            ga.out_count      = &iq->q[0].primary;
            ga.qs             = iq;
            ga.cam            = d->camera;
            ga.sx = sx, ga.sy = sy;
            ga.width = rs->width, ga.height = rs->height, ga.spi = rs->spi;
            ga.iteration = 0, ga.frame = rs->frame, ga.seed = rs->user_seed;
            ga.row_offset = row_offset, ga.row_stride = row_stride;
            ga.first_local_id     = first;
            ga.rays_per_iteration = (int32_t)std::max<int64_t>(per_it, 1);
            ga.n                  = n;
            launch_generate(ga, st);
            const PrimaryCols in = d->primaryCols(0);
            TraverseArgs ta{};
            ta.scene = d->dscene;
            ta.rayA = in.rayA, ta.rayB = in.rayB, ta.meta = in.meta;
            ta.count        = &iq->q[0].primary;
            ta.work_counter = &iq->work[0].w[0][0];
            ta.index_list   = d->deep_rays.ptr;
            ta.index_count  = &iq->deep_count;
            ta.qs           = iq;
            ta.hit = in.hit, ta.hit_v = in.hit_v;
            ta.sphere_work_counter = &iq->work[4].w[0][0];
            launchTraverse(d, ta, false, false, d->traverseGrid(), &iq->work[1].w[0][0], st, d->deep_grid, d->deep_primary);
            InfoArgs ia{};
            ia.scene   = d->dscene;
            ia.in      = in;
            ia.count   = &iq->q[0].primary;
            ia.normals = reinterpret_cast<float4*>(d->info_tmp[0].ptr);
            ia.albedo  = reinterpret_cast<float4*>(d->info_tmp[1].ptr);
            ia.id_base = first;
            ia.inv_spi = inv;
            launch_info(ia, d->num_cus * 8, st);
            for (int k = 0; k < 2; ++k) {
                ResolveArgs ra{};
                ra.accum             = reinterpret_cast<const float4*>(d->info_tmp[k].ptr);
                ra.fb                = d->aov[k].ptr;
                ra.width             = rs->width;
                ra.spi               = rs->spi;
                ra.row_offset        = row_offset;
                ra.row_stride        = row_stride;
                ra.first_local_pixel = first / rs->spi;
                ra.pixels            = n / (uint32_t)rs->spi;
                ra.local_pixels      = (uint32_t)(per_it / rs->spi);
                ra.iterations        = 1;
                launch_resolve(ra, st);
                d->aov_host_dirty[k] = true;
            }
        }
        // the camera-ray counter of this extra pass is not part of the render's statistics (info_qs is never collected)
    }

    d->fb_host_dirty = true;
    int64_t step = 0;
    for (int64_t first = 0; first < total; first += step) {
        step = std::min<int64_t>(chunk_rays, total - first);
        // the photon mapper: a chunk ends at its iteration's end even when the capacity does not divide an iteration — each iteration
        // has its own photon map and merge radius, and its light pass runs in front of the chunk that starts it
        if (ppm && per_it > 0)
            step = std::min<int64_t>(step, per_it - first % per_it);
        const uint32_t n = (uint32_t)step;

        // this chunk's flight slot: wait (host side) until the chunk n_flights back has left it
        const int slot         = (int)(d->chunk_seq % (uint64_t)d->n_flights);
        igd_device::Flight& fl = d->flight[slot];
        igd_device::Flight& prev = d->flight[(slot + d->n_flights - 1) % d->n_flights];
        hipStream_t side       = d->side[slot];
        collect(d, fl);
        ++d->chunk_seq;
        QueueState* qs = fl.qs;
        float4* accum  = reinterpret_cast<float4*>(fl.accum.ptr);
        // "Direct Weights" / "NEE Weights" (ig_technique.aov_mis): two more accumulators per sample, resolved like the colour
        const bool mis_aovs = d->dscene.tech.aov_mis != 0;
        float4* accum_mis[2] = { nullptr, nullptr };
        if (mis_aovs)
            for (int k = 0; k < 2; ++k) {
                fl.accum_mis[k].alloc((size_t)n * 4);
                accum_mis[k] = reinterpret_cast<float4*>(fl.accum_mis[k].ptr);
            }

        auto timed = [&](int kind, hipStream_t on, const std::function<void()>& fn) {
            if (!stats) {
                fn();
                return;
            }
            const hipEvent_t e0 = d->nextEvent(), e1 = d->nextEvent();
            HIP_CHECK(hipEventRecord(e0, on));
            fn();
            HIP_CHECK(hipEventRecord(e1, on));
            fl.spans.push_back({ kind, { e0, e1 } });
        };

        fl.pending = true; // from here on the slot owns work that collect() has to wait for
        HIP_CHECK(hipEventRecord(fl.done, st)); // placeholder so that an early throw leaves a valid event
        // (the camera emitter clears sample i's accumulator as it writes ray i: one pass less over 16 B per sample)
        const bool clear_in_generate = !light_tracer && d->clear_in_generate;
        if (!clear_in_generate)
            HIP_CHECK(hipMemsetAsync(fl.accum.ptr, 0, (size_t)n * 4 * sizeof(float), st));
        for (int k = 0; k < 2; ++k)
            if (accum_mis[k])
                HIP_CHECK(hipMemsetAsync(accum_mis[k], 0, (size_t)n * 4 * sizeof(float), st));
        HIP_CHECK(hipMemsetAsync(qs, 0, sizeof(QueueState), st));

        if (ppm && per_it > 0 && first % per_it == 0)
            photonPass(rs->iteration + (int)(first / per_it));

        int in_slot = 0;
        GenerateArgs ga{};
        ga.out            = d->primaryCols(in_slot);
        ga.out_count      = &qs->q[in_slot].primary;
        ga.qs             = qs;
        ga.cam            = d->camera;
        ga.sx             = sx;
        ga.sy             = sy;
        ga.width          = rs->width;
        ga.height         = rs->height;
        ga.spi            = rs->spi;
        ga.iteration      = rs->iteration;
        ga.frame          = rs->frame;
        ga.seed           = rs->user_seed;
        ga.row_offset     = row_offset;
        ga.row_stride     = row_stride;
        ga.first_local_id = first;
        ga.rays_per_iteration = (int32_t)std::max<int64_t>(per_it, 1);
        ga.n              = n;
        ga.list_rays      = list_mode ? d->list_rays.ptr : nullptr;
        // a camera whose rays all leave one point writes only the directions (kernels.h CameraStream)
        CameraStream cam_stream{};
        {
            const ig_camera& c = d->camera;
            const bool one_point = (c.type == IG_CAMERA_PERSPECTIVE && !(c.aperture_radius > 1.1920928955e-07f)) || (c.type == IG_CAMERA_FISHLENS && !c.fisheye_mask);
            cam_stream.compact     = d->camera_compact && one_point && !list_mode && !light_tracer ? 1 : 0;
            cam_stream.rnd_counter = c.pixel_sampler == IG_PIXEL_SAMPLER_HALTON ? 1u : 3u; // Tea{seed, 1} after the sampler's two draws (none for Halton)
            cam_stream.first_id    = first;
            cam_stream.rayA        = make_float4(c.eye[0], c.eye[1], c.eye[2], c.near_clip);
        }
        ga.compact = cam_stream.compact;
        ga.accum_clear = clear_in_generate ? accum : nullptr;
        if (light_tracer) {
            GenerateLightArgs gl{};
            gl.scene     = d->dscene;
            gl.out       = ga.out;
            gl.out_count = ga.out_count;
            gl.qs        = qs;
            gl.width     = rs->width;
            gl.spi       = rs->spi;
            gl.iteration = rs->iteration;
            gl.frame     = rs->frame;
            gl.seed      = rs->user_seed;
            gl.first_local_id     = first;
            gl.rays_per_iteration = ga.rays_per_iteration;
            gl.n                  = n;
            timed(0, st, [&] { launch_generate_light(gl, st); });
        } else {
            timed(0, st, [&] { launch_generate(ga, st); });
        }

        float wire_footprint = 0;
        if (d->dscene.tech.type == IG_TECHNIQUE_WIREFRAME) {
            // camera.differential: (right scale.x, up scale.y) (camera/perspective.art:59-64), (right, up) (orthogonal.art:38-43);
            // right = normalize(dir x up); footprint_u = |dx x dy| with vec3_len = sqrt of the fma chain of vec3_dot
            const ig_camera& c = d->camera;
            const float r[3]   = { c.dir[1] * c.up[2] - c.dir[2] * c.up[1], c.dir[2] * c.up[0] - c.dir[0] * c.up[2], c.dir[0] * c.up[1] - c.dir[1] * c.up[0] };
            const float rl     = 1 / std::sqrt(std::fma(r[0], r[0], std::fma(r[1], r[1], r[2] * r[2])));
            const float kx = c.type == IG_CAMERA_ORTHOGONAL ? 1.0f : sx, ky = c.type == IG_CAMERA_ORTHOGONAL ? 1.0f : sy;
            const float dx[3] = { (r[0] * rl) * kx, (r[1] * rl) * kx, (r[2] * rl) * kx }, dy[3] = { c.up[0] * ky, c.up[1] * ky, c.up[2] * ky };
            const float cr[3] = { dx[1] * dy[2] - dx[2] * dy[1], dx[2] * dy[0] - dx[0] * dy[2], dx[0] * dy[1] - dx[1] * dy[0] };
            wire_footprint    = std::sqrt(std::fma(cr[0], cr[0], std::fma(cr[1], cr[1], cr[2] * cr[2])));
        }
        ShadeFrame frame{ rs->width, rs->spi, rs->iteration, rs->frame, rs->user_seed, row_offset, row_stride, (int32_t)std::max<int64_t>(per_it, 1), wire_footprint };
        frame.finish();
        uint32_t live          = n;
        bool run_tail          = false;
        int tail_from_round    = 0; // rounds submitted before the hand-over to the tail: 0 = its input is what k_generate wrote
        // One bounce round on stream `on` over the given stream buffers: closest-hit traversal (K2) -> sort +
        // shade + compact (K3, K4, K5, K9) -> any-hit traversal of the shadow rays + splat (K6).
        struct RoundBufs {
            PrimaryCols prim[2];
            SecondaryCols sec;
            uint32_t* deep_rays;
            float4* sec_hit;
        };
        uint32_t round_bound = n; // upper bound of the round's primary stream, hence of its shadow rays (light tracer: what its connection sort covers)
        // raysort.hip: the order a traversal launch takes the stream in
        auto sortedOrder = [&](hipStream_t on, const float4* rayA, const float4* rayB, const uint32_t* count) -> const uint32_t* {
            RaySortArgs ra{};
            ra.rayA = rayA, ra.rayB = rayB, ra.count = count;
            const float cells = (float)(1u << d->ray_sort_bits);
            for (int k = 0; k < 3; ++k) {
                const float ext = d->scene_bbox[3 + k] - d->scene_bbox[k];
                ra.box_min[k]   = d->scene_bbox[k];
                ra.box_scale[k] = ext > 0 ? cells / ext : 0.0f;
            }
            ra.cell_bits  = (uint32_t)d->ray_sort_bits;
            ra.octant_low = (uint32_t)d->ray_sort_octant_low;
            ra.dir_bits   = (uint32_t)std::min(d->ray_sort_dir_bits, (32 - 3 * d->ray_sort_bits) / 2);
            for (int k = 0; k < 2; ++k)
                ra.keys[k] = d->rs_keys[k].ptr, ra.idx[k] = d->rs_idx[k].ptr;
            ra.state   = d->rs_state.ptr;
            ra.wg_hist = d->rs_state.ptr + 256;
            const uint32_t* order = nullptr;
            timed(6, on, [&] { order = launch_ray_sort(ra, d->sortGrid(), on); });
            return order;
        };
        auto launchRound = [&](hipStream_t on, const RoundBufs& b, int in_slot, int trav_grid, int shade_grid, QueueState* mirror, bool bounce_rays_only) {
            const PrimaryCols in = b.prim[in_slot];
            const bool sort_round = ray_sort && bounce_rays_only && round_bound >= d->ray_sort_min;
            TraverseArgs ta{};
            ta.scene = d->dscene;
            ta.rayA = in.rayA, ta.rayB = in.rayB, ta.meta = in.meta;
            if (bounce_rays_only) {
                // every ray k_shade appends carries IG_RAY_FLAG_BOUNCE (shade_kernel.h): from round 1 on the traversal need not read the
                // meta column for the visibility flags (16 of the 48 bytes it reads per ray)
                ta.meta             = nullptr;
                ta.uniform_flags    = IG_RAY_FLAG_BOUNCE;
                ta.uniform_tmax     = std::numeric_limits<float>::max(); // (FLT_MAX, what k_shade used to write into rayB.w; the path's seed travels there now)
                ta.use_uniform_tmax = 1;
            } else if (cam_stream.compact) {
                // round 0 of a compact camera stream: rayB is all there is
                ta.rayA          = nullptr;
                ta.uniform_rayA  = cam_stream.rayA;
                ta.meta          = nullptr;
                ta.uniform_flags = IG_RAY_FLAG_CAMERA;
            }
            ta.count        = &qs->q[in_slot].primary;
            ta.work_counter = &qs->work[0].w[0][0];
            ta.index_list   = b.deep_rays;
            ta.index_count  = &qs->deep_count;
            ta.qs           = qs;
            ta.hit = in.hit, ta.hit_v = in.hit_v;
            ta.hit_pack = d->hit_pack_bits;
            ta.sphere_work_counter = &qs->work[4].w[0][0];
            if (sort_round && (d->ray_sort_which & 1))
                ta.sort_idx = sortedOrder(on, in.rayA, in.rayB, &qs->q[in_slot].primary);
            timed(1, on, [&] { launchTraverse(d, ta, false, counters, trav_grid, &qs->work[1].w[0][0], on, d->deep_grid, d->deep_primary); });

            ShadeArgs sa{};
            sa.scene     = d->dscene;
            sa.in        = in;
            sa.out       = b.prim[in_slot ^ 1];
            sa.sec       = b.sec;
            sa.in_count  = &qs->q[in_slot].primary;
            sa.out_count = &qs->q[in_slot ^ 1].primary;
            sa.qs        = qs;
            sa.accum     = accum;
            sa.id_base   = first;
            sa.frame     = frame;
            sa.inv_spi   = inv;
            sa.accum_direct = accum_mis[0];
            if (light_tracer) {
                // make_perspective_camera (camera/perspective.art:29-31): view = (normalize(dir x up), up, dir)
                const ig_camera& c = d->camera;
                const float r[3]   = { c.dir[1] * c.up[2] - c.dir[2] * c.up[1], c.dir[2] * c.up[0] - c.dir[0] * c.up[2], c.dir[0] * c.up[1] - c.dir[1] * c.up[0] };
                const float rl     = 1 / std::sqrt(std::fma(r[0], r[0], std::fma(r[1], r[1], r[2] * r[2])));
                for (int k = 0; k < 3; ++k) {
                    sa.lt_cam.eye[k]      = c.eye[k];
                    sa.lt_cam.view[k]     = r[k] * rl;
                    sa.lt_cam.view[3 + k] = c.up[k];
                    sa.lt_cam.view[6 + k] = c.dir[k];
                }
                sa.lt_cam.sx = sx, sa.lt_cam.sy = sy;
                sa.lt_cam.width = rs->width, sa.lt_cam.height = rs->height;
            }
            sa.ppm = ppm_args;
            sa.hit_pack = d->hit_pack_bits;
            sa.in_kind  = bounce_rays_only ? kStreamShaded : (light_tracer ? kStreamLight : kStreamCamera);
            sa.cam_stream = cam_stream;
            sa.skip_misses = d->skip_misses && d->dscene.infinite_light_count == 0 && (d->dscene.tech.type == IG_TECHNIQUE_PATH || d->dscene.tech.type == IG_TECHNIQUE_VOLPATH) ? 1 : 0;
            timed(2, on, [&] {
                if (ppm)
                    launch_shade_ppm(sa, shade_grid, on);
                else {
                    if (by_class && d->shade_classes == 1u && !d->sort_single_class) {
                        // every material of the scene is of the basic class: its kernel takes the stream as it lies (sa.sort_idx stays null) —
                        // no sort, no index indirection
                    } else if (by_class) {
                        // K3: the round's hits sorted by material, so that each class kernel shades its own dense run
                        BinSortArgs ba{};
                        ba.hit             = in.hit;
                        ba.hit_pack        = d->hit_pack_bits;
                        ba.count           = &qs->q[in_slot].primary;
                        ba.entity_material = d->dscene.entity_material;
                        ba.material_count  = d->dscene.material_count;
                        ba.keys            = d->sort_keys.ptr;
                        ba.sort_idx        = d->sort_idx.ptr;
                        ba.state           = d->sort_state.ptr;
                        ba.wg_hist         = d->sort_state.ptr + kSortStateWords;
                        ba.bin_order       = d->sort_tables.ptr;
                        ba.bin_class       = d->sort_tables.ptr + kSortBins;
                        launch_bin_sort(ba, d->sortGrid(), on);
                        sa.sort_idx  = d->sort_idx.ptr;
                        sa.cls_range = d->sort_state.ptr;
                    }
                    launch_shade(sa, shade_grid, d->full_bsdfs, on, by_class ? d->shade_classes : 0u);
                }
                launch_round_end(qs, in_slot, on);
            });

            TraverseArgs tb{};
            tb.scene = d->dscene;
            tb.rayA = b.sec.rayA, tb.rayB = b.sec.rayB, tb.meta = nullptr;
            tb.uniform_flags = d->dscene.tech.type == IG_TECHNIQUE_AO ? IG_RAY_FLAG_BOUNCE : IG_RAY_FLAG_SHADOW; // aotracer.art:13
            tb.count         = &qs->q[in_slot ^ 1].secondary; // generated by this round's k_shade
            tb.work_counter  = &qs->work[2].w[0][0];
            tb.index_list    = b.deep_rays;
            tb.index_count   = &qs->deep_count;
            tb.qs            = qs;
            tb.col     = b.sec.col;
            tb.hit     = b.sec_hit; // only with a sphere pass (null otherwise)
            tb.sphere_work_counter = &qs->work[5].w[0][0];
            tb.accum   = accum;
            tb.accum_nee = accum_mis[1];
            tb.id_base = first;
            tb.inv_spi = inv;
            if (light_tracer) {
                // K6 records verdicts only; K7 / K8 below add the unoccluded connections into their pixels' slots in a fixed order
                tb.accum = nullptr;
                tb.hit   = reinterpret_cast<float4*>(d->secondary_hit.ptr);
            }
            if (ray_sort && round_bound >= d->ray_sort_min && (d->ray_sort_which & 2))
                tb.sort_idx = sortedOrder(on, b.sec.rayA, b.sec.rayB, &qs->q[in_slot ^ 1].secondary);
            timed(3, on, [&] {
                launchTraverse(d, tb, true, counters, trav_grid, &qs->work[3].w[0][0], on, d->deep_grid, d->deep_primary);
                if (light_tracer)
                    HIP_CHECK(launch_lt_splat(b.sec.col, reinterpret_cast<const float4*>(d->secondary_hit.ptr), b.sec.path_id, &qs->q[in_slot ^ 1].secondary, round_bound,
                                              d->lt_keys.ptr, d->lt_vals.ptr, d->lt_temp.ptr, d->lt_temp.count, accum, first, inv, on));
                launch_secondary_end(qs, in_slot ^ 1, mirror, on);
            });
            d->stats.rounds++;
            d->stats.traverse_primary_launches++;
            d->stats.traverse_secondary_launches++;
        };

        const RoundBufs main_bufs{ { d->primaryCols(0), d->primaryCols(1) }, d->secondaryCols(), d->deep_rays.ptr,
                                   d->dscene.sphere_node_count ? reinterpret_cast<float4*>(d->secondary_hit.ptr) : nullptr };
        // The host runs one round ahead of what it knows: the queue sizes after round r are copied back
        // asynchronously and looked at only after round r + 1 has been submitted, so the stream never drains while
        // the host waits. The hand-over decision therefore uses the size one round old — an upper bound of the
        // current one (a stream never grows), which is all the tail's buffers and grid need. A round submitted
        // on an already empty stream costs a few empty launches.
        uint32_t known_live = n; // upper bound of the stream size after the last submitted round
        for (int round = 0;; ++round) {
            if (known_live == 0)
                break;
            // (the tail kernels keep one accumulator per path and have no debug views: such scenes run their rounds to the end instead)
            if (known_live <= d->tail_threshold && !mis_aovs && d->dscene.tech.type != IG_TECHNIQUE_DEBUG && d->dscene.tech.type != IG_TECHNIQUE_LIGHTTRACER && d->dscene.tech.type != IG_TECHNIQUE_WIREFRAME && !ppm && !d->dscene.expr_code) {
                live            = known_live;
                run_tail        = true;
                tail_from_round = round;
                break;
            }
            // size after this round -> pinned slot (round & 1), written by the round's last kernel itself
            round_bound = known_live;
            launchRound(st, main_bufs, in_slot, d->traverseGrid(), d->shadeGrid(), d->host_store_dev + igd_device::kMaxFlights + (round & 1), round > 0);
            in_slot ^= 1;
            HIP_CHECK(hipEventRecord(d->poll_event[round & 1], st));
            if (round >= 1) {
                const int r = round - 1;
                HIP_CHECK(hipEventSynchronize(d->poll_event[r & 1]));
                const QueueState& hq = d->host_store[igd_device::kMaxFlights + (r & 1)];
                if (hq.error_flags & 1u)
                    break; // reported by collect()
                known_live = hq.q[(r & 1) ^ 1].primary; // in_slot after round r (starts at 0, flips every round)
                d->noteDeep(hq.deep_total, (unsigned long long)n + hq.bounce_rays + hq.shadow_rays);
            }
            live = known_live;
            if (round > d->dscene.tech.max_depth + 8)
                throw HipError{ IGD_ERR_DEVICE, "igd_render: wavefront loop did not terminate" };
        }

        // ---- second half of the chunk, on the side stream: the next chunk's rounds start meanwhile
        TailArgs tl{};
        int tail_grid = 0;
        if (run_tail) {
            // few paths left: follow each to its end in one launch instead of ~50 more rounds. Its input is
            // moved out of the primary stream, which the next chunk overwrites.
            d->ensureTailInput(fl, live);
            const PrimaryCols keep = igd_device::colsAt(fl.tail_in.ptr, fl.tail_capacity);
            launch_copy_paths(d->primaryCols(in_slot), keep, &qs->q[in_slot].primary, live, tail_from_round == 0 ? cam_stream : CameraStream{}, st);
            tl.scene        = d->dscene;
            tl.in           = keep;
            tl.in_count     = &qs->q[in_slot].primary;
            tl.work_counter = nullptr; // set per pass
            tl.qs           = qs;
            tl.accum        = accum;
            tl.id_base      = first;
            tl.frame        = frame;
            tl.inv_spi      = inv;
            tl.count_paths  = 1;
            tl.wide_lanes   = (uint32_t)d->tail_wide;
            tl.wide8_lanes  = (uint32_t)d->tail_wide8;
            tl.in_kind      = tail_from_round > 0 ? kStreamShaded : kStreamCamera; // (the light tracer never runs the tail)
            tl.deep_lane_base = d->dscene.deep_tail_base + (uint32_t)slot * d->tail_lanes; // concurrent tails: own columns
            fl.tail_ctr.alloc(2 * kMaxTailPasses);
            HIP_CHECK(hipMemsetAsync(fl.tail_ctr.ptr, 0, 2 * kMaxTailPasses * sizeof(uint32_t), st));
            // one-wave workgroups, 2 waves/SIMD (VGPR bound) = 8 per CU; fewer when the stream is tiny
            tail_grid = std::max(1, std::min(d->num_cus * d->tail_waves_per_cu, (int)((live + 63) / 64)));
        }
        HIP_CHECK(hipEventRecord(fl.rounds_done, st));
        HIP_CHECK(hipStreamWaitEvent(side, fl.rounds_done, 0));
        if (run_tail)
            timed(5, side, [&] {
                // pass j reads buffer j & 1 and appends its survivors to the other one; the last pass is unbounded
                const int depth_left = std::max(1, d->dscene.tech.max_depth);
                const int passes     = d->tail_split > 0 ? std::min(kMaxTailPasses, (depth_left + d->tail_split - 1) / d->tail_split) : 1;
                const PrimaryCols other  = igd_device::colsAt(fl.tail_long.ptr, fl.tail_capacity);
                const PrimaryCols buf[2] = { tl.in, other };
                for (int j = 0; j < passes; ++j) {
                    TailArgs p     = tl;
                    p.in           = buf[j & 1];
                    p.out          = buf[(j & 1) ^ 1];
                    p.in_count     = j == 0 ? tl.in_count : fl.tail_ctr.ptr + 2 * (j - 1);
                    p.out_count    = fl.tail_ctr.ptr + 2 * j;
                    p.work_counter = fl.tail_ctr.ptr + 2 * j + 1;
                    p.max_bounces  = j + 1 < passes ? d->tail_split : 0;
                    p.count_paths  = j == 0;
                    p.pass         = j;
                    p.in_kind      = j == 0 ? tl.in_kind : kStreamShaded;
                    int pass_grid  = tail_grid;
                    if (d->tail_adapt && j > 0 && d->tail_share[0] > 0)
                        pass_grid = std::max(64, std::min(tail_grid, (int)std::ceil(d->tail_share[j] * (double)live / d->tail_density)));
                    static const bool tail_debug = std::getenv("IGD_TAIL_DEBUG") != nullptr;
                    if (tail_debug)
                        std::fprintf(stderr, "[tail] pass %d: share %.6f live %llu grid %d of %d\n", j, d->tail_share[j], (unsigned long long)live, pass_grid, tail_grid);
                    (d->q8_nodes ? launch_tail_q8 : launch_tail)(p, counters, d->full_bsdfs, pass_grid, side);
                }
            });

        ResolveArgs ra{};
        ra.accum             = accum;
        ra.fb                = d->fb.ptr;
        ra.width             = rs->width;
        ra.spi               = rs->spi;
        ra.row_offset        = row_offset;
        ra.row_stride        = row_stride;
        ra.first_local_pixel = first / rs->spi;
        ra.pixels            = n / (uint32_t)rs->spi;
        ra.local_pixels      = (uint32_t)std::max<int64_t>(per_it / rs->spi, 1);
        ra.iterations        = (per_it > 0 && (int64_t)n > per_it) ? (uint32_t)((int64_t)n / per_it) : 1u; // whole iterations, see the chunking above
        // framebuffer updates stay in chunk order (pixels of successive iterations coincide, and the float sum
        // must not depend on which tail finished first)
        if (prev.used && &prev != &fl)
            HIP_CHECK(hipStreamWaitEvent(side, prev.resolved, 0));
        timed(4, side, [&] { launch_resolve(ra, side); });
        for (int k = 0; k < 2; ++k)
            if (accum_mis[k]) {
                ResolveArgs rm = ra;
                rm.accum       = accum_mis[k];
                rm.fb          = d->aov[2 + k].ptr;
                launch_resolve(rm, side);
                d->aov_host_dirty[2 + k] = true;
            }
        HIP_CHECK(hipEventRecord(fl.resolved, side));
        fl.used = true;
        HIP_CHECK(hipMemcpyAsync(fl.host, qs, kQueueStateHead, hipMemcpyDeviceToHost, side));
        HIP_CHECK(hipEventRecord(fl.done, side));
    }
    HIP_CHECK(hipGetLastError());

    if (!d->async_tail || list_mode)
        finish(d);
    d->stats.ms_total += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count();
}

void traverseList(igd_device* d, int64_t count, const float* rays, uint32_t ray_flags, int any_hit,
                  int32_t* ent_id, int32_t* prim_id, float* t, float* u, float* v, int repeat, double* kernel_ms)
{
    if (!d->has_scene)
        throw HipError{ IGD_ERR_NO_SCENE, "igd_traverse: no scene assigned" };
    if (count < 0 || count >= ((int64_t)1 << 31) || (count > 0 && !rays))
        throw HipError{ IGD_ERR_INVALID_ARG, "igd_traverse: bad ray list" };
    if (kernel_ms)
        *kernel_ms = 0;
    if (count == 0)
        return;
    const size_t n = (size_t)count;

    // host list (8 floats per ray) -> rayA = (org, tmin), rayB = (dir, tmax) columns in HBM
    std::vector<float> cols(n * 8);
    for (size_t i = 0; i < n; ++i) {
        const float* r = rays + i * 8;
        float* ra      = cols.data() + i * 4;
        float* rb      = cols.data() + n * 4 + i * 4;
        ra[0] = r[0], ra[1] = r[1], ra[2] = r[2], ra[3] = r[6];
        rb[0] = r[3], rb[1] = r[4], rb[2] = r[5], rb[3] = r[7];
    }
    DevBuf<float> in, out;
    DevBuf<uint32_t> deep_rays;
    deep_rays.alloc(n);
    in.upload(cols.data(), cols.size());
    out.alloc(n * 5);
    clearOnStream(d->stream, out.ptr, 0xFF, n * 5 * sizeof(float));

    finish(d);
    hipStream_t st = d->stream;
    QueueState* qs = d->flight[0].qs;
    const uint32_t cnt = (uint32_t)n;
    HIP_CHECK(hipMemsetAsync(qs, 0, sizeof(QueueState), st));
    HIP_CHECK(hipMemcpyAsync(&qs->q[0].primary, &cnt, 4, hipMemcpyHostToDevice, st));

    TraverseArgs ta{};
    ta.scene         = d->dscene;
    ta.rayA          = reinterpret_cast<const float4*>(in.ptr);
    ta.rayB          = reinterpret_cast<const float4*>(in.ptr + n * 4);
    ta.meta          = nullptr;
    ta.uniform_flags = ray_flags;
    ta.count         = &qs->q[0].primary;
    ta.work_counter  = &qs->work[0].w[0][0];
    ta.index_list    = deep_rays.ptr;
    ta.index_count   = &qs->deep_count;
    ta.qs            = qs;
    ta.hit           = reinterpret_cast<float4*>(out.ptr);
    ta.hit_v         = out.ptr + n * 4;
    ta.sphere_work_counter = &qs->work[any_hit ? 5 : 4].w[0][0];

    const bool stats = d->setup.acquire_stats >= 2;
    if (repeat < 1)
        repeat = 1;
    float total_ms = 0;
    for (int r = 0; r < repeat; ++r) {
        HIP_CHECK(hipMemsetAsync(&qs->work[0], 0, sizeof(qs->work), st));
        HIP_CHECK(hipMemsetAsync(&qs->deep_count, 0, sizeof(qs->deep_count), st));
        HIP_CHECK(hipEventRecord(d->event(0), st));
        launchTraverse(d, ta, any_hit != 0, stats && r == 0, d->traverseGrid(), &qs->work[1].w[0][0], st, d->deep_grid, d->deep_primary);
        HIP_CHECK(hipEventRecord(d->event(1), st));
        HIP_CHECK(hipStreamSynchronize(st));
        float ms = 0;
        HIP_CHECK(hipEventElapsedTime(&ms, d->event(0), d->event(1)));
        total_ms += ms;
    }
    HIP_CHECK(hipGetLastError());
    if (kernel_ms)
        *kernel_ms = total_ms / repeat;

    QueueState host_qs;
    readQueueState(d, qs, host_qs);
    if (host_qs.error_flags & 1u)
        throw HipError{ IGD_ERR_DEVICE, "igd_traverse: traversal stack overflow (BVH deeper than the LDS stack)" };
    d->stats.nodes_primary += host_qs.nodes[0], d->stats.nodes_secondary += host_qs.nodes[1];
    d->stats.tris_primary += host_qs.tris[0], d->stats.tris_secondary += host_qs.tris[1];
    d->stats.leaves_primary += host_qs.leaves[0], d->stats.leaves_secondary += host_qs.leaves[1];
    d->stats.unoccluded += host_qs.unoccluded;
    if (any_hit)
        d->stats.traverse_secondary_launches += (uint64_t)repeat;
    else
        d->stats.traverse_primary_launches += (uint64_t)repeat;

    std::vector<float> host(n * 5);
    HIP_CHECK(hipMemcpy(host.data(), out.ptr, host.size() * sizeof(float), hipMemcpyDeviceToHost));
    for (size_t i = 0; i < n; ++i) {
        const float* h = host.data() + i * 4;
        if (ent_id)
            std::memcpy(&ent_id[i], &h[0], 4);
        if (prim_id)
            std::memcpy(&prim_id[i], &h[1], 4);
        if (!any_hit) {
            if (t)
                t[i] = h[2];
            if (u)
                u[i] = h[3];
            if (v)
                v[i] = host[n * 4 + i];
        }
    }
}

int guarded(const char* what, const std::function<void()>& fn)
{
    g_error.clear();
    try {
        fn();
        return IGD_OK;
    } catch (const HipError& e) {
        g_error = std::string(what) + ": " + e.msg;
        return e.code;
    } catch (const std::exception& e) {
        g_error = std::string(what) + ": " + e.what();
        return IGD_ERR_DEVICE;
    }
}

bool isColorName(const char* name) { return !name || !*name || std::strcmp(name, "Color") == 0 || std::strcmp(name, "Default") == 0; }

// The film-sized buffer a framebuffer accessor addresses by name: the colour buffer or, with igd_setup.info_aovs, "Normals" /
// "Albedo" (Device.cpp:1385-1451 looks AOVs up by name and logs unknown ones).
struct FilmBuffer {
    DevBuf<float>* dev        = nullptr;
    std::vector<float>* host  = nullptr;
    bool* dirty               = nullptr;
};
FilmBuffer filmBuffer(igd_device* d, const char* name)
{
    if (isColorName(name))
        return FilmBuffer{ &d->fb, &d->fb_host, &d->fb_host_dirty };
    if (d->setup.info_aovs) {
        if (std::strcmp(name, "Normals") == 0)
            return FilmBuffer{ &d->aov[0], &d->aov_host[0], &d->aov_host_dirty[0] };
        if (std::strcmp(name, "Albedo") == 0)
            return FilmBuffer{ &d->aov[1], &d->aov_host[1], &d->aov_host_dirty[1] };
        if (std::strcmp(name, "Denoised") == 0) {
            if (!d->aov[4].ptr && d->fb.ptr) { // first access: one more film-sized buffer, zeroed
                HIP_CHECK(hipSetDevice(d->setup.gpu_index));
                d->aov[4].alloc((size_t)d->fb_w * d->fb_h * 3);
                clearOnStream(d->stream, d->aov[4].ptr, 0, (size_t)d->fb_w * d->fb_h * 3 * sizeof(float));
                d->aov_host[4].assign((size_t)d->fb_w * d->fb_h * 3, 0.0f);
                d->aov_host_dirty[4] = true;
            }
            d->denoised_wanted = true;
            return FilmBuffer{ &d->aov[4], &d->aov_host[4], &d->aov_host_dirty[4] };
        }
    }
    if (d->has_scene && d->dscene.tech.aov_mis) { // PathTechnique.cpp:24-25
        if (std::strcmp(name, "Direct Weights") == 0)
            return FilmBuffer{ &d->aov[2], &d->aov_host[2], &d->aov_host_dirty[2] };
        if (std::strcmp(name, "NEE Weights") == 0)
            return FilmBuffer{ &d->aov[3], &d->aov_host[3], &d->aov_host_dirty[3] };
    }
    throw HipError{ IGD_ERR_INVALID_ARG, std::string("unknown AOV '") + name + "'" };
}

} // namespace

namespace {
struct NamedBuffer {
    void* ptr      = nullptr;
    uint64_t bytes = 0;
};
// device-resident tables under the names the reference uses for them (SceneDatabase tables, film buffers)
NamedBuffer namedBuffer(igd_device* d, const char* name)
{
    if (!name)
        return {};
    const std::string n(name);
    auto of = [](auto& b) { return NamedBuffer{ b.ptr, (uint64_t)b.count * sizeof(*b.ptr) }; };
    if (n == "entities")
        return of(d->entities);
    if (n == "shapes")
        return NamedBuffer{ d->shape_data.ptr, d->shape_data.count ? (uint64_t)d->shape_data.count - 64 : 0 }; // (64 bytes of tail padding)
    if (n == "trimesh_primbvh") {
        // the device keeps the table with its triangle packets re-ordered (igd_assign_scene); whoever asks for the named table gets
        // the reference's bytes, rebuilt once per scene
        if (d->geom.ptr && d->primbvh_bytes && !d->geom_ref.ptr) {
            std::vector<uint8_t> host(d->primbvh_bytes);
            HIP_CHECK(hipMemcpy(host.data(), d->geom.ptr, host.size(), hipMemcpyDeviceToHost));
            for (const auto& span : d->tri_spans) { // (byte offset of the first packet, packets)
                for (uint32_t t = 0; t < span.second; ++t) {
                    float dev[48], ref[48];
                    uint8_t* at = host.data() + span.first + (size_t)t * sizeof(ig_tri4);
                    std::memcpy(dev, at, sizeof(dev));
                    for (int h = 0; h < 2; ++h)
                        for (int k = 0; k < 12; ++k)
                            for (int j = 0; j < 2; ++j)
                                ref[k * 4 + 2 * h + j] = dev[h * 24 + k * 2 + j];
                    std::memcpy(at, ref, sizeof(ref));
                }
            }
            d->geom_ref.upload(host.data(), host.size());
        }
        return NamedBuffer{ d->geom_ref.ptr, d->geom_ref.ptr ? (uint64_t)d->primbvh_bytes : 0 };
    }
    if (n == "scene_bvh_nodes")
        return NamedBuffer{ d->geom.ptr ? d->geom.ptr + d->scene_node8_off : nullptr, (uint64_t)d->dscene.scene_node_count * sizeof(ig_node8) };
    if (n == "scene_bvh_leaves")
        return of(d->leaves);
    if (n == "materials")
        return of(d->materials);
    if (n == "lights")
        return of(d->lights);
    if (n == "Color" || n == "Default")
        return of(d->fb);
    if (d->setup.info_aovs && n == "Normals")
        return of(d->aov[0]);
    if (d->setup.info_aovs && n == "Albedo")
        return of(d->aov[1]);
    if (d->setup.info_aovs && n == "Denoised")
        return of(d->aov[4]); // (empty until a framebuffer accessor has asked for it)
    if (d->dscene.tech.aov_mis && n == "Direct Weights")
        return of(d->aov[2]);
    if (d->dscene.tech.aov_mis && n == "NEE Weights")
        return of(d->aov[3]);
    return {};
}
} // namespace

extern "C" {

uint32_t igd_get_abi_version(void) { return IGD_ABI_VERSION; }

int32_t igd_device_count(void)
{
    g_error.clear();
    int n = 0;
    const hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        g_error = std::string("hipGetDeviceCount: ") + hipGetErrorString(e);
        return 0;
    }
    int usable = 0;
    for (int i = 0; i < n; ++i) {
        hipDeviceProp_t p;
        if (hipGetDeviceProperties(&p, i) == hipSuccess && std::strncmp(p.gcnArchName, "gfx950", 6) == 0)
            ++usable;
    }
    if (usable == 0)
        g_error = "no gfx950 (MI355X) device visible";
    return usable;
}

igd_device* igd_create(const igd_setup* setup)
{
    igd_device* dev = nullptr;
    const int rc    = guarded("igd_create", [&] {
        if (!setup)
            throw HipError{ IGD_ERR_INVALID_ARG, "setup is NULL" };
        int n = 0;
        if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
            throw HipError{ IGD_ERR_NO_DEVICE, "no HIP device visible; the HIP backend has no CPU fallback" };
        if (setup->gpu_index < 0 || setup->gpu_index >= n)
            throw HipError{ IGD_ERR_NO_DEVICE, "gpu_index out of range" };
        hipDeviceProp_t p;
        HIP_CHECK(hipGetDeviceProperties(&p, setup->gpu_index));
        if (std::strncmp(p.gcnArchName, "gfx950", 6) != 0)
            throw HipError{ IGD_ERR_NO_DEVICE, std::string("device is ") + p.gcnArchName + ", kernels are built for gfx950 only" };
        HIP_CHECK(hipSetDevice(setup->gpu_index));
        auto d     = std::make_unique<igd_device>();
        d->setup   = *setup;
        d->num_cus = p.multiProcessorCount;
        HIP_CHECK(hipStreamCreateWithFlags(&d->stream, hipStreamNonBlocking));
        {
            int lo = 0, hi = 0;
            HIP_CHECK(hipDeviceGetStreamPriorityRange(&lo, &hi));
            if (const char* e = std::getenv("IGD_FLIGHTS"))
                d->n_flights = std::min(igd_device::kMaxFlights, std::max(2, std::atoi(e)));
            // The tails of a long render finish sooner on a high-priority queue, and the rounds they run under lose nothing by it:
            // 10 891 / 10 917 / 10 944 Mrays/s at 256 steps for low / normal / high (three runs each, A/B section 28); no difference for
            // a single wavefront. IGD_SIDE_PRIORITY=low|normal|high.
            int prio = hi;
            if (const char* e = std::getenv("IGD_SIDE_PRIORITY"))
                prio = std::strcmp(e, "high") == 0 ? hi : std::strcmp(e, "normal") == 0 ? (lo + hi) / 2 : lo;
            for (int k = 0; k < d->n_flights; ++k)
                HIP_CHECK(hipStreamCreateWithPriority(&d->side[k], hipStreamNonBlocking, prio));
        }
        constexpr int F = igd_device::kMaxFlights;
        d->qs_store.alloc(F);
        clearOnStream(d->stream, d->qs_store.ptr, 0, F * sizeof(QueueState));
        HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&d->host_store), (F + 3) * sizeof(QueueState), hipHostMallocMapped));
        HIP_CHECK(hipHostGetDevicePointer(reinterpret_cast<void**>(&d->host_store_dev), d->host_store, 0));
        std::memset(d->host_store, 0, (F + 3) * sizeof(QueueState));
        for (auto& e : d->poll_event)
            HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        for (int k = 0; k < d->n_flights; ++k) {
            d->flight[k].qs   = d->qs_store.ptr + k;
            d->flight[k].host = d->host_store + k;
            HIP_CHECK(hipEventCreateWithFlags(&d->flight[k].rounds_done, hipEventDisableTiming));
            HIP_CHECK(hipEventCreateWithFlags(&d->flight[k].resolved, hipEventDisableTiming));
            HIP_CHECK(hipEventCreateWithFlags(&d->flight[k].done, hipEventDisableTiming));
        }
        if (const char* e = std::getenv("IGD_SHADE_GRID"))
            d->shade_mult = std::max(1, std::atoi(e));
        if (const char* e = std::getenv("IGD_DEEP_GRID"))
            d->deep_grid = std::max(1, std::atoi(e));
        if (const char* e = std::getenv("IGD_DEEP_PRIMARY")) { // 0 never (default), 1 always, -1 / auto: from the overflow counts
            d->deep_primary_mode = (e[0] == 'a' || e[0] == '-') ? -1 : (std::atoi(e) ? 1 : 0);
            d->deep_primary      = d->deep_primary_mode == 1;
        }
        if (const char* e = std::getenv("IGD_BATCH_RAYS"))
            d->batch_rays = std::strtoull(e, nullptr, 10);
        if (const char* e = std::getenv("IGD_TAIL_THRESHOLD"))
            d->tail_threshold = (uint32_t)std::strtoul(e, nullptr, 10);
        if (const char* e = std::getenv("IGD_TAIL_WAVES"))
            d->tail_waves_per_cu = std::max(1, std::atoi(e));
        if (const char* e = std::getenv("IGD_TAIL_SPLIT"))
            d->tail_split = std::max(0, std::atoi(e));
        if (const char* e = std::getenv("IGD_NODE_FORMAT")) // "full": always Node8 records; anything else: auto
            d->node_format_mode = std::strcmp(e, "full") == 0 || std::strcmp(e, "0") == 0 ? 0 : -1;
        if (const char* e = std::getenv("IGD_NODE_REPEAT"))
            d->node_repeat = std::min(16, std::atoi(e));
        if (const char* e = std::getenv("IGD_RAY_SORT"))
            d->ray_sort_env = std::atoi(e) != 0 ? 1 : 0;
        if (const char* e = std::getenv("IGD_RAY_SORT_BITS"))
            d->ray_sort_bits = std::max(1, std::min(9, std::atoi(e)));
        if (const char* e = std::getenv("IGD_RAY_SORT_ORDER"))
            d->ray_sort_octant_low = std::strcmp(e, "cell") == 0 ? 1 : 0;
        if (const char* e = std::getenv("IGD_RAY_SORT_DIR_BITS"))
            d->ray_sort_dir_bits = std::max(0, std::min(8, std::atoi(e)));
        if (const char* e = std::getenv("IGD_RAY_SORT_MIN"))
            d->ray_sort_min = (uint32_t)std::max(0, std::atoi(e));
        if (const char* e = std::getenv("IGD_RAY_SORT_STREAMS"))
            d->ray_sort_which = std::atoi(e) & 3;
        if (const char* e = std::getenv("IGD_WORK_SHARDS"))
            d->work_shards_env = std::atoi(e) > 1 ? kWorkShards : 1;
        if (const char* e = std::getenv("IGD_CLEAR_IN_GENERATE"))
            d->clear_in_generate = std::atoi(e) != 0;
        if (const char* e = std::getenv("IGD_CAMERA_COMPACT"))
            d->camera_compact = std::atoi(e) != 0;
        if (const char* e = std::getenv("IGD_HIT_PACK"))
            d->hit_pack_allowed = std::atoi(e) != 0;
        if (const char* e = std::getenv("IGD_SKIP_MISSES"))
            d->skip_misses = std::atoi(e) != 0;
        if (const char* e = std::getenv("IGD_SORT_SINGLE_CLASS"))
            d->sort_single_class = std::atoi(e) != 0;
        if (const char* e = std::getenv("IGD_SHADE_CLASSES"))
            d->shade_by_class = std::atoi(e) != 0;
        if (const char* e = std::getenv("IGD_TAIL_WIDE"))
            d->tail_wide = std::min(64, std::max(0, std::atoi(e)));
        if (const char* e = std::getenv("IGD_TAIL_WIDE8"))
            d->tail_wide8 = std::max(0, std::min(64, std::atoi(e)));
        if (const char* e = std::getenv("IGD_TAIL_ADAPT"))
            d->tail_adapt = std::atoi(e) != 0;
        if (const char* e = std::getenv("IGD_TAIL_DENSITY"))
            d->tail_density = std::min(64.0, std::max(0.125, std::atof(e)));
        if (const char* e = std::getenv("IGD_ASYNC_TAIL"))
            d->async_tail = std::atoi(e) != 0;
        dev = d.release();
    });
    if (rc != IGD_OK) {
        delete dev;
        return nullptr;
    }
    return dev;
}

void igd_destroy(igd_device* dev) { delete dev; }

int32_t igd_assign_scene(igd_device* dev, const igd_scene* scene)
{
    return guarded("igd_assign_scene", [&] {
        if (!dev || !scene)
            throw HipError{ IGD_ERR_INVALID_ARG, "NULL argument" };
        HIP_CHECK(hipSetDevice(dev->setup.gpu_index));
        finish(dev);
        dev->has_scene = false;
        assignScene(dev, scene);
    });
}

int32_t igd_render(igd_device* dev, const igd_render_settings* settings)
{
    return guarded("igd_render", [&] {
        if (!dev || !settings)
            throw HipError{ IGD_ERR_INVALID_ARG, "NULL argument" };
        HIP_CHECK(hipSetDevice(dev->setup.gpu_index));
        submit(dev, settings);
    });
}

int32_t igd_resize(igd_device* dev, int32_t width, int32_t height)
{
    return guarded("igd_resize", [&] {
        if (!dev || width <= 0 || height <= 0)
            throw HipError{ IGD_ERR_INVALID_ARG, "bad size" };
        HIP_CHECK(hipSetDevice(dev->setup.gpu_index));
        finish(dev);
        if (width != dev->fb_w || height != dev->fb_h) {
            dev->fb.release();
            dev->fb_w = dev->fb_h = 0;
        }
        resizeFb(dev, width, height);
        clearOnStream(dev->stream, dev->fb.ptr, 0, (size_t)width * height * 3 * sizeof(float));
        dev->fb_host_dirty = true;
        for (int k = 0; k < 5; ++k)
            if (dev->aov[k].ptr) {
                clearOnStream(dev->stream, dev->aov[k].ptr, 0, (size_t)width * height * 3 * sizeof(float));
                dev->aov_host_dirty[k] = true;
            }
    });
}

int32_t igd_release_all(igd_device* dev)
{
    return guarded("igd_release_all", [&] {
        if (!dev)
            throw HipError{ IGD_ERR_INVALID_ARG, "NULL device" };
        HIP_CHECK(hipSetDevice(dev->setup.gpu_index));
        finish(dev);
        dev->releaseStreams();
        for (auto& f : dev->flight) {
            f.tail_in.release();
            f.tail_long.release();
            f.tail_capacity = 0;
        }
        dev->list_rays.release();
        dev->geom.release();
        dev->shape_data.release();
        dev->leaves.release();
        dev->dev_leaves.release();
        dev->dev_leaf_scan.release();
        dev->dev_sphere_leaf_scan.release();
        dev->geom_ref.release();
        dev->sphere_leaves.release();
        dev->dev_sphere_leaves.release();
        dev->secondary_hit.release();
        dev->entities.release();
        dev->shape_offsets.release();
        dev->materials.release();
        dev->entity_material.release();
        dev->entity_ext.release();
        dev->entity_rec.release();
        dev->prim_records.release();
        dev->lights.release();
        dev->light_hierarchy.release();
        dev->light_codes.release();
        dev->light_cdf.release();
        dev->media.release();
        dev->expr_code.release();
        dev->textures.release();
        dev->texture_data.release();
        // the photon mapper's buffers (up to 32 + 32 + 16 bytes per photon) and the record of re-ordered packets
        dev->ppm_photons.release();
        dev->ppm_sorted.release();
        dev->ppm_keys.release();
        dev->ppm_cell_count.release();
        dev->ppm_cell_offset.release();
        dev->ppm_valid.release();
        dev->ppm_temp.release();
        dev->ppm_qs.release();
        dev->tri_spans.clear();
        dev->cdf_data.release();
        dev->info_tmp[0].release();
        dev->info_tmp[1].release();
        dev->info_qs.release();
        dev->has_scene = false;
    });
}

// (a recorded but not yet executed request already determines the size of the framebuffer)
int32_t igd_framebuffer_width(const igd_device* dev) { return dev ? (dev->pending.active ? dev->pending.rs.width : dev->fb_w) : 0; }
int32_t igd_framebuffer_height(const igd_device* dev) { return dev ? (dev->pending.active ? dev->pending.rs.height : dev->fb_h) : 0; }

const float* igd_framebuffer_host(igd_device* dev, const char* name, int32_t sync)
{
    const float* result = nullptr;
    guarded("igd_framebuffer_host", [&] {
        if (!dev)
            throw HipError{ IGD_ERR_INVALID_ARG, "NULL device" };
        const FilmBuffer b = filmBuffer(dev, name); // unknown AOV -> error (Device.cpp:1391-1395)
        HIP_CHECK(hipSetDevice(dev->setup.gpu_index));
        finish(dev);
        if (!b.dev->ptr)
            throw HipError{ IGD_ERR_INVALID_ARG, "no framebuffer yet (render or resize first)" };
        if (sync && *b.dirty) {
            HIP_CHECK(hipMemcpy(b.host->data(), b.dev->ptr, b.host->size() * sizeof(float), hipMemcpyDeviceToHost));
            *b.dirty = false;
        }
        result = b.host->data();
    });
    return result;
}

float* igd_framebuffer_device(igd_device* dev, const char* name)
{
    float* result = nullptr;
    // the pointer is about to be read by other device work: make the framebuffer final first
    guarded("igd_framebuffer_device", [&] {
        if (!dev)
            throw HipError{ IGD_ERR_INVALID_ARG, "NULL device" };
        const FilmBuffer b = filmBuffer(dev, name);
        HIP_CHECK(hipSetDevice(dev->setup.gpu_index));
        finish(dev);
        result = b.dev->ptr;
    });
    return result;
}

int32_t igd_clear_framebuffer(igd_device* dev, const char* name)
{
    return guarded("igd_clear_framebuffer", [&] {
        if (!dev)
            throw HipError{ IGD_ERR_INVALID_ARG, "NULL device" };
        const FilmBuffer b = filmBuffer(dev, name);
        HIP_CHECK(hipSetDevice(dev->setup.gpu_index));
        finish(dev);
        if (b.dev->ptr)
            clearOnStream(dev->stream, b.dev->ptr, 0, (size_t)dev->fb_w * dev->fb_h * 3 * sizeof(float));
        std::fill(b.host->begin(), b.host->end(), 0.0f);
        *b.dirty = true;
    });
}

int32_t igd_sync_framebuffer_to_device(igd_device* dev, const char* name, const float* data)
{
    return guarded("igd_sync_framebuffer_to_device", [&] {
        if (!dev || !data)
            throw HipError{ IGD_ERR_INVALID_ARG, "NULL argument" };
        const FilmBuffer b = filmBuffer(dev, name);
        HIP_CHECK(hipSetDevice(dev->setup.gpu_index));
        finish(dev);
        if (!b.dev->ptr)
            throw HipError{ IGD_ERR_INVALID_ARG, "no framebuffer yet (resize first)" };
        HIP_CHECK(hipMemcpy(b.dev->ptr, data, (size_t)dev->fb_w * dev->fb_h * 3 * sizeof(float), hipMemcpyHostToDevice));
        *b.dirty = true;
    });
}

uint64_t igd_buffer_size(igd_device* dev, const char* name)
{
    g_error.clear();
    if (!dev)
        return 0;
    return dev->has_scene || (name && (std::strcmp(name, "Color") == 0 || std::strcmp(name, "Default") == 0)) ? namedBuffer(dev, name).bytes : 0;
}

int32_t igd_buffer_copy(igd_device* dev, const char* name, void* dst, uint64_t max_bytes)
{
    return guarded("igd_buffer_copy", [&] {
        if (!dev || !name || (!dst && max_bytes))
            throw HipError{ IGD_ERR_INVALID_ARG, "NULL argument" };
        HIP_CHECK(hipSetDevice(dev->setup.gpu_index));
        finish(dev);
        const NamedBuffer b = namedBuffer(dev, name);
        if (!b.ptr)
            throw HipError{ IGD_ERR_INVALID_ARG, std::string("unknown buffer '") + name + "'" };
        const uint64_t n = std::min(b.bytes, max_bytes);
        if (n)
            HIP_CHECK(hipMemcpy(dst, b.ptr, n, hipMemcpyDeviceToHost));
    });
}

void* igd_buffer_ptr(igd_device* dev, const char* name, uint64_t* size_in_bytes)
{
    void* result = nullptr;
    if (size_in_bytes)
        *size_in_bytes = 0;
    guarded("igd_buffer_ptr", [&] {
        if (!dev || !name)
            throw HipError{ IGD_ERR_INVALID_ARG, "NULL argument" };
        HIP_CHECK(hipSetDevice(dev->setup.gpu_index));
        finish(dev);
        const NamedBuffer b = namedBuffer(dev, name);
        if (!b.ptr)
            throw HipError{ IGD_ERR_INVALID_ARG, std::string("unknown buffer '") + name + "'" };
        result = b.ptr;
        if (size_in_bytes)
            *size_in_bytes = b.bytes;
    });
    return result;
}

int32_t igd_get_stats(igd_device* dev, igd_stats* out)
{
    g_error.clear();
    if (!dev || !out) {
        g_error = "igd_get_stats: NULL argument";
        return IGD_ERR_INVALID_ARG;
    }
    // counters and timers of chunks still in flight are folded in first
    const int rc = guarded("igd_get_stats", [&] {
        HIP_CHECK(hipSetDevice(dev->setup.gpu_index));
        finish(dev);
    });
    *out = dev->stats;
    {
        // the byte model of the streams (igd_stats::stream_bytes): which columns exist follows from the scene and the camera alone
        const ig_camera& c   = dev->camera;
        const bool one_point = (c.type == IG_CAMERA_PERSPECTIVE && !(c.aperture_radius > 1.1920928955e-07f)) || (c.type == IG_CAMERA_FISHLENS && !c.fisheye_mask);
        const bool compact   = dev->camera_compact && one_point && dev->dscene.tech.type != IG_TECHNIQUE_LIGHTTRACER;
        const uint32_t hit   = dev->hit_pack_bits ? 16u : 20u;
        out->stream_bytes[0] = compact ? 16u : 48u;
        out->stream_bytes[1] = 32u;
        out->stream_bytes[2] = hit;
        out->stream_bytes[3] = (compact ? 16u : 48u) + hit;
        out->stream_bytes[4] = 64u + hit;
        out->stream_bytes[5] = 64u;
        out->stream_bytes[6] = 48u;
        out->stream_bytes[7] = 32u;
    }
    return rc;
}

int32_t igd_reset_stats(igd_device* dev)
{
    g_error.clear();
    if (!dev) {
        g_error = "igd_reset_stats: NULL device";
        return IGD_ERR_INVALID_ARG;
    }
    const int rc = guarded("igd_reset_stats", [&] {
        HIP_CHECK(hipSetDevice(dev->setup.gpu_index));
        finish(dev);
    });
    dev->stats = igd_stats{};
    return rc;
}

int32_t igd_traverse(igd_device* dev, int64_t count, const float* rays, uint32_t ray_flags, int32_t any_hit,
                     int32_t* ent_id, int32_t* prim_id, float* t, float* u, float* v, int32_t repeat, double* kernel_ms)
{
    return guarded("igd_traverse", [&] {
        if (!dev)
            throw HipError{ IGD_ERR_INVALID_ARG, "NULL device" };
        HIP_CHECK(hipSetDevice(dev->setup.gpu_index));
        traverseList(dev, count, rays, ray_flags, any_hit, ent_id, prim_id, t, u, v, repeat, kernel_ms);
    });
}

int32_t igd_set_parameter_i32(igd_device* dev, const char* name, int32_t value)
{
    return guarded("igd_set_parameter_i32", [&] {
        if (!dev || !name)
            throw HipError{ IGD_ERR_INVALID_ARG, "NULL argument" };
        // kernel arguments are captured by value at launch, so work already submitted keeps its parameters; iterations
        // that were only recorded so far are submitted first, with the old values
        // -- and only when a value the device reads actually changes: a caller that re-sends its whole registry before every
        // iteration (Runtime::stepVariant does) must not break the batch (IRenderDevice::render hands over the ParameterSet each call)
        int32_t* dst = nullptr;
        if (std::strcmp(name, "__tech_max_depth") == 0)
            dst = &dev->dscene.tech.max_depth;
        else if (std::strcmp(name, "__debug_mode") == 0) // DebugTechnique.cpp:27-31 (igview switches the view through it)
            dst = &dev->dscene.tech.debug_mode;
        else if (std::strcmp(name, "__tech_min_depth") == 0)
            dst = &dev->dscene.tech.min_depth;
        if (!dst || *dst == value)
            return;
        if (dst == &dev->dscene.tech.max_depth && dev->dscene.tech.type == IG_TECHNIQUE_VOLPATH && value > 0xFFFF)
            throw HipError{ IGD_ERR_UNSUPPORTED, "__tech_max_depth: the volumetric path tracer supports a max_depth of at most 65535" };
        HIP_CHECK(hipSetDevice(dev->setup.gpu_index));
        flushPending(dev);
        *dst = value;
    });
}

int32_t igd_set_parameter_f32(igd_device* dev, const char* name, float value)
{
    return guarded("igd_set_parameter_f32", [&] {
        if (!dev || !name)
            throw HipError{ IGD_ERR_INVALID_ARG, "NULL argument" };
        float* dst = nullptr;
        if (std::strcmp(name, "__tech_clamp") == 0)
            dst = &dev->dscene.tech.clamp;
        else if (std::strcmp(name, "__camera_scale") == 0) // OrthogonalCamera.cpp:28,39
            dst = &dev->camera.scale;
        if (!dst || std::memcmp(dst, &value, sizeof(float)) == 0)
            return;
        HIP_CHECK(hipSetDevice(dev->setup.gpu_index));
        flushPending(dev);
        *dst = value;
    });
}

int32_t igd_set_parameter_vec3(igd_device* dev, const char* name, const float value[3])
{
    return guarded("igd_set_parameter_vec3", [&] {
        if (!dev || !name || !value)
            throw HipError{ IGD_ERR_INVALID_ARG, "NULL argument" };
        float* dst = nullptr;
        if (std::strcmp(name, "__camera_eye") == 0)
            dst = dev->camera.eye;
        else if (std::strcmp(name, "__camera_dir") == 0)
            dst = dev->camera.dir;
        else if (std::strcmp(name, "__camera_up") == 0)
            dst = dev->camera.up;
        if (!dst || std::memcmp(dst, value, 3 * sizeof(float)) == 0)
            return;
        HIP_CHECK(hipSetDevice(dev->setup.gpu_index));
        flushPending(dev);
        std::memcpy(dst, value, 3 * sizeof(float));
    });
}

int32_t igd_synchronize(igd_device* dev)
{
    return guarded("igd_synchronize", [&] {
        if (!dev)
            throw HipError{ IGD_ERR_INVALID_ARG, "NULL device" };
        HIP_CHECK(hipSetDevice(dev->setup.gpu_index));
        finish(dev);
    });
}

int32_t igd_comm_available(void)
{
    std::string why;
    if (igdev::comm_available(why)) {
        g_error.clear();
        return 1;
    }
    g_error = why;
    return 0;
}

int32_t igd_comm_unique_id(uint8_t id[IGD_COMM_ID_BYTES])
{
    if (!id) {
        g_error = "igd_comm_unique_id: null id";
        return IGD_ERR_INVALID_ARG;
    }
    return guarded("igd_comm_unique_id", [&] { comm_unique_id(id); });
}

int32_t igd_comm_init(igd_device* dev, const uint8_t id[IGD_COMM_ID_BYTES], int32_t rank, int32_t world_size)
{
    if (!dev || !id) {
        g_error = "igd_comm_init: null device or id";
        return IGD_ERR_INVALID_ARG;
    }
    return guarded("igd_comm_init", [&] {
        HIP_CHECK(hipSetDevice(dev->setup.gpu_index));
        comm_destroy(dev->comm);
        dev->comm = nullptr;
        dev->comm = comm_create(id, rank, world_size);
    });
}

int32_t igd_comm_world_size(igd_device* dev)
{
    if (!dev || !dev->comm)
        return 0;
    int n = 0;
    return guarded("igd_comm_world_size", [&] { n = comm_world_size(dev->comm); }) == IGD_OK ? n : 0;
}

int32_t igd_comm_gather_rows(igd_device* dev, int32_t dst_rank)
{
    if (!dev || !dev->comm) {
        g_error = "igd_comm_gather_rows: no communicator (igd_comm_init)";
        return IGD_ERR_INVALID_ARG;
    }
    return guarded("igd_comm_gather_rows", [&] {
        HIP_CHECK(hipSetDevice(dev->setup.gpu_index));
        finish(dev); // everything rendered so far is in the framebuffer
        if (!dev->fb.ptr)
            throw HipError{ IGD_ERR_INVALID_ARG, "no framebuffer yet" };
        static const bool loopback = std::getenv("IGD_COMM_LOOPBACK") != nullptr; // tests on one GPU: rank dst's own rows travel too
        comm_gather_rows(dev->comm, dev->fb.ptr, dev->fb_w, dev->fb_h, dst_rank, dev->stream, loopback);
        dev->fb_host_dirty = true;
    });
}

int32_t igd_comm_allreduce_f64(igd_device* dev, double* values, int32_t count, int32_t op)
{
    if (!dev || !dev->comm || (!values && count > 0)) {
        g_error = "igd_comm_allreduce_f64: no communicator (igd_comm_init) or null values";
        return IGD_ERR_INVALID_ARG;
    }
    return guarded("igd_comm_allreduce_f64", [&] {
        HIP_CHECK(hipSetDevice(dev->setup.gpu_index));
        comm_allreduce_f64(dev->comm, values, count, op, dev->stream);
    });
}

int32_t igd_comm_destroy(igd_device* dev)
{
    if (!dev)
        return IGD_ERR_INVALID_ARG;
    comm_destroy(dev->comm);
    dev->comm = nullptr;
    return IGD_OK;
}

int32_t igd_node_bytes(const igd_device* dev) { return dev && dev->has_scene ? (dev->q8_nodes ? 128 : 256) : 0; }

const char* igd_last_error(void) { return g_error.c_str(); }

} // extern "C"
