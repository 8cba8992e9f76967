// traverse.hip — persistent-threads traversal kernels (closest hit and any hit) for gfx950.
//
// Replaces the reference kernels K2/K6 (`gpu_traverse_primary` / `_secondary`,
// src/artic/driver/mapping_gpu.art:52-121 over src/artic/traversal/mapping_gpu.art:67-219):
// one ray per lane (traverse_core.h); each wave reserves batches of ray indices from a
// device-resident counter (one atomic per 256 rays) and re-fills idle lanes from its batch
// (__ballot + lane rank) instead of one thread per ray and one launch per 1 M-ray batch.
#include "traverse_core.h"

namespace igdev {

#ifndef IG_REFILL_IDLE
#define IG_REFILL_IDLE 24
#endif
#ifndef IG_REFILL_IDLE_ANY
#define IG_REFILL_IDLE_ANY IG_REFILL_IDLE
#endif
constexpr int kRefillIdleClosest = IG_REFILL_IDLE;     // refill when at least this many lanes of a wave are idle
constexpr int kRefillIdleAny     = IG_REFILL_IDLE_ANY; // ... in the any-hit launches (shorter rays)
#ifndef IG_ATOMIC_SPLAT
#define IG_ATOMIC_SPLAT 0
#endif
#ifndef IG_EARLY_SPLAT
#define IG_EARLY_SPLAT 1
#endif
#ifndef IG_EARLY_ACCUM
#define IG_EARLY_ACCUM 0
#endif
constexpr bool kEarlyAccum  = IG_EARLY_ACCUM != 0;  // any hit: the accumulator slot is read with the ray as well
constexpr bool kAtomicSplat = IG_ATOMIC_SPLAT != 0; // sums into the per-sample accumulators as no-return float atomics
constexpr bool kEarlySplat  = IG_EARLY_SPLAT != 0;  // any hit: the colour of a shadow ray is loaded with the ray
constexpr int kMaxRayBatch = 1024; // ray indices reserved per atomic (one word sustains ~88 atomics/us)

template <bool ANY_HIT, bool STATS, bool DEEP, bool SPHERES = false>
__global__ void __launch_bounds__(kBlockThreads, kTraverseOcc) k_traverse(const TraverseArgs a)
{
    __shared__ StackLds s_stack;

    const int tid  = threadIdx.x;
    const int lane = tid & 63;

    const uint32_t count = *a.count;
    if (DEEP && count == 0)
        return; // the usual case: no ray of the first launch needed the deep stack
    // Guided self-scheduling: a wave asks for (what it believes is left) / (4 x waves) rays, between 64 and
    // kMaxRayBatch. Large batches keep the shared counter cold while the stream is long; towards its end the
    // batches shrink to one wave's worth, so the launch does not end with a few waves chewing on 1024 rays
    // each while the rest of the chip idles.
    const uint32_t total_waves = gridDim.x * (kBlockThreads / 64);
    uint32_t last_base         = 0; // where the previous reservation of this wave started

    Traverser<ANY_HIT, STATS, kBlockThreads, DEEP, SPHERES> tr;
    tr.attach_deep(a.scene.deep_stack + (blockIdx.x * kBlockThreads + tid), a.scene.deep_stride);
    tr.init_counters();
    tr.clk_start();
    bool has_ray          = false;
    uint32_t ray_idx      = 0;
    uint32_t st_unoccluded = 0;
    float4 splat          = any_float4(); // any hit: the shadow ray's colour and slot, fetched with the ray (kEarlySplat)
    float4 slot_value     = any_float4(); // ... and what the slot holds (kEarlyAccum)
    uint32_t snap_nodes = 0, snap_tris = 0, snap_leaves = 0; // work counters at the start of the current ray
    bool fatal = false;

    // wave-local batch of reserved ray indices (uniform across the wave)
    uint32_t batch_next = 0, batch_end = 0;
    bool exhausted = false;

    for (;;) {
        tr.mark(0); // epilogue of the previous pass (hit stores / splat), loop bookkeeping
        // ---- refill idle lanes
        const unsigned long long idle = __ballot(!has_ray);
        const int n_idle              = __popcll(idle);
        if (n_idle >= (ANY_HIT ? kRefillIdleAny : kRefillIdleClosest) && !(exhausted && batch_next >= batch_end)) {
            if (batch_next >= batch_end) {
                const uint32_t left = count > last_base ? count - last_base : 0u;
                uint32_t kRayBatch  = left / (total_waves * 4u);
                kRayBatch           = kRayBatch < 64u ? 64u : (kRayBatch > (uint32_t)kMaxRayBatch ? (uint32_t)kMaxRayBatch : kRayBatch);
                kRayBatch &= ~63u;
                uint32_t base = 0;
                if (lane == 0)
                    base = atomicAdd(a.work_counter, (uint32_t)kRayBatch);
                base       = (uint32_t)__builtin_amdgcn_readfirstlane((int)base); // wave-uniform by construction: keeps the batch bookkeeping (and the loop) scalar
                last_base  = base;
                batch_next = base < count ? base : count;
                batch_end  = base + kRayBatch < count ? base + kRayBatch : count;
                if (base + kRayBatch >= count)
                    exhausted = true;
            }
            const uint32_t avail = batch_end - batch_next;
            const uint32_t take  = avail < (uint32_t)n_idle ? avail : (uint32_t)n_idle;
            const uint32_t rank  = (uint32_t)__popcll(idle & ((1ull << lane) - 1ull));
            if (!has_ray && rank < take) {
                const uint32_t idx = (DEEP && a.index_list) ? a.index_list[batch_next + rank] : batch_next + rank; // (DEEP as the primary kernel: no list)
                ray_idx = idx;
                has_ray = true;
                const float4 ra = a.rayA[idx], rb = a.rayB[idx];
                if (SPHERES) {
                    // init_hit = what the triangle pass found; its distance is the ray's tmax from here on
                    const float4 h = a.hit[idx];
                    tr.begin(a.scene, s_stack, tid, f3{ ra.x, ra.y, ra.z }, f3{ rb.x, rb.y, rb.z }, ra.w, h.z,
                             a.meta ? (uint32_t)a.meta[idx].y : a.uniform_flags);
                    tr.set_initial_hit((int)igm_bits(h.x), (int)igm_bits(h.y), h.w, ANY_HIT ? 0.0f : a.hit_v[idx]);
                } else {
                    tr.begin(a.scene, s_stack, tid, f3{ ra.x, ra.y, ra.z }, f3{ rb.x, rb.y, rb.z }, ra.w, rb.w,
                             a.meta ? (uint32_t)a.meta[idx].y : a.uniform_flags);
                }
                if (ANY_HIT && kEarlySplat && a.accum) {
                    splat = a.col[idx];
                    // ... and the accumulator slot it may be added to (the slot is this ray's alone for the length of the launch): one more
                    // round trip per refill, shared by all the rays of the refill, instead of one per pass in which a lane finishes
                    if (kEarlyAccum && !a.atomic_splat)
                        slot_value = a.accum[(int64_t)(int32_t)igm_bits(splat.w) - a.id_base];
                }
                if (STATS && !DEEP)
                    snap_nodes = tr.st_nodes, snap_tris = tr.st_tris, snap_leaves = tr.st_leaves;
            }
            batch_next += take;
        }
        // (no `continue` for the wave that got no ray out of a refill: a second back edge made the compiler rotate the ~30 state
        // registers through copies at the end of every pass; an empty step() costs one settle body and happens once per launch)
        if (!__any(has_ray) && exhausted && batch_next >= batch_end)
            break;

        // every lane steps: a lane without a ray is `finished` in mode 0, which no section of step() acts on. (Wrapping the call
        // in `if (has_ray)` made the compiler copy the whole traversal state at the merge, ~35 v_mov per pass.)
        tr.mark(5); // refill: batch reservation, ray loads, begin()
        tr.step(a.scene, s_stack, tid);
        if (has_ray) {
            if (tr.finished && tr.overflow) {
                has_ray = false;
                if (DEEP || SPHERES) {
                    fatal = true; // deeper than LDS + global part together
                } else {
                    // hand the ray to the DEEP launch; what this lane counted for it does not count
                    a.index_list[atomicAdd(a.index_count, 1u)] = ray_idx;
                    if (STATS)
                        tr.st_nodes = snap_nodes, tr.st_tris = snap_tris, tr.st_leaves = snap_leaves;
                }
            } else if (tr.finished) {
                has_ray = false;
                if (ANY_HIT) {
                    if (a.hit)
                        a.hit[ray_idx] = make_float4(igm_float((uint32_t)tr.hit_ent), igm_float((uint32_t)tr.hit_prim), tr.tmax, tr.hit_u);
                    if (tr.hit_prim < 0 && a.sphere_pass != 1) { // (with a sphere pass to come, the verdict is its)
                        if (STATS)
                            ++st_unoccluded;
                        if (a.accum) {
                            // gpu_traverse_secondary splat (mapping_gpu.art:96-117) into the per-sample
                            // accumulator: plain read-modify-write, the slot is owned by this ray.
                            // The colour came with the ray and the sums are no-return float atomics performed in L2 (the slot is owned by
                            // this ray, so the sum is the read-modify-write's): a finished lane costs its wave no round trip. As a load of
                            // the colour, a dependent load of the slot and a store, each event held the whole wave for two HBM latencies,
                            // in order in front of its next node loads (waves waiting 63 %, VALU issue 0.65, profiles/r03_rocprofv3_pmc.txt).
                            const float4 c = kEarlySplat ? splat : a.col[ray_idx];
                            float4* dst    = a.accum + ((int64_t)(int32_t)igm_bits(c.w) - a.id_base);
                            // (the light tracer's connections, on_advanced_shadow_miss, technique/lighttracer.art:116-120, add into the
                            // slots of one pixel from many paths: atomics for correctness there)
                            if (kAtomicSplat || a.atomic_splat) {
                                unsafeAtomicAdd(&dst->x, c.x * a.inv_spi);
                                unsafeAtomicAdd(&dst->y, c.y * a.inv_spi);
                                unsafeAtomicAdd(&dst->z, c.z * a.inv_spi);
                            } else {
                                float4 v = (kEarlySplat && kEarlyAccum) ? slot_value : *dst;
                                v.x += c.x * a.inv_spi;
                                v.y += c.y * a.inv_spi;
                                v.z += c.z * a.inv_spi;
                                *dst = v;
                            }
                            if (a.accum_nee) { // aov_nee.splat in on_shadow_miss (technique/pathtracer.art:212-218)
                                float4* nd = a.accum_nee + ((int64_t)(int32_t)igm_bits(c.w) - a.id_base);
                                if (kAtomicSplat) {
                                    unsafeAtomicAdd(&nd->x, c.x * a.inv_spi);
                                    unsafeAtomicAdd(&nd->y, c.y * a.inv_spi);
                                    unsafeAtomicAdd(&nd->z, c.z * a.inv_spi);
                                } else {
                                    float4 w = *nd;
                                    w.x += c.x * a.inv_spi;
                                    w.y += c.y * a.inv_spi;
                                    w.z += c.z * a.inv_spi;
                                    *nd = w;
                                }
                            }
                        }
                    }
                } else {
                    a.hit[ray_idx]   = make_float4(igm_float((uint32_t)tr.hit_ent), igm_float((uint32_t)tr.hit_prim), tr.tmax, tr.hit_u);
                    a.hit_v[ray_idx] = tr.hit_v;
                }
            }
        }
    }

    if (fatal)
        atomicOr(&a.qs->error_flags, 1u);
#ifdef IG_TRAV_CLOCKS
    tr.mark(0);
    if (lane == 0 && !DEEP && !SPHERES)
        for (int k = 0; k < 6; ++k)
            atomicAdd(ANY_HIT ? &a.qs->section_lanes[k] : &a.qs->section_passes[k], tr.clk_acc[k]);
#endif

    if (STATS) {
        const uint32_t n = wave_sum_u32(tr.st_nodes), t = wave_sum_u32(tr.st_tris), l = wave_sum_u32(tr.st_leaves), uo = wave_sum_u32(st_unoccluded);
        if (lane == 0) {
            atomicAdd(&a.qs->nodes[ANY_HIT], (unsigned long long)n);
            atomicAdd(&a.qs->tris[ANY_HIT], (unsigned long long)t);
            atomicAdd(&a.qs->leaves[ANY_HIT], (unsigned long long)l);
            if (ANY_HIT)
                atomicAdd(&a.qs->unoccluded, (unsigned long long)uo);
            for (int k = 0; k < 3; ++k) {
                atomicAdd(&a.qs->section_passes[(ANY_HIT ? 3 : 0) + k], (unsigned long long)tr.sec_pass[k]);
                atomicAdd(&a.qs->section_lanes[(ANY_HIT ? 3 : 0) + k], (unsigned long long)tr.sec_lane[k]);
            }
        }
    }
}

int traverse_workgroups_per_cu() { return kTraverseOcc; }

template <bool DEEP, bool SPHERES = false>
static void launch_one(const TraverseArgs& args, bool any_hit, bool stats, int grid_blocks, hipStream_t stream)
{
    const dim3 grid((unsigned)grid_blocks), block(kBlockThreads);
    if (any_hit) {
        if (stats)
            hipLaunchKernelGGL((k_traverse<true, true, DEEP, SPHERES>), grid, block, 0, stream, args);
        else
            hipLaunchKernelGGL((k_traverse<true, false, DEEP, SPHERES>), grid, block, 0, stream, args);
    } else {
        if (stats)
            hipLaunchKernelGGL((k_traverse<false, true, DEEP, SPHERES>), grid, block, 0, stream, args);
        else
            hipLaunchKernelGGL((k_traverse<false, false, DEEP, SPHERES>), grid, block, 0, stream, args);
    }
}

// Two launches: the LDS-stack kernel over the whole stream, then the DEEP kernel over the rays the first one
// could not finish (almost always none: it reads one counter and exits). `deep_work_counter` must be zero.
// `deep_primary`: the scene's rays are known to outgrow the LDS stack (device.hip watches the overflow counts): ONE launch of the
// DEEP instantiation over the whole stream, whose lanes spill into their HBM columns as they go, instead of finishing most rays,
// listing the rest and re-traversing those from the root.
void launch_traverse(const TraverseArgs& args_in, bool any_hit, bool stats, int grid_blocks, uint32_t* deep_work_counter, hipStream_t stream, int deep_grid_blocks, bool deep_primary)
{
    TraverseArgs args = args_in;
    const bool spheres = args.scene.sphere_node_count != 0;
    args.sphere_pass   = spheres ? 1 : 0;
    if (deep_primary) {
        TraverseArgs all = args;
        all.index_list   = nullptr;
        launch_one<true>(all, any_hit, stats, grid_blocks, stream);
    } else {
        launch_one<false>(args, any_hit, stats, grid_blocks, stream);
        TraverseArgs deep = args;
        deep.count        = args.index_count;
        deep.work_counter = deep_work_counter;
        // (normally the same grid: the workgroups of an empty DEEP launch only read the counter)
        launch_one<true>(deep, any_hit, stats, grid_blocks < deep_grid_blocks ? grid_blocks : deep_grid_blocks, stream);
    }
    if (spheres) {
        // the other SceneGeometry (driver/mapping_cpu.art:385-403): the sphere BVH, starting from the hits of the pass above.
        // Its stack never leaves LDS (a BVH over entities, not triangles); a ray that would need more raises the error flag.
        TraverseArgs sp = args;
        sp.sphere_pass  = 2;
        sp.work_counter = args.sphere_work_counter;
        launch_one<false, true>(sp, any_hit, stats, grid_blocks, stream);
    }
}

} // namespace igdev
