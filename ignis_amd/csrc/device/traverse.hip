// traverse.hip — persistent-threads traversal kernels (closest hit and any hit) for gfx950.
//
// Replaces the reference kernels K2/K6 (`gpu_traverse_primary` / `_secondary`,
// src/artic/driver/mapping_gpu.art:52-121 over src/artic/traversal/mapping_gpu.art:67-219):
// one ray per lane (traverse_core.h); each wave reserves batches of ray indices from a
// device-resident counter (one atomic per 256 rays) and re-fills idle lanes from its batch
// (__ballot + lane rank) instead of one thread per ray and one launch per 1 M-ray batch.
// Compiled twice: as is (Node8 records, launch_traverse) and with -DIG_QNODE=1 (the 128-byte quantised node records of
// DevScene::node_format, launch_traverse_q8); device.hip picks by the scene.
#include "traverse_core.h"
#ifdef IG_TRAV_TIMELINE
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#endif

#ifndef IG_QNODE
#define IG_QNODE 0
#endif
#ifndef IG_DEFER_HIT_STORE
#define IG_DEFER_HIT_STORE 1 // the results of finished rays go out at the wave's next refill (one store instruction for all of them) instead of in the pass they end in (0); profiles/r05_experiment_ab.txt section 23
#endif
#ifndef IG_SPLAT_PREFETCH
#define IG_SPLAT_PREFETCH 1 // the shadow ray's accumulator fetched with the ray (0: read in the epilogue); profiles/r05_experiment_ab.txt section 16
#endif

namespace igdev {

constexpr bool kQNode = IG_QNODE != 0;

#ifndef IG_GUIDED_DIV
#define IG_GUIDED_DIV 4
#endif
#ifndef IG_MIN_RAY_BATCH
#define IG_MIN_RAY_BATCH 64 // smallest reservation of ray indices (the end of a stream)
#endif
#ifndef IG_REFILL_IDLE
#define IG_REFILL_IDLE 24
#endif
#ifndef IG_REFILL_IDLE_ANY
#define IG_REFILL_IDLE_ANY IG_REFILL_IDLE
#endif
constexpr int kRefillIdleClosest = IG_REFILL_IDLE;     // refill when at least this many lanes of a wave are idle
constexpr int kRefillIdleAny     = IG_REFILL_IDLE_ANY; // ... in the any-hit launches (shorter rays)
#ifndef IG_MAX_RAY_BATCH
#define IG_MAX_RAY_BATCH 1024
#endif
constexpr int kMaxRayBatch = IG_MAX_RAY_BATCH; // ray indices reserved per atomic (one word sustains ~88 atomics/us)

#ifdef IG_TRAV_TIMELINE
// variant build (tools/trav_timeline.sh): when each wave of a launch started, ran out of rays to fetch, ended, and how many rays it took
// (100 MHz wall clock); launch_traverse prints the summary of every non-DEEP launch to the file IGD_TRAV_TIMELINE names
constexpr int kTimelineWaves = 1 << 14;
__device__ unsigned long long g_timeline[kTimelineWaves * 4];
#endif

// SORTED: the launch takes its rays through TraverseArgs::sort_idx (raysort.hip). An instantiation of its own: as a run-time test of the
// pointer in the refill it cost the headline's closest-hit launches 6 % (profiles/r06_experiment_ab.txt section 2) — the kernel's scalar
// registers are that tight.
template <bool ANY_HIT, bool STATS, bool DEEP, bool SPHERES = false, bool QNODE = false, bool SORTED = false>
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
    // each while the rest of the chip idles. The stream is cut into kWorkShards contiguous shares with a counter each (kernels.h
    // WorkCounters): a wave draws from the share it starts on until that is used up, then from the next one, one probe per pass.
    const uint32_t total_waves = gridDim.x * (kBlockThreads / 64);
    const uint32_t n_shards    = a.work_shards > 1u ? (uint32_t)kWorkShards : 1u; // (1: one front over the whole stream, what a BVH beyond the caches wants)
    const uint32_t shard_waves = total_waves >= n_shards ? total_waves / n_shards : 1u;
    const uint32_t shard_rays  = ((count + n_shards - 1u) / n_shards + 63u) & ~63u; // rays per share (the last ones may be short or empty)
    // (workgroups go to the XCDs round-robin: blockIdx mod 8 is the XCD, whose workgroups share kWorkShards / 8 shares)
    uint32_t shard = n_shards == 1u ? 0u : ((blockIdx.x % 8u) * (uint32_t)(kWorkShards / 8) + (blockIdx.x / 8u) % (uint32_t)(kWorkShards / 8));
    uint32_t shards_left = n_shards; // shares this wave has not seen the end of
    uint32_t last_off          = 0; // where, in its share, the previous reservation of this wave started

    Traverser<ANY_HIT, STATS, kBlockThreads, DEEP, SPHERES, QNODE> tr;
    tr.attach_deep(a.scene.deep_stack + (blockIdx.x * kBlockThreads + tid), a.scene.deep_stride);
    tr.init_counters();
    tr.clk_start();
    tr.prof_start();
#ifdef IG_TRAV_TIMELINE
    const unsigned long long tl_start = wall_clock64();
    unsigned long long tl_dry = 0, tl_rays = 0;
#endif
    mask_t has_ray         = 0; // lanes with a ray in flight (a lane mask in scalar registers, like the traverser's)
    uint32_t ray_idx       = 0;
    uint32_t st_unoccluded = 0;
#if IG_SPLAT_PREFETCH
    float4 acc_pre = make_float4(0, 0, 0, 0);
#endif
    float4 splat           = make_float4(0, 0, 0, 0); // any hit: the shadow ray's colour and slot, fetched with the ray (one round trip per refill instead of one per finished lane)
    uint32_t snap_nodes = 0, snap_tris = 0, snap_leaves = 0; // work counters at the start of the current ray
    bool fatal = false;

    // wave-local batch of reserved ray indices (uniform across the wave)
    uint32_t batch_next = 0, batch_end = 0;
    bool exhausted = false;
    mask_t unsent = 0; // IG_DEFER_HIT_STORE: lanes whose ray has ended and whose hit is still in the traverser's registers
    // what a finished ray leaves behind (called for the lanes concerned, inside a region): the closest hit's row; the shadow ray's verdict
    // and, unoccluded, its colour into the sample's accumulator. With IG_DEFER_HIT_STORE the lanes wait in `unsent` until the wave's next
    // refill (their results are in registers the refill overwrites, not before) and go out in one instruction each instead of one per pass.
    // (closest hit only: deferring the shadow rays' read-modify-write costs their launch 10 %, section 23)
    constexpr bool kDefer = IG_DEFER_HIT_STORE && !ANY_HIT;
    const auto commit = [&]() {
        if (ANY_HIT) {
            if (a.hit)
                a.hit[ray_idx] = make_float4(igm_float((uint32_t)tr.hit_ent), igm_float((uint32_t)tr.hit_prim), tr.tmax, tr.hit_u);
            if (tr.hit_prim < 0 && a.sphere_pass != 1) { // (with a sphere pass to come, the verdict is its)
                if (STATS)
                    ++st_unoccluded;
                if (a.accum) {
                    // gpu_traverse_secondary splat (mapping_gpu.art:96-117) into the per-sample accumulator: a plain
                    // read-modify-write, the slot is owned by this ray's sample for the length of the launch. The colour came
                    // with the ray. (No-return float atomics instead are bit-identical but slower: the epilogue +25 %,
                    // profiles/r03_experiment_shade.txt.)
                    const float4 c = splat;
                    float4* dst    = a.accum + ((int64_t)(int32_t)igm_bits(c.w) - a.id_base);
                    // (the light tracer's connections add into the slots of OTHER paths' pixels: its launches pass no accumulator and
                    // record verdicts only, launch_lt_splat adds them in a fixed order afterwards, photon.hip)
#if IG_SPLAT_PREFETCH
                    float4 v = acc_pre;
#else
                    float4 v = *dst;
#endif
                    v.x += c.x * a.inv_spi;
                    v.y += c.y * a.inv_spi;
                    v.z += c.z * a.inv_spi;
                    *dst = v;
                    if (a.accum_nee) { // aov_nee.splat in on_shadow_miss (technique/pathtracer.art:212-218)
                        float4* nd = a.accum_nee + ((int64_t)(int32_t)igm_bits(c.w) - a.id_base);
                        float4 w   = *nd;
                        w.x += c.x * a.inv_spi;
                        w.y += c.y * a.inv_spi;
                        w.z += c.z * a.inv_spi;
                        *nd = w;
                    }
                }
            }
        } else if (a.hit_pack) {
            a.hit[ray_idx] = pack_hit(a.hit_pack, tr.hit_ent, tr.hit_prim, tr.tmax, tr.hit_u, tr.hit_v);
        } else {
            a.hit[ray_idx]   = make_float4(igm_float((uint32_t)tr.hit_ent), igm_float((uint32_t)tr.hit_prim), tr.tmax, tr.hit_u);
            a.hit_v[ray_idx] = tr.hit_v;
        }
    };

    for (;;) {
        tr.mark(0); // epilogue of the previous pass (hit stores / splat), loop bookkeeping
        tr.prof(0);
        // ---- refill idle lanes
        IG_MARK("pass.head");
        const mask_t idle = ~has_ray;
        const int n_idle  = lanes_in(idle);
        if (n_idle >= (ANY_HIT ? kRefillIdleAny : kRefillIdleClosest) && !(exhausted && batch_next >= batch_end)) {
            if (batch_next >= batch_end) {
                const uint32_t lo  = shard * shard_rays;
                const uint32_t len = lo < count ? (count - lo < shard_rays ? count - lo : shard_rays) : 0u; // rays of this share
                bool share_done    = true;
                if (len) {
                    const uint32_t left = len > last_off ? len - last_off : 0u;
                    uint32_t kRayBatch  = left / (shard_waves * (uint32_t)IG_GUIDED_DIV);
                    kRayBatch           = kRayBatch < (uint32_t)IG_MIN_RAY_BATCH ? (uint32_t)IG_MIN_RAY_BATCH : (kRayBatch > (uint32_t)kMaxRayBatch ? (uint32_t)kMaxRayBatch : kRayBatch);
                    kRayBatch &= ~63u;
                    uint32_t off = 0;
                    if (lane == 0)
                        off = atomicAdd(a.work_counter + shard * (uint32_t)kWorkShardWords, (uint32_t)kRayBatch);
                    off        = (uint32_t)__builtin_amdgcn_readfirstlane((int)off); // wave-uniform by construction: keeps the batch bookkeeping (and the loop) scalar
                    last_off   = off;
                    batch_next = lo + (off < len ? off : len);
                    batch_end  = lo + (off + kRayBatch < len ? off + kRayBatch : len);
                    share_done = off + kRayBatch >= len;
                }
                if (share_done) {
                    // (this reservation was the share's last, or came too late: the next one goes to the next share)
                    shard    = shard + 1u == n_shards ? 0u : shard + 1u;
                    last_off = 0xFFFFFFFFu; // (how far the others have come there is not known: the smallest reservation first)
                    if (--shards_left == 0)
                        exhausted = true;
                }
            }
            tr.prof(1);
            IG_MARK("refill");
            const uint32_t avail = batch_end - batch_next;
            const uint32_t take  = avail < (uint32_t)n_idle ? avail : (uint32_t)n_idle;
            const uint32_t rank  = __builtin_amdgcn_mbcnt_hi((uint32_t)(idle >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle, 0u)); // idle lanes below this one
            const mask_t fill    = lanes_where(rank < take) & idle;
            float4 ra = make_float4(0, 0, 0, 0), rb = ra, h = ra;
            uint32_t flags = a.uniform_flags;
            float hv       = 0;
            if (kDefer) {
                if (in(unsent))
                    commit();
                region_end();
                unsent = 0;
            }
            if (in(fill)) {
                uint32_t idx = batch_next + rank;
                if (DEEP && a.index_list) // (DEEP as the primary kernel: no list)
                    idx = a.index_list[idx];
                else if (SORTED) // the stream in key order (raysort.hip)
                    idx = a.sort_idx[idx];
                ray_idx = idx;
                tr.prof(2, true);
                ra = a.rayA ? a.rayA[idx] : a.uniform_rayA, rb = a.rayB[idx]; // (no rayA column: a compact camera stream, kernels.h CameraStream)
                if (a.meta)
                    flags = (uint32_t)a.meta[idx].y;
                if (SPHERES) {
                    h  = a.hit[idx];
                    hv = ANY_HIT ? 0.0f : a.hit_v[idx];
                }
                if (ANY_HIT && a.accum) {
                    splat = a.col[idx];
#if IG_SPLAT_PREFETCH
                    // the accumulator the ray would add into, fetched with the ray: the epilogue's add-and-store then waits for nothing (the
                    // slot is this ray's sample's for the length of the launch: one shadow ray per path and round)
                    acc_pre = a.accum[(int64_t)(int32_t)igm_bits(splat.w) - a.id_base];
#endif
                }
                if (STATS && !DEEP)
                    snap_nodes = tr.st_nodes, snap_tris = tr.st_tris, snap_leaves = tr.st_leaves;
            }
            region_end();
            if (SPHERES) {
                // init_hit = what the triangle pass found; its distance is the ray's tmax from here on
                tr.begin(fill, a.scene, s_stack, tid, f3{ ra.x, ra.y, ra.z }, f3{ rb.x, rb.y, rb.z }, ra.w, h.z, flags);
                tr.set_initial_hit(fill, (int)igm_bits(h.x), (int)igm_bits(h.y), h.w, hv);
            } else {
                tr.begin(fill, a.scene, s_stack, tid, f3{ ra.x, ra.y, ra.z }, f3{ rb.x, rb.y, rb.z }, ra.w, a.use_uniform_tmax ? a.uniform_tmax : rb.w, flags);
            }
            has_ray |= fill;
            batch_next += take;
#ifdef IG_TRAV_TIMELINE
            tl_rays += take;
#endif
            IG_MARK("refill.end");
        }
        // (no `continue` for the wave that got no ray out of a refill: a second back edge makes the compiler rotate the loop-carried
        // state registers through copies at the end of every pass; an empty step() is a few scalar instructions, once per launch)
        if (!has_ray && exhausted && batch_next >= batch_end)
            break;
#ifdef IG_TRAV_TIMELINE
        if (!tl_dry && exhausted && batch_next >= batch_end)
            tl_dry = wall_clock64();
#endif

        tr.mark(5); // refill: batch reservation, ray loads, begin()
        tr.step(a.scene, s_stack, tid);

        // ---- rays that ended in this pass
        IG_MARK("epilogue");
        const mask_t ended = has_ray & ~tr.active();
        const mask_t lost  = ended & tr.overflow;
        has_ray &= ~ended;
        if (ended) {
            if (DEEP || SPHERES) {
                if (lost)
                    fatal = true; // deeper than LDS + global part together
            } else {
                if (in(lost)) {
                    // hand the ray to the DEEP launch; what this lane counted for it does not count
                    a.index_list[atomicAdd(a.index_count, 1u)] = ray_idx;
                    if (STATS)
                        tr.st_nodes = snap_nodes, tr.st_tris = snap_tris, tr.st_leaves = snap_leaves;
                }
                region_end();
            }
            if (kDefer) {
                unsent |= ended & ~lost;
            } else {
                if (in(ended & ~lost))
                    commit();
                region_end();
            }
        }
    }
    if (kDefer) {
        if (in(unsent))
            commit();
        region_end();
    }

#ifdef IG_TRAV_TIMELINE
    if (!DEEP && !SPHERES && lane == 0) {
        const unsigned w = blockIdx.x * (kBlockThreads / 64) + tid / 64;
        if (w < (unsigned)kTimelineWaves) {
            const unsigned long long tl_end = wall_clock64();
            g_timeline[w * 4 + 0] = tl_start, g_timeline[w * 4 + 1] = tl_dry ? tl_dry : tl_end, g_timeline[w * 4 + 2] = tl_end, g_timeline[w * 4 + 3] = tl_rays;
        }
    }
#endif
    if (fatal)
        atomicOr(&a.qs->error_flags, 1u);
#ifdef IG_TRAV_PROFILE
    if (!DEEP && !SPHERES && ANY_HIT == (IG_TRAV_PROFILE == 2)) {
        for (int k = 0; k < 12; ++k) {
            const uint32_t n = wave_sum_u32(tr.ev[k]);
            if (lane == 0)
                atomicAdd(k < 6 ? &a.qs->section_passes[k] : &a.qs->section_lanes[k - 6], (unsigned long long)n);
        }
    }
#endif
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
        }
        for (int k = 0; k < 3; ++k) {
            const uint32_t p = wave_sum_u32(tr.sec_pass[k]), q = wave_sum_u32(tr.sec_lane[k]);
            if (lane == 0) {
                atomicAdd(&a.qs->section_passes[(ANY_HIT ? 3 : 0) + k], (unsigned long long)p);
                atomicAdd(&a.qs->section_lanes[(ANY_HIT ? 3 : 0) + k], (unsigned long long)q);
            }
        }
    }
}

#if !IG_QNODE
int traverse_workgroups_per_cu() { return kTraverseOcc; }
#endif

template <bool DEEP, bool SPHERES = false>
static void launch_one(const TraverseArgs& args, bool any_hit, bool stats, int grid_blocks, hipStream_t stream)
{
    const dim3 grid((unsigned)grid_blocks), block(kBlockThreads);
    if constexpr (!DEEP && !SPHERES) {
        if (args.sort_idx) {
            if (any_hit) {
                if (stats)
                    hipLaunchKernelGGL((k_traverse<true, true, false, false, kQNode, true>), grid, block, 0, stream, args);
                else
                    hipLaunchKernelGGL((k_traverse<true, false, false, false, kQNode, true>), grid, block, 0, stream, args);
            } else {
                if (stats)
                    hipLaunchKernelGGL((k_traverse<false, true, false, false, kQNode, true>), grid, block, 0, stream, args);
                else
                    hipLaunchKernelGGL((k_traverse<false, false, false, false, kQNode, true>), grid, block, 0, stream, args);
            }
            return;
        }
    }
    if (any_hit) {
        if (stats)
            hipLaunchKernelGGL((k_traverse<true, true, DEEP, SPHERES, kQNode>), grid, block, 0, stream, args);
        else
            hipLaunchKernelGGL((k_traverse<true, false, DEEP, SPHERES, kQNode>), grid, block, 0, stream, args);
    } else {
        if (stats)
            hipLaunchKernelGGL((k_traverse<false, true, DEEP, SPHERES, kQNode>), grid, block, 0, stream, args);
        else
            hipLaunchKernelGGL((k_traverse<false, false, DEEP, SPHERES, kQNode>), grid, block, 0, stream, args);
    }
}

// Two launches: the LDS-stack kernel over the whole stream, then the DEEP kernel over the rays the first one
// could not finish (almost always none: it reads one counter and exits). `deep_work_counter` must be zero.
// `deep_primary`: the scene's rays are known to outgrow the LDS stack (device.hip watches the overflow counts): ONE launch of the
// DEEP instantiation over the whole stream, whose lanes spill into their HBM columns as they go, instead of finishing most rays,
// listing the rest and re-traversing those from the root.
#if IG_QNODE
#define launch_traverse launch_traverse_q8
#endif
void launch_traverse(const TraverseArgs& args_in, bool any_hit, bool stats, int grid_blocks, uint32_t* deep_work_counter, hipStream_t stream, int deep_grid_blocks, bool deep_primary)
{
    TraverseArgs args = args_in;
    const bool spheres = args.scene.sphere_node_count != 0;
    args.sphere_pass   = spheres ? 1 : 0;
    if (deep_primary || spheres)
        args.sort_idx = nullptr; // (the DEEP-as-primary launch and the sphere pass take the stream as it lies)
    if (deep_primary) {
        TraverseArgs all = args;
        all.index_list   = nullptr;
        launch_one<true>(all, any_hit, stats, grid_blocks, stream);
    } else {
        launch_one<false>(args, any_hit, stats, grid_blocks, stream);
#ifdef IG_TRAV_TIMELINE
        if (const char* path = std::getenv("IGD_TRAV_TIMELINE")) {
            (void)hipStreamSynchronize(stream);
            const int waves = std::min(kTimelineWaves, grid_blocks * (kBlockThreads / 64));
            std::vector<unsigned long long> h((size_t)waves * 4);
            (void)hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(g_timeline), h.size() * 8);
            unsigned long long t0 = ~0ull, t1 = 0, rays = 0;
            for (int w = 0; w < waves; ++w)
                t0 = std::min(t0, h[w * 4]), t1 = std::max(t1, h[w * 4 + 2]), rays += h[w * 4 + 3];
            std::vector<double> start, dry, end, dry_to_end;
            for (int w = 0; w < waves; ++w) {
                start.push_back((h[w * 4] - t0) * 0.01), dry.push_back((h[w * 4 + 1] - t0) * 0.01), end.push_back((h[w * 4 + 2] - t0) * 0.01);
                dry_to_end.push_back((h[w * 4 + 2] - h[w * 4 + 1]) * 0.01);
            }
            auto pct = [](std::vector<double> v, double q) { std::sort(v.begin(), v.end()); return v[(size_t)(q * (v.size() - 1))]; };
            if (FILE* f = std::fopen(path, "a")) {
                std::fprintf(f, "%s rays %llu span %.1f us | wave start p50 %.1f max %.1f | out of rays p1 %.1f p50 %.1f p99 %.1f | end p1 %.1f p50 %.1f p90 %.1f p99 %.1f max %.1f | dry->end p50 %.1f p90 %.1f p99 %.1f max %.1f | alive at 10%% steps of the span:",
                             any_hit ? "any-hit" : "closest", rays, (t1 - t0) * 0.01, pct(start, 0.5), pct(start, 1.0), pct(dry, 0.01), pct(dry, 0.5), pct(dry, 0.99), pct(end, 0.01), pct(end, 0.5), pct(end, 0.9),
                             pct(end, 0.99), pct(end, 1.0), pct(dry_to_end, 0.5), pct(dry_to_end, 0.9), pct(dry_to_end, 0.99), pct(dry_to_end, 1.0));
                for (int k = 1; k <= 10; ++k) {
                    const double t = (t1 - t0) * 0.01 * k / 10.0 - 1e-9;
                    int alive = 0;
                    for (int w = 0; w < waves; ++w)
                        alive += start[w] <= t && end[w] > t;
                    std::fprintf(f, " %d", alive);
                }
                std::fprintf(f, "\n");
                std::fclose(f);
            }
        }
#endif
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
