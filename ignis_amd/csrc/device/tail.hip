// tail.hip — follows the paths that are left once the wavefront stream has become small, one lane per path.
//
// The wavefront loop pays three kernel launches per bounce; with max_depth 64 (diamond_scene) the paths
// bouncing inside the dielectrics keep it alive for ~50 more rounds of ever emptier launches, each costing
// a cold-start latency chain (the reference has the same long tail, src/artic/driver/mapping_gpu.art:756-866,
// plus its host round trips). Once the live stream is small, every remaining path is instead followed by
// one lane:
//   closest-hit traversal -> shade_vertex -> (any-hit traversal of the NEE ray) -> next bounce ...
// using the same device code as the wavefront kernels (traverse_core.h, shade_core.h), so every path
// produces bit-identical contributions and counters; only the scheduling differs.
//
// The host (device.hip) runs it on a second stream, overlapping the next chunk's rounds, as a sequence of
// passes: a pass follows each path for at most `max_bounces` bounces and appends the survivors, compacted,
// to the input of the next pass, so that the long paths (geometric length distribution) end up in a few
// full waves instead of pinning every wave of the grid behind its longest lane.
// Compiled twice, like traverse.hip: as is (launch_tail) and with -DIG_QNODE=1 for scenes whose inner nodes are the quantised
// records (launch_tail_q8; no wide single-ray traversal there, wide_core.h reads Node8 rows).
#include "shade_core.h"
#include "traverse_core.h"
#include "wide_core.h"
#include "group_core.h"

#ifndef IG_QNODE
#define IG_QNODE 0
#endif

namespace igdev {

constexpr bool kQNode = IG_QNODE != 0;

// One wave per workgroup: a straggling path then pins 12 KiB of LDS and one wave slot, not a 256-lane
// workgroup's 48 KiB, which matters because the tail overlaps the next chunk's traversal launches.
#ifndef IG_TAIL_OCC
#define IG_TAIL_OCC 3 // waves per SIMD the tail kernels are built for (168 VGPRs)
#endif
#ifndef IG_TAIL_BLOCK
#define IG_TAIL_BLOCK 64 // threads per workgroup of k_tail (its waves never meet: no barrier, LDS rows by thread)
#endif
#ifndef IG_TAIL_OCC_FULL
#define IG_TAIL_OCC_FULL IG_TAIL_OCC // ... of the full-BSDF instantiations (they spill ~ 370 VGPRs at 168: A/B r06 section 10)
#endif
constexpr int kTailBlock = IG_TAIL_BLOCK;

template <bool STATS, bool FULL, bool QNODE = false>
__global__ void __launch_bounds__(kTailBlock, FULL ? IG_TAIL_OCC_FULL : IG_TAIL_OCC) k_tail(const TailArgs a)
{
    __shared__ StackOf<kTailBlock> s_stack;

    const int tid      = threadIdx.x;
    const int lane     = tid & 63;
    const DevScene& sc = a.scene;
    const uint32_t n   = *a.in_count;
    uint2* deep_col    = sc.deep_stack + (a.deep_lane_base + blockIdx.x * kTailBlock + tid);

    uint32_t c_bounce = 0, c_shadow = 0, c_unoccluded = 0;
    uint32_t c_nodes[2] = { 0, 0 }, c_tris[2] = { 0, 0 }, c_leaves[2] = { 0, 0 };
    bool overflow = false;

    // Persistent lanes: a lane whose path ended takes the next unprocessed path (wave-aggregated fetch from
    // the device counter), so a wave stays full until the stream is empty instead of idling behind its
    // longest path.
    bool have = false;
    PathVertexIn in{};
    float tmin = 0, tmax = 0;
    uint32_t flags = 0, seed = 0;
    float4 acc     = make_float4(0, 0, 0, 0);
    bool exhausted = false;
    int hops       = 0; // bounces this lane has followed its current path for

    // Few paths, many waves (the late passes: 20 000 paths, then 3 000, ... for the 3 072 resident waves): a wave takes its share of
    // the pass, n / waves paths at a time, instead of 64 -- the traversal sections and the shading branches are wave-uniform, so a
    // wave with three paths in flight finishes a bounce in a fraction of the time of a full one, and what a pass costs is the
    // serial chain of its longest path. Which wave follows a path changes nothing about the path (its accumulator slot is its own).
    const uint32_t waves      = gridDim.x * (uint32_t)(kTailBlock / 64), wave_id = blockIdx.x * (uint32_t)(kTailBlock / 64) + (uint32_t)(tid >> 6);
    const int cap             = (int)min(64u, max(1u, (n + waves - 1) / waves));
    const uint32_t handed_out = waves * (uint32_t)cap; // paths the waves take by position, before the counter
    bool first                = true;
    if (handed_out >= n && wave_id * (uint32_t)cap >= n) {
        // nothing by position and nothing to refill from: this wave has no part in the pass
        if (blockIdx.x == 0 && tid == 0) {
            if (a.count_paths)
                a.qs->tail_rays += n;
            a.qs->tail_pass_in[a.pass] = n;
        }
        return;
    }

#ifdef IG_TAIL_CLOCKS
    // where a tail wave's cycles go (variant build, tools/tail_clocks.py): shader clock around the parts of a bounce, memory drained first
    unsigned long long tclk[6] = { 0, 0, 0, 0, 0, 0 }, tclk_last = __builtin_readcyclecounter();
#define TAIL_MARK(k)                                                    \
    do {                                                                \
        __builtin_amdgcn_s_waitcnt(0);                                  \
        const unsigned long long now_ = __builtin_readcyclecounter();   \
        tclk[k] += now_ - tclk_last;                                    \
        tclk_last = now_;                                               \
    } while (0)
#else
#define TAIL_MARK(k) ((void)0)
#endif
    for (;;) {
        TAIL_MARK(4); // loop, spill of long paths
        const unsigned long long idle = __ballot(!have);
        const int n_idle              = __popcll(idle);
        const int want                = cap - (64 - n_idle); // paths this wave may add to the ones it follows
        if (want > 0 && !exhausted) {
            // The first `cap` paths of a wave are the wave's by position: no counter, so a pass that fits the grid (every late one)
            // runs without a single atomic.
            uint32_t base = wave_id * (uint32_t)cap;
            if (first) {
                first = false;
                if (handed_out >= n)
                    exhausted = true;
            } else {
                if (lane == 0)
                    base = atomicAdd(a.work_counter, (uint32_t)want);
                base = __shfl(base, 0) + handed_out;
                if (base + (uint32_t)want >= n)
                    exhausted = true;
            }
            const uint32_t rank = (uint32_t)__popcll(idle & ((1ull << lane) - 1ull));
            const uint32_t i    = base + rank;
            if (!have && rank < (uint32_t)want && i < n) {
                have            = true;
                const float4 ra = a.in.rayA[i], rb = a.in.rayB[i];
                const int4 meta = a.in.meta[i];
                const float4 pay = a.in_kind == kStreamCamera ? make_float4(0, 1, 1, 1) : a.in.pay[i]; // (kernels.h kStream*)
                in.ray_id  = meta.x;
                in.org     = f3{ ra.x, ra.y, ra.z };
                in.dir     = f3{ rb.x, rb.y, rb.z };
                in.rnd     = (uint32_t)meta.z;
                in.inv_pdf = pay.x;
                in.contrib = Col{ pay.y, pay.z, pay.w };
                in.depth   = meta.w;
                in.eta     = a.in_kind == kStreamShaded ? igm_float((uint32_t)meta.y) : 1.0f;
                tmin = ra.w, tmax = a.in_kind == kStreamShaded ? kFltMax : rb.w;
                seed = a.in_kind == kStreamShaded ? igm_bits(rb.w) : 0u; // (kernels.h kStream*: a shaded stream keeps the path's seed there)
                flags = a.in_kind == kStreamShaded ? (uint32_t)IG_RAY_FLAG_BOUNCE : (uint32_t)meta.y;
                acc   = a.accum[(int64_t)in.ray_id - a.id_base]; // owned by this path until it ends
                hops  = 0;
            }
        }
        if (!__any(have))
            break;

        // (the traversals are called by the whole wave with the mask of the lanes that take part, traverse_core.h; shading sits under `have`)
        TAIL_MARK(0); // refill
        const mask_t have_m = lanes_where(have);
        {
            // the lanes whose ray the per-lane machine traverses: all of them, or — a handful of paths — those whose whole-wave traversal
            // ran out of its stack (practically never: it holds kLdsStack * 64 - 64 entries; the per-lane machine then spills to HBM)
            mask_t per_lane = have_m;
            if (!QNODE && lanes_in(have_m) <= (int)a.wide_lanes) {
                // a handful of paths: their rays one after the other, each traversed by the whole wave
                mask_t todo = have_m;
                per_lane    = 0;
                while (todo) {
                    const int l = __builtin_ctzll(todo);
                    todo &= todo - 1ull;
                    WideTraverser<STATS> w;
                    w.run(sc, s_stack, wide_bcast(in.org, l), wide_bcast(in.dir, l), wide_bcast(tmin, l), wide_bcast(tmax, l), (uint32_t)wide_bcast((int)flags, l));
                    if (lane == l && !w.overflow) {
                        in.ent  = w.hit_ent;
                        in.prim = w.hit_prim;
                        in.t = w.tmax, in.u = w.hit_u, in.v = w.hit_v;
                        if (STATS) {
                            c_nodes[0] += w.st_nodes;
                            c_tris[0] += w.st_tris;
                            c_leaves[0] += w.st_leaves;
                        }
                    }
                    region_end();
                    if (wide_any(w.overflow)) // (the traversal's state is wave-uniform) what it counted does not count: the ray starts again below
                        per_lane |= 1ull << l;
#ifdef IG_TAIL_CLOCKS
                    tclk[5] += 1;
#endif
                }
                TAIL_MARK(1);
            } else if (!QNODE && lanes_in(have_m) <= (int)a.wide8_lanes) {
                // some more paths (group_core.h): eight rays at a time, eight lanes each. Group g of a batch takes the ray of the g-th lane
                // that still has one; the lanes of the batch read their results back from their groups
                mask_t todo = have_m;
                per_lane    = 0;
                while (todo) {
                    int src      = -1;
                    mask_t batch = 0;
                    for (int k = 0; k < 8 && todo; ++k) {
                        const int l = __builtin_ctzll(todo);
                        todo &= todo - 1ull;
                        batch |= 1ull << l;
                        src = (lane >> 3) == k ? l : src;
                    }
                    const int from = src < 0 ? 0 : src;
                    GroupTraverser<STATS> w;
                    w.run(sc, s_stack, src >= 0, f3{ __shfl(in.org.x, from), __shfl(in.org.y, from), __shfl(in.org.z, from) },
                          f3{ __shfl(in.dir.x, from), __shfl(in.dir.y, from), __shfl(in.dir.z, from) }, __shfl(tmin, from), __shfl(tmax, from), (uint32_t)__shfl((int)flags, from));
                    // the lane of the batch's r-th ray takes what group r found
                    const int mine   = 8 * (__popcll(batch & ((1ull << lane) - 1ull)) & 7); // (a lane above the whole batch counts 8 of them: it takes nothing, any group will do)
                    const int g_ent  = __shfl(w.hit_ent, mine), g_prim = __shfl(w.hit_prim, mine);
                    const float g_t = __shfl(w.tmax, mine), g_u = __shfl(w.hit_u, mine), g_v = __shfl(w.hit_v, mine);
                    const bool g_ovf = __shfl((int)w.overflow, mine) != 0;
                    const uint32_t g_n = (uint32_t)__shfl((int)w.st_nodes, mine), g_tr = (uint32_t)__shfl((int)w.st_tris, mine), g_l = (uint32_t)__shfl((int)w.st_leaves, mine);
                    const bool took = igdev::in(batch);
                    if (took && !g_ovf) {
                        in.ent  = g_ent;
                        in.prim = g_prim;
                        in.t = g_t, in.u = g_u, in.v = g_v;
                        if (STATS) {
                            c_nodes[0] += g_n;
                            c_tris[0] += g_tr;
                            c_leaves[0] += g_l;
                        }
                    }
                    region_end();
                    per_lane |= lanes_where(took && g_ovf); // (what such a traversal counted does not count: the ray starts again below)
#ifdef IG_TAIL_CLOCKS
                    tclk[5] += 1;
#endif
                }
                TAIL_MARK(1);
            }
            if (per_lane) {
                Traverser<false, STATS, kTailBlock, true, false, QNODE> tr;
                tr.init_counters();
                tr.attach_deep(deep_col, sc.deep_stride);
                tr.begin(per_lane, sc, s_stack, tid, in.org, in.dir, tmin, tmax, flags);
                while (tr.active()) {
                    tr.step(sc, s_stack, tid);
#ifdef IG_TAIL_CLOCKS
                    tclk[5] += 1; // passes of the closest-hit traversal
#endif
                }
                TAIL_MARK(1); // closest-hit traversal
                if (igdev::in(per_lane)) {
                    in.ent  = tr.hit_ent;
                    in.prim = tr.hit_prim;
                    in.t = tr.tmax, in.u = tr.hit_u, in.v = tr.hit_v;
                    overflow |= tr.overflowed();
                    if (STATS) {
                        c_nodes[0] += tr.st_nodes;
                        c_tris[0] += tr.st_tris;
                        c_leaves[0] += tr.st_leaves;
                    }
                }
                region_end();
            }
            if (sc.sphere_node_count) {
                // the sphere geometry, from the hit so far (traverse.hip launches it as a second pass)
                Traverser<false, STATS, kTailBlock, false, true> tp;
                tp.init_counters();
                tp.begin(have_m, sc, s_stack, tid, in.org, in.dir, tmin, in.t, flags);
                tp.set_initial_hit(have_m, in.ent, in.prim, in.u, in.v);
                while (tp.active())
                    tp.step(sc, s_stack, tid);
                if (have) {
                    in.ent  = tp.hit_ent;
                    in.prim = tp.hit_prim;
                    in.t = tp.tmax, in.u = tp.hit_u, in.v = tp.hit_v;
                    overflow |= tp.overflowed();
                    if (STATS) {
                        c_nodes[0] += tp.st_nodes;
                        c_leaves[0] += tp.st_leaves;
                    }
                }
                region_end();
            }
        }

        PathVertexOut out;
        out.shadow = out.bounce = out.has_radiance = false;
        out.b_seed = 0u;
        if (have) {
            in.seed = seed;
            shade_vertex<FULL>(sc, a.frame, in, out);
            if (out.has_radiance) {
                acc.x += out.radiance.r * a.inv_spi;
                acc.y += out.radiance.g * a.inv_spi;
                acc.z += out.radiance.b * a.inv_spi;
            }
        }
        region_end();
        TAIL_MARK(2); // shading

        const bool shadow     = have && out.shadow;
        const mask_t shadow_m = lanes_where(shadow);
        if (shadow_m) {
            const uint32_t sflags = sc.tech.type == IG_TECHNIQUE_AO ? IG_RAY_FLAG_BOUNCE : IG_RAY_FLAG_SHADOW;
            Traverser<true, STATS, kTailBlock, true, false, QNODE> ts;
            ts.init_counters();
            ts.attach_deep(deep_col, sc.deep_stride);
            ts.begin(shadow_m, sc, s_stack, tid, out.s_org, out.s_dir, kRayOffset, out.s_tmax, sflags);
            while (ts.active())
                ts.step(sc, s_stack, tid);
            bool occluded = ts.hit_prim >= 0;
            if (shadow) {
                ++c_shadow;
                overflow |= ts.overflowed();
                if (STATS) {
                    c_nodes[1] += ts.st_nodes;
                    c_tris[1] += ts.st_tris;
                    c_leaves[1] += ts.st_leaves;
                }
            }
            region_end();
            if (sc.sphere_node_count) {
                Traverser<true, STATS, kTailBlock, false, true> tq;
                tq.init_counters();
                tq.begin(shadow_m, sc, s_stack, tid, out.s_org, out.s_dir, kRayOffset, ts.tmax, sflags);
                tq.set_initial_hit(shadow_m, ts.hit_ent, ts.hit_prim, 0, 0);
                while (tq.active())
                    tq.step(sc, s_stack, tid);
                occluded = tq.hit_prim >= 0;
                if (shadow) {
                    overflow |= tq.overflowed();
                    if (STATS) {
                        c_nodes[1] += tq.st_nodes;
                        c_leaves[1] += tq.st_leaves;
                    }
                }
                region_end();
            }
            if (shadow && !occluded) {
                ++c_unoccluded;
                acc.x += out.s_col.r * a.inv_spi;
                acc.y += out.s_col.g * a.inv_spi;
                acc.z += out.s_col.b * a.inv_spi;
            }
            region_end();
        }

        TAIL_MARK(3); // any-hit traversal + splat
        if (have) {
            if (!out.bounce) {
                a.accum[(int64_t)in.ray_id - a.id_base] = acc;
                have                                    = false;
            } else {
                ++c_bounce;
                in.org     = out.b_org;
                in.dir     = out.b_dir;
                in.rnd     = out.b_rnd;
                seed       = out.b_seed;
                in.inv_pdf = out.b_inv_pdf;
                in.contrib = out.b_contrib;
                in.depth   = out.b_depth;
                in.eta     = out.b_eta;
                tmin       = FULL ? out.b_tmin : kRayOffset;
                tmax       = kFltMax;
                flags      = IG_RAY_FLAG_BOUNCE;
                ++hops;
            }
        }
        region_end();

        // long paths leave this launch: they would pin the wave (and its registers / LDS) for milliseconds
        const bool spill               = have && a.max_bounces > 0 && hops >= a.max_bounces;
        const unsigned long long mspill = __ballot(spill);
        if (mspill) {
            uint32_t base = 0;
            if (lane == 0)
                base = atomicAdd(a.out_count, (uint32_t)__popcll(mspill));
            base = __shfl(base, 0);
            if (spill) {
                const uint32_t o = base + (uint32_t)__popcll(mspill & ((1ull << lane) - 1ull));
                a.out.rayA[o] = make_float4(in.org.x, in.org.y, in.org.z, tmin);
                a.out.rayB[o] = make_float4(in.dir.x, in.dir.y, in.dir.z, igm_float(seed)); // (kStreamShaded: a bounce ray's tmax is FLT_MAX)
                a.out.meta[o] = make_int4(in.ray_id, (int32_t)igm_bits(in.eta), (int32_t)in.rnd, in.depth); // (kStreamShaded; a spilled path has bounced: its flags are IG_RAY_FLAG_BOUNCE)
                a.out.pay[o]  = make_float4(in.inv_pdf, in.contrib.r, in.contrib.g, in.contrib.b);
                a.accum[(int64_t)in.ray_id - a.id_base] = acc;
                have                                    = false;
            }
        }
    }

    if (overflow)
        atomicOr(&a.qs->error_flags, 1u);
#ifdef IG_TAIL_CLOCKS
    // one pass is reported per build (-DIG_TAIL_CLOCKS=<pass index>): sums over the waves that took part, their count and the longest
    // wave's total, in the section counters' words
    if (lane == 0 && a.pass == IG_TAIL_CLOCKS) {
        for (int k = 0; k < 6; ++k)
            atomicAdd(&a.qs->section_passes[k], tclk[k]);
        atomicAdd(&a.qs->section_lanes[0], 1ull);
        atomicMax(&a.qs->section_lanes[1], tclk[0] + tclk[1] + tclk[2] + tclk[3] + tclk[4]);
    }
#endif

    const uint32_t b = wave_sum_u32(c_bounce), s = wave_sum_u32(c_shadow), u = wave_sum_u32(c_unoccluded);
    if (lane == 0) {
        if (b) atomicAdd(&a.qs->bounce_rays, (unsigned long long)b);
        if (s) atomicAdd(&a.qs->shadow_rays, (unsigned long long)s);
        if (u) atomicAdd(&a.qs->unoccluded, (unsigned long long)u);
    }
    if (STATS) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const uint32_t nn = wave_sum_u32(c_nodes[k]), tt = wave_sum_u32(c_tris[k]), ll = wave_sum_u32(c_leaves[k]);
            if (lane == 0) {
                atomicAdd(&a.qs->nodes[k], (unsigned long long)nn);
                atomicAdd(&a.qs->tris[k], (unsigned long long)tt);
                atomicAdd(&a.qs->leaves[k], (unsigned long long)ll);
            }
        }
    }
    if (blockIdx.x == 0 && tid == 0) {
        if (a.count_paths)
            a.qs->tail_rays += n;
        a.qs->tail_pass_in[a.pass] = n;
    }
}

template __global__ void k_tail<false, false, kQNode>(const TailArgs);
template __global__ void k_tail<true, false, kQNode>(const TailArgs);
template __global__ void k_tail<false, true, kQNode>(const TailArgs);
template __global__ void k_tail<true, true, kQNode>(const TailArgs);

#if IG_QNODE
#define launch_tail launch_tail_q8
#endif
void launch_tail(const TailArgs& args, bool stats, bool full_bsdfs, int grid_blocks, hipStream_t stream)
{
    const dim3 grid((unsigned)((grid_blocks + kTailBlock / 64 - 1) / (kTailBlock / 64))), block(kTailBlock); // grid_blocks counts waves
    if (full_bsdfs) {
        if (stats)
            hipLaunchKernelGGL((k_tail<true, true, kQNode>), grid, block, 0, stream, args);
        else
            hipLaunchKernelGGL((k_tail<false, true, kQNode>), grid, block, 0, stream, args);
    } else {
        if (stats)
            hipLaunchKernelGGL((k_tail<true, false, kQNode>), grid, block, 0, stream, args);
        else
            hipLaunchKernelGGL((k_tail<false, false, kQNode>), grid, block, 0, stream, args);
    }
}

} // namespace igdev
