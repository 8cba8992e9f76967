// traverse.hip — two-level BVH8 traversal for gfx950 (closest hit and any hit).
//
// Replaces the reference kernels K2/K6 (`gpu_traverse_primary` / `_secondary`,
// src/artic/driver/mapping_gpu.art:52-121 over src/artic/traversal/mapping_gpu.art:67-219)
// with a persistent-threads design:
//   * one ray per lane; each wave reserves batches of ray indices from a device-resident counter
//     (one atomic per 256 rays) and re-fills idle lanes from its batch (__ballot + lane rank)
//     instead of one thread per ray and one launch per batch;
//   * the traversal stack lives in LDS, [entry][thread] layout (conflict-free ds_read/write_b64),
//     instead of the reference's 64-entry private array that spills to scratch;
//   * scene level and shape level share ONE inner-node section and ONE stack: entering an instance
//     saves the scene top, pushes a sentinel and switches the node base offset. Every loop
//     iteration runs the sections in pipeline order  entity leaf -> inner node -> triangle packet,
//     with cheap state transitions in between, so a lane can walk a whole instance (leaf test,
//     shape root, triangles) in one iteration and the lanes of a wave stay in phase;
//   * nodes / Tri4 packets / entity leaves are fetched as 16-byte vectors from one HBM blob
//     (SGPR base + 32-bit VGPR offset).
//
// Per-ray semantics (visit order, culling points, acceptance `t <= tmax`, hence tie-breaking) are
// those of the reference CPU device, `cpu_traverse_helper(_prim)` with vector width 1
// (src/artic/traversal/mapping_cpu.art:282-518), on the same Node8 / Tri4 / EntityLeaf1 bytes,
// so hits AND the visited-node / tested-triangle counts equal the CPU oracle's
// (DESIGN.md "Traversal order").
#include "dev_math.h"
#include "kernels.h"

namespace igdev {

constexpr int kLdsStack     = 24;  // LDS: 24 entries * 256 threads * 8 B = 48 KiB per workgroup (3 workgroups per CU)
constexpr int kBlockThreads = 256;
constexpr int kRefillIdle   = 16;  // refill when at least this many lanes of a wave are idle
constexpr int kRayBatch     = 256; // ray indices reserved per atomic

template <bool ANY_HIT, bool STATS>
__global__ void __launch_bounds__(kBlockThreads) k_traverse(const TraverseArgs a)
{
    __shared__ uint2 s_stack[kLdsStack][kBlockThreads];

    const int tid  = threadIdx.x;
    const int lane = tid & 63;

    const uint32_t count = *a.count;
    const uint8_t* geom  = a.scene.geom;
    // ---- per-lane ray state
    bool has_ray     = false;
    uint32_t ray_idx = 0;
    RayT gray{}, cur{};
    float tmin = 0, tmax = 0;
    uint32_t rflags = 0;
    float hit_u = 0, hit_v = 0;
    int hit_prim = -1, hit_ent = -1;
    int top_node = 0;
    float top_tmin = kFltMax;
    int ptr = -1;
    int level = 0; // 0 scene BVH, 1 shape BVH
    int mode  = 0; // 0 stack driven, 1 inside a triangle leaf, 2 inside an entity leaf run
    int ent_cursor = 0, tri_cursor = 0;
    uint32_t node_off = 0, tri_off = 0;
    int cur_ent = -1;
    bool ent_last = true;
    bool need_cull = true;
    bool finished = false;
    bool overflow = false;
    uint32_t st_nodes = 0, st_tris = 0, st_leaves = 0, st_unoccluded = 0;

    // The reference's stack has 64 entries and no overflow check (traversal/stack.art:53-54). Here the
    // stack is 24 entries of LDS per lane; deeper pushes raise error bit 0 (igd_render fails loudly).
    auto push_entry = [&](int n, float t) {
        ++ptr;
        if (ptr < kLdsStack)
            s_stack[ptr][tid] = make_uint2((uint32_t)n, igm_bits(t));
        else
            overflow = true;
    };
    auto pop_top = [&]() {
        const uint2 e = s_stack[ptr < kLdsStack ? ptr : kLdsStack - 1][tid];
        top_node      = (int)e.x;
        top_tmin      = igm_float(e.y);
        --ptr;
    };

    // Cheap state transitions up to the next heavy action: an entity-leaf step (mode 2), an inner
    // node on top (mode 0, top_node > 0), a triangle packet (mode 1), or the end of the ray.
    // The cull points are exactly the reference's (mapping_cpu.art:326-347): at level entry, after
    // a leaf and after an inner node that pushed nothing.
    auto settle = [&]() {
        while (mode == 0 && !finished) {
            if (need_cull) {
                while (top_node != 0 && !(top_tmin <= tmax))
                    pop_top();
                need_cull = false;
            }
            if (top_node == 0) {
                if (level == 1) {
                    // shape BVH exhausted: back to the scene leaf run (mapping_cpu.art:489-508)
                    level = 0;
                    pop_top(); // saved scene-level top
                    cur      = gray;
                    node_off = a.scene.scene_nodes_off;
                    if (ent_last)
                        need_cull = true;
                    else
                        mode = 2;
                } else {
                    finished = true;
                }
            } else if (top_node > 0) {
                break; // inner node pending
            } else {
                // leaf on top (mapping_cpu.art:379-381): an entry that starts behind the current
                // hit is dropped, its items have no effect in the reference either
                const bool active = top_tmin <= tmax;
                if (level)
                    tri_cursor = ~top_node;
                else
                    ent_cursor = ~top_node;
                pop_top();
                if (active)
                    mode = level ? 1 : 2;
                else
                    need_cull = true;
            }
        }
    };

    // wave-local batch of reserved ray indices (uniform across the wave)
    uint32_t batch_next = 0, batch_end = 0;
    bool exhausted = false;

    for (;;) {
        // ---- refill idle lanes
        const unsigned long long idle = __ballot(!has_ray);
        const int n_idle              = __popcll(idle);
        if (n_idle >= kRefillIdle && !(exhausted && batch_next >= batch_end)) {
            if (batch_next >= batch_end) {
                uint32_t base = 0;
                if (lane == 0)
                    base = atomicAdd(a.work_counter, (uint32_t)kRayBatch);
                base       = __shfl(base, 0);
                batch_next = base < count ? base : count;
                batch_end  = base + kRayBatch < count ? base + kRayBatch : count;
                if (base + kRayBatch >= count)
                    exhausted = true;
            }
            const uint32_t avail = batch_end - batch_next;
            const uint32_t take  = avail < (uint32_t)n_idle ? avail : (uint32_t)n_idle;
            const uint32_t rank  = (uint32_t)__popcll(idle & ((1ull << lane) - 1ull));
            if (!has_ray && rank < take) {
                const uint32_t idx = batch_next + rank;
                ray_idx = idx;
                has_ray = true;
                const f3 org = f3{ a.ox[idx], a.oy[idx], a.oz[idx] };
                const f3 dir = f3{ a.dx[idx], a.dy[idx], a.dz[idx] };
                gray   = make_ray_terms(org, dir);
                cur    = gray;
                tmin   = a.tmin[idx];
                tmax   = a.tmax[idx];
                rflags = a.flags ? a.flags[idx] : a.uniform_flags;
                hit_u = hit_v = 0;
                hit_prim = hit_ent = -1;
                level = 0, mode = 0;
                need_cull = true;
                finished  = false;
                node_off  = a.scene.scene_nodes_off;
                // stack.push(root, ray.tmin) on an empty stack: sentinel below, root on top
                ptr      = -1;
                top_node = 0, top_tmin = kFltMax;
                push_entry(top_node, top_tmin);
                top_node = a.scene.scene_node_count ? 1 : 0;
                top_tmin = tmin;
            }
            batch_next += take;
        }
        if (!__any(has_ray)) {
            if (exhausted && batch_next >= batch_end)
                break;
            continue;
        }

        if (has_ray) {
            settle();

            // ---- one entity leaf of the current run (mapping_cpu.art:481-515)
            if (mode == 2) {
                const float4* lf = reinterpret_cast<const float4*>(a.scene.leaves + ent_cursor);
                const uint2 ext  = a.scene.leaf_ext[ent_cursor];
                ++ent_cursor;
                const float4 l0 = lf[0], l1 = lf[1], l5 = lf[5];
                const int entity_id   = (int)igm_bits(l0.w);
                const uint32_t lflags = igm_bits(l5.x);
                ent_last              = entity_id < 0;
                if (STATS)
                    ++st_leaves;
                bool enter = false;
                // check_ray_visibility (traversal/ray.art:51)
                if ((rflags & IG_RAY_FLAG_TYPE_MASK) == ((rflags & lflags) & IG_RAY_FLAG_TYPE_MASK)) {
                    float entry, exit;
                    slab_test(gray, tmin, tmax, l0.x, l1.x, l0.y, l1.y, l0.z, l1.z, entry, exit);
                    enter = (entry <= exit) & (exit >= 0) & (entry <= tmax);
                }
                if (enter) {
                    const float4 l2 = lf[2], l3 = lf[3], l4 = lf[4];
                    m34 m;
                    m.c0 = f3{ l2.x, l2.y, l2.z };
                    m.c1 = f3{ l2.w, l3.x, l3.y };
                    m.c2 = f3{ l3.z, l3.w, l4.x };
                    m.c3 = f3{ l4.y, l4.z, l4.w };
                    // transform_ray (traversal/ray.art:56-59): direction not normalised, t stays global
                    cur     = make_ray_terms(xform_point(m, gray.org), xform_dir(m, gray.dir));
                    cur_ent = entity_id & 0x7FFFFFFF;
                    // save the scene-level top, then a fresh stack: sentinel + shape root
                    push_entry(top_node, top_tmin);
                    push_entry(0, kFltMax);
                    top_node  = 1;
                    top_tmin  = tmin;
                    level     = 1;
                    mode      = 0;
                    need_cull = true;
                    node_off  = ext.x;
                    tri_off   = ext.y;
                } else if (ent_last) {
                    mode      = 0;
                    need_cull = true;
                }
                settle();
            }

            // ---- one inner node: fetch 256 B, test 8 children (mapping_cpu.art:350-377)
            if (mode == 0 && !finished) {
                const uint8_t* np = geom + node_off + (uint32_t)(top_node - 1) * 256u;
                pop_top();
                const float4* nf = reinterpret_cast<const float4*>(np);
                const int4* nc   = reinterpret_cast<const int4*>(np) + 12;
                if (STATS)
                    ++st_nodes;
                bool pushed = false;
                // two halves of four children keep the live register set small (occupancy 4 waves/SIMD)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    float bnd[6][4];
#pragma unroll
                    for (int k = 0; k < 6; ++k) {
                        const float4 x = nf[2 * k + h];
                        bnd[k][0] = x.x, bnd[k][1] = x.y, bnd[k][2] = x.z, bnd[k][3] = x.w;
                    }
                    const int4 c4   = nc[h];
                    const int ch[4] = { c4.x, c4.y, c4.z, c4.w };
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        float entry, exit;
                        slab_test(cur, tmin, tmax, bnd[0][i], bnd[1][i], bnd[2][i], bnd[3][i], bnd[4][i], bnd[5][i], entry, exit);
                        const bool hit = (ch[i] != 0) & !(exit < entry);
                        if (hit) {
                            // push (becomes the top) if nearer than the current top, else push_after
                            const bool front = ANY_HIT || (top_tmin > entry);
                            push_entry(front ? top_node : ch[i], front ? top_tmin : entry);
                            if (front) {
                                top_node = ch[i];
                                top_tmin = entry;
                            }
                            pushed = true;
                        }
                    }
                }
                if (!pushed)
                    need_cull = true;
                settle();
            }

            // ---- one Tri4 packet of a leaf (mapping_cpu.art:379-410)
            if (mode == 1) {
                const uint8_t* tp = geom + tri_off + (uint32_t)tri_cursor * 208u;
                ++tri_cursor;
                const float4* tf = reinterpret_cast<const float4*>(tp);
                float q[12][4];
#pragma unroll
                for (int k = 0; k < 12; ++k) {
                    const float4 x = tf[k];
                    q[k][0] = x.x, q[k][1] = x.y, q[k][2] = x.z, q[k][3] = x.w;
                }
                const int4 pid4 = reinterpret_cast<const int4*>(tp)[12];
                const int pid[4] = { pid4.x, pid4.y, pid4.z, pid4.w };
                bool valid = true;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    valid = valid & (pid[i] != -1);
                    if (valid && !(ANY_HIT && finished)) {
                        if (STATS)
                            ++st_tris;
                        float t, u, v;
                        if (tri_test(cur, tmin, tmax, f3{ q[0][i], q[1][i], q[2][i] }, f3{ q[3][i], q[4][i], q[5][i] },
                                     f3{ q[6][i], q[7][i], q[8][i] }, f3{ q[9][i], q[10][i], q[11][i] }, t, u, v)) {
                            tmax     = t;
                            hit_u    = u;
                            hit_v    = v;
                            hit_prim = pid[i] & 0x7FFFFFFF;
                            hit_ent  = cur_ent;
                            if (ANY_HIT)
                                finished = true;
                        }
                    }
                }
                if (pid[3] < 0) {
                    mode      = 0;
                    need_cull = true;
                }
            }

            if (finished) {
                has_ray = false;
                if (ANY_HIT) {
                    if (a.prim_id)
                        a.prim_id[ray_idx] = hit_prim;
                    if (a.ent_id)
                        a.ent_id[ray_idx] = hit_ent;
                    if (hit_prim < 0) {
                        if (STATS)
                            ++st_unoccluded;
                        if (a.accum) {
                            // gpu_traverse_secondary splat (mapping_gpu.art:96-117) into the per-sample
                            // accumulator: plain read-modify-write, the slot is owned by this ray.
                            float* dst = a.accum + ((int64_t)a.ray_id[ray_idx] - a.id_base) * 3;
                            dst[0] += a.cr[ray_idx] * a.inv_spi;
                            dst[1] += a.cg[ray_idx] * a.inv_spi;
                            dst[2] += a.cb[ray_idx] * a.inv_spi;
                        }
                    }
                } else {
                    a.ent_id[ray_idx]  = hit_ent;
                    a.prim_id[ray_idx] = hit_prim;
                    a.t[ray_idx]       = tmax;
                    a.u[ray_idx]       = hit_u;
                    a.v[ray_idx]       = hit_v;
                }
            }
        }
    }

    if (overflow)
        atomicOr(&a.qs->error_flags, 1u);

    if (STATS) {
        // wave-level reduction, one atomic per wave and counter
        auto wave_sum = [&](uint32_t v) {
#pragma unroll
            for (int off = 32; off > 0; off >>= 1)
                v += __shfl_down(v, off);
            return v;
        };
        const uint32_t n = wave_sum(st_nodes), t = wave_sum(st_tris), l = wave_sum(st_leaves), uo = wave_sum(st_unoccluded);
        if (lane == 0) {
            atomicAdd(&a.qs->nodes[ANY_HIT], (unsigned long long)n);
            atomicAdd(&a.qs->tris[ANY_HIT], (unsigned long long)t);
            atomicAdd(&a.qs->leaves[ANY_HIT], (unsigned long long)l);
            if (ANY_HIT)
                atomicAdd(&a.qs->unoccluded, (unsigned long long)uo);
        }
    }
}

template __global__ void k_traverse<false, false>(const TraverseArgs);
template __global__ void k_traverse<false, true>(const TraverseArgs);
template __global__ void k_traverse<true, false>(const TraverseArgs);
template __global__ void k_traverse<true, true>(const TraverseArgs);

void launch_traverse(const TraverseArgs& args, bool any_hit, bool stats, int grid_blocks, hipStream_t stream)
{
    const dim3 grid((unsigned)grid_blocks), block(kBlockThreads);
    if (any_hit) {
        if (stats)
            hipLaunchKernelGGL((k_traverse<true, true>), grid, block, 0, stream, args);
        else
            hipLaunchKernelGGL((k_traverse<true, false>), grid, block, 0, stream, args);
    } else {
        if (stats)
            hipLaunchKernelGGL((k_traverse<false, true>), grid, block, 0, stream, args);
        else
            hipLaunchKernelGGL((k_traverse<false, false>), grid, block, 0, stream, args);
    }
}

} // namespace igdev
