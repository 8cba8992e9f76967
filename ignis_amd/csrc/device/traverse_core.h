// traverse_core.h — per-lane two-level BVH8 traversal state machine (gfx950).
//
// One ray per lane. The stack lives in LDS, [entry][thread] layout (conflict-free
// ds_read/write_b64). Scene level and shape level share ONE inner-node section and ONE stack:
// entering an instance saves the scene top, pushes a sentinel and switches the node base offset.
// step() runs the sections in pipeline order  entity leaf -> inner node -> triangle packet, with
// cheap state transitions (settle) in between, so a lane can walk a whole instance (leaf test,
// shape root, triangles) in one step and the lanes of a wave stay in phase.
//
// Per-ray semantics (visit order, culling points, acceptance `t <= tmax`, hence tie-breaking) are
// those of the reference CPU device, `cpu_traverse_helper(_prim)` with vector width 1
// (src/artic/traversal/mapping_cpu.art:282-518), on the same Node8 / Tri4 / EntityLeaf1 bytes,
// so hits AND the visited-node / tested-triangle counts equal the CPU oracle's
// (DESIGN.md "Traversal order").
#pragma once

#include "dev_math.h"
#include "kernels.h"

namespace igdev {

constexpr int kPostponeNum   = 1;   // a section needs kPostponeNum / 2^kPostponeShift of the wave's active lanes
constexpr int kPostponeShift = 1;   // (0 disables postponing)
constexpr int kLdsStack     = 24;  // 24 entries * 256 threads * 8 B = 48 KiB per workgroup (3 workgroups per CU)
constexpr int kBlockThreads = 256;

// per-lane traversal stacks of one workgroup of BLOCK lanes, entry-major so that a wave's accesses are conflict free
template <int BLOCK>
using StackOf = uint2[kLdsStack][BLOCK];
using StackLds = StackOf<kBlockThreads>;

// DEEP: entries above the LDS part spill to global memory (push_entry); the persistent kernels run without it
// and re-traverse the rare rays that need it in a second, DEEP launch (traverse.hip), because the extra branch in
// every push / pop costs 5 % on scenes that never need it.
template <bool ANY_HIT, bool STATS, int BLOCK = kBlockThreads, bool DEEP = false>
struct Traverser {
    using Stack = StackOf<BLOCK>;
    // ---- ray + hit
    // `gray` (the ray in scene space) is never written after begin(); `loc` (the ray in the current shape's
    // space) only when an entity leaf is entered. Nothing in settle() touches either, which keeps the twelve
    // registers of each out of every control-flow merge of the state machine.
    RayT gray, loc;
    float tmin, tmax; // tmax == distance of the accepted hit (ray.tmax shrinks with it)
    uint32_t rflags;
    float hit_u, hit_v;
    int hit_prim, hit_ent;
    // ---- hit of the shape-level traversal in flight (local_hit / local ray.tmax of handle_local)
    float ltmax, l_u, l_v;
    int l_prim;
    int lbase;  // stack pointer of the saved scene-level top
    bool lterm;  // any-hit: the shape-level traversal found its hit
    // ---- control
    int top_node;
    float top_tmin;
    int ptr;
    int level; // 0 scene BVH, 1 shape BVH
    int mode;  // 0 stack driven, 1 inside a triangle leaf, 2 inside an entity leaf run
    int ent_cursor, tri_cursor;
    uint32_t node_off, tri_off;
    int cur_ent;
    bool ent_last, need_cull, finished, overflow;
    uint2* deep; // this lane's column of the deep-stack buffer
    uint32_t deep_stride;
    uint32_t st_nodes, st_tris, st_leaves;

    IG_DEV void init_counters()
    {
        overflow = false;
        st_nodes = st_tris = st_leaves = 0;
        finished = true;
    }

    // The reference's stack has 64 entries and no overflow check (traversal/stack.art:53-54). Here the first
    // kLdsStack entries of a lane live in LDS; entries above that (deep BVHs only) go to the lane's column of a
    // global buffer (kDeepStack more entries, [entry][lane] so that a wave's accesses coalesce). Beyond both,
    // `overflow` is set: the launch raises error bit 0 and igd_render fails loudly.
    IG_DEV void attach_deep(uint2* lane_column, uint32_t stride)
    {
        deep        = lane_column;
        deep_stride = stride;
    }
    IG_DEV void push_entry(Stack& st, int tid, int n, float t)
    {
        ++ptr;
        const uint2 e = make_uint2((uint32_t)n, igm_bits(t));
        if (ptr < kLdsStack)
            st[ptr][tid] = e;
        else if (DEEP && ptr < kLdsStack + kDeepStack)
            deep[(size_t)(ptr - kLdsStack) * deep_stride] = e;
        else {
            // out of stack: the ray ends here (popping a clamped slot again and again would never terminate);
            // the caller sees finished && overflow and re-traverses it with the DEEP variant or reports the error
            overflow = true;
            finished = true;
        }
    }
    IG_DEV void pop_top(Stack& st, int tid)
    {
        uint2 e;
        if (!DEEP)
            e = st[ptr < kLdsStack ? ptr : kLdsStack - 1][tid];
        else if (ptr < kLdsStack)
            e = st[ptr][tid];
        else
            e = deep[(size_t)((ptr < kLdsStack + kDeepStack ? ptr : kLdsStack + kDeepStack - 1) - kLdsStack) * deep_stride];
        top_node = (int)e.x;
        top_tmin = igm_float(e.y);
        --ptr;
    }

    IG_DEV void begin(const DevScene& sc, Stack& st, int tid, f3 org, f3 dir, float tmin_, float tmax_, uint32_t flags)
    {
        gray   = make_ray_terms(org, dir);
        loc    = gray;
        tmin   = tmin_;
        tmax   = tmax_;
        rflags = flags;
        overflow = false;
        hit_u = hit_v = 0;
        hit_prim = hit_ent = -1;
        level = 0, mode = 0;
        ent_cursor = tri_cursor = 0;
        tri_off    = 0;
        cur_ent    = -1;
        ent_last   = true;
        need_cull  = true;
        finished   = false;
        ltmax = l_u = l_v = 0;
        l_prim = -1;
        lbase  = 0;
        lterm  = false;
        node_off   = 0; // Node8[] of the entered shape; the scene level uses sc.scene_nodes_off
        // stack.push(root, ray.tmin) on an empty stack: sentinel below, root on top
        ptr      = -1;
        top_node = 0, top_tmin = kFltMax;
        push_entry(st, tid, top_node, top_tmin);
        top_node = sc.scene_node_count ? 1 : 0;
        top_tmin = tmin;
    }

    // Cheap state transitions up to the next heavy action: an entity-leaf step (mode 2), an inner
    // node on top (mode 0, top_node > 0), a triangle packet (mode 1), or the end of the ray.
    // The cull points are exactly the reference's (mapping_cpu.art:326-347): at level entry, after
    // a leaf and after an inner node that pushed nothing.
    IG_DEV void settle(const DevScene& sc, Stack& st, int tid)
    {
        while (mode == 0 && !finished) {
            if (level == 1 && lterm) {
                // any-hit: the shape-level traversal returned early; unwind its stack entries
                ptr      = lbase;
                top_node = 0;
            } else if (need_cull) {
                const float cull_t = level ? ltmax : tmax;
                while (top_node != 0 && !(top_tmin <= cull_t))
                    pop_top(st, tid);
                need_cull = false;
            }
            if (top_node == 0) {
                if (level == 1) {
                    // shape BVH done: back to the scene leaf run (mapping_cpu.art:489-508). The local hit is
                    // accepted only if its (rounded) distance does not exceed the current one.
                    level = 0;
                    lterm = false;
                    pop_top(st, tid); // saved scene-level top
                    if (l_prim != -1 && ltmax <= tmax) {
                        tmax     = ltmax;
                        hit_u    = l_u;
                        hit_v    = l_v;
                        hit_prim = l_prim;
                        hit_ent  = cur_ent;
                        if (ANY_HIT)
                            finished = true;
                    }
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
                const bool active = top_tmin <= (level ? ltmax : tmax);
                if (level)
                    tri_cursor = ~top_node;
                else
                    ent_cursor = ~top_node;
                pop_top(st, tid);
                if (active)
                    mode = level ? 1 : 2;
                else
                    need_cull = true;
            }
        }
    }

    // One pipeline pass: entity leaf -> inner node -> triangle packet. Call while !finished.
    IG_DEV void step(const DevScene& sc, Stack& st, int tid)
    {
        const uint8_t* geom = sc.geom;
        settle(sc, st, tid);

        // Postponing: a section runs only when enough lanes of the wave want it (they wait in their mode until
        // then), so the wave does not pay a whole section for a handful of lanes. If no section reaches the
        // quorum the threshold drops to one lane for this pass, which guarantees progress.
        int quorum = 1;
        if (kPostponeShift > 0) {
            const int active = __popcll(__ballot(true));
            const int n_ent  = __popcll(__ballot(mode == 2));
            const int n_node = __popcll(__ballot(mode == 0 && !finished));
            const int n_tri  = __popcll(__ballot(mode == 1));
            const int most   = n_ent > n_node ? (n_ent > n_tri ? n_ent : n_tri) : (n_node > n_tri ? n_node : n_tri);
            quorum           = (active * kPostponeNum) >> kPostponeShift;
            if (quorum < 1 || most < quorum)
                quorum = 1;
        }

        // ---- entity leaves of the current run, up to the first one the ray enters (mapping_cpu.art:481-515)
        if (mode == 2 && (quorum <= 1 || __popcll(__ballot(mode == 2)) >= quorum)) {
            // leaves whose box (or visibility mask) rejects the ray cost only this short loop
            const float4* lf;
            uint2 ext;
            int entity_id;
            bool enter;
            do {
                lf  = reinterpret_cast<const float4*>(sc.leaves + ent_cursor);
                ext = sc.leaf_ext[ent_cursor];
                ++ent_cursor;
                const float4 l0 = lf[0], l1 = lf[1], l5 = lf[5];
                entity_id             = (int)igm_bits(l0.w);
                const uint32_t lflags = igm_bits(l5.x);
                ent_last              = entity_id < 0;
                if (STATS)
                    ++st_leaves;
                enter = false;
                // check_ray_visibility (traversal/ray.art:51)
                if ((rflags & IG_RAY_FLAG_TYPE_MASK) == ((rflags & lflags) & IG_RAY_FLAG_TYPE_MASK)) {
                    float entry, exit;
                    slab_test(gray, tmin, tmax, l0.x, l1.x, l0.y, l1.y, l0.z, l1.z, entry, exit);
                    enter = (entry <= exit) & (exit >= 0) & (entry <= tmax);
                }
            } while (!enter && !ent_last);
            if (enter) {
                const float4 l2 = lf[2], l3 = lf[3], l4 = lf[4];
                m34 m;
                m.c0 = f3{ l2.x, l2.y, l2.z };
                m.c1 = f3{ l2.w, l3.x, l3.y };
                m.c2 = f3{ l3.z, l3.w, l4.x };
                m.c3 = f3{ l4.y, l4.z, l4.w };
                // transform_ray (traversal/ray.art:56-59): direction not normalised, t stays global
                loc     = make_ray_terms(xform_point(m, gray.org), xform_dir(m, gray.dir));
                cur_ent = entity_id & 0x7FFFFFFF;
                // save the scene-level top, then a fresh stack: sentinel + shape root
                push_entry(st, tid, top_node, top_tmin);
                lbase  = ptr;
                ltmax  = tmax; // invalid_hit(local_ray.tmax)
                l_prim = -1;
                lterm  = false;
                push_entry(st, tid, 0, kFltMax);
                top_node  = 1;
                top_tmin  = tmin;
                level     = 1;
                mode      = 0;
                need_cull = true;
                node_off  = ext.x;
                tri_off   = ext.y;
            } else {
                mode      = 0;
                need_cull = true;
            }
            settle(sc, st, tid);
        }

        // ---- one inner node: fetch 256 B, test 8 children (mapping_cpu.art:350-377)
        if (mode == 0 && !finished && (quorum <= 1 || __popcll(__ballot(mode == 0 && !finished)) >= quorum)) {
            const uint8_t* np = geom + (level ? node_off : sc.scene_nodes_off) + (uint32_t)(top_node - 1) * 256u;
            pop_top(st, tid);
            const float4* nf = reinterpret_cast<const float4*>(np);
            const int4* nc   = reinterpret_cast<const int4*>(np) + 12;
            if (STATS)
                ++st_nodes;
            bool pushed           = false;
            const float node_tmax = level ? ltmax : tmax;
            const f3 inv = level ? loc.inv_dir : gray.inv_dir;
            const f3 io  = level ? loc.inv_org : gray.inv_org;
            // The slab test of the reference takes min / max of the two plane distances per axis
            // (intersection.art:38-58); which plane is the near one is decided by the sign of inv_dir alone
            // (fma is monotonic and lo <= hi), so the near / far rows are picked by address instead and six
            // of the eighteen min / max per child disappear. Results are bit-identical for real children
            // (empty slots are masked by child == 0).
            const int ox = inv.x < 0 ? 1 : 0, oy = inv.y < 0 ? 1 : 0, oz = inv.z < 0 ? 1 : 0;
            // two halves of four children keep the live register set small
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const float4 nx = nf[2 * ox + h], fx = nf[2 * (1 - ox) + h];
                const float4 ny = nf[2 * (2 + oy) + h], fy = nf[2 * (3 - oy) + h];
                const float4 nz = nf[2 * (4 + oz) + h], fz = nf[2 * (5 - oz) + h];
                const float nb[3][4] = { { nx.x, nx.y, nx.z, nx.w }, { ny.x, ny.y, ny.z, ny.w }, { nz.x, nz.y, nz.z, nz.w } };
                const float fb[3][4] = { { fx.x, fx.y, fx.z, fx.w }, { fy.x, fy.y, fy.z, fy.w }, { fz.x, fz.y, fz.z, fz.w } };
                const int4 c4   = nc[h];
                const int ch[4] = { c4.x, c4.y, c4.z, c4.w };
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float entry = igm_max(igm_max(igm_fma(inv.x, nb[0][i], io.x), igm_fma(inv.y, nb[1][i], io.y)), igm_max(igm_fma(inv.z, nb[2][i], io.z), tmin));
                    const float exit  = igm_min(igm_min(igm_fma(inv.x, fb[0][i], io.x), igm_fma(inv.y, fb[1][i], io.y)), igm_min(igm_fma(inv.z, fb[2][i], io.z), node_tmax));
                    const bool hit = (ch[i] != 0) & !(exit < entry);
                    if (hit) {
                        // push (becomes the top) if nearer than the current top, else push_after
                        const bool front = ANY_HIT || (top_tmin > entry);
                        push_entry(st, tid, front ? top_node : ch[i], front ? top_tmin : entry);
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
            settle(sc, st, tid);
        }

        // ---- the Tri4 packets of a leaf (mapping_cpu.art:379-410)
        const bool run_tri = quorum <= 1 || __popcll(__ballot(mode == 1)) >= quorum;
        while (run_tri && mode == 1) {
            const uint8_t* tp = geom + tri_off + (uint32_t)tri_cursor * 208u;
            ++tri_cursor;
            const float4* tf = reinterpret_cast<const float4*>(tp);
            float q[12][4];
#pragma unroll
            for (int k = 0; k < 12; ++k) {
                const float4 x = tf[k];
                q[k][0] = x.x, q[k][1] = x.y, q[k][2] = x.z, q[k][3] = x.w;
            }
            const int4 pid4  = reinterpret_cast<const int4*>(tp)[12];
            const int pid[4] = { pid4.x, pid4.y, pid4.z, pid4.w };
            bool valid       = true;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                valid = valid & (pid[i] != -1);
                if (valid && !(ANY_HIT && lterm)) {
                    if (STATS)
                        ++st_tris;
                    float t, u, v;
                    if (tri_test(loc, tmin, ltmax, f3{ q[0][i], q[1][i], q[2][i] }, f3{ q[3][i], q[4][i], q[5][i] },
                                 f3{ q[6][i], q[7][i], q[8][i] }, f3{ q[9][i], q[10][i], q[11][i] }, t, u, v)) {
                        ltmax  = t;
                        l_u    = u;
                        l_v    = v;
                        l_prim = pid[i] & 0x7FFFFFFF;
                        if (ANY_HIT)
                            lterm = true;
                    }
                }
            }
            if (pid[3] < 0 || (ANY_HIT && lterm)) {
                mode      = 0;
                need_cull = true;
            }
        }
        if (ANY_HIT && lterm)
            settle(sc, st, tid); // return to the scene level now: the hit may end the ray
    }
};

// wave-level sum of a per-lane counter
IG_DEV uint32_t wave_sum_u32(uint32_t v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
        v += __shfl_down(v, off);
    return v;
}

} // namespace igdev
