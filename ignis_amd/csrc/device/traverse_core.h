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
//
// Control flow is wave-uniform by construction: every branch and loop condition below is a __ballot / __any over the
// wave, the lanes a section does not concern run it predicated (safe addresses, results merged with selects), and only
// stores sit under per-lane conditions. Written with per-lane `if`s the same machine spends a third of its VALU
// instructions on register copies: the compiler's structurizer versions the ~35 state registers at every divergent
// region and copies them back at the merge (60 v_mov per settle() iteration, measured in the ISA).
#pragma once

#include <type_traits>

#include "dev_math.h"
#include "kernels.h"

namespace igdev {

#ifndef IG_POSTPONE_NUM
#define IG_POSTPONE_NUM 1
#endif
#ifndef IG_POSTPONE_SHIFT
#define IG_POSTPONE_SHIFT 1
#endif
constexpr int kPostponeNum   = IG_POSTPONE_NUM;   // a section needs kPostponeNum / 2^kPostponeShift of the wave's active lanes
constexpr int kPostponeShift = IG_POSTPONE_SHIFT; // (0 disables postponing)
#ifndef IG_RAY_TERMS_LDS
#define IG_RAY_TERMS_LDS 0
#endif
#ifndef IG_LDS_STACK
#define IG_LDS_STACK (IG_RAY_TERMS_LDS ? 14 : 20)
#endif
#ifndef IG_TRAV_OCC
#define IG_TRAV_OCC 4
#endif
// 20 entries * 256 threads * 8 B = 40 KiB per workgroup: four workgroups fill the 160 KiB of a CU. The scene-space ray terms (12 floats per
// lane) sat next to a 14-entry stack in LDS while the kernels needed 150 registers; since the loop restructuring of round 3 they need
// 104 - 112, the terms fit into registers below the 128 of four waves per SIMD, and their 12 KiB are six more stack entries: the rays
// of the 16 M-triangle stand-in need 16 - 24, so far fewer of them are listed and re-traversed by the DEEP launch (diamond_scene: 11).
// IG_RAY_TERMS_LDS=1 is the old arrangement.
constexpr bool kRayTermsLds = IG_RAY_TERMS_LDS != 0;
constexpr int kLdsStack     = IG_LDS_STACK;
constexpr int kTraverseOcc  = IG_TRAV_OCC;   // workgroups of 256 per CU = waves per SIMD the kernel is built for
constexpr int kBlockThreads = 256;
#ifndef IG_MASK_LOADS
#define IG_MASK_LOADS 1
#endif
#ifndef IG_LEAF_REPEAT
#define IG_LEAF_REPEAT 0
#endif
#ifndef IG_LEAF_REPEAT_MIN
#define IG_LEAF_REPEAT_MIN 1
#endif
constexpr bool kLeafRepeat   = IG_LEAF_REPEAT != 0; // the entity-leaf section repeats while a quorum of lanes is at a leaf run again
constexpr int kLeafRepeatMin = IG_LEAF_REPEAT_MIN;  // ... and at least this many
#ifndef IG_SINGLE_ROWS_EARLY
#define IG_SINGLE_ROWS_EARLY 0
#endif
constexpr bool kSingleRowsEarly = IG_SINGLE_ROWS_EARLY != 0; // entity-leaf section: rows 6 / 7 of an entered leaf with rows 2 - 5
#ifndef IG_SETTLE_CULL_TWO
#define IG_SETTLE_CULL_TWO 0
#endif
constexpr bool kSettleCullTwo = IG_SETTLE_CULL_TWO != 0; // settle(): two culled entries per trip
#ifndef IG_POSTPONE_FALLBACK_BEST
#define IG_POSTPONE_FALLBACK_BEST 0
#endif
constexpr bool kFallbackBest = IG_POSTPONE_FALLBACK_BEST != 0;
#ifndef IG_NODE_REPEAT
#define IG_NODE_REPEAT 0
#endif
#ifndef IG_NODE_REPEAT_SHARE
#define IG_NODE_REPEAT_SHARE 6
#endif
constexpr int kNodeRepeat      = IG_NODE_REPEAT;       // inner-node section: extra visits per pass while most of the wave wants one
constexpr int kNodeRepeatShare = IG_NODE_REPEAT_SHARE; // ... in eighths of 64 lanes
#ifndef IG_NODE_PUSH_FAST
#define IG_NODE_PUSH_FAST 1
#endif
constexpr bool kNodePushFast = IG_NODE_PUSH_FAST != 0; // inner-node section: stack rows by address, one bound check per node
#ifndef IG_SCAN_LEAVES
#define IG_SCAN_LEAVES 2
#endif
constexpr int kScanLeaves = IG_SCAN_LEAVES; // entity-leaf section: leaves of a run fetched per round trip
#ifndef IG_REUSE_RCP
#define IG_REUSE_RCP 1
#endif
constexpr bool kReuseRcp = IG_REUSE_RCP != 0; // entity-leaf section: scene-space reciprocals for instances that keep the direction
constexpr bool kMaskLoads = IG_MASK_LOADS != 0; // experiments: 0 = every lane loads (from a safe address where the section does not concern it)

// Per-lane LDS of one workgroup of BLOCK lanes. `e`: the traversal stacks, entry-major so that a wave's accesses are conflict
// free. `g`: the scene-space ray terms (written once per ray by begin(), read by the entity-leaf section and by inner nodes of
// the scene level) — twelve registers per lane that the shape-level sections, where a ray spends most of its steps, do not
// have to carry: g[0] = (inv_dir, inv_org.x), g[1] = (inv_org.yz, org.xy), g[2] = (org.z, dir).
template <int BLOCK>
struct StackTerms {
    float4 g[3][BLOCK];
};
struct StackNoTerms {};
template <int BLOCK>
struct StackOf : std::conditional_t<kRayTermsLds, StackTerms<BLOCK>, StackNoTerms> {
    uint2 e[kLdsStack][BLOCK];
};
using StackLds = StackOf<kBlockThreads>;

// An unspecified value that costs no instruction: what a register that is loaded under a per-lane condition holds in the other
// lanes. (Left uninitialised in C++, such a register becomes a loop-carried value the compiler zero-fills with v_mov's.)
IG_DEV float any_float()
{
    float u;
    asm volatile("" : "=v"(u));
    return u;
}
IG_DEV float4 any_float4() { return make_float4(any_float(), any_float(), any_float(), any_float()); }
IG_DEV int4 any_int4() { return make_int4((int)igm_bits(any_float()), (int)igm_bits(any_float()), (int)igm_bits(any_float()), (int)igm_bits(any_float())); }

IG_DEV int sel(bool c, int a, int b) { return c ? a : b; }
IG_DEV uint32_t sel(bool c, uint32_t a, uint32_t b) { return c ? a : b; }
IG_DEV float sel(bool c, float a, float b) { return c ? a : b; }

// DEEP: entries above the LDS part spill to global memory; the persistent kernels run without it
// and re-traverse the rare rays that need it in a second, DEEP launch (traverse.hip), because the extra work in
// every push / pop costs 5 % on scenes that never need it.
// SPHERES: the scene BVH over the analytic-sphere entities (igd_scene.sphere_*): its leaves are intersected right in the
// entity-leaf section (make_scene_local_handler_sphere, shapes/sphere.art:139-148), there is no shape level and no triangle
// section, and the ray starts from the hit the triangle pass left (driver/mapping_cpu.art:385-403).
template <bool ANY_HIT, bool STATS, int BLOCK = kBlockThreads, bool DEEP = false, bool SPHERES = false>
struct Traverser {
    using Stack = StackOf<BLOCK>;
    // ---- ray + hit
    RayT scene_ray; // the scene-space ray and its slab-test terms (in LDS instead with IG_RAY_TERMS_LDS)
    // `loc`: the ray in the current shape's space, written when an entity leaf is entered (the scene-space terms are in LDS)
    RayT loc;
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
    uint32_t sec_pass[3], sec_lane[3]; // STATS: executions of the three sections by this wave / lanes that had work in them (wave-uniform)
#ifdef IG_TRAV_CLOCKS
    // where a wave's cycles go (variant build, tools/trav_clocks.py): the shader clock at phase ends, memory counters drained first
    unsigned long long clk_last, clk_acc[6];
    IG_DEV void clk_start()
    {
        for (int k = 0; k < 6; ++k)
            clk_acc[k] = 0;
        clk_last = __builtin_readcyclecounter();
    }
    IG_DEV void mark(int k)
    {
        __builtin_amdgcn_s_waitcnt(0);
        const unsigned long long now = __builtin_readcyclecounter();
        clk_acc[k] += now - clk_last;
        clk_last = now;
    }
#else
    IG_DEV void clk_start() {}
    IG_DEV void mark(int) {}
#endif

    IG_DEV void init_counters()
    {
        overflow = false;
        st_nodes = st_tris = st_leaves = 0;
        for (int k = 0; k < 3; ++k)
            sec_pass[k] = sec_lane[k] = 0;
        // a lane without a ray: finished, mode 0 -> step() leaves it alone
        finished  = true;
        mode      = 0;
        level     = 0;
        lterm     = false;
        need_cull = false;
        top_node  = 0;
        top_tmin  = 0;
        ptr       = 0;
        ent_last  = true;
        ent_cursor = tri_cursor = 0;
        node_off = tri_off = 0;
        cur_ent  = -1;
        lbase    = 0;
        tmin = tmax = 0;
        rflags = 0;
        hit_u = hit_v = 0;
        hit_prim = hit_ent = -1;
        ltmax = l_u = l_v = 0;
        l_prim = -1;
        loc    = RayT{ f3{ 0, 0, 0 }, f3{ 0, 0, 0 }, f3{ 0, 0, 0 }, f3{ 0, 0, 0 } };
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

    // `on` lanes push (n, t). Out of stack: the ray ends here (popping a clamped slot again and again would never
    // terminate); the caller sees finished && overflow and re-traverses it with the DEEP variant or reports the error.
    IG_DEV void push_entry(Stack& st, int tid, bool on, int n, float t)
    {
        ptr += on ? 1 : 0;
        const uint2 e = make_uint2((uint32_t)n, igm_bits(t));
        if (on && ptr < kLdsStack)
            st.e[ptr][tid] = e;
        if (DEEP) {
            if (on && ptr >= kLdsStack && ptr < kLdsStack + kDeepStack)
                deep[(size_t)(ptr - kLdsStack) * deep_stride] = e;
        }
        const bool out = on & (ptr >= (DEEP ? kLdsStack + kDeepStack : kLdsStack));
        overflow       = overflow | out;
        finished       = finished | out;
    }
    // `on` lanes pop the top entry into (top_node, top_tmin)
    IG_DEV void pop_top(Stack& st, int tid, bool on)
    {
        const int lds_slot = ptr < 0 ? 0 : (ptr < kLdsStack ? ptr : kLdsStack - 1);
        uint2 e            = st.e[lds_slot][tid];
        if (DEEP) {
            if (__any(on && ptr >= kLdsStack)) {
                if (on && ptr >= kLdsStack)
                    e = deep[(size_t)((ptr < kLdsStack + kDeepStack ? ptr : kLdsStack + kDeepStack - 1) - kLdsStack) * deep_stride];
            }
        }
        top_node = sel(on, (int)e.x, top_node);
        top_tmin = sel(on, igm_float(e.y), top_tmin);
        ptr -= on ? 1 : 0;
    }

    // Starts a ray on the calling lanes (callers wrap this in their refill condition).
    IG_DEV void begin(const DevScene& sc, Stack& st, int tid, f3 org, f3 dir, float tmin_, float tmax_, uint32_t flags)
    {
        const RayT g = make_ray_terms(org, dir);
        if constexpr (kRayTermsLds) {
            st.g[0][tid] = make_float4(g.inv_dir.x, g.inv_dir.y, g.inv_dir.z, g.inv_org.x);
            st.g[1][tid] = make_float4(g.inv_org.y, g.inv_org.z, g.org.x, g.org.y);
            st.g[2][tid] = make_float4(g.org.z, g.dir.x, g.dir.y, g.dir.z);
        } else {
            scene_ray = g;
        }
        loc    = g;
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
        push_entry(st, tid, true, 0, kFltMax);
        top_node = (SPHERES ? sc.sphere_node_count : sc.scene_node_count) ? 1 : 0;
        top_tmin = tmin;
    }
    // init_hit of a later geometry pass: the hit found so far (tmax passed to begin() is its distance)
    IG_DEV void set_initial_hit(int ent, int prim, float u, float v)
    {
        hit_ent  = ent;
        hit_prim = prim;
        hit_u    = u;
        hit_v    = v;
        if (ANY_HIT)
            finished = finished | (prim >= 0); // already occluded: nothing left to find
    }

    // a lane is settled when it waits for a section (or is done): only stack-driven lanes have transitions to make
    IG_DEV bool unsettled() const { return (mode == 0) & !finished & !((top_node > 0) & !need_cull & !((level == 1) & lterm)); }

    // Cheap state transitions up to the next heavy action: an entity-leaf step (mode 2), an inner
    // node on top (mode 0, top_node > 0), a triangle packet (mode 1), or the end of the ray.
    // The cull points are exactly the reference's (mapping_cpu.art:326-347): at level entry, after
    // a leaf and after an inner node that pushed nothing. Each pass of the loop makes the transitions that need at most
    // one pop per lane:
    //   unwind (any-hit: the shape-level traversal returned early)  -> falls into `ret`
    //   cull   : the top starts behind the current hit              -> pop, stay culling
    //   ret    : sentinel on top at shape level                     -> pop the saved scene top, accept the local hit, next leaf
    //   fin    : sentinel on top at scene level                     -> the ray is done
    //   leaf   : leaf on top                                        -> pop, enter its items (or cull on if it starts behind the hit)
    IG_DEV void settle(const DevScene& sc, Stack& st, int tid)
    {
        // (a do-while on purpose: the body leaves settled lanes alone, so running it once too often is harmless, while the
        // rotated `while` form made the compiler copy the fourteen loop-carried state registers in every iteration's header)
        do {
            const bool run    = (mode == 0) & !finished;
            const bool unwind = run & (level == 1) & lterm;
            ptr      = sel(unwind, lbase, ptr);
            top_node = sel(unwind, 0, top_node);
            const float cull_t = level ? ltmax : tmax;
            const bool behind  = !(top_tmin <= cull_t);
            const bool culling = run & !unwind & need_cull & (top_node != 0) & behind;
            need_cull          = need_cull & !(run & !unwind & !culling);
            const bool rest = run & !culling;
            const bool ret  = rest & (top_node == 0) & (level == 1);
            const bool fin  = rest & (top_node == 0) & (level == 0);
            const bool leaf = rest & (top_node < 0);
            // leaf on top (mapping_cpu.art:379-381): an entry that starts behind the current
            // hit is dropped, its items have no effect in the reference either
            tri_cursor = sel(leaf & (level == 1), ~top_node, tri_cursor);
            ent_cursor = sel(leaf & (level == 0), ~top_node, ent_cursor);
            if (kSettleCullTwo && !DEEP) {
                // pop_top, reading the entry below as well: when the entry that becomes the top is itself behind the hit (the
                // siblings pushed behind a nearest child mostly are, once that child has produced a hit) it is culled in the same
                // step instead of costing the wave another trip around this loop
                const bool on     = culling | ret | leaf;
                const int row1    = ptr < 0 ? 0 : (ptr < kLdsStack ? ptr : kLdsStack - 1);
                const int row2    = row1 > 0 ? row1 - 1 : 0;
                const uint2 e1    = st.e[row1][tid], e2 = st.e[row2][tid];
                const bool second = culling & (ptr >= 1) & (e1.x != 0u) & !(igm_float(e1.y) <= cull_t);
                top_node = sel(on, (int)(second ? e2.x : e1.x), top_node);
                top_tmin = sel(on, igm_float(second ? e2.y : e1.y), top_tmin);
                ptr -= on ? (second ? 2 : 1) : 0;
            } else {
                pop_top(st, tid, culling | ret | leaf);
            }
            // shape BVH done: back to the scene leaf run (mapping_cpu.art:489-508). The local hit is
            // accepted only if its (rounded) distance does not exceed the current one.
            const bool accept = ret & (l_prim != -1) & (ltmax <= tmax);
            tmax     = sel(accept, ltmax, tmax);
            hit_u    = sel(accept, l_u, hit_u);
            hit_v    = sel(accept, l_v, hit_v);
            hit_prim = sel(accept, l_prim, hit_prim);
            hit_ent  = sel(accept, cur_ent, hit_ent);
            finished = finished | fin | (ANY_HIT & accept);
            lterm    = lterm & !ret;
            // (an any-hit ray that just accepted its hit is done: it must not be taken for a lane waiting at its next leaf)
            mode     = sel(leaf & !behind, level ? 1 : 2, sel(ret & !ent_last & !(ANY_HIT & accept), 2, mode));
            need_cull = need_cull | (ret & ent_last) | (leaf & behind);
            level     = sel(ret, 0, level);
        } while (__any(unsettled()));
    }

    // One pipeline pass: entity leaf -> inner node -> triangle packet. Every lane of the wave calls it; lanes without
    // a ray (finished) are left alone.
    IG_DEV void step(const DevScene& sc, Stack& st, int tid)
    {
        const uint8_t* geom = sc.geom;
        settle(sc, st, tid);

        // Postponing: a section runs only when enough lanes of the wave want it (they wait in their mode until
        // then), so the wave does not pay a whole section for a handful of lanes. If no section reaches the
        // quorum the threshold drops to one lane for this pass, which guarantees progress.
        int quorum = 1;
        if (kPostponeShift > 0) {
            const int active = __popcll(__ballot(!finished));
            const int n_ent  = __popcll(__ballot(mode == 2));
            const int n_node = __popcll(__ballot((mode == 0) & !finished));
            const int n_tri  = __popcll(__ballot(mode == 1));
            const int most   = n_ent > n_node ? (n_ent > n_tri ? n_ent : n_tri) : (n_node > n_tri ? n_node : n_tri);
            quorum           = (active * kPostponeNum) >> kPostponeShift;
            if (quorum < 1 || most < quorum)
                quorum = kFallbackBest ? (most > 0 ? most : 1) : 1; // no section reaches the quorum: all of them run (or, kFallbackBest, the best filled one only)
        }

        mark(4); // settle at the top of a pass + quorum
        // ---- entity leaves of the current run, up to the first one the ray enters (mapping_cpu.art:481-515)
        if (__popcll(__ballot(mode == 2)) >= quorum) {
            RayT gray = scene_ray;
            if constexpr (kRayTermsLds) {
                const float4 g0 = st.g[0][tid], g1 = st.g[1][tid], g2 = st.g[2][tid];
                gray.inv_dir = f3{ g0.x, g0.y, g0.z };
                gray.inv_org = f3{ g0.w, g1.x, g1.y };
                gray.org     = f3{ g1.z, g1.w, g2.x };
                gray.dir     = f3{ g2.y, g2.z, g2.w };
            }
            // A run whose boxes all reject the ray ends in settle(), which often pops the next run: the section repeats while a
            // quorum of lanes is at a leaf run again, instead of those lanes waiting a whole pass (both other sections, the refill
            // test, the epilogue) for every run of the scene BVH they walk past — diamond_scene: 4.6 leaves in runs of one or two.
            do {
            const bool here = mode == 2;
            if (STATS)
                sec_pass[0] += 1, sec_lane[0] += (uint32_t)__popcll(__ballot(here));
            bool scanning = here;
            bool in_tris  = false; // entered a one-leaf shape whose only box the ray hits: straight on to its triangles
            bool entered  = false; // SPHERES: the lane tested a sphere in this pass
            // (the outer loop repeats only when a lane's one-leaf shape was missed inside its entity box and the run has leaves left)
            do {
                // leaves whose box (or visibility mask) rejects the ray cost only this short loop
                bool enter    = false;
                int enter_at  = 0;
                int entity_id = 0;
                do { // (at least one lane is scanning: the quorum is >= 1)
                    const int at     = scanning ? ent_cursor : 0;
                    const float4* ls = (SPHERES ? sc.sphere_leaf_scan : sc.leaf_scan) + at * 2; // rows 0 and 1 of the records, packed: four leaves per 128-byte line
                    // (loads sit under per-lane conditions, like stores: a lane the section does not concern issues no memory
                    // request — the L1 / texture path, not the VALU, is what this kernel keeps busiest, profiles/r03_pmc_*.txt —
                    // and what it then computes from the undefined registers is discarded by the selects below)
                    // kScanLeaves: the rows of the next leaves come with the same round trip (a scan is a chain of dependent loads, 4.6
                    // leaves per ray on diamond_scene); they are looked at only if this one rejects the ray and the run goes on. The
                    // records behind the last leaf of the table are padding.
                    float4 lr[kScanLeaves][2];
#pragma unroll
                    for (int k = 0; k < kScanLeaves; ++k)
                        lr[k][0] = any_float4(), lr[k][1] = any_float4();
                    if (!kMaskLoads || scanning) {
#pragma unroll
                        for (int k = 0; k < kScanLeaves; ++k)
                            lr[k][0] = ls[2 * k], lr[k][1] = ls[2 * k + 1];
                    }
#pragma unroll
                    for (int half = 0; half < kScanLeaves; ++half) {
                        const float4 r0 = lr[half][0], r1 = lr[half][1];
                        ent_cursor += scanning ? 1 : 0;
                        const int id          = (int)igm_bits(r0.w);
                        const uint32_t lflags = igm_bits(r1.w);
                        ent_last              = scanning ? (id < 0) : ent_last;
                        if (STATS)
                            st_leaves += scanning ? 1u : 0u;
                        // check_ray_visibility (traversal/ray.art:51)
                        const bool visible = (rflags & IG_RAY_FLAG_TYPE_MASK) == ((rflags & lflags) & IG_RAY_FLAG_TYPE_MASK);
                        float entry, exit;
                        slab_test(gray, tmin, tmax, r0.x, r1.x, r0.y, r1.y, r0.z, r1.z, entry, exit);
                        const bool inside = scanning & visible & (entry <= exit) & (exit >= 0) & (entry <= tmax);
                        enter             = enter | inside;
                        enter_at          = sel(inside, at + half, enter_at);
                        entity_id         = sel(inside, id, entity_id);
                        scanning          = scanning & !inside & !(id < 0);
                    }
                } while (__any(scanning));
                if (__any(enter)) {
                    const float4* lf = (SPHERES ? sc.sphere_leaves : sc.leaves) + enter_at * kDevLeafRows;
                    float4 l2 = any_float4(), l3 = any_float4(), l4 = any_float4(), l5 = any_float4();
                    float4 blo = any_float4(), bhi = any_float4(); // (the one child box of a one-leaf shape comes with the same round trip)
                    if (!kMaskLoads || enter) {
                        l2 = lf[2], l3 = lf[3], l4 = lf[4], l5 = lf[5];
                        if (!SPHERES && kSingleRowsEarly)
                            blo = lf[6], bhi = lf[7];
                    }
                    const uint2 ext = make_uint2(igm_bits(l5.x), igm_bits(l5.y));
                    m34 m;
                    m.c0 = f3{ l2.x, l2.y, l2.z };
                    m.c1 = f3{ l2.w, l3.x, l3.y };
                    m.c2 = f3{ l3.z, l3.w, l4.x };
                    m.c3 = f3{ l4.y, l4.z, l4.w };
                    if (SPHERES) {
                        // intersect_sphere (shapes/sphere.art:107-137) with the ray in shape space: direction not normalised, t global
                        const f3 lorg = xform_point(m, gray.org), ldir = xform_dir(m, gray.dir);
                        const float4 sp = *reinterpret_cast<const float4*>(sc.shape_data + (enter ? ext.x : 0u)); // centre, radius
                        const f3 L     = lorg - f3{ sp.x, sp.y, sp.z };
                        const float S  = -dot3(L, ldir);
                        const float D2 = dot3(ldir, ldir);
                        const float L2 = dot3(L, L);
                        const float R2 = sp.w * sp.w * D2;
                        const float M2 = L2 * D2 - S * S;
                        const float Q   = igm_sqrt(R2 - M2);
                        const float t0_ = (S - Q) / D2;
                        const float t1_ = (S + Q) / D2;
                        const float t0 = t0_ > t1_ ? t1_ : t0_, t1 = t0_ > t1_ ? t0_ : t1_;
                        const float th = t0 < tmin ? t1 : t0;
                        // accepted if in range (local_hit.distance <= hit.distance is implied by th <= tmax)
                        const bool ok = enter & !((S < 0) | (M2 > R2)) & (th >= tmin) & (th <= tmax);
                        // sphere_map_uv (sphere.art:1-6)
                        const f3 n        = (L + ldir * th) * (1 / sp.w);
                        const float theta = igm_acos(n.z);
                        float phi         = igm_atan2(-n.x, n.y);
                        phi               = phi < 0 ? phi + 2 * kPi : phi;
                        tmax     = sel(ok, th, tmax);
                        hit_u    = sel(ok, phi / (2 * kPi), hit_u);
                        hit_v    = sel(ok, theta / kPi, hit_v);
                        hit_prim = sel(ok, 0, hit_prim);
                        hit_ent  = sel(ok, entity_id & 0x7FFFFFFF, hit_ent);
                        if (ANY_HIT)
                            finished = finished | ok;
                        entered = enter;
                    } else {
                        // transform_ray (traversal/ray.art:56-59): direction not normalised, t stays global
                        // A direction the matrix hands back bit for bit (an instance that is only translated: every entity of
                        // diamond_scene) has the reciprocals the scene-space ray already has: the three IEEE divisions are run only
                        // when some entering lane of the wave needs them.
                        RayT nl;
                        nl.org = xform_point(m, gray.org);
                        nl.dir = xform_dir(m, gray.dir);
                        const bool same_dir = (igm_bits(nl.dir.x) == igm_bits(gray.dir.x)) & (igm_bits(nl.dir.y) == igm_bits(gray.dir.y)) & (igm_bits(nl.dir.z) == igm_bits(gray.dir.z));
                        nl.inv_dir = gray.inv_dir;
                        if (!kReuseRcp || __any(enter & !same_dir))
                            nl.inv_dir = f3{ safe_rcp(nl.dir.x), safe_rcp(nl.dir.y), safe_rcp(nl.dir.z) };
                        nl.inv_org = -(nl.org * nl.inv_dir);
                        // A shape whose BVH is ONE node with ONE triangle leaf (a wall, a light quad; marked in bit 0 of row 5 of its leaf record by
                        // igd_assign_scene) skips the inner-node section: its root visit is the slab test of that one child, done here
                        // with the operations of the inner-node section. Hit: the state the root visit and the pop of the leaf
                        // would have left (saved scene top on the stack, sentinel on top, in the triangle leaf). Miss: the state
                        // `ret` would have restored, i.e. as if the entity's box had rejected the ray, and the run is scanned on.
                        const bool single = enter & ((ext.x & 1u) != 0);
                        bool missed       = false;
                        int only_leaf     = 0;
                        if (__any(single)) {
                            if (!kSingleRowsEarly) {
                                if (!kMaskLoads || single)
                                    blo = lf[6], bhi = lf[7];
                            }
                            only_leaf = (int)igm_bits(l5.z);
                            // (near / far plane by the sign of the inverse direction, as the inner-node section picks its rows)
                            const bool ox = nl.inv_dir.x < 0, oy = nl.inv_dir.y < 0, oz = nl.inv_dir.z < 0;
                            const float nx = sel(ox, bhi.x, blo.x), fx = sel(ox, blo.x, bhi.x);
                            const float ny = sel(oy, bhi.y, blo.y), fy = sel(oy, blo.y, bhi.y);
                            const float nz = sel(oz, bhi.z, blo.z), fz = sel(oz, blo.z, bhi.z);
                            const float entry = igm_max(igm_max(igm_fma(nl.inv_dir.x, nx, nl.inv_org.x), igm_fma(nl.inv_dir.y, ny, nl.inv_org.y)), igm_max(igm_fma(nl.inv_dir.z, nz, nl.inv_org.z), tmin));
                            const float exit  = igm_min(igm_min(igm_fma(nl.inv_dir.x, fx, nl.inv_org.x), igm_fma(nl.inv_dir.y, fy, nl.inv_org.y)), igm_min(igm_fma(nl.inv_dir.z, fz, nl.inv_org.z), tmax));
                            missed     = single & (exit < entry);
                            if (STATS)
                                st_nodes += single ? 1u : 0u;
                        }
                        const bool go   = enter & !missed;
                        const bool tris = single & !missed;
                        loc.org.x = sel(go, nl.org.x, loc.org.x), loc.org.y = sel(go, nl.org.y, loc.org.y), loc.org.z = sel(go, nl.org.z, loc.org.z);
                        loc.dir.x = sel(go, nl.dir.x, loc.dir.x), loc.dir.y = sel(go, nl.dir.y, loc.dir.y), loc.dir.z = sel(go, nl.dir.z, loc.dir.z);
                        loc.inv_dir.x = sel(go, nl.inv_dir.x, loc.inv_dir.x), loc.inv_dir.y = sel(go, nl.inv_dir.y, loc.inv_dir.y), loc.inv_dir.z = sel(go, nl.inv_dir.z, loc.inv_dir.z);
                        loc.inv_org.x = sel(go, nl.inv_org.x, loc.inv_org.x), loc.inv_org.y = sel(go, nl.inv_org.y, loc.inv_org.y), loc.inv_org.z = sel(go, nl.inv_org.z, loc.inv_org.z);
                        cur_ent = sel(go, entity_id & 0x7FFFFFFF, cur_ent);
                        // save the scene-level top, then a fresh stack: sentinel + shape root (one-leaf shapes: the sentinel is already
                        // back on top, the root and its leaf entry have come and gone)
                        push_entry(st, tid, go, top_node, top_tmin);
                        lbase  = sel(go, ptr, lbase);
                        ltmax  = sel(go, tmax, ltmax); // invalid_hit(local_ray.tmax)
                        l_prim = sel(go, -1, l_prim);
                        lterm  = lterm & !go;
                        push_entry(st, tid, go & !tris, 0, kFltMax);
                        top_node = sel(go, tris ? 0 : 1, top_node);
                        top_tmin = sel(go, tris ? kFltMax : tmin, top_tmin);
                        level    = sel(go, 1, level);
                        node_off = sel(go, ext.x & ~1u, node_off);
                        tri_off  = sel(go, ext.y, tri_off);
                        tri_cursor = sel(tris, ~only_leaf, tri_cursor);
                        in_tris    = in_tris | tris;
                        // the one-leaf shape was missed: on with the run, if it has leaves left
                        scanning = missed & !ent_last;
                    }
                }
            } while (!SPHERES && __any(scanning));
            if (SPHERES) {
                // a leaf run continues after a sphere test (the shape level of the triangle pass comes back through settle()'s
                // `ret`; here the run is resumed directly): lanes that entered and have leaves left scan on in the next pass
                const bool more = here & entered & !ent_last & !finished;
                mode            = sel(here & !more, 0, mode);
                need_cull       = need_cull | (here & !more);
            } else {
                mode      = sel(here, in_tris ? 1 : 0, mode);
                need_cull = need_cull | (here & !in_tris);
            }
            settle(sc, st, tid);
            } while (kLeafRepeat && !SPHERES && __popcll(__ballot(mode == 2)) >= (quorum > kLeafRepeatMin ? quorum : kLeafRepeatMin));
        }

        mark(1); // entity-leaf section (with its settle)
        // ---- one inner node: fetch 256 B, test 8 children (mapping_cpu.art:350-377)
        if (__popcll(__ballot((mode == 0) & !finished)) >= quorum) {
            // kNodeRepeat: where rays walk many nodes between leaves (deep trees) most lanes are at an inner node again after the visit;
            // the section then repeats, up to kNodeRepeat more times, while at least kNodeRepeatShare / 8 of the wave is — without the trip
            // through the other sections' tests, the quorum and the loop's refill / finish checks in between
            int again = kNodeRepeat;
            do {
            const bool here   = (mode == 0) & !finished; // settled: an inner node is on top
            const uint8_t* np = geom + (SPHERES ? sc.sphere_nodes_off : (level ? node_off : sc.scene_nodes_off)) + (here ? (uint32_t)(top_node - 1) * 256u : 0u);
            pop_top(st, tid, here);
            const float4* nf = reinterpret_cast<const float4*>(np);
            const int4* nc   = reinterpret_cast<const int4*>(np) + 12;
            if (STATS) {
                st_nodes += here ? 1u : 0u;
                sec_pass[1] += 1, sec_lane[1] += (uint32_t)__popcll(__ballot(here));
            }
            bool pushed           = false;
            // (after the pop: ptr names the row of the last entry, -1 for none)
            const int slot_at  = (ptr * BLOCK + tid) * (int)sizeof(uint2); // (negative for an empty stack: signed comparisons)
            const int slot_end = (kLdsStack * BLOCK + tid) * (int)sizeof(uint2);
            int slot           = slot_at;
            const float node_tmax = level ? ltmax : tmax;
            f3 inv = loc.inv_dir, io = loc.inv_org;
            if (__any(here & (level == 0))) { // scene-level node: the terms of the untransformed ray
                f3 ginv = scene_ray.inv_dir, gio = scene_ray.inv_org;
                if constexpr (kRayTermsLds) {
                    const float4 g0 = st.g[0][tid];
                    const float2 g1 = *reinterpret_cast<const float2*>(&st.g[1][tid]);
                    ginv = f3{ g0.x, g0.y, g0.z }, gio = f3{ g0.w, g1.x, g1.y };
                }
                const bool s = level == 0;
                inv = f3{ sel(s, ginv.x, inv.x), sel(s, ginv.y, inv.y), sel(s, ginv.z, inv.z) };
                io  = f3{ sel(s, gio.x, io.x), sel(s, gio.y, io.y), sel(s, gio.z, io.z) };
            }
            // The slab test of the reference takes min / max of the two plane distances per axis
            // (intersection.art:38-58); which plane is the near one is decided by the sign of inv_dir alone
            // (fma is monotonic and lo <= hi), so the near / far rows are picked by address instead and six
            // of the eighteen min / max per child disappear. Results are bit-identical for real children
            // (empty slots are masked by child == 0).
            const int ox = inv.x < 0 ? 1 : 0, oy = inv.y < 0 ? 1 : 0, oz = inv.z < 0 ? 1 : 0;
            // two halves of four children keep the live register set small
            int4 c4lo = any_int4(), c4hi = any_int4(); // both halves' child ids with the first batch of loads: the test for the second half does not cost a round trip of its own
            if (!kMaskLoads || here)
                c4lo = nc[0], c4hi = nc[1];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int4 c4 = h ? c4hi : c4lo;
                if (h == 1 && !__any(here & ((c4.x | c4.y | c4.z | c4.w) != 0)))
                    break; // no lane has a child in the second half
                float4 nx = any_float4(), fx = any_float4(), ny = any_float4(), fy = any_float4(), nz = any_float4(), fz = any_float4();
                if (!kMaskLoads || here) {
                    nx = nf[2 * ox + h], fx = nf[2 * (1 - ox) + h];
                    ny = nf[2 * (2 + oy) + h], fy = nf[2 * (3 - oy) + h];
                    nz = nf[2 * (4 + oz) + h], fz = nf[2 * (5 - oz) + h];
                }
                const float nb[3][4] = { { nx.x, nx.y, nx.z, nx.w }, { ny.x, ny.y, ny.z, ny.w }, { nz.x, nz.y, nz.z, nz.w } };
                const float fb[3][4] = { { fx.x, fx.y, fx.z, fx.w }, { fy.x, fy.y, fy.z, fy.w }, { fz.x, fz.y, fz.z, fz.w } };
                const int ch[4] = { c4.x, c4.y, c4.z, c4.w };
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float entry = igm_max(igm_max(igm_fma(inv.x, nb[0][i], io.x), igm_fma(inv.y, nb[1][i], io.y)), igm_max(igm_fma(inv.z, nb[2][i], io.z), tmin));
                    const float exit  = igm_min(igm_min(igm_fma(inv.x, fb[0][i], io.x), igm_fma(inv.y, fb[1][i], io.y)), igm_min(igm_fma(inv.z, fb[2][i], io.z), node_tmax));
                    const bool hit = here & (ch[i] != 0) & !(exit < entry);
                    // push (becomes the top) if nearer than the current top, else push_after
                    const bool front = ANY_HIT || (top_tmin > entry);
                    if (kNodePushFast && !DEEP) {
                        // the stack slot as an LDS address that moves up by one row per hit child; how far the node got, and whether that
                        // was too far, is read off the address once after the eight children (push_entry keeps count, tests the bound and
                        // flags the overflow per entry: 4 of its 10 instructions per child)
                        slot += hit ? BLOCK * (int)sizeof(uint2) : 0;
                        if (hit & (slot < slot_end)) {
                            uint32_t* w = reinterpret_cast<uint32_t*>(reinterpret_cast<uint8_t*>(&st.e[0][0]) + slot);
                            w[0] = (uint32_t)(front ? top_node : ch[i]), w[1] = igm_bits(front ? top_tmin : entry);
                        }
                    } else {
                        push_entry(st, tid, hit, front ? top_node : ch[i], front ? top_tmin : entry);
                        pushed = pushed | hit;
                    }
                    top_node = sel(hit & front, ch[i], top_node);
                    top_tmin = sel(hit & front, entry, top_tmin);
                }
            }
            if (kNodePushFast && !DEEP) {
                pushed         = slot != slot_at;
                ptr += (slot - slot_at) / (BLOCK * (int)sizeof(uint2)); // (lanes the section does not concern: + 0)
                const bool out = here & (slot >= slot_end); // out of stack: the ray ends here (see push_entry)
                overflow       = overflow | out;
                finished       = finished | out;
            }
            need_cull = need_cull | (here & !pushed);
            settle(sc, st, tid);
            } while (kNodeRepeat > 0 && again-- > 0 && __popcll(__ballot((mode == 0) & !finished)) >= kNodeRepeatShare * 8);
        }

        mark(2); // inner-node section (with its settle)
        // ---- the Tri4 packets of a leaf (mapping_cpu.art:379-410)
        if (!SPHERES && __popcll(__ballot(mode == 1)) >= quorum) {
            do { // (at least one lane is in a triangle leaf: the quorum is >= 1)
                const bool here   = mode == 1;
                if (STATS)
                    sec_pass[2] += 1, sec_lane[2] += (uint32_t)__popcll(__ballot(here));
                const uint8_t* tp = geom + tri_off + (here ? (uint32_t)tri_cursor * 208u : 0u);
                tri_cursor += here ? 1 : 0;
                // two triangles of the packet at a time (a 96-byte half of the re-ordered packet): 24 live registers instead
                // of 48, and the second half is not even fetched when no lane's packet holds more than two triangles
                int4 pid4 = any_int4();
                if (!kMaskLoads || here)
                    pid4 = reinterpret_cast<const int4*>(tp)[12];
                const int pid[4] = { pid4.x, pid4.y, pid4.z, pid4.w };
                bool valid       = here;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    if (h == 1 && !__any(valid & (pid[2] != -1) & !(ANY_HIT & lterm)))
                        break;
                    // half h of the packet: 96 contiguous bytes, float 2 k + j = row k of triangle 2 h + j (igd_assign_scene re-orders the
                    // reference's Tri4 that way): six 16-byte loads per half
                    const float4* th = reinterpret_cast<const float4*>(tp) + 6 * h;
                    float4 c[6];
#pragma unroll
                    for (int m = 0; m < 6; ++m)
                        c[m] = any_float4();
                    if (!kMaskLoads || here) {
#pragma unroll
                        for (int m = 0; m < 6; ++m)
                            c[m] = th[m];
                    }
                    float q[12][2];
#pragma unroll
                    for (int m = 0; m < 6; ++m)
                        q[2 * m][0] = c[m].x, q[2 * m][1] = c[m].y, q[2 * m + 1][0] = c[m].z, q[2 * m + 1][1] = c[m].w;
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const int i   = 2 * h + j;
                        valid         = valid & (pid[i] != -1);
                        const bool on = valid & !(ANY_HIT & lterm);
                        if (STATS)
                            st_tris += on ? 1u : 0u;
                        float t = 0, u = 0, v = 0;
                        const bool ok = tri_test_flat(loc, tmin, ltmax, f3{ q[0][j], q[1][j], q[2][j] }, f3{ q[3][j], q[4][j], q[5][j] },
                                                 f3{ q[6][j], q[7][j], q[8][j] }, f3{ q[9][j], q[10][j], q[11][j] }, t, u, v) & on;
                        ltmax  = sel(ok, t, ltmax);
                        l_u    = sel(ok, u, l_u);
                        l_v    = sel(ok, v, l_v);
                        l_prim = sel(ok, pid[i] & 0x7FFFFFFF, l_prim);
                        if (ANY_HIT)
                            lterm = lterm | ok;
                    }
                }
                const bool leave = here & ((pid[3] < 0) | (ANY_HIT & lterm));
                mode             = sel(leave, 0, mode);
                need_cull        = need_cull | leave;
            } while (__any(mode == 1));
            if (ANY_HIT) {
                if (__any(lterm))
                    settle(sc, st, tid); // return to the scene level now: the hit may end the ray
            }
        }
        mark(3); // triangle section
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
