// traverse_core.h — per-lane two-level BVH8 traversal state machine (gfx950).
//
// One ray per lane. The stack lives in LDS, [entry][thread] layout (conflict-free
// ds_read/write_b64). Scene level and shape level share ONE inner-node section and ONE stack:
// entering an instance saves the scene top, pushes a sentinel and switches the node base offset.
// step() runs the sections in pipeline order  entity leaf -> inner node -> triangle packet, so a
// lane can walk a whole instance (leaf test, shape root, triangles) in one step and the lanes of
// a wave stay in phase.
//
// Per-ray semantics (visit order, culling points, acceptance `t <= tmax`, hence tie-breaking) are
// those of the reference CPU device, `cpu_traverse_helper(_prim)` with vector width 1
// (src/artic/traversal/mapping_cpu.art:282-518), on the same Node8 / Tri4 / EntityLeaf1 bytes,
// so hits AND the visited-node / tested-triangle counts equal the CPU oracle's
// (DESIGN.md "Traversal order").
//
// Control flow (round 4): WHETHER a section runs is decided per wave (__ballot against the quorum); INSIDE a section the
// lanes it concerns run it under the hardware's EXEC mask — plain per-lane `if`s and loops — and update their state
// registers in place. A lane that leaves a section is `settled`: the cheap stack transitions up to its next heavy action
// (settle()) run inside the section, for its lanes only. Rounds 2 / 3 ran every section for all 64 lanes and merged with
// selects (a fifth of the kernel's VALU instructions were v_cndmask, another tenth v_mov).
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
#ifndef IG_LDS_STACK
#define IG_LDS_STACK 20
#endif
#ifndef IG_TRAV_OCC
#define IG_TRAV_OCC 4
#endif
// 20 entries * 256 threads * 8 B = 40 KiB per workgroup: four workgroups fill the 160 KiB of a CU. The rays of the 16 M-triangle
// stand-in need 16 - 24 entries (diamond_scene: 11); what does not fit is re-traversed by the DEEP launch.
constexpr int kLdsStack     = IG_LDS_STACK;
constexpr int kTraverseOcc  = IG_TRAV_OCC;   // workgroups of 256 per CU = waves per SIMD the kernel is built for
constexpr int kBlockThreads = 256;
constexpr int kScanLeaves   = 2; // entity-leaf section: leaves of a run fetched per round trip (the host builder keeps runs <= 2)

// Per-lane LDS of one workgroup of BLOCK lanes: the traversal stacks, entry-major so that a wave's accesses are conflict free.
template <int BLOCK>
struct StackOf {
    uint2 e[kLdsStack][BLOCK];
};
using StackLds = StackOf<kBlockThreads>;

// DEEP: entries above the LDS part spill to global memory; the persistent kernels run without it
// and re-traverse the rare rays that need it in a second, DEEP launch (traverse.hip), because the extra work in
// every push / pop costs 5 % on scenes that never need it.
// SPHERES: the scene BVH over the analytic-sphere entities (igd_scene.sphere_*): its leaves are intersected right in the
// entity-leaf section (make_scene_local_handler_sphere, shapes/sphere.art:139-148), there is no shape level and no triangle
// section, and the ray starts from the hit the triangle pass left (driver/mapping_cpu.art:385-403).
template <bool ANY_HIT, bool STATS, int BLOCK = kBlockThreads, bool DEEP = false, bool SPHERES = false>
struct Traverser {
    using Stack = StackOf<BLOCK>;
    static constexpr int kRow    = BLOCK * (int)sizeof(uint2);  // bytes between two entries of a lane's stack
    static constexpr int kLdsEnd = kLdsStack * kRow;            // first byte offset (+ lane) behind the LDS part
    // what a lane waits for
    static constexpr int kNode = 0; // an inner node is on top of its stack
    static constexpr int kTri  = 1; // inside a triangle leaf
    static constexpr int kLeaf = 2; // inside an entity-leaf run
    static constexpr int kDone = 3; // the ray ended (or the lane has none)
    static constexpr int kSettle = 4; // stack driven, transitions pending (inside step() only)

    // ---- ray + hit
    RayT scene_ray;   // the scene-space ray and its slab-test terms
    f3 lorg, ldir;    // the ray in the current shape's space (triangle tests)
    f3 inv, io;       // slab-test terms of the level the lane is on: scene_ray's, or the transformed ray's
    float tmin;
    float tmax;       // cull distance of the current level. Scene level: distance of the accepted hit (ray.tmax shrinks with it);
                      // shape level: the local ray's (local_hit / local ray.tmax of handle_local)
    float scene_tmax; // shape level: the scene level's distance, kept for the return
    uint32_t rflags;
    float hit_u, hit_v;
    int hit_prim, hit_ent;
    float l_u, l_v; // hit of the shape-level traversal in flight
    int l_prim;
    int lbase;   // stack position of the saved scene-level top
    bool lterm;  // any-hit: the shape-level traversal found its hit
    // ---- control
    int top_node;
    float top_tmin;
    int sp;      // byte offset of the entry below the top inside Stack::e (this lane's column): (entry * BLOCK + tid) * 8
    int sp_end;  // ... of the first entry this lane's stack does not have
    int level;   // 0 scene BVH, 1 shape BVH
    int mode;
    int ent_cursor, tri_cursor;
    uint32_t nodes_off; // Node8[] of the level the lane is on (byte offset inside geom)
    uint32_t tri_off;
    int cur_ent;
    bool ent_last, need_cull, overflow;
    uint2* deep; // this lane's column of the deep-stack buffer
    uint32_t deep_stride;
    uint32_t st_nodes, st_tris, st_leaves;
    uint32_t sec_pass[3], sec_lane[3]; // STATS: executions of the three sections / lanes that had work in them (per lane: see count_section)
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

    IG_DEV bool finished() const { return mode == kDone; }

    // STATS: one execution of section k by the calling lanes. The counters are per lane (callers sum them over the wave): every
    // calling lane counts itself, the first of them the execution.
    IG_DEV void count_section(int k)
    {
        if (STATS) {
            const unsigned long long m = __ballot(true);
            sec_lane[k] += 1u;
            sec_pass[k] += (__lane_id() == (unsigned)(__ffsll((long long)m) - 1)) ? 1u : 0u;
        }
    }

    IG_DEV void init_counters()
    {
        overflow = false;
        st_nodes = st_tris = st_leaves = 0;
        for (int k = 0; k < 3; ++k)
            sec_pass[k] = sec_lane[k] = 0;
        mode = kDone; // a lane without a ray: no section of step() acts on it
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

    IG_DEV uint2& slot(Stack& st, int at) { return *reinterpret_cast<uint2*>(reinterpret_cast<uint8_t*>(&st.e[0][0]) + at); }

    // push (n, t) below the cached top. Out of stack: the ray ends here (popping a clamped slot again and again would never
    // terminate); the caller sees finished() && overflow and re-traverses it with the DEEP variant or reports the error.
    IG_DEV void push_entry(Stack& st, int n, float t)
    {
        sp += kRow;
        const uint2 e = make_uint2((uint32_t)n, igm_bits(t));
        if (sp < sp_end) {
            slot(st, sp) = e;
        } else {
            bool out = true;
            if (DEEP) {
                const uint32_t k = (uint32_t)(sp - sp_end) / (uint32_t)kRow;
                if (k < (uint32_t)kDeepStack) {
                    deep[(size_t)k * deep_stride] = e;
                    out                           = false;
                }
            }
            if (out)
                overflow = true, mode = kDone;
        }
    }
    // pop the entry below the top into (top_node, top_tmin)
    IG_DEV void pop_top(Stack& st)
    {
        uint2 e;
        if (!DEEP || sp < sp_end) {
            e = slot(st, sp);
        } else {
            const uint32_t k = (uint32_t)(sp - sp_end) / (uint32_t)kRow;
            e                = deep[(size_t)(k < (uint32_t)kDeepStack ? k : (uint32_t)kDeepStack - 1u) * deep_stride];
        }
        top_node = (int)e.x;
        top_tmin = igm_float(e.y);
        sp -= kRow;
    }

    // Starts a ray on the calling lanes (callers wrap this in their refill condition).
    IG_DEV void begin(const DevScene& sc, Stack& st, int tid, f3 org, f3 dir, float tmin_, float tmax_, uint32_t flags)
    {
        scene_ray = make_ray_terms(org, dir);
        inv       = scene_ray.inv_dir;
        io        = scene_ray.inv_org;
        tmin      = tmin_;
        tmax      = tmax_;
        rflags    = flags;
        overflow  = false;
        hit_u = hit_v = 0;
        hit_prim = hit_ent = -1;
        level     = 0;
        ent_last  = true;
        lterm     = false;
        nodes_off = SPHERES ? sc.sphere_nodes_off : sc.scene_nodes_off;
        // stack.push(root, ray.tmin) on an empty stack: sentinel below, root on top
        sp     = tid * (int)sizeof(uint2);
        sp_end = sp + kLdsEnd;
        slot(st, sp) = make_uint2(0u, igm_bits(kFltMax));
        top_node = 1;
        top_tmin = tmin;
        // the cull at level entry (mapping_cpu.art:326-347): a root that starts behind tmax is popped, which leaves the sentinel
        need_cull = false;
        mode      = (((SPHERES ? sc.sphere_node_count : sc.scene_node_count) != 0) & (tmin_ <= tmax_)) ? kNode : kDone;
    }
    // init_hit of a later geometry pass: the hit found so far (tmax passed to begin() is its distance)
    IG_DEV void set_initial_hit(int ent, int prim, float u, float v)
    {
        hit_ent  = ent;
        hit_prim = prim;
        hit_u    = u;
        hit_v    = v;
        if (ANY_HIT) {
            if (prim >= 0)
                mode = kDone; // already occluded: nothing left to find
        }
    }

    // Cheap state transitions of a stack-driven lane (mode kNode on entry) up to its next heavy action: an inner node on
    // top (kNode), a triangle packet (kTri), an entity-leaf run (kLeaf), or the end of the ray (kDone).
    // The cull points are exactly the reference's (mapping_cpu.art:326-347): at level entry, after
    // a leaf and after an inner node that pushed nothing.
    //   unwind (any-hit: the shape-level traversal returned early)  -> falls into `ret`
    //   cull   : the top starts behind the current hit              -> pop, stay culling
    //   ret    : sentinel on top at shape level                     -> pop the saved scene top, accept the local hit, next leaf
    //   fin    : sentinel on top at scene level                     -> the ray is done
    //   leaf   : leaf on top                                        -> pop, enter its items (or cull on if it starts behind the hit)
    IG_DEV void settle(const DevScene& sc, Stack& st)
    {
        // (a wave-uniform loop around a loop-free per-lane body: with a per-lane loop here the structurizer versions the state
        // registers the body may write — the return to the scene level writes a dozen — at the loop's header and exits)
        do {
            if (mode == kSettle)
                settle_once(sc, st);
        } while (__any(mode == kSettle));
    }
    IG_DEV void settle_once(const DevScene& sc, Stack& st)
    {
        // (decide, pop once, apply: one flat region per kind of transition, and the control words merged with selects, keep the
        // compiler from versioning registers across nested regions)
        if (ANY_HIT) {
            if ((level == 1) & lterm)
                sp = lbase, top_node = 0;
        }
        const bool sentinel = top_node == 0;
        const bool ret      = sentinel & (level == 1);
        const bool fin      = sentinel & (level == 0);
        const bool behind   = !(top_tmin <= tmax);
        const bool culling  = !sentinel & need_cull & behind;
        const bool leaf     = !sentinel & !culling & (top_node < 0);
        const bool node     = !sentinel & !culling & (top_node > 0);
        const int cursor    = ~top_node;
        if (culling | ret | leaf)
            pop_top(st);
        bool accept = false;
        if (ret) {
            // shape BVH done: back to the scene leaf run (mapping_cpu.art:489-508). The local hit is
            // accepted only if its (rounded) distance does not exceed the current one.
            accept = (l_prim != -1) & (tmax <= scene_tmax);
            hit_u    = accept ? l_u : hit_u;
            hit_v    = accept ? l_v : hit_v;
            hit_prim = accept ? l_prim : hit_prim;
            hit_ent  = accept ? cur_ent : hit_ent;
            tmax     = accept ? tmax : scene_tmax;
            inv       = scene_ray.inv_dir;
            io        = scene_ray.inv_org;
            nodes_off = sc.scene_nodes_off;
            level     = 0;
            lterm     = false;
        }
        // leaf on top (mapping_cpu.art:379-381): an entry that starts behind the current
        // hit is dropped, its items have no effect in the reference either
        const bool enter_leaf = leaf & !behind;
        tri_cursor = (enter_leaf & (level != 0)) ? cursor : tri_cursor;
        ent_cursor = (enter_leaf & (level == 0)) ? cursor : ent_cursor;
        // (an any-hit ray that just accepted its hit is done: it must not be taken for a lane waiting at its next leaf)
        const bool done = fin | (ANY_HIT & accept);
        const bool next = ret & !done & !ent_last; // on with the leaf run
        need_cull = (need_cull & culling) | (ret & !done & ent_last) | (leaf & behind);
        mode      = done ? kDone : (next ? kLeaf : (enter_leaf ? (level ? kTri : kLeaf) : (node ? kNode : kSettle)));
    }

    // The three sections below are called by the whole wave; all their loops are wave-uniform (conditions are __any over the wave),
    // the per-lane work inside sits under per-lane conditions.

    // ---- entity leaves of the current run, up to the first one the ray enters (mapping_cpu.art:481-515); lanes in kLeaf
    IG_DEV void leaf_section(const DevScene& sc, Stack& st)
    {
        const RayT& gray = scene_ray;
        if (mode == kLeaf)
            count_section(0);
        bool scanning = mode == kLeaf;
        // (the outer loop repeats only when a lane's one-leaf shape was missed inside its entity box and the run has leaves left,
        // or, SPHERES, after a sphere test)
        do {
            // leaves whose box (or visibility mask) rejects the ray cost only this short loop
            bool enter    = false;
            int enter_at  = 0;
            int entity_id = 0;
            do {
                if (scanning) {
                    // rows 0 and 1 of the records, packed: four leaves per 128-byte line. kScanLeaves: the rows of the next leaves come
                    // with the same round trip (a scan is a chain of dependent loads); they are looked at only if this one rejects the
                    // ray and the run goes on. The records behind the last leaf of the table are padding.
                    const int at     = ent_cursor;
                    const float4* ls = (SPHERES ? sc.sphere_leaf_scan : sc.leaf_scan) + at * 2;
                    float4 lr[kScanLeaves][2];
#pragma unroll
                    for (int k = 0; k < kScanLeaves; ++k)
                        lr[k][0] = ls[2 * k], lr[k][1] = ls[2 * k + 1];
#pragma unroll
                    for (int k = 0; k < kScanLeaves; ++k) {
                        if (k == 0 || scanning) {
                            const float4 r0 = lr[k][0], r1 = lr[k][1];
                            ent_cursor += 1;
                            const int id          = (int)igm_bits(r0.w);
                            const uint32_t lflags = igm_bits(r1.w);
                            ent_last              = id < 0;
                            if (STATS)
                                st_leaves += 1u;
                            // check_ray_visibility (traversal/ray.art:51)
                            const bool visible = (rflags & IG_RAY_FLAG_TYPE_MASK) == ((rflags & lflags) & IG_RAY_FLAG_TYPE_MASK);
                            float entry, exit;
                            slab_test(gray, tmin, tmax, r0.x, r1.x, r0.y, r1.y, r0.z, r1.z, entry, exit);
                            const bool inside = visible & (entry <= exit) & (exit >= 0) & (entry <= tmax);
                            if (inside)
                                enter = true, enter_at = at + k, entity_id = id;
                            scanning = !inside & !(id < 0);
                        }
                    }
                }
            } while (__any(scanning));
            if (enter) {
                const float4* lf = (SPHERES ? sc.sphere_leaves : sc.leaves) + enter_at * kDevLeafRows;
                const float4 l2 = lf[2], l3 = lf[3], l4 = lf[4], l5 = lf[5];
                const uint2 ext = make_uint2(igm_bits(l5.x), igm_bits(l5.y));
                m34 m;
                m.c0 = f3{ l2.x, l2.y, l2.z };
                m.c1 = f3{ l2.w, l3.x, l3.y };
                m.c2 = f3{ l3.z, l3.w, l4.x };
                m.c3 = f3{ l4.y, l4.z, l4.w };
                if (SPHERES) {
                    // intersect_sphere (shapes/sphere.art:107-137) with the ray in shape space: direction not normalised, t global
                    const f3 so = xform_point(m, gray.org), sd = xform_dir(m, gray.dir);
                    const float4 sp4 = *reinterpret_cast<const float4*>(sc.shape_data + ext.x); // centre, radius
                    const f3 L     = so - f3{ sp4.x, sp4.y, sp4.z };
                    const float S  = -dot3(L, sd);
                    const float D2 = dot3(sd, sd);
                    const float L2 = dot3(L, L);
                    const float R2 = sp4.w * sp4.w * D2;
                    const float M2 = L2 * D2 - S * S;
                    const float Q   = igm_sqrt(R2 - M2);
                    const float t0_ = (S - Q) / D2;
                    const float t1_ = (S + Q) / D2;
                    const float t0 = t0_ > t1_ ? t1_ : t0_, t1 = t0_ > t1_ ? t0_ : t1_;
                    const float th = t0 < tmin ? t1 : t0;
                    // accepted if in range (local_hit.distance <= hit.distance is implied by th <= tmax)
                    const bool ok = !((S < 0) | (M2 > R2)) & (th >= tmin) & (th <= tmax);
                    if (ok) {
                        // sphere_map_uv (sphere.art:1-6)
                        const f3 n        = (L + sd * th) * (1 / sp4.w);
                        const float theta = igm_acos(n.z);
                        float phi         = igm_atan2(-n.x, n.y);
                        phi               = phi < 0 ? phi + 2 * kPi : phi;
                        tmax     = th;
                        hit_u    = phi / (2 * kPi);
                        hit_v    = theta / kPi;
                        hit_prim = 0;
                        hit_ent  = entity_id & 0x7FFFFFFF;
                        if (ANY_HIT)
                            mode = kDone;
                    }
                    // a leaf run continues after a sphere test (the shape level of the triangle pass comes back through settle()'s `ret`)
                    scanning = !ent_last & !(ANY_HIT & ok);
                } else {
                    // transform_ray (traversal/ray.art:56-59): direction not normalised, t stays global
                    // A direction the matrix hands back bit for bit (an instance that is only translated: every entity of
                    // diamond_scene) has the reciprocals the scene-space ray already has: the three IEEE divisions are run only
                    // when some entering lane of the wave needs them.
                    // (written straight into the lane's shape-space ray: at the scene level nothing reads lorg / ldir, and the slab-test
                    // terms are put back if the shape turns out to be missed)
                    lorg = xform_point(m, gray.org);
                    ldir = xform_dir(m, gray.dir);
                    const bool same_dir = (igm_bits(ldir.x) == igm_bits(gray.dir.x)) & (igm_bits(ldir.y) == igm_bits(gray.dir.y)) & (igm_bits(ldir.z) == igm_bits(gray.dir.z));
                    if (!same_dir)
                        inv = f3{ safe_rcp(ldir.x), safe_rcp(ldir.y), safe_rcp(ldir.z) };
                    io = -(lorg * inv);
                    // A shape whose BVH is ONE node with ONE triangle leaf (a wall, a light quad; marked in bit 0 of row 5 of its leaf record by
                    // igd_assign_scene) skips the inner-node section: its root visit is the slab test of that one child, done here
                    // with the operations of the inner-node section. Hit: the state the root visit and the pop of the leaf
                    // would have left (saved scene top on the stack, sentinel on top, in the triangle leaf). Miss: the state
                    // `ret` would have restored, i.e. as if the entity's box had rejected the ray, and the run is scanned on.
                    const bool single = (ext.x & 1u) != 0;
                    bool missed       = false;
                    if (single) {
                        const float4 blo = lf[6], bhi = lf[7];
                        // (near / far plane by the sign of the inverse direction, as the inner-node section picks its rows)
                        const bool ox = inv.x < 0, oy = inv.y < 0, oz = inv.z < 0;
                        const float nx = ox ? bhi.x : blo.x, fx = ox ? blo.x : bhi.x;
                        const float ny = oy ? bhi.y : blo.y, fy = oy ? blo.y : bhi.y;
                        const float nz = oz ? bhi.z : blo.z, fz = oz ? blo.z : bhi.z;
                        const float entry = igm_max(igm_max(igm_fma(inv.x, nx, io.x), igm_fma(inv.y, ny, io.y)), igm_max(igm_fma(inv.z, nz, io.z), tmin));
                        const float exit  = igm_min(igm_min(igm_fma(inv.x, fx, io.x), igm_fma(inv.y, fy, io.y)), igm_min(igm_fma(inv.z, fz, io.z), tmax));
                        missed            = exit < entry;
                        if (STATS)
                            st_nodes += 1u;
                    }
                    if (!missed) {
                        cur_ent = entity_id & 0x7FFFFFFF;
                        // save the scene-level top, then a fresh stack: sentinel + shape root (one-leaf shapes: the sentinel is already
                        // back on top, the root and its leaf entry have come and gone)
                        mode = single ? kTri : kSettle; // (a push that runs out of stack makes it kDone)
                        push_entry(st, top_node, top_tmin);
                        lbase      = sp;
                        scene_tmax = tmax; // invalid_hit(local_ray.tmax): the local distance starts from the scene level's
                        l_prim     = -1;
                        lterm      = false;
                        level      = 1;
                        nodes_off  = ext.x & ~1u;
                        tri_off    = ext.y;
                        if (single) {
                            top_node   = 0;
                            top_tmin   = kFltMax;
                            tri_cursor = ~(int)igm_bits(l5.z);
                        } else {
                            push_entry(st, 0, kFltMax);
                            top_node  = 1;
                            top_tmin  = tmin;
                            need_cull = true; // the cull at level entry
                        }
                    } else {
                        // the one-leaf shape was missed: on with the run, if it has leaves left
                        inv      = gray.inv_dir;
                        io       = gray.inv_org;
                        scanning = !ent_last;
                    }
                }
            }
        } while (__any(scanning));
        if (mode == kLeaf) {
            // the run is over and nothing was entered
            mode      = kSettle;
            need_cull = true;
        }
        settle(sc, st);
    }

    // ---- one inner node: fetch 256 B, test 8 children (mapping_cpu.art:350-377); lanes in kNode
    IG_DEV void node_section(const DevScene& sc, Stack& st)
    {
        if (mode == kNode) {
            const uint8_t* np = sc.geom + nodes_off + (uint32_t)(top_node - 1) * 256u;
            pop_top(st);
            const float4* nf = reinterpret_cast<const float4*>(np);
            const int4* nc   = reinterpret_cast<const int4*>(np) + 12;
            if (STATS)
                st_nodes += 1u;
            count_section(1);
            const int sp_before = sp;
            // The slab test of the reference takes min / max of the two plane distances per axis
            // (intersection.art:38-58); which plane is the near one is decided by the sign of inv_dir alone
            // (fma is monotonic and lo <= hi), so the near / far rows are picked by address instead and six
            // of the eighteen min / max per child disappear. Results are bit-identical for real children
            // (empty slots are masked by child == 0).
            const int ox = inv.x < 0 ? 1 : 0, oy = inv.y < 0 ? 1 : 0, oz = inv.z < 0 ? 1 : 0;
            // both halves' child ids with the first batch of loads: the test for the second half does not cost a round trip of its own
            const int4 c4lo = nc[0], c4hi = nc[1];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int4 c4 = h ? c4hi : c4lo;
                // two halves of four children keep the live register set small; children are packed from slot 0 (the first zero ends the
                // list, mapping_cpu.art:357)
                if (h == 1 && c4.x == 0)
                    break;
                const float4 nx = nf[2 * ox + h], fx = nf[2 * (1 - ox) + h];
                const float4 ny = nf[2 * (2 + oy) + h], fy = nf[2 * (3 - oy) + h];
                const float4 nz = nf[2 * (4 + oz) + h], fz = nf[2 * (5 - oz) + h];
                const float nb[3][4] = { { nx.x, nx.y, nx.z, nx.w }, { ny.x, ny.y, ny.z, ny.w }, { nz.x, nz.y, nz.z, nz.w } };
                const float fb[3][4] = { { fx.x, fx.y, fx.z, fx.w }, { fy.x, fy.y, fy.z, fy.w }, { fz.x, fz.y, fz.z, fz.w } };
                const int ch[4] = { c4.x, c4.y, c4.z, c4.w };
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float entry = igm_max(igm_max(igm_fma(inv.x, nb[0][i], io.x), igm_fma(inv.y, nb[1][i], io.y)), igm_max(igm_fma(inv.z, nb[2][i], io.z), tmin));
                    const float exit  = igm_min(igm_min(igm_fma(inv.x, fb[0][i], io.x), igm_fma(inv.y, fb[1][i], io.y)), igm_min(igm_fma(inv.z, fb[2][i], io.z), tmax));
                    const bool hit    = (ch[i] != 0) & !(exit < entry);
                    if (hit) {
                        // push (becomes the top) if nearer than the current top, else push_after
                        const bool front = ANY_HIT || (top_tmin > entry);
                        const int pn     = front ? top_node : ch[i];
                        const float pt   = front ? top_tmin : entry;
                        if (DEEP) {
                            push_entry(st, pn, pt);
                        } else {
                            // (how far the node got, and whether that was too far, is read off the address once after the eight children)
                            sp += kRow;
                            if (sp < sp_end)
                                slot(st, sp) = make_uint2((uint32_t)pn, igm_bits(pt));
                        }
                        if (front)
                            top_node = ch[i], top_tmin = entry;
                    }
                }
            }
            if (!DEEP) {
                if (sp >= sp_end)
                    overflow = true, mode = kDone; // out of stack: the ray ends here (see push_entry)
            }
            if (mode != kDone) {
                need_cull = sp == sp_before; // nothing pushed: cull (mapping_cpu.art:377)
                mode      = kSettle;
            }
        }
        settle(sc, st);
    }

    // ---- the Tri4 packets of a leaf (mapping_cpu.art:379-410); lanes in kTri
    IG_DEV void tri_section(const DevScene& sc, Stack& st)
    {
        RayT lr;
        lr.org = lorg, lr.dir = ldir;
        do {
            if (mode == kTri) {
                count_section(2);
                const uint8_t* tp = sc.geom + tri_off + (uint32_t)tri_cursor * 208u;
                tri_cursor += 1;
                // two triangles of the packet at a time (a 96-byte half of the re-ordered packet): 24 live registers instead
                // of 48, and the second half is not even fetched when the packet holds no more than two triangles
                const int4 pid4  = reinterpret_cast<const int4*>(tp)[12];
                const int pid[4] = { pid4.x, pid4.y, pid4.z, pid4.w };
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    // (valid triangles are packed from slot 0: the first -1 ends the packet, mapping_cpu.art:386)
                    // (the first half comes with the ids: a packet has at least one triangle)
                    if (h == 1 && (pid[2] == -1 || (ANY_HIT && lterm)))
                        break;
                    // half h of the packet: 96 contiguous bytes, float 2 k + j = row k of triangle 2 h + j (igd_assign_scene re-orders the
                    // reference's Tri4 that way): six 16-byte loads per half
                    const float4* th = reinterpret_cast<const float4*>(tp) + 6 * h;
                    float4 c[6];
#pragma unroll
                    for (int m = 0; m < 6; ++m)
                        c[m] = th[m];
                    float q[12][2];
#pragma unroll
                    for (int m = 0; m < 6; ++m)
                        q[2 * m][0] = c[m].x, q[2 * m][1] = c[m].y, q[2 * m + 1][0] = c[m].z, q[2 * m + 1][1] = c[m].w;
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const int i   = 2 * h + j;
                        const bool on = (pid[i] != -1) & !(ANY_HIT & lterm);
                        if (on) {
                            if (STATS)
                                st_tris += 1u;
                            float t, u, v;
                            if (tri_test(lr, tmin, tmax, f3{ q[0][j], q[1][j], q[2][j] }, f3{ q[3][j], q[4][j], q[5][j] },
                                         f3{ q[6][j], q[7][j], q[8][j] }, f3{ q[9][j], q[10][j], q[11][j] }, t, u, v)) {
                                tmax   = t;
                                l_u    = u;
                                l_v    = v;
                                l_prim = pid[i] & 0x7FFFFFFF;
                                if (ANY_HIT)
                                    lterm = true;
                            }
                        }
                    }
                }
                if ((pid[3] < 0) | (ANY_HIT & lterm)) {
                    mode      = kSettle;
                    need_cull = true;
                }
            }
        } while (__any(mode == kTri));
        settle(sc, st);
    }

    // One pipeline pass: entity leaf -> inner node -> triangle packet. Every lane of the wave calls it; lanes without
    // a ray (kDone) are left alone.
    IG_DEV void step(const DevScene& sc, Stack& st, int tid)
    {
        // Postponing: a section runs only when enough lanes of the wave want it (they wait in their mode until
        // then), so the wave does not pay a whole section for a handful of lanes. If no section reaches the
        // quorum the threshold drops to one lane for this pass, which guarantees progress.
        int quorum = 1;
        if (kPostponeShift > 0) {
            const int n_ent  = __popcll(__ballot(mode == kLeaf));
            const int n_node = __popcll(__ballot(mode == kNode));
            const int n_tri  = __popcll(__ballot(mode == kTri));
            const int active = n_ent + n_node + n_tri; // (a lane with a ray waits in exactly one of the three)
            const int most   = n_ent > n_node ? (n_ent > n_tri ? n_ent : n_tri) : (n_node > n_tri ? n_node : n_tri);
            quorum           = (active * kPostponeNum) >> kPostponeShift;
            if (quorum < 1 || most < quorum)
                quorum = 1; // no section reaches the quorum: all of them run
        }
        mark(4); // quorum at the top of a pass
        if (__popcll(__ballot(mode == kLeaf)) >= quorum)
            leaf_section(sc, st);
        mark(1); // entity-leaf section (with its settle)
        if (__popcll(__ballot(mode == kNode)) >= quorum)
            node_section(sc, st);
        mark(2); // inner-node section (with its settle)
        if (!SPHERES) {
            if (__popcll(__ballot(mode == kTri)) >= quorum)
                tri_section(sc, st);
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
