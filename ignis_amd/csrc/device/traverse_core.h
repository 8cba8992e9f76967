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
// Control flow (round 4): every per-lane boolean of the machine — what a lane waits for, need_cull, the level, ... — is a 64-bit
// lane mask in scalar registers (`mask_t`), combined on the scalar unit; every branch and loop is wave-uniform (a mask is or is
// not empty); and the per-lane work of a section sits in flat regions `if (in(mask))`, i.e. under the hardware's EXEC mask,
// where the lanes concerned update their state registers in place. A lane that leaves a section is `settled`: the cheap
// stack transitions up to its next heavy action (settle(): a tight pop loop for culled entries, then one classification)
// run right there, for those lanes only. Rounds 2 / 3 ran every section for all 64 lanes and merged with selects (a fifth of
// the kernel's VALU instructions were v_cndmask, another tenth v_mov); per-lane `if`s around per-lane
// loops make the structurizer version the state registers instead (measured in the ISA, tools/isa_histogram.py).
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
using mask_t = unsigned long long;
IG_DEV mask_t lanes_where(bool c) { return __builtin_amdgcn_ballot_w64(c); }          // the active lanes for which c holds
IG_DEV bool in(mask_t m) { return __builtin_amdgcn_inverse_ballot_w64(m); }           // is the calling lane in m (m wave-uniform)
IG_DEV int lanes_in(mask_t m) { return __builtin_popcountll(m); }
// Two rules keep the masks in scalar registers. The compiler's uniformity analysis calls every phi at the join of a divergent
// branch divergent, and its CFG simplification folds an empty join block into the next one — the head of a loop, the join of a
// wave-uniform `if` — whose phis carry the masks. So (i) a region `if (in(m)) { ... }` is closed by region_end(), an empty
// volatile asm that keeps the region's join a block of its own, and (ii) masks are never assigned under any `if`: their updates
// are unconditional scalar code (an empty region costs its two scalar instructions either way).
IG_DEV void region_end() { asm volatile(""); }
// a register with no particular value, at no instruction: for state that is written before it is read by the lanes that count
// (a v_mov of 0 per register otherwise: the one-time prologue, the early rows of the leaf scan)
IG_DEV float any_f32()
{
    float x;
    asm volatile("" : "=v"(x));
    return x;
}
IG_DEV f3 any_f3() { return f3{ any_f32(), any_f32(), any_f32() }; }
// -DIG_ISA_MARKS: comment lines in the assembly at the borders of the kernel's parts, for tools/isa_regions.py (static
// instructions per part; with the event counts of tools/trav_events.py: the dynamic mix). Analysis builds only.
#ifdef IG_ISA_MARKS
#define IG_MARK(name) asm volatile("; @@ " name)
#else
#define IG_MARK(name) ((void)0)
#endif

// 16 bytes at base + off + 16 * row: a wave-uniform base and a 32-bit per-lane byte offset, so the load takes its address as
// SGPR pair + VGPR offset + immediate (no 64-bit address arithmetic per lane)
IG_DEV float4 ld16(const void* base, uint32_t off, int row = 0) { return reinterpret_cast<const float4*>(static_cast<const uint8_t*>(base) + off)[row]; }
IG_DEV int4 ld16i(const void* base, uint32_t off, int row = 0) { return reinterpret_cast<const int4*>(static_cast<const uint8_t*>(base) + off)[row]; }

// 16 bytes at a wave-uniform address through the scalar cache (s_load_dwordx4: the constant address space tells the compiler so): the
// vector L1 path, which the traversal loads to a third to a half of its 64 B / clk (and keeps pending 97 % of the time; TCP / TD
// counters, tools/tcp_pmc.sh), does not see the request
typedef float scalar_f4 __attribute__((ext_vector_type(4)));
typedef int scalar_i4 __attribute__((ext_vector_type(4)));
IG_DEV float4 ld16s(const void* base, uint32_t off)
{
    const scalar_f4 v = *(const __attribute__((address_space(4))) scalar_f4*)(uintptr_t)(static_cast<const uint8_t*>(base) + off);
    return make_float4(v.x, v.y, v.z, v.w);
}
IG_DEV int4 ld16si(const void* base, uint32_t off)
{
    const scalar_i4 v = *(const __attribute__((address_space(4))) scalar_i4*)(uintptr_t)(static_cast<const uint8_t*>(base) + off);
    return make_int4(v.x, v.y, v.z, v.w);
}
#ifndef IG_ROOT_SCALAR
#define IG_ROOT_SCALAR 1 // the visit of the scene root inside begin(), from scalar loads, when the new rays agree on their direction's octant
#endif

// QNODE: the inner nodes are the 128-byte quantised records igd_assign_scene packs when every Node8 of the scene is losslessly
// representable that way (kernels.h, DevScene::node_format): one cache line and six 16-byte requests per visit instead of two lines and
// fourteen. The planes are decoded to the very floats the Node8 holds (fma(q, 2^e, origin): the host's builder wrote those), then tested
// as usual: hits and counters stay those of the reference order on the same tables.
template <bool ANY_HIT, bool STATS, int BLOCK = kBlockThreads, bool DEEP = false, bool SPHERES = false, bool QNODE = false>
struct Traverser {
    static constexpr uint32_t kNodeBytes = QNODE ? 128u : 256u;
    using Stack = StackOf<BLOCK>;
    static constexpr int kRow    = BLOCK * (int)sizeof(uint2);  // bytes between two entries of a lane's stack
    // an entry's byte offset is entry * kRow + tid * 8 with tid * 8 < kRow, so `offset < kLdsEnd` says `entry < kLdsStack` for every lane
    static constexpr int kLdsEnd = kLdsStack * kRow;
    // a node's twelve rows / a packet's twelve rows in one round trip (k_traverse), or the second half after a look at its ids (k_tail, whose
    // register file is the shading code's: the 24 more live registers are 60 B more scratch there and cost the late passes 5 %)
    static constexpr bool kOneTrip = BLOCK == kBlockThreads;

    // ---- what the lanes wait for (a lane with a ray is in exactly one of the three between two sections; in none: no ray, or done)
    mask_t m_node; // an inner node is on top of the stack
    mask_t m_tri;  // inside a triangle leaf
    mask_t m_leaf; // inside an entity-leaf run
    // ---- per-lane flags
    mask_t need_cull; // the cull points of the reference (mapping_cpu.art:326-347): at level entry, after a leaf, after a node that pushed nothing
    mask_t ent_last;  // the entity leaf scanned last ends its run
    mask_t level1;    // in a shape BVH (else: the scene BVH)
    mask_t lterm;     // any-hit: the shape-level traversal found its hit
    mask_t overflow;  // the ray ran out of stack (it is in no mode mask any more)
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
    // ---- control
    uint2 top;   // the cached top of the stack: (node, bits of its entry distance) — one register pair, so a pop is one ds_read_b64 into it
    int sp;      // byte offset of the entry below the top inside Stack::e (this lane's column): (entry * BLOCK + tid) * 8
    int ent_cursor, tri_cursor;
    uint32_t nodes_off; // Node8[] of the level the lane is on (byte offset inside geom)
    uint32_t tri_off;
    int cur_ent;
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
#ifdef IG_TRAV_PROFILE
    // how often each part of the kernel runs (variant build, tools/trav_events.py): event k counted once per execution by the
    // calling lanes' first (so a block under a per-lane condition counts once per wave that enters it); `lanes` counts the callers
    uint32_t ev[12];
    IG_DEV void prof_start()
    {
        for (int k = 0; k < 12; ++k)
            ev[k] = 0;
    }
    IG_DEV void prof(int k, bool lanes = false)
    {
        const mask_t m = lanes_where(true);
        ev[k] += (lanes || __lane_id() == (unsigned)(__ffsll((long long)m) - 1)) ? 1u : 0u;
    }
#else
    IG_DEV void prof_start() {}
    IG_DEV void prof(int, bool = false) {}
#endif

    IG_DEV mask_t active() const { return m_node | m_tri | m_leaf; } // lanes whose ray is not finished
    IG_DEV bool finished() const { return !in(active()); }
    IG_DEV bool overflowed() const { return in(overflow); }

    // STATS: one execution of section k by the calling lanes. The counters are per lane (callers sum them over the wave): every
    // calling lane counts itself, the first of them the execution.
    IG_DEV void count_section(int k)
    {
        if (STATS) {
            const mask_t m = lanes_where(true);
            sec_lane[k] += 1u;
            sec_pass[k] += (__lane_id() == (unsigned)(__ffsll((long long)m) - 1)) ? 1u : 0u;
        }
    }

    // Once per kernel: no lane has a ray. (Every register of the state gets a value: lanes outside a region compute with theirs
    // where a mask is formed from a comparison before it is restricted to the region.)
    IG_DEV void init_counters()
    {
        m_node = m_tri = m_leaf = 0;
        need_cull = level1 = lterm = overflow = 0;
        ent_last  = ~0ull;
        st_nodes = st_tris = st_leaves = 0;
        for (int k = 0; k < 3; ++k)
            sec_pass[k] = sec_lane[k] = 0;
        // (ray and hit: begin() writes them for every lane that gets a ray, and no other lane's are looked at)
        scene_ray = RayT{ any_f3(), any_f3(), any_f3(), any_f3() };
        lorg = any_f3(), ldir = any_f3(), inv = any_f3(), io = any_f3();
        tmin = any_f32(), tmax = any_f32(), scene_tmax = any_f32();
        rflags = 0;
        hit_u = any_f32(), hit_v = any_f32(), l_u = any_f32(), l_v = any_f32();
        hit_prim = hit_ent = l_prim = -1;
        lbase = sp = 0;
        top = make_uint2(0u, 0u);
        ent_cursor = tri_cursor = 0;
        nodes_off = tri_off = 0;
        cur_ent = -1;
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

    // (per lane, inside a region) push (n, t) below the cached top; returns true when the lane is out of stack: its ray ends
    // there (popping a clamped slot again and again would never terminate), the caller takes it out of the mode masks and into
    // `overflow`, and the kernel re-traverses it with the DEEP variant or reports the error.
    IG_DEV bool push_entry(Stack& st, int n, float t)
    {
        sp += kRow;
        const uint2 e = make_uint2((uint32_t)n, igm_bits(t));
        bool out      = false;
        if (sp < kLdsEnd) {
            slot(st, sp) = e;
        } else {
            out = true;
            if (DEEP) {
                const uint32_t k = (uint32_t)(sp - kLdsEnd) / (uint32_t)kRow;
                if (k < (uint32_t)kDeepStack) {
                    deep[(size_t)k * deep_stride] = e;
                    out                           = false;
                }
            }
        }
        return out;
    }
    // (per lane, inside a region) pop the entry below the top into (top_node, top_tmin)
    IG_DEV void pop_top(Stack& st)
    {
        uint2 e;
        if (!DEEP || sp < kLdsEnd) {
            e = slot(st, sp);
        } else {
            const uint32_t k = (uint32_t)(sp - kLdsEnd) / (uint32_t)kRow;
            e                = deep[(size_t)(k < (uint32_t)kDeepStack ? k : (uint32_t)kDeepStack - 1u) * deep_stride];
        }
        top = e;
        sp -= kRow;
    }

    // Starts a ray on the lanes of `lanes` (wave-uniform; called by the whole wave). The ray arguments need values on those lanes only.
    IG_DEV void begin(mask_t lanes, const DevScene& sc, Stack& st, int tid, f3 org, f3 dir, float tmin_, float tmax_, uint32_t flags)
    {
        if (in(lanes)) {
            scene_ray = make_ray_terms(org, dir);
            inv       = scene_ray.inv_dir;
            io        = scene_ray.inv_org;
            tmin      = tmin_;
            tmax      = tmax_;
            rflags    = flags;
            hit_u = hit_v = 0;
            hit_prim = hit_ent = -1;
            nodes_off = SPHERES ? sc.sphere_nodes_off : sc.scene_nodes_off;
            // stack.push(root, ray.tmin) on an empty stack: sentinel below, root on top
            sp = tid * (int)sizeof(uint2);
            slot(st, sp) = make_uint2(0u, igm_bits(kFltMax));
            top = make_uint2(1u, igm_bits(tmin_));
        }
        region_end();
        need_cull &= ~lanes;
        level1 &= ~lanes;
        lterm &= ~lanes;
        overflow &= ~lanes;
        ent_last |= lanes;
        // the cull at level entry (mapping_cpu.art:326-347): a root that starts behind tmax is popped, which leaves the sentinel: done
        const mask_t start = (SPHERES ? sc.sphere_node_count : sc.scene_node_count) != 0 ? (lanes_where(tmin_ <= tmax_) & lanes) : 0ull;
        m_tri &= ~lanes;
        m_leaf &= ~lanes;
        if (IG_ROOT_SCALAR && !SPHERES && !QNODE) {
            const mask_t nx = lanes_where(inv.x < 0) & start, ny = lanes_where(inv.y < 0) & start, nz = lanes_where(inv.z < 0) & start;
            // (one visit per octant among the new rays, for up to two or for any number of octants, loses: 9 030 / 8 180 against 9 370 Mrays/s —
            // the arithmetic of a visit is paid per execution, profiles/r04_experiment_ab.txt section 19)
            if (start != 0ull && (nx == 0ull || nx == start) && (ny == 0ull || ny == start) && (nz == 0ull || nz == start)) {
                m_node &= ~lanes;
                root_visit(sc, st, tid, start, nx != 0ull, ny != 0ull, nz != 0ull);
                return;
            }
        }
        m_node = (m_node & ~lanes) | start;
    }
    // init_hit of a later geometry pass: the hit found so far (tmax passed to begin() is its distance)
    IG_DEV void set_initial_hit(mask_t lanes, int ent, int prim, float u, float v)
    {
        if (in(lanes)) {
            hit_ent  = ent;
            hit_prim = prim;
            hit_u    = u;
            hit_v    = v;
        }
        region_end();
        if (ANY_HIT)
            m_node &= ~(lanes_where(prim >= 0) & lanes); // already occluded: nothing left to find
    }

    // Cheap state transitions of the stack-driven lanes `work` (in no mode mask on entry) up to their next heavy action: an
    // inner node on top (m_node), a triangle packet (m_tri), an entity-leaf run (m_leaf), or the end of the ray (no mask).
    // The cull points are exactly the reference's (mapping_cpu.art:326-347): at level entry, after
    // a leaf and after an inner node that pushed nothing.
    //   unwind (any-hit: the shape-level traversal returned early)  -> falls into `ret`
    //   cull   : the top starts behind the current hit              -> pop, stay culling (the tight loop)
    //   ret    : sentinel on top at shape level                     -> pop the saved scene top, accept the local hit, next leaf — or cull on at the scene level
    //   fin    : sentinel on top at scene level                     -> the ray is done
    //   leaf   : leaf on top                                        -> pop, enter its items (or cull on if it starts behind the hit)
    IG_DEV void settle(const DevScene& sc, Stack& st, mask_t work)
    {
        if (ANY_HIT) {
            if (in(work & level1 & lterm))
                sp = lbase, top.x = 0u;
            region_end();
        }
        while (work) {
            prof(11);
            IG_MARK("settle.cull");
            // entries that start behind the current hit: a compare and a pop each
            for (;;) {
                const mask_t cull = lanes_where(top.x != 0u) & lanes_where(!(igm_float(top.y) <= tmax)) & work & need_cull; // (one compare per ballot: a ballot of a conjunction is materialised first)
                if (!cull)
                    break;
                if (in(cull))
                    pop_top(st);
                region_end();
            }
            IG_MARK("settle.classify");
            const mask_t behind   = lanes_where(!(igm_float(top.y) <= tmax)) & work; // (possible only for lanes that were not culling)
            const mask_t sentinel = lanes_where(top.x == 0u) & work;
            const mask_t leaf     = lanes_where((int)top.x < 0) & work;
            const mask_t ret      = sentinel & level1;
            // leaf on top (mapping_cpu.art:379-381): an entry that starts behind the current
            // hit is dropped, its items have no effect in the reference either
            const mask_t enter = leaf & ~behind;
            const mask_t tris = enter & level1, ents = enter & ~level1;
            if (in(leaf)) {
                const int cursor = (int)~top.x;
                pop_top(st);
                tri_cursor = in(tris) ? cursor : tri_cursor;
                ent_cursor = in(ents) ? cursor : ent_cursor;
            }
            region_end();
            IG_MARK("settle.ret");
            // shape BVH done: back to the scene leaf run (mapping_cpu.art:489-508). The local hit is
            // accepted only if its (rounded) distance does not exceed the current one.
            // (the verdict as a mask, and the two outcomes as regions of their own: the hit registers are written in place, not selected)
            const mask_t accept = lanes_where(l_prim != -1) & lanes_where(tmax <= scene_tmax) & ret;
            if (in(ret)) {
                pop_top(st);
                inv       = scene_ray.inv_dir;
                io        = scene_ray.inv_org;
                nodes_off = sc.scene_nodes_off;
            }
            region_end();
            if (in(accept)) {
                hit_u    = l_u;
                hit_v    = l_v;
                hit_prim = l_prim;
                hit_ent  = cur_ent;
            }
            region_end();
            if (in(ret & ~accept))
                tmax = scene_tmax;
            region_end();
            // (an any-hit ray that just accepted its hit is done: it must not be taken for a lane waiting at its next leaf)
            const mask_t on = ANY_HIT ? ret & ~accept : ret;
            m_node |= work & ~sentinel & ~leaf;
            m_tri |= tris;
            m_leaf |= ents | (on & ~ent_last); // a leaf to enter; on with the leaf run after a shape
            level1 &= ~ret;
            lterm &= ~ret;
            // the lanes stop culling here, except: a leaf that was dropped; a shape whose run is over (on at the scene level)
            const mask_t next = (leaf & behind) | (on & ent_last);
            need_cull         = (need_cull & ~work) | next;
            work              = next;
            IG_MARK("settle.end");
        }
    }

    // The three sections below are called by the whole wave.

    // ---- entity leaves of the current run, up to the first one the ray enters (mapping_cpu.art:481-515); the lanes of m_leaf
    IG_DEV void leaf_section(const DevScene& sc, Stack& st)
    {
        const RayT& gray  = scene_ray;
        const mask_t here = m_leaf;
        if (STATS) {
            if (in(here))
                count_section(0);
            region_end();
        }
        prof(3);
        IG_MARK("leaf.scan");
        mask_t scanning  = here;
        mask_t to_settle = 0, to_tri = 0;
        // (the outer loop repeats only when a lane's one-leaf shape was missed inside its entity box and the run has leaves left,
        // or, SPHERES, after a sphere test)
        do {
            // leaves whose box (or visibility mask) rejects the ray cost only this short loop
            mask_t enter  = 0;
            int enter_at  = 0;
            int entity_id = 0;
            // rows 2 - 7 of the leaf the scan looks at first come with its scan rows: the common case (that leaf is entered) then costs one
            // round trip instead of three (+1 %, any hit -3 %; profiles/r04_experiment_ab.txt)
            float4 early[6];
            int early_at = -1;
#pragma unroll
            for (int k = 0; k < 6; ++k)
                early[k] = make_float4(any_f32(), any_f32(), any_f32(), any_f32()); // (read only where early_at says they were loaded)
            do {
                prof(4);
                bool inside = false, last = false;
                if (in(scanning)) {
                    // rows 0 and 1 of the records, packed: four leaves per 128-byte line. kScanLeaves: the rows of the next leaves come
                    // with the same round trip (a scan is a chain of dependent loads); they are looked at only if this one rejects the
                    // ray and the run goes on. The records behind the last leaf of the table are padding.
                    const int at     = ent_cursor;
                    const void* ls      = SPHERES ? sc.sphere_leaf_scan : sc.leaf_scan;
                    const uint32_t lsat = (uint32_t)at * 32u;
                    float4 lr[kScanLeaves][2];
#pragma unroll
                    for (int k = 0; k < kScanLeaves; ++k)
                        lr[k][0] = ld16(ls, lsat, 2 * k), lr[k][1] = ld16(ls, lsat, 2 * k + 1);
                    if (!SPHERES) {
#pragma unroll
                        for (int k = 0; k < 6; ++k)
                            early[k] = ld16(sc.leaves, (uint32_t)at * (uint32_t)(kDevLeafRows * 16), 2 + k);
                        early_at = at;
                    }
#pragma unroll
                    for (int k = 0; k < kScanLeaves; ++k) {
                        if (k == 0 || (!inside & !last)) {
                            const float4 r0 = lr[k][0], r1 = lr[k][1];
                            ent_cursor += 1;
                            const int id          = (int)igm_bits(r0.w);
                            const uint32_t lflags = igm_bits(r1.w);
                            last                  = id < 0;
                            if (STATS)
                                st_leaves += 1u;
                            // check_ray_visibility (traversal/ray.art:51)
                            const bool visible = (rflags & IG_RAY_FLAG_TYPE_MASK) == ((rflags & lflags) & IG_RAY_FLAG_TYPE_MASK);
                            float entry, exit;
                            slab_test(gray, tmin, tmax, r0.x, r1.x, r0.y, r1.y, r0.z, r1.z, entry, exit);
                            inside = visible & (entry <= exit) & (exit >= 0) & (entry <= tmax);
                            if (inside)
                                enter_at = at + k, entity_id = id;
                        }
                    }
                }
                region_end();
                const mask_t ins = lanes_where(inside) & scanning, lst = lanes_where(last) & scanning;
                ent_last = (ent_last & ~scanning) | lst;
                enter |= ins;
                to_settle |= lst & ~ins; // the run is over and nothing was entered
                scanning &= ~(ins | lst);
            } while (scanning);
            {
                prof(5);
                IG_MARK("leaf.enter");
                bool missed = false, single = false, out = false, ok = false;
                if (in(enter)) {
                    const void* lf      = SPHERES ? sc.sphere_leaves : sc.leaves;
                    const uint32_t lfat = (uint32_t)enter_at * (uint32_t)(kDevLeafRows * 16);
                    // (the early rows where they are the entered leaf's, loaded over otherwise: no copies either way)
                    float4 l2 = early[0], l3 = early[1], l4 = early[2], l5 = early[3], l6 = early[4], l7 = early[5];
                    const bool have_early = !SPHERES && enter_at == early_at;
                    if (!have_early)
                        l2 = ld16(lf, lfat, 2), l3 = ld16(lf, lfat, 3), l4 = ld16(lf, lfat, 4), l5 = ld16(lf, lfat, 5);
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
                        ok = !((S < 0) | (M2 > R2)) & (th >= tmin) & (th <= tmax);
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
                        }
                    } else {
                        // transform_ray (traversal/ray.art:56-59): direction not normalised, t stays global.
                        // (written straight into the lane's shape-space ray: at the scene level nothing reads lorg / ldir, and the slab-test
                        // terms are put back if the shape turns out to be missed)
                        lorg = xform_point(m, gray.org);
                        ldir = xform_dir(m, gray.dir);
                        // A direction the matrix hands back bit for bit (an instance that is only translated: every entity of
                        // diamond_scene) has the reciprocals the scene-space ray already has: the three IEEE divisions are run only
                        // by the lanes that need them.
                        const bool same_dir = (igm_bits(ldir.x) == igm_bits(gray.dir.x)) & (igm_bits(ldir.y) == igm_bits(gray.dir.y)) & (igm_bits(ldir.z) == igm_bits(gray.dir.z));
                        if (!same_dir)
                            inv = f3{ safe_rcp(ldir.x), safe_rcp(ldir.y), safe_rcp(ldir.z) };
                        io = -(lorg * inv);
                        // A shape whose BVH is ONE node with ONE triangle leaf (a wall, a light quad; marked in bit 0 of row 5 of its leaf record by
                        // igd_assign_scene) skips the inner-node section: its root visit is the slab test of that one child, done here
                        // with the operations of the inner-node section. Hit: the state the root visit and the pop of the leaf
                        // would have left (saved scene top on the stack, sentinel on top, in the triangle leaf). Miss: the state
                        // `ret` would have restored, i.e. as if the entity's box had rejected the ray, and the run is scanned on.
                        single = (ext.x & 1u) != 0;
                        if (single) {
                            float4 blo = l6, bhi = l7;
                            if (!have_early)
                                blo = ld16(lf, lfat, 6), bhi = ld16(lf, lfat, 7);
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
                            out        = push_entry(st, (int)top.x, igm_float(top.y));
                            lbase      = sp;
                            scene_tmax = tmax; // invalid_hit(local_ray.tmax): the local distance starts from the scene level's
                            l_prim     = -1;
                            nodes_off  = ext.x & ~1u;
                            tri_off    = ext.y;
                            if (single) {
                                top        = make_uint2(0u, igm_bits(kFltMax));
                                tri_cursor = ~(int)igm_bits(l5.z);
                            } else {
                                out |= push_entry(st, 0, kFltMax);
                                top = make_uint2(1u, igm_bits(tmin));
                            }
                        } else {
                            inv = gray.inv_dir;
                            io  = gray.inv_org;
                        }
                    }
                }
                region_end();
                if (SPHERES) {
                    // a leaf run continues after a sphere test (the shape level of the triangle pass comes back through settle()'s `ret`)
                    const mask_t on = ANY_HIT ? enter & ~lanes_where(ok) : enter; // (an any-hit ray that found its hit is done)
                    scanning        = on & ~ent_last;
                    to_settle |= on & ent_last;
                } else {
                    const mask_t ms = lanes_where(missed) & enter, sg = lanes_where(single) & enter, o = lanes_where(out) & enter;
                    const mask_t go = enter & ~ms;
                    level1 |= go;
                    lterm &= ~go;
                    overflow |= o;
                    to_tri |= go & sg & ~o;
                    to_settle |= go & ~sg & ~o; // at the root of the entered shape: the cull at level entry
                    // the one-leaf shape was missed: on with the run, if it has leaves left
                    scanning = ms & ~ent_last;
                    to_settle |= ms & ent_last;
                }
            }
        } while (scanning);
        m_leaf &= ~here;
        m_tri |= to_tri;
        need_cull |= to_settle;
        IG_MARK("leaf.settle");
        settle(sc, st, to_settle);
        IG_MARK("leaf.end");
    }

    // (per lane, inside a region) the children of one inner node: slab tests and nearest-first insertion against the cached top
    // (mapping_cpu.art:350-377). rows(h, ...) fetches the near / far plane rows of half h (children 4 h .. 4 h + 3).
    template <bool SECTION = true, class Rows>
    IG_DEV void test_children(Stack& st, const int4 c4lo, const int4 c4hi, bool& out, Rows&& rows)
    {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int4 c4 = h ? c4hi : c4lo;
            // two halves of four children keep the live register set small; children are packed from slot 0 (the first zero ends the
            // list, mapping_cpu.art:357)
            if (h == 1 && c4.x == 0)
                break;
            if (h == 1 && SECTION) { // (the profile builds' event and the mark of the inner-node section's second half)
                prof(7);
                IG_MARK("node.half1");
            }
            float4 nx, fx, ny, fy, nz, fz;
            rows(h, nx, fx, ny, fy, nz, fz);
            const float nb[3][4] = { { nx.x, nx.y, nx.z, nx.w }, { ny.x, ny.y, ny.z, ny.w }, { nz.x, nz.y, nz.z, nz.w } };
            const float fb[3][4] = { { fx.x, fx.y, fx.z, fx.w }, { fy.x, fy.y, fy.z, fy.w }, { fz.x, fz.y, fz.z, fz.w } };
            const int ch[4] = { c4.x, c4.y, c4.z, c4.w };
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                // (v_pk_fma_f32 for two children at once measured no faster: it issues like two fmas, and half of the scalar
                // operands end up in register pairs built with v_mov; profiles/r04_experiment_ab.txt)
                const float entry = igm_max(igm_max(igm_fma(inv.x, nb[0][i], io.x), igm_fma(inv.y, nb[1][i], io.y)), igm_max(igm_fma(inv.z, nb[2][i], io.z), tmin));
                const float exit  = igm_min(igm_min(igm_fma(inv.x, fb[0][i], io.x), igm_fma(inv.y, fb[1][i], io.y)), igm_min(igm_fma(inv.z, fb[2][i], io.z), tmax));
                const bool hit    = (ch[i] != 0) & !(exit < entry);
                if (hit) {
                    // push (becomes the top) if nearer than the current top, else push_after
                    const bool front = ANY_HIT || (igm_float(top.y) > entry);
                    const int pn     = front ? (int)top.x : ch[i];
                    const float pt   = front ? igm_float(top.y) : entry;
                    if (DEEP) {
                        out |= push_entry(st, pn, pt);
                    } else {
                        // (how far the node got, and whether that was too far, is read off the address once after the eight children)
                        sp += kRow;
                        if (sp < kLdsEnd)
                            slot(st, sp) = make_uint2((uint32_t)pn, igm_bits(pt));
                    }
                    if (front)
                        top = make_uint2((uint32_t)ch[i], igm_bits(entry));
                }
            }
        }
    }

    // The visit of the scene root for the rays begin() just started, all of one direction octant: the node is the same for every
    // lane and so are the rows the sign of the direction picks, hence scalar loads (the root is a seventh to two fifths of the node
    // visits of a ray, and every one of them is 14 requests of the lane to the vector L1). Same operations in the same order as the
    // inner-node section's, then settle(): the lanes leave begin() where their first section execution would have left them.
    IG_DEV void root_visit(const DevScene& sc, Stack& st, int tid, mask_t lanes, bool neg_x, bool neg_y, bool neg_z)
    {
        const uint32_t node_at = sc.scene_nodes_off;
        const uint32_t sx = neg_x ? 32u : 0u, sy = neg_y ? 32u : 0u, sz = neg_z ? 32u : 0u;
        const int4 c4lo = ld16si(sc.geom, node_at + 192u), c4hi = ld16si(sc.geom, node_at + 208u);
        bool pushed = false, out = false;
        if (in(lanes)) {
            // (the root popped: the sentinel is the cached top, the lane's stack is empty)
            top = make_uint2(0u, igm_bits(kFltMax));
            sp  = tid * (int)sizeof(uint2) - kRow;
            if (STATS)
                st_nodes += 1u;
            count_section(1);
            const int sp_before = sp;
            test_children<false>(st, c4lo, c4hi, out, [&](int h, float4& nx, float4& fx, float4& ny, float4& fy, float4& nz, float4& fz) {
                const uint32_t hb = node_at + 16u * (uint32_t)h;
                nx = ld16s(sc.geom, hb + sx), fx = ld16s(sc.geom, hb + 32u - sx);
                ny = ld16s(sc.geom, hb + 64u + sy), fy = ld16s(sc.geom, hb + 96u - sy);
                nz = ld16s(sc.geom, hb + 128u + sz), fz = ld16s(sc.geom, hb + 160u - sz);
            });
            if (!DEEP)
                out = sp >= kLdsEnd;
            pushed = sp != sp_before;
        }
        region_end();
        const mask_t o = lanes_where(out) & lanes, live = lanes & ~o;
        overflow |= o;
        need_cull |= live & ~lanes_where(pushed);
        settle(sc, st, live);
    }

    // ---- one inner node: fetch 256 B, test 8 children (mapping_cpu.art:350-377); the lanes of m_node
    IG_DEV void node_section(const DevScene& sc, Stack& st)
    {
        const mask_t here = m_node;
        prof(6);
        IG_MARK("node.half0");
        bool pushed = false, out = false;
        if (in(here)) {
            const uint32_t node_at = nodes_off + (top.x - 1u) * kNodeBytes; // byte offset of the node inside geom
            pop_top(st);
            if (STATS)
                st_nodes += 1u;
            count_section(1);
            const int sp_before = sp;
            if constexpr (QNODE) {
                // rows of a quantised node: 0 (origin.xyz, the three scale exponents), 1 - 3 x / y / z: (lo[0..3], lo[4..7], hi[0..3], hi[4..7])
                // one byte per plane, 4 - 5 child ids; plane = fma(q, 2^e, origin), bit for bit the float of the Node8 it was packed from
                const float4 hd = ld16(sc.geom, node_at, 0);
                const int4 qx = ld16i(sc.geom, node_at, 1), qy = ld16i(sc.geom, node_at, 2), qz = ld16i(sc.geom, node_at, 3);
                const int4 c4lo = ld16i(sc.geom, node_at, 4), c4hi = ld16i(sc.geom, node_at, 5);
                const uint32_t ex = igm_bits(hd.w);
                const float sx = igm_float((ex & 0xFFu) << 23), sy = igm_float(((ex >> 8) & 0xFFu) << 23), sz = igm_float(((ex >> 16) & 0xFFu) << 23);
                const bool ox = inv.x < 0, oy = inv.y < 0, oz = inv.z < 0;
                test_children(st, c4lo, c4hi, out, [&](int h, float4& nx, float4& fx, float4& ny, float4& fy, float4& nz, float4& fz) {
                    const auto dec = [](uint32_t w, float s, float o) {
                        // (byte k of w as a float: v_cvt_f32_ubyte<k>)
                        return make_float4(igm_fma((float)(w & 0xFFu), s, o), igm_fma((float)((w >> 8) & 0xFFu), s, o),
                                           igm_fma((float)((w >> 16) & 0xFFu), s, o), igm_fma((float)(w >> 24), s, o));
                    };
                    const uint32_t lx = (uint32_t)(h ? qx.y : qx.x), hx = (uint32_t)(h ? qx.w : qx.z);
                    const uint32_t ly = (uint32_t)(h ? qy.y : qy.x), hy = (uint32_t)(h ? qy.w : qy.z);
                    const uint32_t lz = (uint32_t)(h ? qz.y : qz.x), hz = (uint32_t)(h ? qz.w : qz.z);
                    nx = dec(ox ? hx : lx, sx, hd.x), fx = dec(ox ? lx : hx, sx, hd.x);
                    ny = dec(oy ? hy : ly, sy, hd.y), fy = dec(oy ? ly : hy, sy, hd.y);
                    nz = dec(oz ? hz : lz, sz, hd.z), fz = dec(oz ? lz : hz, sz, hd.z);
                });
            } else {
                // The slab test of the reference takes min / max of the two plane distances per axis
                // (intersection.art:38-58); which plane is the near one is decided by the sign of inv_dir alone
                // (fma is monotonic and lo <= hi), so the near / far rows are picked by address instead and six
                // of the eighteen min / max per child disappear. Results are bit-identical for real children
                // (empty slots are masked by child == 0).
                // (rows of a Node8, 16 bytes each: x lo [0, 1], x hi [2, 3], y lo [4, 5], y hi [6, 7], z lo [8, 9], z hi [10, 11], child ids [12, 13])
                const uint32_t sx = inv.x < 0 ? 32u : 0u, sy = inv.y < 0 ? 32u : 0u, sz = inv.z < 0 ? 32u : 0u;
                const uint32_t near_x = node_at + sx, far_x = node_at + 32u - sx;
                const uint32_t near_y = node_at + 64u + sy, far_y = node_at + 96u - sy;
                const uint32_t near_z = node_at + 128u + sz, far_z = node_at + 160u - sz;
                // both halves' child ids with the first batch of loads: the test for the second half does not cost a round trip of its own
                const int4 c4lo = ld16i(sc.geom, node_at, 12), c4hi = ld16i(sc.geom, node_at, 13);
                // (the second half's rows by 32-bit offsets of their own: an address shared with the first half's loads across the branch in
                // between is materialised as a 64-bit pointer per lane)
                // all twelve rows with the child ids: one round trip per visit. (Nearly every node has more than four children — 2.09 of 2.10 visits
                // per 64 rays on the headline, 29.8 of 29.9 on the stand-in — so fetching the second half after the look at its ids bought nothing
                // and cost a second round trip: stand-in +3.8 %, headline +0.5 %, profiles/r04_experiment_ab.txt section 21.)
                const auto load_half = [&](int h, float4& nx, float4& fx, float4& ny, float4& fy, float4& nz, float4& fz) {
                    // (the second half's rows by 32-bit offsets of their own: an address shared with the first half's loads across a branch in
                    // between is materialised as a 64-bit pointer per lane)
                    const uint32_t hb = 16u * (uint32_t)h;
                    nx = ld16(sc.geom, near_x + hb), fx = ld16(sc.geom, far_x + hb);
                    ny = ld16(sc.geom, near_y + hb), fy = ld16(sc.geom, far_y + hb);
                    nz = ld16(sc.geom, near_z + hb), fz = ld16(sc.geom, far_z + hb);
                };
                if constexpr (kOneTrip) {
                    float4 rw[2][6];
#pragma unroll
                    for (int h = 0; h < 2; ++h)
                        load_half(h, rw[h][0], rw[h][1], rw[h][2], rw[h][3], rw[h][4], rw[h][5]);
                    test_children(st, c4lo, c4hi, out, [&](int h, float4& nx, float4& fx, float4& ny, float4& fy, float4& nz, float4& fz) {
                        nx = rw[h][0], fx = rw[h][1], ny = rw[h][2], fy = rw[h][3], nz = rw[h][4], fz = rw[h][5];
                    });
                } else {
                    test_children(st, c4lo, c4hi, out, load_half);
                }
            }
            if (!DEEP)
                out = sp >= kLdsEnd; // out of stack: the ray ends here (see push_entry)
            pushed = sp != sp_before;
        }
        region_end();
        const mask_t o = lanes_where(out) & here, live = here & ~o;
        overflow |= o;
        need_cull |= live & ~lanes_where(pushed); // nothing pushed: cull (mapping_cpu.art:377)
        m_node = 0;
        IG_MARK("node.settle");
        settle(sc, st, live);
        IG_MARK("node.end");
    }

    // ---- the Tri4 packets of a leaf (mapping_cpu.art:379-410); the lanes of m_tri
    IG_DEV void tri_section(const DevScene& sc, Stack& st)
    {
        RayT lr;
        lr.org = lorg, lr.dir = ldir;
        const mask_t here = m_tri;
        mask_t work       = here;
        prof(8);
        do {
            prof(9);
            IG_MARK("tri.half0");
            bool leave = false, found = false;
            if (in(work)) {
                count_section(2);
                const uint32_t tri_at = tri_off + (uint32_t)tri_cursor * 208u; // byte offset of the packet inside geom
                tri_cursor += 1;
                // two triangles of the packet at a time (a 96-byte half of the re-ordered packet) are tested, the second half only when the
                // packet holds more than two triangles
                const int4 pid4  = ld16i(sc.geom, tri_at, 12);
                const int pid[4] = { pid4.x, pid4.y, pid4.z, pid4.w };
                // the whole packet with its ids: one round trip (85 % of the packets visited on the headline and 99 % on the stand-in hold more
                // than two triangles; +0.8 % / +0.9 %, profiles/r04_experiment_ab.txt section 21)
                float4 call[kOneTrip ? 12 : 1];
                if constexpr (kOneTrip) {
#pragma unroll
                    for (int m = 0; m < 12; ++m)
                        call[m] = ld16(sc.geom, tri_at, m);
                }
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    // (valid triangles are packed from slot 0: the first -1 ends the packet, mapping_cpu.art:386; the first half comes
                    // with the ids)
                    if (h == 1 && (pid[2] == -1 || (ANY_HIT && found)))
                        break;
                    if (h == 1) {
                        prof(10);
                        IG_MARK("tri.half1");
                    }
                    // half h of the packet: 96 contiguous bytes, float 2 k + j = row k of triangle 2 h + j (igd_assign_scene re-orders the
                    // reference's Tri4 that way): six 16-byte loads per half
                    float4 c[6];
#pragma unroll
                    for (int m = 0; m < 6; ++m)
                        c[m] = kOneTrip ? call[kOneTrip ? 6 * h + m : 0] : ld16(sc.geom, tri_at, 6 * h + m);
                    float q[12][2];
#pragma unroll
                    for (int m = 0; m < 6; ++m)
                        q[2 * m][0] = c[m].x, q[2 * m][1] = c[m].y, q[2 * m + 1][0] = c[m].z, q[2 * m + 1][1] = c[m].w;
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const int i   = 2 * h + j;
                        const bool on = (pid[i] != -1) & !(ANY_HIT & found);
                        if (STATS)
                            st_tris += on ? 1u : 0u;
                        // (the test up to its verdict runs for every lane of the region, valid triangle or not: skipping it would pay off only
                        // when no lane of the wave has one, and as straight-line code it keeps the loads above whole and in front)
                        TriCandidate cand;
                        const bool ok = tri_test_candidate(lr, tmin, tmax, f3{ q[0][j], q[1][j], q[2][j] }, f3{ q[3][j], q[4][j], q[5][j] },
                                                           f3{ q[6][j], q[7][j], q[8][j] }, f3{ q[9][j], q[10][j], q[11][j] }, cand) & on;
                        if (ok) {
                            tri_test_finish(cand, tmax, l_u, l_v);
                            l_prim = pid[i] & 0x7FFFFFFF;
                            found  = true;
                        }
                    }
                }
                leave = (pid[3] < 0) | (ANY_HIT & found);
            }
            region_end();
            if (ANY_HIT)
                lterm |= lanes_where(found) & work;
            work &= ~lanes_where(leave);
        } while (work);
        m_tri = 0;
        need_cull |= here;
        IG_MARK("tri.settle");
        settle(sc, st, here);
        IG_MARK("tri.end");
    }

    // One pipeline pass: entity leaf -> inner node -> triangle packet. The whole wave calls it; lanes without a ray are in no mask.
    IG_DEV void step(const DevScene& sc, Stack& st, int tid)
    {
        // Postponing: a section runs only when enough lanes of the wave want it (they wait in their mode until
        // then), so the wave does not pay a whole section for a handful of lanes. If no section reaches the
        // quorum the threshold drops to one lane for this pass, which guarantees progress.
        IG_MARK("pass.quorum");
        int quorum = 1;
        if (kPostponeShift > 0) {
            const int n_ent = lanes_in(m_leaf), n_node = lanes_in(m_node), n_tri = lanes_in(m_tri);
            const int most  = n_ent > n_node ? (n_ent > n_tri ? n_ent : n_tri) : (n_node > n_tri ? n_node : n_tri);
            quorum          = ((n_ent + n_node + n_tri) * kPostponeNum) >> kPostponeShift;
            if (quorum < 1 || most < quorum)
                quorum = 1; // no section reaches the quorum: all of them run
        }
        mark(4); // quorum at the top of a pass
        if (lanes_in(m_leaf) >= quorum)
            leaf_section(sc, st);
        mark(1); // entity-leaf section (with its settle)
        if (lanes_in(m_node) >= quorum) {
            node_section(sc, st);
            // The section once more (up to DevScene::node_repeat times) while 24 or more of the wave's rays wait at a node again: on a BVH
            // that lives in HBM a ray descends node after node and a pass costs its fixed part every time (16 M-triangle stand-in +2.4 %);
            // on a cache-resident scene the repeats run for too few lanes (-0.3 %), the host leaves the count at 0 there.
            for (uint32_t k = 0; k < sc.node_repeat && lanes_in(m_node) >= 24; ++k)
                node_section(sc, st);
        }
        mark(2); // inner-node section (with its settle)
        if (!SPHERES) {
            if (lanes_in(m_tri) >= quorum)
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
