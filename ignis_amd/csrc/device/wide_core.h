// wide_core.h — two-level BVH8 traversal of ONE ray by a whole wave (gfx950).
//
// The late passes of k_tail follow a handful of paths per wave, and what a pass costs is the serial chain of its longest
// path: a bounce of a lone lane through Traverser<> is ~17 passes of the section machine at ~500 instructions each, issued
// by a wave that has its SIMD to itself (profiles/r04_tail_chain.txt). Here the wave's lanes share one ray instead:
//   inner node : lane c tests child c (seven 4-byte loads, one slab test), then the hit children are inserted one after the
//                other in slot order on wave-uniform state — the reference's order (mapping_cpu.art:350-377)
//   Tri4 packet: lane t runs the test of triangle t up to its verdict; the candidates are accepted in slot order against
//                the distance the earlier ones left (mapping_cpu.art:379-410)
//   entity run : lane k looks at leaf k of the run (the builder keeps runs at <= 2 leaves by default, IGH_SCENE_MAX_LEAF / IGH_BVH_REFERENCE allow 8: the loop takes any length), the first one the ray enters is entered
// Everything else — the stack (a linear array in the wave's LDS), the cull points, the level switch — is the per-lane
// machine's logic (traverse_core.h settle()) on wave-uniform values, i.e. scalar branches. Same visit order, same
// arithmetic per test, hence the same hit and the same node / triangle / leaf counts as Traverser<false, ...> (the GPU
// suite runs every feature test through both, IGD_TAIL_WIDE=64 / 0).
// Closest hit on the triangle geometry only (the sphere pass and the shadow rays stay with Traverser<>).
#pragma once

#include "traverse_core.h"

namespace igdev {

IG_DEV int wide_bcast(int v, int l) { return __builtin_amdgcn_readlane(v, l); }
IG_DEV float wide_bcast(float v, int l) { return igm_float((uint32_t)__builtin_amdgcn_readlane((int)igm_bits(v), l)); }
IG_DEV f3 wide_bcast(f3 v, int l) { return f3{ wide_bcast(v.x, l), wide_bcast(v.y, l), wide_bcast(v.z, l) }; }
IG_DEV float wide_ldf(const void* base, uint32_t off) { return *reinterpret_cast<const float*>(static_cast<const uint8_t*>(base) + off); }
IG_DEV int wide_ldi(const void* base, uint32_t off) { return *reinterpret_cast<const int*>(static_cast<const uint8_t*>(base) + off); }
IG_DEV bool wide_any(bool c) { return lanes_where(c) != 0ull; } // a wave-uniform condition as a scalar branch

template <bool STATS>
struct WideTraverser {
    // the wave's stack: the StackOf<64> of its workgroup as a linear array; the last 64 words take the stores of the lanes that
    // have nothing to store (one store instruction for the wave, no change of EXEC)
    static constexpr int kEntries = kLdsStack * 64 - 64;
    enum { kDone = 0, kNode = 1, kTri = 2, kLeaf = 3 };

    // the result (wave-uniform)
    float tmax, hit_u, hit_v;
    int hit_prim, hit_ent;
    bool overflow;
    uint32_t st_nodes, st_tris, st_leaves;

    IG_DEV void run(const DevScene& sc, StackOf<64>& st, f3 org, f3 dir, float tmin, float tmax_in, uint32_t rflags)
    {
        uint2* const stk    = &st.e[0][0];
        const uint32_t lane = __lane_id();
        const RayT gray     = make_ray_terms(org, dir);
        f3 inv = gray.inv_dir, io = gray.inv_org, lorg = org, ldir = dir;
        tmax  = tmax_in;
        hit_u = hit_v = 0;
        hit_prim = hit_ent = -1;
        overflow = false;
        st_nodes = st_tris = st_leaves = 0;
        float scene_tmax = 0, l_u = 0, l_v = 0;
        int l_prim = -1, cur_ent = -1;
        uint32_t nodes_off = sc.scene_nodes_off, tri_off = 0;
        int ent_cursor = 0, tri_cursor = 0;
        bool level1 = false, ent_last = true, need_cull = false;

        const auto put = [&](int at, uint2 e) { stk[lane == 0 ? at : kEntries + (int)lane] = e; };
        // stack.push(root, ray.tmin) on an empty stack: sentinel below, root on top
        put(0, make_uint2(0u, igm_bits(kFltMax)));
        int sp    = 0; // index of the entry below the cached top
        uint2 top = make_uint2(1u, igm_bits(tmin));
        const auto pop = [&]() {
            top = stk[sp];
            sp -= 1;
        };
        const auto push = [&](uint32_t n, uint32_t t) {
            sp += 1;
            if (sp < kEntries)
                put(sp, make_uint2(n, t));
            else
                overflow = true;
        };

        IG_MARK("wide.begin");
        int mode = (sc.scene_node_count != 0 && wide_any(tmin <= tmax)) ? kNode : kDone;
        while (mode != kDone) {
            if (mode == kNode) {
                IG_MARK("wide.node");
                // ---- one inner node (mapping_cpu.art:350-377): lane c & 7 tests child c
                const uint32_t at = nodes_off + (top.x - 1u) * 256u + (lane & 7u) * 4u;
                // (rows of a Node8, 16 bytes each: x lo [0, 1], x hi [2, 3], y lo [4, 5], y hi [6, 7], z lo [8, 9], z hi [10, 11], child ids [12, 13];
                // near / far plane by the sign of the inverse direction, as in the per-lane machine)
                const uint32_t sx = inv.x < 0 ? 32u : 0u, sy = inv.y < 0 ? 32u : 0u, sz = inv.z < 0 ? 32u : 0u;
                const float nx = wide_ldf(sc.geom, at + sx), fx = wide_ldf(sc.geom, at + 32u - sx);
                const float ny = wide_ldf(sc.geom, at + 64u + sy), fy = wide_ldf(sc.geom, at + 96u - sy);
                const float nz = wide_ldf(sc.geom, at + 128u + sz), fz = wide_ldf(sc.geom, at + 160u - sz);
                const int id   = wide_ldi(sc.geom, at + 192u);
                pop();
                if (STATS)
                    st_nodes += 1u;
                const float entry = igm_max(igm_max(igm_fma(inv.x, nx, io.x), igm_fma(inv.y, ny, io.y)), igm_max(igm_fma(inv.z, nz, io.z), tmin));
                const float exit  = igm_min(igm_min(igm_fma(inv.x, fx, io.x), igm_fma(inv.y, fy, io.y)), igm_min(igm_fma(inv.z, fz, io.z), tmax));
                uint32_t hits     = (uint32_t)(lanes_where((id != 0) & !(exit < entry)) & 0xFFull);
                need_cull         = hits == 0u; // nothing pushed: cull (mapping_cpu.art:377)
                while (hits) {
                    const int c = __builtin_ctz(hits);
                    hits &= hits - 1u;
                    const float e = wide_bcast(entry, c);
                    const int ch  = wide_bcast(id, c);
                    // push (becomes the top) if nearer than the current top, else push_after
                    if (wide_any(igm_float(top.y) > e)) {
                        push(top.x, top.y);
                        top = make_uint2((uint32_t)ch, igm_bits(e));
                    } else {
                        push((uint32_t)ch, igm_bits(e));
                    }
                }
            } else if (mode == kTri) {
                // ---- the Tri4 packets of a leaf (mapping_cpu.art:379-410): lane t & 3 tests triangle t of the packet
                IG_MARK("wide.tri");
                RayT lr;
                lr.org = lorg, lr.dir = ldir;
                const uint32_t t4   = lane & 3u;
                const uint32_t toff = 96u * (t4 >> 1) + 4u * (t4 & 1u); // half t / 2 of the re-ordered packet, float 2 k + (t & 1) = row k
                bool last;
                do {
                    const uint32_t tri_at = tri_off + (uint32_t)tri_cursor * 208u;
                    tri_cursor += 1;
                    float q[12];
#pragma unroll
                    for (int k = 0; k < 12; ++k)
                        q[k] = wide_ldf(sc.geom, tri_at + toff + 8u * (uint32_t)k);
                    const int pid = wide_ldi(sc.geom, tri_at + 192u + 4u * t4);
                    const bool on = pid != -1;
                    TriCandidate cand;
                    const bool ok = tri_test_candidate(lr, tmin, tmax, f3{ q[0], q[1], q[2] }, f3{ q[3], q[4], q[5] }, f3{ q[6], q[7], q[8] }, f3{ q[9], q[10], q[11] }, cand) & on;
                    uint32_t cands = (uint32_t)(lanes_where(ok) & 0xFull);
                    if (STATS)
                        st_tris += (uint32_t)__builtin_popcount((uint32_t)(lanes_where(on) & 0xFull));
                    last       = wide_bcast(pid, 3) < 0;
                    bool first = true; // (its verdict above was against the distance it meets in slot order too)
                    while (cands) {
                        const int t = __builtin_ctz(cands);
                        cands &= cands - 1u;
                        const TriCandidate c{ wide_bcast(cand.t, t), wide_bcast(cand.u, t), wide_bcast(cand.v, t), wide_bcast(cand.adet, t) };
                        if (first || wide_any(c.t <= c.adet * tmax)) {
                            tri_test_finish(c, tmax, l_u, l_v);
                            l_prim = wide_bcast(pid, t) & 0x7FFFFFFF;
                        }
                        first = false;
                    }
                } while (!last);
                need_cull = true;
            } else {
                // ---- entity leaves of the current run, up to the first one the ray enters (mapping_cpu.art:481-515): lane k & 1 looks at leaf k
                IG_MARK("wide.leaf");
                for (;;) {
                    const int at       = ent_cursor;
                    const uint32_t k2  = lane & 1u;
                    const uint32_t lat = ((uint32_t)at + k2) * 32u;
                    const float4 r0 = ld16(sc.leaf_scan, lat, 0), r1 = ld16(sc.leaf_scan, lat, 1);
                    // rows 2 - 5 of the first leaf come with the scan rows (it is the one entered, as a rule)
                    float4 l2 = ld16(sc.leaves, (uint32_t)at * (uint32_t)(kDevLeafRows * 16), 2), l3 = ld16(sc.leaves, (uint32_t)at * (uint32_t)(kDevLeafRows * 16), 3);
                    float4 l4 = ld16(sc.leaves, (uint32_t)at * (uint32_t)(kDevLeafRows * 16), 4), l5 = ld16(sc.leaves, (uint32_t)at * (uint32_t)(kDevLeafRows * 16), 5);
                    const int id          = (int)igm_bits(r0.w);
                    const uint32_t lflags = igm_bits(r1.w);
                    // check_ray_visibility (traversal/ray.art:51)
                    const bool visible = (rflags & IG_RAY_FLAG_TYPE_MASK) == ((rflags & lflags) & IG_RAY_FLAG_TYPE_MASK);
                    float entry, exit;
                    slab_test(gray, tmin, tmax, r0.x, r1.x, r0.y, r1.y, r0.z, r1.z, entry, exit);
                    const bool inside = visible & (entry <= exit) & (exit >= 0) & (entry <= tmax);
                    const uint32_t bi = (uint32_t)(lanes_where(inside) & 3ull), bl = (uint32_t)(lanes_where(id < 0) & 3ull);
                    // the second leaf is looked at only if the first one rejects the ray and the run goes on
                    const int looked  = ((bi | bl) & 1u) ? 1 : 2;
                    const int enter_k = (bi & 1u) ? 0 : (!(bl & 1u) && (bi & 2u)) ? 1 : -1;
                    ent_cursor += looked;
                    if (STATS)
                        st_leaves += (uint32_t)looked;
                    ent_last = ((bl >> (looked - 1)) & 1u) != 0u;
                    if (enter_k < 0) {
                        if (ent_last)
                            break; // the run is over and nothing was entered
                        continue;
                    }
                    const int entity_id = wide_bcast(id, enter_k);
                    if (enter_k != 0) {
                        const uint32_t lfat = (uint32_t)(at + 1) * (uint32_t)(kDevLeafRows * 16);
                        l2 = ld16(sc.leaves, lfat, 2), l3 = ld16(sc.leaves, lfat, 3), l4 = ld16(sc.leaves, lfat, 4), l5 = ld16(sc.leaves, lfat, 5);
                    }
                    m34 m;
                    m.c0 = f3{ l2.x, l2.y, l2.z };
                    m.c1 = f3{ l2.w, l3.x, l3.y };
                    m.c2 = f3{ l3.z, l3.w, l4.x };
                    m.c3 = f3{ l4.y, l4.z, l4.w };
                    // transform_ray (traversal/ray.art:56-59): direction not normalised, t stays global
                    lorg = xform_point(m, gray.org);
                    ldir = xform_dir(m, gray.dir);
                    const bool same_dir = (igm_bits(ldir.x) == igm_bits(gray.dir.x)) & (igm_bits(ldir.y) == igm_bits(gray.dir.y)) & (igm_bits(ldir.z) == igm_bits(gray.dir.z));
                    if (wide_any(!same_dir))
                        inv = f3{ safe_rcp(ldir.x), safe_rcp(ldir.y), safe_rcp(ldir.z) };
                    io = -(lorg * inv);
                    // save the scene-level top, then a fresh stack: sentinel + shape root (a one-leaf shape takes the general way here:
                    // its root visit leaves the state and the counts the per-lane machine's shortcut sets up)
                    cur_ent = entity_id & 0x7FFFFFFF;
                    push(top.x, top.y);
                    scene_tmax = tmax; // invalid_hit(local_ray.tmax): the local distance starts from the scene level's
                    l_prim     = -1;
                    nodes_off  = igm_bits(l5.x) & ~1u;
                    tri_off    = igm_bits(l5.y);
                    push(0u, igm_bits(kFltMax));
                    top    = make_uint2(1u, igm_bits(tmin));
                    level1 = true;
                    break;
                }
                need_cull = true; // after a run, and the cull at level entry
            }
            IG_MARK("wide.settle");
            if (overflow)
                break;

            // ---- the stack transitions up to the next heavy action (traverse_core.h settle(), mapping_cpu.art:326-347)
            for (;;) {
                if (need_cull) {
                    // entries that start behind the current hit
                    while (top.x != 0u && wide_any(!(igm_float(top.y) <= tmax)))
                        pop();
                }
                if (top.x != 0u && (int)top.x > 0) {
                    mode = kNode;
                    break;
                }
                if ((int)top.x < 0) {
                    // leaf on top (mapping_cpu.art:379-381): an entry that starts behind the current hit is dropped
                    const bool behind = wide_any(!(igm_float(top.y) <= tmax));
                    const int cursor  = (int)~top.x;
                    pop();
                    if (!behind) {
                        if (level1)
                            tri_cursor = cursor, mode = kTri;
                        else
                            ent_cursor = cursor, mode = kLeaf;
                        break;
                    }
                    need_cull = true;
                    continue;
                }
                // the sentinel
                if (!level1) {
                    mode = kDone;
                    break;
                }
                // shape BVH done: back to the scene leaf run (mapping_cpu.art:489-508). The local hit is accepted only if its
                // (rounded) distance does not exceed the current one.
                pop();
                if (l_prim != -1 && wide_any(tmax <= scene_tmax)) {
                    hit_u = l_u, hit_v = l_v;
                    hit_prim = l_prim;
                    hit_ent  = cur_ent;
                } else {
                    tmax = scene_tmax;
                }
                inv       = gray.inv_dir;
                io        = gray.inv_org;
                nodes_off = sc.scene_nodes_off;
                level1    = false;
                if (!ent_last) {
                    mode = kLeaf; // on with the leaf run
                    break;
                }
                need_cull = true;
            }
        }
        IG_MARK("wide.end");
    }
};

} // namespace igdev
