// group_core.h — two-level BVH8 traversal of EIGHT rays by one wave, eight lanes per ray (gfx950).
//
// Between the whole wave on one ray (wide_core.h: a wave that follows <= 4 paths) and one ray per lane (traverse_core.h: full waves)
// sit the middle passes of k_tail, whose waves follow 5 - 16 paths: the per-lane section machine then issues its ~500 instructions per
// pass for a handful of lanes, and a bounce costs 70 - 100 us whatever their number (profiles/r04_tail_chain.txt). Here a group of
// eight lanes shares a ray, the natural width of the tables:
//   inner node : lane c of the group tests child c of the Node8 (seven 4-byte loads, one slab test)
//   Tri4 packet: lanes 0 - 3 test triangle t of the packet (lanes 4 - 7 repeat them: no divergence inside a group)
//   entity run : lanes k & 1 look at leaf k of the run
// and the hit children / candidate triangles are committed one after the other in slot order on GROUP-uniform state — every lane of
// a group holds a copy of its ray's state and updates it with the same operations —, which is the reference's order
// (mapping_cpu.art:350-410) and wide_core.h's: the same hits and the same node / triangle / leaf counts as the other two machines, bit
// for bit (the `tail-wide8` schedule of the GPU suite runs every feature test this way). Control flow is per group: a group in its
// node step and a group in its triangle step take turns inside one pass of the loop (EXEC masks), nobody waits for a quorum.
// The stack of group g is its eighth of the wave's LDS stack array (kLdsStack * 8 entries); a ray that outgrows it is handed to the
// per-lane DEEP machine by the caller, like wide_core.h's.
// Closest hit on the triangle geometry only (the sphere pass and the shadow rays stay with Traverser<>).
#pragma once

#include "traverse_core.h"

namespace igdev {

template <bool STATS>
struct GroupTraverser {
    static constexpr int kGroup   = 8;
    static constexpr int kEntries = kLdsStack * 64 / kGroup; // stack entries of a group
    enum { kDone = 0, kNode = 1, kTri = 2, kLeaf = 3 };

    // the result, the same in the eight lanes of a group
    float tmax, hit_u, hit_v;
    int hit_prim, hit_ent;
    bool overflow;
    uint32_t st_nodes, st_tris, st_leaves;

    // `active`: this lane's group has a ray (org, dir, tmin, tmax_in, rflags: the same values in its eight lanes)
    IG_DEV void run(const DevScene& sc, StackOf<64>& st, bool active, f3 org, f3 dir, float tmin, float tmax_in, uint32_t rflags)
    {
        const uint32_t lane = __lane_id(), c8 = lane & 7u, gfirst = lane & ~7u;
        uint2* const stk    = &st.e[0][0] + (lane >> 3) * (uint32_t)kEntries;
        // a value of lane c of the own group, and the group's eight verdicts as a byte
        const auto gf = [&](float v, uint32_t c) { return __shfl(v, (int)(gfirst | c)); };
        const auto gi = [&](int v, uint32_t c) { return __shfl(v, (int)(gfirst | c)); };
        const auto gmask = [&](bool p) { return (uint32_t)(lanes_where(p) >> gfirst) & 0xFFu; };

        const RayT gray = make_ray_terms(org, dir);
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

        // stack.push(root, ray.tmin) on an empty stack: sentinel below, root on top (every lane of the group stores the same word)
        stk[0]    = make_uint2(0u, igm_bits(kFltMax));
        int sp    = 0; // index of the entry below the cached top
        uint2 top = make_uint2(1u, igm_bits(tmin));
        const auto pop = [&]() {
            top = stk[sp];
            sp -= 1;
        };
        const auto push = [&](uint32_t n, uint32_t t) {
            sp += 1;
            if (sp < kEntries)
                stk[sp] = make_uint2(n, t);
            else
                overflow = true;
        };

        int mode = (active && sc.scene_node_count != 0 && tmin <= tmax) ? kNode : kDone;
        while (__any(mode != kDone)) {
            if (mode == kNode) {
                // ---- one inner node (mapping_cpu.art:350-377): lane c of the group tests child c
                const uint32_t at = nodes_off + (top.x - 1u) * 256u + c8 * 4u;
                // (rows of a Node8, 16 bytes each: x lo [0, 1], x hi [2, 3], y lo [4, 5], y hi [6, 7], z lo [8, 9], z hi [10, 11], child ids [12, 13];
                // near / far plane by the sign of the inverse direction, as in the other machines)
                const uint32_t sx = inv.x < 0 ? 32u : 0u, sy = inv.y < 0 ? 32u : 0u, sz = inv.z < 0 ? 32u : 0u;
                const float nx = *reinterpret_cast<const float*>(sc.geom + at + sx), fx = *reinterpret_cast<const float*>(sc.geom + at + 32u - sx);
                const float ny = *reinterpret_cast<const float*>(sc.geom + at + 64u + sy), fy = *reinterpret_cast<const float*>(sc.geom + at + 96u - sy);
                const float nz = *reinterpret_cast<const float*>(sc.geom + at + 128u + sz), fz = *reinterpret_cast<const float*>(sc.geom + at + 160u - sz);
                const int id   = *reinterpret_cast<const int*>(sc.geom + at + 192u);
                pop();
                if (STATS)
                    st_nodes += 1u;
                const float entry = igm_max(igm_max(igm_fma(inv.x, nx, io.x), igm_fma(inv.y, ny, io.y)), igm_max(igm_fma(inv.z, nz, io.z), tmin));
                const float exit  = igm_min(igm_min(igm_fma(inv.x, fx, io.x), igm_fma(inv.y, fy, io.y)), igm_min(igm_fma(inv.z, fz, io.z), tmax));
                uint32_t hits     = gmask((id != 0) & !(exit < entry));
                need_cull         = hits == 0u; // nothing pushed: cull (mapping_cpu.art:377)
                while (hits) {
                    const uint32_t c = (uint32_t)__builtin_ctz(hits);
                    hits &= hits - 1u;
                    const float e = gf(entry, c);
                    const int ch  = gi(id, c);
                    // push (becomes the top) if nearer than the current top, else push_after
                    if (igm_float(top.y) > e) {
                        push(top.x, top.y);
                        top = make_uint2((uint32_t)ch, igm_bits(e));
                    } else {
                        push((uint32_t)ch, igm_bits(e));
                    }
                }
            } else if (mode == kTri) {
                // ---- the Tri4 packets of a leaf (mapping_cpu.art:379-410): lane t & 3 tests triangle t of the packet
                RayT lr;
                lr.org = lorg, lr.dir = ldir;
                const uint32_t t4   = c8 & 3u;
                const uint32_t toff = 96u * (t4 >> 1) + 4u * (t4 & 1u); // half t / 2 of the re-ordered packet, float 2 k + (t & 1) = row k
                bool last;
                do {
                    const uint32_t tri_at = tri_off + (uint32_t)tri_cursor * 208u;
                    tri_cursor += 1;
                    float q[12];
#pragma unroll
                    for (int k = 0; k < 12; ++k)
                        q[k] = *reinterpret_cast<const float*>(sc.geom + tri_at + toff + 8u * (uint32_t)k);
                    const int pid = *reinterpret_cast<const int*>(sc.geom + tri_at + 192u + 4u * t4);
                    const bool on = pid != -1;
                    TriCandidate cand;
                    const bool ok  = tri_test_candidate(lr, tmin, tmax, f3{ q[0], q[1], q[2] }, f3{ q[3], q[4], q[5] }, f3{ q[6], q[7], q[8] }, f3{ q[9], q[10], q[11] }, cand) & on;
                    uint32_t cands = gmask(ok) & 0xFu;
                    if (STATS)
                        st_tris += (uint32_t)__builtin_popcount(gmask(on) & 0xFu);
                    last       = gi(pid, 3u) < 0;
                    bool first = true; // (its verdict above was against the distance it meets in slot order too)
                    while (cands) {
                        const uint32_t t = (uint32_t)__builtin_ctz(cands);
                        cands &= cands - 1u;
                        const TriCandidate c{ gf(cand.t, t), gf(cand.u, t), gf(cand.v, t), gf(cand.adet, t) };
                        const int cpid = gi(pid, t);
                        if (first || c.t <= c.adet * tmax) {
                            tri_test_finish(c, tmax, l_u, l_v);
                            l_prim = cpid & 0x7FFFFFFF;
                        }
                        first = false;
                    }
                } while (!last);
                need_cull = true;
            } else if (mode == kLeaf) {
                // ---- entity leaves of the current run, up to the first one the ray enters (mapping_cpu.art:481-515): lane k & 1 looks at leaf k
                for (;;) {
                    const int at       = ent_cursor;
                    const uint32_t k2  = c8 & 1u;
                    const uint32_t lat = ((uint32_t)at + k2) * 32u;
                    const float4 r0 = ld16(sc.leaf_scan, lat, 0), r1 = ld16(sc.leaf_scan, lat, 1);
                    const int id          = (int)igm_bits(r0.w);
                    const uint32_t lflags = igm_bits(r1.w);
                    // check_ray_visibility (traversal/ray.art:51)
                    const bool visible = (rflags & IG_RAY_FLAG_TYPE_MASK) == ((rflags & lflags) & IG_RAY_FLAG_TYPE_MASK);
                    float entry, exit;
                    slab_test(gray, tmin, tmax, r0.x, r1.x, r0.y, r1.y, r0.z, r1.z, entry, exit);
                    const bool inside = visible & (entry <= exit) & (exit >= 0) & (entry <= tmax);
                    const uint32_t bi = gmask(inside) & 3u, bl = gmask(id < 0) & 3u;
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
                    const int entity_id = gi(id, (uint32_t)enter_k);
                    const uint32_t lfat = (uint32_t)(at + enter_k) * (uint32_t)(kDevLeafRows * 16);
                    const float4 l2 = ld16(sc.leaves, lfat, 2), l3 = ld16(sc.leaves, lfat, 3), l4 = ld16(sc.leaves, lfat, 4), l5 = ld16(sc.leaves, lfat, 5);
                    m34 m;
                    m.c0 = f3{ l2.x, l2.y, l2.z };
                    m.c1 = f3{ l2.w, l3.x, l3.y };
                    m.c2 = f3{ l3.z, l3.w, l4.x };
                    m.c3 = f3{ l4.y, l4.z, l4.w };
                    // transform_ray (traversal/ray.art:56-59): direction not normalised, t stays global
                    lorg = xform_point(m, gray.org);
                    ldir = xform_dir(m, gray.dir);
                    const bool same_dir = (igm_bits(ldir.x) == igm_bits(gray.dir.x)) & (igm_bits(ldir.y) == igm_bits(gray.dir.y)) & (igm_bits(ldir.z) == igm_bits(gray.dir.z));
                    if (!same_dir)
                        inv = f3{ safe_rcp(ldir.x), safe_rcp(ldir.y), safe_rcp(ldir.z) };
                    io = -(lorg * inv);
                    // save the scene-level top, then a fresh stack: sentinel + shape root
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
            if (mode != kDone) {
                if (overflow) {
                    mode = kDone;
                } else {
                    // ---- the stack transitions up to the next heavy action (traverse_core.h settle(), mapping_cpu.art:326-347)
                    for (;;) {
                        if (need_cull) {
                            // entries that start behind the current hit
                            while (top.x != 0u && !(igm_float(top.y) <= tmax))
                                pop();
                        }
                        if (top.x != 0u && (int)top.x > 0) {
                            mode = kNode;
                            break;
                        }
                        if ((int)top.x < 0) {
                            // leaf on top (mapping_cpu.art:379-381): an entry that starts behind the current hit is dropped
                            const bool behind = !(igm_float(top.y) <= tmax);
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
                        if (l_prim != -1 && tmax <= scene_tmax) {
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
            }
        }
    }
};

} // namespace igdev
