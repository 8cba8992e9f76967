// Host BVH construction: binary sweep-SAH build -> N-ary collapse -> Node8 /
// Tri4 / EntityLeaf1 packing.
//
// The reference builds with madmann91/bvh v2 `DefaultBuilder` (un-vendored,
// pinned only as GIT_TAG master in cmake/GetDependencies.cmake:53-58; call
// sites src/runtime/bvh/TriBVHAdapter.h:216-220, SceneBVHAdapter.h:122-126 --
// the `config` object built there is never passed, so library defaults apply:
// sweep-SAH, min_leaf_size 1, max_leaf_size 8, cost_ratio 1). The library is
// not available here, so its published algorithms are restated: the top-down
// sweep-SAH build and the reinsertion optimiser DefaultBuilder runs behind it
// at its default quality (Meister & Bittner 2018; here the moves of a batch are
// applied one after the other in gain order with a conflict set, as there). Tree topology is "parity
// unpinned" all the same (the reference has no test that inspects it, and the
// dependency is pinned to a moving branch).
//
// Since the topology cannot be matched anyway, the default build is tuned for the <8,4> layout (bvh.cpp): triangle
// leaves are at least one full Tri4 packet and the N-ary collapse opens children by surface area until a node has
// eight. IGH_BVH_REFERENCE=1 selects the reference-like parameters (min_leaf_size 1, breadth-first collapse of at
// most four openings) instead. Results (hits, radiance) do not depend on the topology; work counters do.
//
// The collapse and the node/leaf writers follow the reference's own code:
//   convert_to_narity   src/runtime/bvh/NArityBvh.h:93-155
//   write_node          src/runtime/bvh/BvhNAdapter.h:37-93
//   write_leaf (Tri4)   src/runtime/bvh/TriBVHAdapter.h:94-151
//   write_leaf (entity) src/runtime/bvh/SceneBVHAdapter.h:68-100
#pragma once

#include "hostmath.h"
#include "ig_tables.h"
#include "mesh.h"

#include <cstdint>
#include <vector>

namespace igh {

struct Bvh2Node {
    float bounds[6]; // min_x, max_x, min_y, max_y, min_z, max_z (libbvh ordering)
    uint32_t first;  // first child (inner) or first primitive (leaf)
    uint32_t prim_count; // 0 for inner nodes
    bool isLeaf() const { return prim_count != 0; }
};

struct Bvh2 {
    std::vector<Bvh2Node> nodes; // children of an inner node are adjacent: first, first + 1
    std::vector<size_t> prim_ids;
};

Bvh2 build_bvh2(const std::vector<BBox>& bboxes, const std::vector<V3>& centers, size_t max_leaf_size = 8, size_t min_leaf_size = 1);
// The reinsertion pass DefaultBuilder runs after the sweep at quality High (bvh.cpp); in place.
void optimize_bvh2(Bvh2& bvh);
// Sum over the nodes of area x (1 | primitives), relative to the root's area
float bvh2_sah_cost(const Bvh2& bvh);

// Snaps every child box of the nodes outward to the node's own 8-bit grid, in place (bvh.cpp).
void quantise_node8(ig_node8* nodes, size_t count);

// Diagnostics of the collapse plan on a tree the reinsertion pass has re-linked (bvh.cpp; tests/test_bvh_builder.py)
void collapse_plan_check(const std::vector<BBox>& boxes, float ratio, int iterations, double out[5]);

// Triangle BVH of a mesh in the reference's <8,4> layout.
void build_tri_bvh8(const TriMesh& mesh, std::vector<ig_node8>& nodes, std::vector<ig_tri4>& tris);

struct EntityObject {
    BBox bbox;
    int32_t entity_id, shape_id, material_id, user1, user2;
    float local[12]; // column-major 3x4 to-local
    uint32_t flags;
};

void build_scene_bvh8(const std::vector<EntityObject>& objs, std::vector<ig_node8>& nodes, std::vector<ig_entity_leaf1>& leaves);

} // namespace igh
