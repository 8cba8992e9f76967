"""Shape of the BVH8 / Tri4 tables the loader builds for a scene: children per node, triangles and packets per leaf, per shape and in all.
usage: python tools/bvh_stats.py scene.json   (environment knobs of the builder apply: IGH_MIN_LEAF, IGH_MAX_LEAF, IGH_SCENE_MAX_LEAF, IGH_BVH_REFERENCE)"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from ignis_amd.tables import LoadedScene  # noqa: E402

sc = LoadedScene.from_file(sys.argv[1], 64, 64)
s = sc.scene
blob = np.frombuffer(C.string_at(s.primbvh, s.primbvh_size), np.uint8)
tot_nodes = tot_children = tot_leaves = tot_packets = tot_tris = 0
hist = np.zeros(9, np.int64)
# where each shape's {header, Node8[], Tri4[]} starts: the scene leaves' user words (offset in floats, TriMeshProvider.cpp:598)
offsets = sorted({((int(s.scene_leaves[i].user[1]) & 0xFFFFFFFF) << 32 | (int(s.scene_leaves[i].user[0]) & 0xFFFFFFFF)) * 4 for i in range(s.scene_leaf_count)})
for off in offsets:
    nodes, packets = (int(x) for x in np.frombuffer(blob[off:off + 8].tobytes(), np.int32))
    nd = np.frombuffer(blob[off + 16:off + 16 + nodes * 256].tobytes(), np.int32).reshape(nodes, 64)
    child = nd[:, 48:56]
    tr = np.frombuffer(blob[off + 16 + nodes * 256:off + 16 + nodes * 256 + packets * 208].tobytes(), np.int32).reshape(packets, 52)
    pid = tr[:, 48:52]
    n_child = (child != 0).sum(1)
    hist += np.bincount(n_child, minlength=9)
    leaves = int((child < 0).sum())
    tris = int((pid != -1).sum())
    tot_nodes += nodes; tot_children += int(n_child.sum()); tot_leaves += leaves; tot_packets += packets; tot_tris += tris
    if len(offsets) <= 16:
        print("  shape at %8d: %5d nodes, %6d triangles in %5d leaves / %5d packets, children per node %.2f" % (off, nodes, tris, leaves, packets, n_child.mean()))
print("shapes' BVHs: %d nodes, %.2f children per node, %d leaves, %.2f triangles and %.2f packets per leaf, %.2f triangles per packet" % (
    tot_nodes, tot_children / max(tot_nodes, 1), tot_leaves, tot_tris / max(tot_leaves, 1), tot_packets / max(tot_leaves, 1), tot_tris / max(tot_packets, 1)))
print("children per node histogram (0..8):", hist.tolist())
print("scene BVH: %d nodes, %d leaves (entities)" % (s.scene_node_count, s.scene_leaf_count))
