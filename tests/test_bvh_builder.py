"""The host's tree builder (ignis_amd/csrc/host/bvh.cpp): what the tables it writes must satisfy whatever its tuning — every triangle in
exactly one packet slot, 1 .. 8 children per Node8, boxes that contain their subtrees — and what round 3's tuning promises: at most
two entities per scene-BVH leaf, and a collapse whose wide nodes have no more summed area than the greedy one's."""
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import SCENES

ROOT = os.path.dirname(SCENES)

_STATS = r"""
import ctypes as C, json, sys
import numpy as np
sys.path.insert(0, %r)
from ignis_amd.tables import LoadedScene
sc = LoadedScene.from_file(sys.argv[1], 64, 64)
s = sc.scene
blob = np.frombuffer(C.string_at(s.primbvh, s.primbvh_size), np.uint8)
out = []
offsets = sorted({((int(s.scene_leaves[i].user[1]) & 0xFFFFFFFF) << 32 | (int(s.scene_leaves[i].user[0]) & 0xFFFFFFFF)) * 4 for i in range(s.scene_leaf_count)})
for off in offsets:  # where each shape's {header, Node8[], Tri4[]} starts: the scene leaves' user words, in floats
    nodes, packets = (int(x) for x in np.frombuffer(blob[off:off + 8].tobytes(), np.int32))
    nd = np.frombuffer(blob[off + 16:off + 16 + nodes * 256].tobytes(), np.float32).reshape(nodes, 64)
    child = nd[:, 48:56].view(np.int32)
    b = nd[:, :48].reshape(nodes, 6, 8)
    ext = np.maximum(b[:, 1::2, :] - b[:, 0::2, :], 0)
    area = ext[:, 0] * ext[:, 1] + ext[:, 1] * ext[:, 2] + ext[:, 2] * ext[:, 0]
    inner = child > 0
    tr = np.frombuffer(blob[off + 16 + nodes * 256:off + 16 + nodes * 256 + packets * 208].tobytes(), np.int32).reshape(packets, 52)
    pid = tr[:, 48:52]
    ids = (pid[pid != -1] & 0x7FFFFFFF)
    # containment: the box slot c of node n holds for an inner child = the union of that child's own slots
    viol = 0
    for nidx in range(nodes):
        for c in range(8):
            k = int(child[nidx, c])
            if k > 0:
                cb = b[k - 1]
                used = child[k - 1] != 0
                lo = cb[0::2][:, used].min(1); hi = cb[1::2][:, used].max(1)
                viol += int((b[nidx, 0::2, c] > lo).any() or (b[nidx, 1::2, c] < hi).any())
    out.append({"box_violations": viol, "nodes": nodes, "children_min": int((child != 0).sum(1).min()), "children_max": int((child != 0).sum(1).max()),
                "inner_area": float(area[inner].sum()), "leaves": int((child < 0).sum()), "triangles": int(len(ids)),
                "unique": int(len(np.unique(ids))), "max_id": int(ids.max())})
runs, n = [], 0
for i in range(s.scene_leaf_count):
    n += 1
    if s.scene_leaves[i].entity_id < 0:
        runs.append(n); n = 0
print(json.dumps({"shapes": out, "runs": runs, "entities": int(s.entity_count)}))
""" % ROOT


def _stats(scene, **env):
    e = dict(os.environ, **env)
    r = subprocess.run([sys.executable, "-c", _STATS, scene], capture_output=True, text=True, env=e, check=True)
    return json.loads(r.stdout.strip().splitlines()[-1])


@pytest.fixture(scope="module")
def terrain(tmp_path_factory):
    d = tmp_path_factory.mktemp("standin")
    tool = os.path.join(ROOT, "tools", "make_standin_scene.py")
    subprocess.run([sys.executable, tool, str(d), "--triangles", "40000", "--instances", "12", "--seed", "5", "--width", "64", "--height", "64"],
                   check=True, capture_output=True)
    return os.path.join(str(d), "standin.json")


@pytest.mark.parametrize("which", ["diamond", "terrain"])
def test_tables_are_complete_whatever_the_tuning(which, terrain):
    scene = terrain if which == "terrain" else os.path.join(SCENES, "diamond_scene.json")
    for env in ({}, {"IGH_COLLAPSE": "greedy"}, {"IGH_BVH_REFERENCE": "1"}, {"IGH_SCENE_MAX_LEAF": "8", "IGH_MIN_LEAF": "2"}, {"IGH_BVH_REINSERT_RATIO": "0.5"}):
        st = _stats(scene, **env)
        assert sum(st["runs"]) == st["entities"]
        for sh in st["shapes"]:
            assert 1 <= sh["children_min"] and sh["children_max"] <= 8, env
            assert sh["triangles"] == sh["unique"] == sh["max_id"] + 1, env  # every triangle once
            assert sh["box_violations"] == 0, env  # a Node8's child box contains the boxes of that child's own children


def test_scene_leaves_hold_at_most_two_entities(terrain):
    for scene in (terrain, os.path.join(SCENES, "diamond_scene.json")):
        assert max(_stats(scene)["runs"]) <= 2
        assert max(_stats(scene, IGH_SCENE_MAX_LEAF="8")["runs"]) > 2  # (what the builder's plain defaults give)


def test_the_collapse_minimises_the_area_of_the_wide_nodes(terrain):
    best, greedy = _stats(terrain), _stats(terrain, IGH_COLLAPSE="greedy")
    a, b = sum(s["inner_area"] for s in best["shapes"]), sum(s["inner_area"] for s in greedy["shapes"])
    na, nb = sum(s["nodes"] for s in best["shapes"]), sum(s["nodes"] for s in greedy["shapes"])
    assert [s["leaves"] for s in best["shapes"]] == [s["leaves"] for s in greedy["shapes"]]  # the binary tree's leaves either way
    assert a <= b * (1 + 1e-5) and na < nb


def test_collapse_plan_on_a_tree_the_reinsertion_pass_has_relinked():
    """Reinserter::apply() re-links nodes, so a child can sit below its parent in the array; the collapse plan's bottom-up
    pass must follow the links, not the indices. 60 k mixed-size boxes at the default ratio / iteration count (and a heavier setting):
    the plan's cost of the root equals a plain recursive evaluation, and every box still contains its children."""
    from ignis_amd import tables
    lib = tables.host_lib()
    lib.igh_test_collapse_plan.argtypes = [C.POINTER(C.c_float), C.c_uint32, C.c_float, C.c_int32, C.POINTER(C.c_double)]
    lib.igh_test_collapse_plan.restype = C.c_int32
    rng = np.random.default_rng(11)
    n = 60_000
    c = rng.uniform(-10, 10, (n, 3))
    r = rng.lognormal(-2.5, 1.2, (n, 3))  # a few large boxes among many small ones: what makes reinsertion find moves
    boxes = np.ascontiguousarray(np.concatenate([c - r, c + r], axis=1), dtype=np.float32)
    relinked = 0
    for ratio, iters in ((0.05, 3), (0.5, 4)):
        out = (C.c_double * 5)()
        assert lib.igh_test_collapse_plan(boxes.ctypes.data_as(C.POINTER(C.c_float)), n, ratio, iters, out) == 0
        plan, rec, below, broken, emitted = list(out)
        assert below > 0  # otherwise this test does not reach the case
        relinked += below
        assert broken == 0
        assert plan == rec
        assert abs(emitted - plan) <= 1e-4 * plan  # what convertToNArity emits is the plan (float sums in another order)
    assert relinked > 100


_QUANT = r"""
import ctypes as C, json, sys
import numpy as np
sys.path.insert(0, %r)
from ignis_amd.tables import LoadedScene
import oracle
sc = LoadedScene.from_file(sys.argv[1], 96, 54)
s = sc.scene
blob = np.frombuffer(C.string_at(s.primbvh, s.primbvh_size), np.uint8)
tables = [np.frombuffer(C.string_at(C.cast(s.scene_nodes, C.c_void_p), s.scene_node_count * 256), np.uint8).view(np.float32).reshape(-1, 64)]
offsets = sorted({((int(s.scene_leaves[i].user[1]) & 0xFFFFFFFF) << 32 | (int(s.scene_leaves[i].user[0]) & 0xFFFFFFFF)) * 4 for i in range(s.scene_leaf_count)})
for off in offsets:
    nodes = int(np.frombuffer(blob[off:off + 4].tobytes(), np.int32)[0])
    tables.append(np.frombuffer(blob[off + 16:off + 16 + nodes * 256].tobytes(), np.float32).reshape(nodes, 64))
nd = np.concatenate(tables)
rays, _ = oracle.generate_rays(sc, 1, 96, 54, 0, 96 * 54, seed=5)
hits = oracle.trace(sc, rays, flags=1)
np.savez(sys.argv[2], nodes=nd, **{k: hits[k] for k in ("ent_id", "prim_id", "t", "u", "v")}, work=np.array([hits["stats"][k] for k in ("nodes", "tris", "leaves")]))
""" % ROOT


def test_quantised_node_boxes_decode_exactly_and_contain_the_exact_ones(terrain, tmp_path):
    """IGH_NODE_QUANT (bvh.cpp quantise_node8; on by default from 64 MB of nodes): every plane of a used slot is fmaf(q, 2^e, origin)
    for a byte q and the grid the node's padding words name — what lets the HIP device hold the node in 128 bytes without loss —,
    the quantised box contains the exact one and is at most two grid steps larger per side, and the same topology gives the same
    hits (the oracle on both tables), at the price of a few more node visits."""
    out = {}
    for q in ("0", "1"):
        path = str(tmp_path / f"q{q}.npz")
        subprocess.run([sys.executable, "-c", _QUANT, terrain, path], env=dict(os.environ, IGH_NODE_QUANT=q), check=True, capture_output=True)
        out[q] = np.load(path)
    exact, quant = out["0"]["nodes"], out["1"]["nodes"]
    assert exact.shape == quant.shape
    child = quant[:, 48:56].view(np.int32)
    assert np.array_equal(child, exact[:, 48:56].view(np.int32))  # same tree
    used = child != 0
    pad = quant[:, 56:64].view(np.uint32)
    assert ((pad[:, 3] & 0xFF000000) == 0x51000000).all() and not (exact[:, 56:64].view(np.uint32)[:, 3] != 0).any()
    with np.errstate(invalid="ignore"):  # (unused slots hold +-inf)
        for a in range(3):
            origin = pad[:, a].copy().view(np.float32).astype(np.float64)
            scale = np.ldexp(1.0, ((pad[:, 3] >> (8 * a)) & 0xFF).astype(np.int64) - 127)
            for k in range(2):
                p = quant[:, (2 * a + k) * 8:(2 * a + k) * 8 + 8].astype(np.float64)
                e = exact[:, (2 * a + k) * 8:(2 * a + k) * 8 + 8].astype(np.float64)
                q = np.rint((p - origin[:, None]) / scale[:, None])
                ok = (q >= 0) & (q <= 255)
                # some byte within one step of the rounded quotient decodes to the plane, bit for bit (origin + q * 2^e is exact in float64)
                match = np.zeros_like(ok)
                for d in (-1, 0, 1):
                    match |= ((origin[:, None] + (q + d) * scale[:, None]).astype(np.float32) == p.astype(np.float32)) & (q + d >= 0) & (q + d <= 255)
                assert (ok & match)[used].all()
                slack = 2 * scale[:, None] + np.abs(e) * 2.0 ** -22
                if k == 0:
                    assert (p <= e)[used].all() and (e - p <= slack)[used].all()
                else:
                    assert (p >= e)[used].all() and (p - e <= slack)[used].all()
    for k in ("ent_id", "prim_id", "t", "u", "v"):
        assert np.array_equal(out["0"][k], out["1"][k]), k
    we, wq = out["0"]["work"], out["1"]["work"]
    assert wq[0] >= we[0] and wq[0] <= 1.2 * we[0]  # node visits: a few more, not many


def test_quantise_node8_is_conservative_or_leaves_the_node_alone():
    """quantise_node8 on hand-made nodes — ordinary boxes, boxes of one point, denormal extents, extents at the top of the float
    range, and nodes with an infinite or NaN plane. Every used slot of a marked node contains the box it replaces (new_lo <= old_lo,
    new_hi >= old_hi) and decodes from its grid; a node that cannot be put on a grid keeps its floats bit for bit and gets NO mark
    (igd_assign_scene then holds the scene as Node8 records: lossless or not at all)."""
    import ctypes as C
    from ignis_amd.tables import host_lib
    rng = np.random.default_rng(11)
    n = 4096
    lo = rng.uniform(-1, 1, (n, 3, 8)).astype(np.float32) * np.float32(10.0) ** rng.integers(-30, 30, (n, 1, 1)).astype(np.float32)
    ext = np.abs(rng.normal(size=(n, 3, 8))).astype(np.float32) * np.float32(10.0) ** rng.integers(-40, 20, (n, 1, 1)).astype(np.float32)
    with np.errstate(over="ignore"):
        hi = (lo + ext * np.abs(lo) + ext).astype(np.float32)
    hi = np.where(np.isfinite(hi), hi, np.float32(3.0e38))
    lo[1], hi[1] = 0.5, 0.5                                    # a point
    lo[2], hi[2] = -3.0e38, 3.0e38                            # the whole float range on every axis
    lo[3], hi[3] = np.float32(1e-45), np.float32(3e-45)       # denormals
    lo[4, 0, :4], hi[4, 0, :4] = -3.4e38, 3.4e38
    bad = [5, 6, 7]
    hi[5, 1, 2] = np.inf
    lo[6, 2, 0] = -np.inf
    lo[7, 0, 7] = np.nan
    child = rng.integers(0, 2, (n, 8)).astype(np.int32) * rng.integers(1, 1000, (n, 8)).astype(np.int32)
    child[:8] = 1  # (the special nodes use every slot)
    bounds = np.empty((n, 6, 8), np.float32)
    bounds[:, 0::2] = lo
    bounds[:, 1::2] = hi
    before = bounds.copy()
    pad = np.zeros((n, 4), np.int32)
    l = host_lib()
    l.igh_test_quantise_nodes.restype = C.c_int32
    assert l.igh_test_quantise_nodes(bounds.ctypes.data_as(C.POINTER(C.c_float)), child.ctypes.data_as(C.POINTER(C.c_int32)), n, pad.ctypes.data_as(C.POINTER(C.c_int32))) == 0
    marked = (pad[:, 3].view(np.uint32) & 0xFF000000) == 0x51000000
    # (node 4: planes at +-3.4e38 — the next grid step above the upper plane is beyond FLT_MAX, so it has no finite conservative grid either)
    assert not marked[bad + [4]].any() and marked[[0, 1, 2, 3]].all() and marked.sum() >= n - len(bad) - 1
    for b in np.flatnonzero(~marked):  # untouched, bit for bit
        assert np.array_equal(bounds[b].view(np.uint32), before[b].view(np.uint32)) and not pad[b].any()
    used = (child != 0)[:, None, :] & marked[:, None, None]
    assert np.isfinite(bounds[np.broadcast_to(used, (n, 3, 8)).repeat(2, axis=1)]).all()
    new_lo, new_hi = bounds[:, 0::2], bounds[:, 1::2]
    old_lo, old_hi = before[:, 0::2], before[:, 1::2]
    u3 = np.broadcast_to(used, (n, 3, 8))
    assert (new_lo <= old_lo)[u3].all() and (new_hi >= old_hi)[u3].all()
    # unused slots keep what they held
    un = np.broadcast_to((child == 0)[:, None, :], (n, 3, 8))
    assert np.array_equal(new_lo.view(np.uint32)[un], old_lo.view(np.uint32)[un]) and np.array_equal(new_hi.view(np.uint32)[un], old_hi.view(np.uint32)[un])
    # every plane of a marked node decodes from its grid: fmaf(q, 2^e, origin) for a byte q (float64 holds origin + q * 2^e exactly or rounds it once)
    for a in range(3):
        origin = pad[:, a].copy().view(np.float32).astype(np.float64)[:, None]
        e = ((pad[:, 3].view(np.uint32) >> (8 * a)) & 0xFF).astype(np.int64) - 127
        scale = np.ldexp(1.0, e)[:, None]
        for planes in (new_lo[:, a], new_hi[:, a]):
            with np.errstate(invalid="ignore", over="ignore"):
                q = np.rint((planes.astype(np.float64) - origin) / scale)
                hit = np.zeros(planes.shape, bool)
                for d in (-1, 0, 1):
                    hit |= ((origin + (q + d) * scale).astype(np.float32) == planes) & (q + d >= 0) & (q + d <= 255)
            assert hit[used[:, 0, :]].all()
