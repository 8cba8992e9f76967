"""CPU-side checks of the drop-in boundary: the C-ABI libraries load and export every symbol the
headers declare (no compute calls without a GPU), and failures are loud."""
import ctypes as C
import json
import os
import re

import pytest

from conftest import ROOT, flat_scene


def _declared(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ig[dh]_[a-z_0-9]+)\s*\(", text)))


def test_device_library_exports_every_declared_symbol():
    from ignis_amd import device
    lib = device.lib()
    names = _declared("igd_device.h")
    assert set(names) == set(device.EXPORTS)
    for n in names:
        assert hasattr(lib, n), n
    assert lib.igd_get_abi_version() == 2  # igd_stats grew in round 6


def test_host_library_exports_every_declared_symbol():
    from ignis_amd import tables
    lib = tables.host_lib()
    for n in _declared("igh_host.h"):
        assert hasattr(lib, n), n


def test_no_gpu_is_a_loud_error_not_a_fallback():
    from ignis_amd import Device, DeviceError, device
    if device.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(DeviceError):
        Device(0)


def test_denoiser_registration_is_host_logic_only():
    """registerDenoiser / hasDenoiser (Runtime::hasDenoiser, Runtime.cpp:756) and DenoiserSettings' defaults (RuntimeSettings.h:6-10);
    the render side of the hook is a GPU test."""
    import ignis_amd
    assert not ignis_amd.hasDenoiser()
    d = ignis_amd.RuntimeOptions.makeDefault().Denoiser
    assert (d.Enabled, d.HighQuality, d.Prefilter) == (False, True, False)
    with pytest.raises(TypeError):
        ignis_amd.registerDenoiser(3)
    ignis_amd.registerDenoiser(lambda c, n, a, s: c)
    try:
        assert ignis_amd.hasDenoiser()
    finally:
        ignis_amd.registerDenoiser(None)
    assert not ignis_amd.hasDenoiser()


def test_product_does_not_touch_the_oracle():
    """The oracle is test infrastructure: nothing under ignis_amd/ may import, link or load it."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "ignis_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".h", ".hip", "Makefile")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "liboracle" not in text and "import oracle" not in text and "oracle/" not in text, os.path.join(dirpath, f)


def test_loader_table_layouts(diamond_scene):
    sc = diamond_scene.scene
    assert (sc.entity_count, sc.shape_count, sc.material_count, sc.light_count) == (9, 7, 4, 1)
    assert list(sc.entity_per_material[:4]) == [1, 3, 2, 3]
    assert diamond_scene.entity_name(0) == "AreaLight" and diamond_scene.entity_name(8) == "Diamond3"
    # prim BVH fix table: header {nodes, tris, 0, 0} then 256-byte nodes and 208-byte Tri4 packets
    blob = diamond_scene.primbvh_bytes()
    import struct
    nodes, tris, p0, p1 = struct.unpack_from("<4I", blob, 0)
    assert p0 == 0 and p1 == 0 and nodes >= 1 and tris >= 1
    # every scene leaf points at a valid prim BVH
    for i in range(sc.scene_leaf_count):
        leaf = sc.scene_leaves[i]
        off = ((leaf.user[1] & 0xFFFFFFFF) << 32 | (leaf.user[0] & 0xFFFFFFFF)) * 4
        n, t, _, _ = struct.unpack_from("<4I", blob, off)
        assert off + 16 + n * 256 + t * 208 <= len(blob)
    assert sc.scene_leaves[sc.scene_leaf_count - 1].entity_id < 0  # sentinel on the last leaf of a run


def test_loader_refuses_what_it_cannot_lower():
    from ignis_amd.tables import LoadedScene
    bad = flat_scene()
    bad["bsdfs"][0] = {"type": "tensortree", "name": "ground"}
    with pytest.raises(RuntimeError, match="not supported"):
        LoadedScene.from_string(json.dumps(bad))
    bad = flat_scene()
    bad["bsdfs"][0]["reflectance"] = "some_texture"
    with pytest.raises(RuntimeError, match="unknown variable 'some_texture'"):
        LoadedScene.from_string(json.dumps(bad))
    with pytest.raises(RuntimeError, match="JSON error"):
        LoadedScene.from_string("{ not json")
    with pytest.raises(RuntimeError):
        LoadedScene.from_file("/nonexistent/scene.json")


def test_transform_parsing_matches_matrix_form():
    from ignis_amd.tables import LoadedScene
    import numpy as np
    a = flat_scene()
    a["entities"][0]["transform"] = [{"translate": [0.5, 0, 0]}, {"scale": 2}]
    b = flat_scene()
    b["entities"][0]["transform"] = [2, 0, 0, 0.5, 0, 2, 0, 0, 0, 0, 2, 0, 0, 0, 0, 1]
    sa, sb = LoadedScene.from_string(json.dumps(a)), LoadedScene.from_string(json.dumps(b))  # keep the owners alive
    ea = np.ctypeslib.as_array(sa.scene.entities, shape=(36,)).copy()
    eb = np.ctypeslib.as_array(sb.scene.entities, shape=(36,)).copy()
    np.testing.assert_array_equal(ea, eb)


def test_loader_bitmap_texture_matches_an_independent_png_decode():
    """Bump-map texture of many_point_lights: the loader's PNG reader + packing (bottom-up rows, sRGB -> linear
    re-quantised to 8 bit, Image.cpp:40-51,714-808) against a decode written with Python's zlib."""
    import struct
    import zlib
    import numpy as np
    from ignis_amd.tables import LoadedScene
    sc = LoadedScene.from_file(os.path.join(ROOT, "scenes", "many_point_lights.json"), 64, 64)
    s = sc.scene
    assert s.texture_count == 2  # [0] the sky light's model image (512 x 256 floats), [1] the bump map
    t = s.textures[1]
    raw = open(os.path.join(ROOT, "scenes", "textures", "bumpmap.png"), "rb").read()
    pos, idat, hdr = 8, b"", None
    while pos < len(raw):
        n, typ = struct.unpack(">I4s", raw[pos:pos + 8])
        body = raw[pos + 8:pos + 8 + n]
        if typ == b"IHDR":
            hdr = struct.unpack(">IIBBBBB", body)
        elif typ == b"IDAT":
            idat += body
        pos += 12 + n
    w, h, depth, ctype = hdr[:4]
    assert (t.width, t.height, t.channels, t.filter) == (w, h, 4, 2) and depth == 8 and ctype == 6  # "trilinear" -> bicubic
    data = np.frombuffer(zlib.decompress(idat), np.uint8).reshape(h, 1 + 4 * w)
    img = np.zeros((h, w * 4), np.int32)
    for y in range(h):
        ft, line = int(data[y, 0]), data[y, 1:].astype(np.int32)
        up = img[y - 1] if y else np.zeros(w * 4, np.int32)
        out = img[y]
        for x in range(w * 4):
            a = out[x - 4] if x >= 4 else 0
            b = up[x]
            c = up[x - 4] if x >= 4 else 0
            if ft == 0:
                p = 0
            elif ft == 1:
                p = a
            elif ft == 2:
                p = b
            elif ft == 3:
                p = (a + b) // 2
            else:
                q = a + b - c
                pa, pb, pc = abs(q - a), abs(q - b), abs(q - c)
                p = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
            out[x] = (line[x] + p) & 255
    img = img.reshape(h, w, 4)[::-1]  # stb's vertical flip
    v = img[..., :3].astype(np.float32) / np.float32(255)
    lin = np.where(v <= np.float32(0.04045), v / np.float32(12.92), ((v + np.float32(0.055)) / np.float32(1.055)) ** np.float32(2.4))
    want = np.concatenate([np.minimum(255, np.floor(lin * np.float32(255))).astype(np.uint8), img[..., 3:].astype(np.uint8)], axis=-1)
    got = np.ctypeslib.as_array(s.texture_data, shape=(s.texture_data_size,))[t.offset:t.offset + w * h * 4].reshape(h, w, 4)
    diff = np.abs(got.astype(np.int32) - want.astype(np.int32))
    assert diff[..., 3].max() == 0
    assert diff.max() <= 1 and (diff > 0).mean() < 1e-3  # powf rounding may move a value across a floor() boundary


def _read_exr(path):
    """Independent reader for the subset igh_save_exr writes: single-part scanline, float channels, uncompressed or ZIP
    (blocks of 16 scanlines: zlib stream -> undo the delta predictor -> interleave the two byte halves)."""
    import zlib
    import struct
    import numpy as np
    d = open(path, "rb").read()
    magic, version = struct.unpack_from("<ii", d, 0)
    assert magic == 20000630 and version == 2
    pos, attrs = 8, {}
    while d[pos] != 0:
        e = d.index(b"\0", pos)
        name = d[pos:e].decode()
        pos = e + 1
        e = d.index(b"\0", pos)
        typ = d[pos:e].decode()
        pos = e + 1
        size, = struct.unpack_from("<i", d, pos)
        pos += 4
        attrs[name] = (typ, d[pos:pos + size])
        pos += size
    pos += 1
    chans, cp = [], 0
    cl = attrs["channels"][1]
    while cl[cp] != 0:
        e = cl.index(b"\0", cp)
        chans.append((cl[cp:e].decode(), struct.unpack_from("<i", cl, e + 1)[0]))
        cp = e + 1 + 16
    x0, y0, x1, y1 = struct.unpack("<4i", attrs["dataWindow"][1])
    w, h = x1 - x0 + 1, y1 - y0 + 1
    assert attrs["compression"][1] in (b"\0", b"\3") and attrs["lineOrder"][1] == b"\0"
    lines = 16 if attrs["compression"][1] == b"\3" else 1
    blocks = (h + lines - 1) // lines
    offsets = struct.unpack_from(f"<{blocks}Q", d, pos)
    planes = {c: np.zeros((h, w), np.float32) for c, _ in chans}
    attrs["_stored_bytes"] = 0
    for off in offsets:
        y, size = struct.unpack_from("<ii", d, off)
        ny = min(lines, y1 - y + 1)
        want = w * 4 * len(chans) * ny
        attrs["_stored_bytes"] += size
        raw = d[off + 8:off + 8 + size]
        if size != want:  # a block that did not shrink is stored as is
            t = np.frombuffer(zlib.decompress(raw), np.uint8)
            assert t.size == want
            t = (np.cumsum(t.astype(np.int64) - 128) + 128).astype(np.uint8)  # t[i] = t[i - 1] + d[i] - 128 (mod 256), t[0] = d[0]
            half = (want + 1) // 2
            out = np.empty(want, np.uint8)
            out[0::2], out[1::2] = t[:half], t[half:]
            raw = out.tobytes()
        rows = np.frombuffer(raw, np.float32).reshape(ny, len(chans), w)
        for k, (c, t) in enumerate(chans):
            assert t == 2  # FLOAT
            planes[c][y - y0:y - y0 + ny] = rows[:, k]
    return planes, attrs


def test_exr_writer_round_trip(tmp_path):
    """Runtime::saveFramebuffer's file (Runtime.cpp:794-876): channels B, G, R, float, scaled by 1 / iterations."""
    import numpy as np
    from ignis_amd.tables import save_exr
    img = np.random.default_rng(1).random((5, 7, 3), dtype=np.float32)
    p = str(tmp_path / "t.exr")
    save_exr(p, img, 0.5, {"igSeed": 3, "igTechniqueType": "path"})
    planes, attrs = _read_exr(p)
    assert sorted(planes) == ["B", "G", "R"]
    for k, c in enumerate("RGB"):
        np.testing.assert_array_equal(planes[c], img[..., k] * np.float32(0.5))
    assert attrs["igSeed"] == ("string", b"3") and attrs["igTechniqueType"][1] == b"path"
    with pytest.raises(RuntimeError):
        save_exr("/nonexistent_dir/x.exr", img)
    # a film-like image (smooth, several 16-line blocks and a ragged last one) shrinks and comes back bit for bit, through the
    # independent reader above and through the loader's own EXR reader; IGH_EXR_COMPRESSION=none keeps the uncompressed form
    yy, xx = np.mgrid[0:37, 0:50].astype(np.float32)
    film = np.stack([np.sin(xx * 0.1) + 2, yy * 0.01, np.float32(0.25) * np.ones_like(xx)], -1).astype(np.float32)
    film[3, 4] = [np.inf, -0.0, 1e-40]
    save_exr(p, film)
    planes, attrs = _read_exr(p)
    assert attrs["compression"][1] == b"\3" and attrs["_stored_bytes"] < film.nbytes // 2
    for k, c in enumerate("RGB"):
        np.testing.assert_array_equal(planes[c].view(np.uint32), film[..., k].view(np.uint32))
    from ignis_amd.tables import read_float_image
    np.testing.assert_array_equal(read_float_image(p)[..., :3].view(np.uint32), film.view(np.uint32))
    os.environ["IGH_EXR_COMPRESSION"] = "none"
    try:
        save_exr(p, film)
    finally:
        del os.environ["IGH_EXR_COMPRESSION"]
    planes, attrs = _read_exr(p)
    assert attrs["compression"][1] == b"\0" and attrs["_stored_bytes"] == film.nbytes
    np.testing.assert_array_equal(planes["G"], film[..., 1])


def test_obj_loader_matches_the_same_mesh_as_ply(tmp_path):
    """OBJ meshes (src/runtime/mesh/ObjFile.cpp): a quad-faced OBJ with v/vt/vn corners, negative indices and a
    shared-vertex fan loads into the same tables as the equivalent hand-written triangle soup in PLY."""
    import numpy as np
    from ignis_amd.tables import LoadedScene
    obj = """# unit quad in the xy plane, split by the loader, plus a triangle using relative indices
v 0 0 0
v 1 0 0
v 1 1 0
v 0 1 0
vn 0 0 1
vt 0 0
vt 1 0
vt 1 1
vt 0 1
f 1/1/1 2/2/1 3/3/1 4/4/1
v 2 0 0
v 3 0 0
v 2 1 0
f -3/1/-1 -2/2/-1 -1/4/-1
"""
    (tmp_path / "m.obj").write_text(obj)
    ply = """ply
format ascii 1.0
element vertex 7
property float x
property float y
property float z
property float nx
property float ny
property float nz
property float s
property float t
element face 3
property list uchar int vertex_indices
end_header
0 0 0 0 0 1 0 0
1 0 0 0 0 1 1 0
1 1 0 0 0 1 1 1
0 1 0 0 0 1 0 1
2 0 0 0 0 1 0 0
3 0 0 0 0 1 1 0
2 1 0 0 0 1 0 1
3 0 1 2
3 0 2 3
3 4 5 6
"""
    (tmp_path / "m.ply").write_text(ply)
    scenes = []
    for fn in ("m.obj", "m.ply"):
        s = flat_scene()
        s["shapes"][0] = {"type": "external", "name": s["shapes"][0]["name"], "filename": fn}
        scenes.append(LoadedScene.from_string(json.dumps(s), str(tmp_path)))
    a, b = scenes
    assert a.scene.shape_data_size == b.scene.shape_data_size
    da = np.ctypeslib.as_array(a.scene.shape_data, shape=(a.scene.shape_data_size,))
    db = np.ctypeslib.as_array(b.scene.shape_data, shape=(b.scene.shape_data_size,))
    np.testing.assert_array_equal(da, db)
    np.testing.assert_array_equal(np.frombuffer(a.primbvh_bytes(), np.uint8), np.frombuffer(b.primbvh_bytes(), np.uint8))


def test_generated_plane_is_detected_as_plane_and_triangle_is_not():
    """src/tests/units/trimesh_plane.cpp: MakePlane(0, X, Y) -> origin 0, axes X / Y, area 1, normal +Z; a triangle is
    not a plane (it becomes a mesh area light, sampled triangle by triangle)."""
    import numpy as np
    from ignis_amd.tables import LoadedScene
    s = flat_scene()
    s["shapes"].append({"type": "rectangle", "name": "L", "origin": [0, 0, 0], "width": 1, "height": 1})
    s["entities"].append({"name": "L", "shape": "L", "bsdf": "ground"})
    s["lights"] = [{"type": "area", "name": "A", "entity": "L", "radiance": [1, 1, 1]}]
    sc = LoadedScene.from_string(json.dumps(s))
    light = sc.scene.lights[0]
    d = np.array(list(light.d), np.float32)
    assert light.type == 0  # IG_LIGHT_PLANE
    np.testing.assert_allclose(d[0:3], [0, 0, 0], atol=1e-7)          # origin
    np.testing.assert_allclose(d[4:7], [1, 0, 0], rtol=1e-6)          # x axis
    np.testing.assert_allclose(d[8:11], [0, 1, 0], rtol=1e-6)         # y axis
    np.testing.assert_allclose([d[3], d[7], d[11]], [0, 0, 1], atol=1e-6)  # normal
    np.testing.assert_allclose(d[23], 1.0, rtol=1e-6)                 # area
    s["shapes"][-1] = {"type": "triangle", "name": "L", "p0": [0, 0, 0], "p1": [1, 0, 0], "p2": [0, 1, 0]}
    tri = LoadedScene.from_string(json.dumps(s))
    assert tri.scene.lights[0].type == 8 and tri.scene.lights[0].entity_id == 1  # IG_LIGHT_MESH_AREA on entity "L"


# ---- procedural shapes (src/runtime/mesh/TriMesh.cpp:819-1131) and "externals" (src/runtime/loader/Parser.cpp:395-463)

def _shape_scene(shape):
    from ignis_amd.tables import LoadedScene
    s = flat_scene()
    s["shapes"] = [dict(shape, name="S")]
    s["entities"] = [{"name": "E", "shape": "S", "bsdf": "ground"}]
    return LoadedScene.from_string(json.dumps(s))


def _edge_use(idx):
    import collections
    use = collections.Counter()
    for a, b, c in idx:
        for p, q in ((a, b), (b, c), (c, a)):
            use[(int(p), int(q))] += 1
    return use


def _is_closed_and_consistently_wound(v, idx):
    """Watertight after welding coincident vertices: every directed edge is used once and its reverse once."""
    import numpy as np
    key = {}
    weld = np.array([key.setdefault(tuple(np.round(p, 5)), len(key)) for p in v])
    faces = [f for f in weld[idx] if len(set(f)) == 3]  # pole triangles of the uv sphere collapse
    use = _edge_use(faces)
    return all(n == 1 and use.get((q, p), 0) == 1 for (p, q), n in use.items())


def _volume(v, idx):
    import numpy as np
    a, b, c = (v[idx[:, k]].astype(np.float64) for k in range(3))
    return float(np.einsum("ij,ij->i", a, np.cross(b, c)).sum() / 6)


@pytest.mark.parametrize("shape,faces", [
    ({"type": "icosphere", "center": [1, 2, 3], "radius": 0.5, "subdivisions": 2}, 20 * 16),
    ({"type": "uvsphere", "center": [1, 2, 3], "radius": 0.5, "stacks": 12, "slices": 10}, 12 * 10 * 2),
])
def test_procedural_spheres(shape, faces):
    import numpy as np
    sc = _shape_scene(shape)
    v, n, idx, uv = sc.shape_mesh(0)
    assert len(idx) == faces and len(n) == len(v) == len(uv)
    c = np.float32(shape["center"])
    np.testing.assert_allclose(np.linalg.norm(v - c, axis=1), 0.5, rtol=1e-5)
    np.testing.assert_allclose(n, (v - c) / 0.5, atol=1e-5)          # outward unit normals
    assert _is_closed_and_consistently_wound(v, idx)
    vol = _volume(v - c, idx)
    assert 0.85 * 4 / 3 * np.pi * 0.125 < vol < 4 / 3 * np.pi * 0.125  # inscribed, outward winding (positive volume)
    assert uv.min() >= 0 and uv.max() <= 1


def test_procedural_disk_cone_cylinder():
    import numpy as np
    n = 24
    v, nrm, idx, _ = _shape_scene({"type": "disk", "origin": [0, 0, 1], "normal": [0, 0, 1], "radius": 2, "sections": n}).shape_mesh(0)
    assert len(idx) == n and np.allclose(v[:, 2], 1) and np.allclose(nrm, [0, 0, 1])
    a, b, c = (v[idx[:, k]].astype(np.float64) for k in range(3))
    area = np.linalg.norm(np.cross(b - a, c - a), axis=1).sum() / 2
    assert area == pytest.approx(n / 2 * 4 * np.sin(2 * np.pi / n), rel=1e-5)          # regular n-gon
    assert np.all(np.cross(b - a, c - a)[:, 2] > 0)                                    # wound around the normal

    v, nrm, idx, _ = _shape_scene({"type": "cylinder", "p0": [0, 0, 0], "p1": [0, 0, 3], "radius": 1, "sections": n}).shape_mesh(0)
    assert len(idx) == 4 * n and _is_closed_and_consistently_wound(v, idx)
    assert abs(_volume(v, idx)) == pytest.approx(3 * n / 2 * np.sin(2 * np.pi / n), rel=1e-5)  # prism over the n-gon
    v2, _, idx2, _ = _shape_scene({"type": "cylinder", "p0": [0, 0, 0], "p1": [0, 0, 3], "bottom_radius": 1, "top_radius": 0.5,
                                   "sections": n, "filled": False}).shape_mesh(0)
    assert len(idx2) == 2 * n and len(v2) == 2 * n and np.allclose(np.hypot(v2[n:, 0], v2[n:, 1]), 0.5, atol=1e-6)

    v, nrm, idx, _ = _shape_scene({"type": "cone", "p0": [0, 0, 0], "p1": [0, 0, 2], "radius": 1, "sections": n}).shape_mesh(0)
    assert len(idx) == 2 * n and _is_closed_and_consistently_wound(v, idx)
    assert abs(_volume(v, idx)) == pytest.approx(2 / 3 * n / 2 * np.sin(2 * np.pi / n), rel=1e-5)
    assert np.allclose(np.linalg.norm(nrm, axis=1), 1, atol=1e-5)


def test_inline_mesh_and_typed_arrays():
    import numpy as np
    tri = {"type": "inline", "indices": {"type": "integer", "values": [0, 1, 2, 1, 3, 2]},
           "vertices": {"type": "number", "values": [0, 0, 0, 1, 0, 0, 0, 1, 0, 1, 1, 0]}}
    v, n, idx, uv = _shape_scene(tri).shape_mesh(0)
    assert idx.tolist() == [[0, 1, 2], [1, 3, 2]] and np.allclose(n, [0, 0, 1]) and v.shape == (4, 3)
    plain = dict(tri, indices=[0, 1, 2, 1, 3, 2], vertices=[0, 0, 0, 1, 0, 0, 0, 1, 0, 1, 1, 0], normals=[0, 0, -1] * 4,
                 texcoords=[0, 0, 1, 0, 0, 1, 1, 1])
    v, n, idx, uv = _shape_scene(plain).shape_mesh(0)
    assert np.allclose(n, [0, 0, -1]) and uv.tolist() == [[0, 0], [1, 0], [0, 1], [1, 1]]
    from ignis_amd.tables import LoadedScene
    bad = flat_scene()
    bad["shapes"] = [{"type": "inline", "name": "Bottom", "indices": [0, 1], "vertices": [0, 0, 0]}]
    with pytest.raises(RuntimeError, match="multiple of 3"):
        LoadedScene.from_string(json.dumps(bad))


def test_externals_are_merged_and_overridden(tmp_path):
    """The including file replaces named objects of the included one; file names stay relative to the declaring file."""
    import numpy as np
    import shutil
    from ignis_amd.tables import LoadedScene
    inc = tmp_path / "inc"
    inc.mkdir()
    shutil.copy(os.path.join(ROOT, "scenes", "meshes", "Bottom.ply"), inc / "floor.ply")
    base = flat_scene()
    base["shapes"] = [{"type": "ply", "name": "Bottom", "filename": "floor.ply"}]
    base["bsdfs"] = [{"type": "diffuse", "name": "ground", "reflectance": [0.1, 0.2, 0.3]}]
    (inc / "base.json").write_text(json.dumps(base))
    top = {"externals": [{"filename": "inc/base.json"}],
           "bsdfs": [{"type": "diffuse", "name": "ground", "reflectance": [0.9, 0.8, 0.7]}],
           "film": {"size": [32, 16]}}
    (tmp_path / "top.json").write_text(json.dumps(top))
    sc = LoadedScene.from_file(str(tmp_path / "top.json"))
    t = sc.scene
    assert (t.entity_count, t.shape_count, t.material_count) == (1, 1, 1) and (t.film_width, t.film_height) == (32, 16)
    assert list(t.materials[0].p[0:3]) == [np.float32(0.9), np.float32(0.8), np.float32(0.7)]
    flat = dict(base, film=top["film"], bsdfs=top["bsdfs"])
    flat["shapes"] = [{"type": "ply", "name": "Bottom", "filename": str(inc / "floor.ply")}]
    ref = LoadedScene.from_string(json.dumps(flat))
    assert sc.primbvh_bytes() == ref.primbvh_bytes()
    with pytest.raises(RuntimeError, match="Could not find path"):
        LoadedScene.from_string(json.dumps({"externals": [{"filename": "nope.json"}]}), str(tmp_path))


def _write_serialized(path, meshes, version=4, double=False):
    """Mitsuba serialized mesh writer for the tests (format as documented with Mitsuba 0.5 and read by
    src/runtime/mesh/MtsSerializedFile.cpp): meshes = [(vertices, normals or None, texcoords or None, triangles)]."""
    import struct
    import zlib
    import numpy as np
    blob, offsets = b"", []
    ft = "<f8" if double else "<f4"
    for v, n, t, tris in meshes:
        flags = (0x2000 if double else 0x1000) | (1 if n is not None else 0) | (2 if t is not None else 0)
        body = struct.pack("<I", flags) + (b"mesh\0" if version >= 4 else b"") + struct.pack("<QQ", len(v), len(tris))
        body += np.asarray(v, ft).tobytes()
        if n is not None:
            body += np.asarray(n, ft).tobytes()
        if t is not None:
            body += np.asarray(t, ft).tobytes()
        body += np.asarray(tris, "<u4").tobytes()
        offsets.append(len(blob))
        blob += struct.pack("<HH", 0x041C, version) + zlib.compress(body)
    blob += b"".join(struct.pack("<Q" if version >= 4 else "<I", o) for o in offsets) + struct.pack("<I", len(offsets))
    open(path, "wb").write(blob)


@pytest.mark.parametrize("version,double", [(4, False), (3, True)])
def test_mitsuba_serialized_meshes(tmp_path, version, double):
    import numpy as np
    from ignis_amd.tables import LoadedScene
    quad = ([[0, 0, 0], [1, 0, 0], [1, 1, 0], [0, 1, 0]], [[0, 0, 2]] * 4, [[0, 0], [1, 0], [1, 1], [0, 1]], [[0, 1, 2], [0, 2, 3]])
    tri = ([[0, 0, 1], [2, 0, 1], [0, 2, 1]], None, None, [[0, 1, 2]])
    path = str(tmp_path / "two.serialized")
    _write_serialized(path, [quad, tri], version, double)
    s = flat_scene()
    s["shapes"] = [{"type": "mitsuba", "name": "A", "filename": path}, {"type": "external", "name": "B", "filename": path, "shape_index": 1}]
    s["entities"] = [{"name": "EA", "shape": "A", "bsdf": "ground"}, {"name": "EB", "shape": "B", "bsdf": "ground"}]
    sc = LoadedScene.from_string(json.dumps(s))
    v, n, idx, uv = sc.shape_mesh(0)
    assert v.tolist() == quad[0] and idx.tolist() == quad[3] and uv.tolist() == quad[2]
    assert np.allclose(n, [0, 0, 1])                     # given normals are normalised (fixNormals)
    v, n, idx, uv = sc.shape_mesh(1)
    assert v.tolist() == tri[0] and idx.tolist() == [[0, 1, 2]] and np.allclose(n, [0, 0, 1])  # computed normals
    s["shapes"][1]["shape_index"] = 2
    with pytest.raises(RuntimeError, match="out of range"):
        LoadedScene.from_string(json.dumps(s))
    open(path, "wb").write(b"\x00" * 32)
    with pytest.raises(RuntimeError, match="not a Mitsuba serialized file"):
        LoadedScene.from_string(json.dumps(s))


def test_constant_expressions_and_the_inline_checkerboard_idiom():
    """Exporter-style strings: constant PExpr arithmetic is folded; "select(checkerboard(uvw * S) == 1, A, B)" is the checkerboard
    texture with scale (S, S), color0 = A, color1 = B (texture/checkerboard.art:1-12); anything else is still refused."""
    import numpy as np
    from ignis_amd.tables import LoadedScene
    s = flat_scene([{"type": "point", "name": "P", "position": [0, 0, -1], "intensity": "(color(1.0, 0.5, 0.25, 1.0) * 100.0) / 4 - 0.5"}])
    s["bsdfs"][0]["reflectance"] = "select(checkerboard(uvw * 10.0) == 1, color(0.8, 0.7, 0.6, 1.0), color(0.2, 0.2, 0.2, 1.0) * 0.5)"
    a = LoadedScene.from_string(json.dumps(s))
    assert list(a.scene.lights[0].d[4:7]) == pytest.approx([24.5, 12.0, 5.75], rel=1e-6)
    m = a.scene.materials[0]
    assert m.flags & 4 and list(m.q) == pytest.approx([0.8, 0.7, 0.6, 0.1, 0.1, 0.1, 10, 10])
    s["textures"] = [{"type": "checkerboard", "name": "chk", "scale_x": 10, "scale_y": 10, "color0": [0.8, 0.7, 0.6], "color1": [0.1, 0.1, 0.1]}]
    s["bsdfs"][0]["reflectance"] = "chk"
    b = LoadedScene.from_string(json.dumps(s))
    assert bytes(a.scene.materials[0]) == bytes(b.scene.materials[0])
    for bad, what in (("select(checkerboard(uvw * 10.0) == 1, some_texture, color(0,0,0,1))", "unknown variable 'some_texture'"),
                      ("color(1, 1)", "no function color"), ("color(1,1,1,1) *", "end of expression")):
        s["bsdfs"][0]["reflectance"] = bad
        with pytest.raises(RuntimeError, match=what):
            LoadedScene.from_string(json.dumps(s))
    s["bsdfs"][0]["reflectance"] = "uv.x * 2"  # what is not constant becomes a program of the expression table (tests/test_pexpr.py)
    prog = LoadedScene.from_string(json.dumps(s))  # (keep the owner of the tables alive)
    assert prog.scene.materials[0].flags & (1 << 8)


def test_sun_position_from_time_and_place():
    """LoaderUtils::getEA without a direction: the PSA algorithm on the light's date, time and place (defaults: 6 May 2020 12:00,
    49.24 N, 7.00 E, UTC+2). Pinned by the reference unit test's known answer (units/sun.cpp), then checked against textbook solar
    geometry (Cooper's declination, hour angle from UTC and longitude; good to about a degree) on other dates and for the obvious symmetries."""
    import numpy as np
    from ignis_amd.tables import LoadedScene

    def sun(**kw):
        s = flat_scene([dict({"type": "cie_clear", "name": "sky"}, **kw)])
        sc = LoadedScene.from_string(json.dumps(s))
        d = np.float64(list(sc.scene.lights[0].d[9:12]))
        return d

    def textbook(day_of_year, utc_hours, lat, lon_east):
        decl = np.radians(23.45) * np.sin(2 * np.pi * (284 + day_of_year) / 365)
        b = 2 * np.pi * (day_of_year - 81) / 364
        eot = 9.87 * np.sin(2 * b) - 7.53 * np.cos(b) - 1.5 * np.sin(b)  # minutes
        h = np.radians(15 * (utc_hours + lon_east / 15 + eot / 60 - 12))
        la = np.radians(lat)
        return np.degrees(np.arcsin(np.sin(la) * np.sin(decl) + np.cos(la) * np.cos(decl) * np.cos(h)))

    # the reference's own known answer first (src/tests/units/sun.cpp:8-37; also tests/test_sky.py): 20.86 deg above the horizon, 10.81 deg west of south
    ka = sun(year=2022, month=11, day=18, hour=13, minute=0, seconds=0, latitude=49.235422, longitude=-6.9965744, timezone=-1)
    np.testing.assert_allclose([ka[0], ka[2], ka[1]], [-0.175382, -0.918072, 0.355506], rtol=1e-3)
    assert np.degrees(np.arcsin(ka[1])) == pytest.approx(20.86, rel=1e-2) and np.degrees(np.arctan2(-ka[0], -ka[2])) == pytest.approx(10.81, rel=1e-2)
    d = sun()
    assert np.linalg.norm(d) == pytest.approx(1, abs=1e-6)
    assert np.degrees(np.arcsin(d[1])) == pytest.approx(textbook(127, 10.0, 49.235422, 6.9965744), abs=1.0)
    assert d[2] < 0 and d[0] > 0  # before solar noon: east of south (south = -z, east = +x in the Y-up sky frame of ElevationAzimuth.h:22-30)
    noon = sun(hour=13, minute=33)  # about solar noon at 7 E in summer time
    assert abs(noon[0]) < 0.03 and noon[1] > d[1]
    winter = sun(month=12, day=21)
    assert np.degrees(np.arcsin(winter[1])) == pytest.approx(textbook(355, 10.0, 49.235422, 6.9965744), abs=1.0)
    ea = sun(elevation=0.5, azimuth=0.25)
    np.testing.assert_allclose(ea, [-np.cos(0.5) * np.sin(0.25), np.sin(0.5), -np.cos(0.5) * np.cos(0.25)], atol=1e-6)


def test_qrotate_and_entities_without_known_parts():
    """"qrotate" is a (w, x, y, z) quaternion; entities naming a missing bsdf or shape are left out, the scene still loads."""
    import numpy as np
    from ignis_amd.tables import LoadedScene
    a, b = flat_scene(), flat_scene()
    h = np.sqrt(0.5)
    a["entities"][0]["transform"] = {"translate": [1, 2, 3], "qrotate": [h, 0, 0, h]}  # 90 degrees about z
    b["entities"][0]["transform"] = {"translate": [1, 2, 3], "rotate": [0, 0, 90]}
    sa, sb = LoadedScene.from_string(json.dumps(a)), LoadedScene.from_string(json.dumps(b))
    ea = np.ctypeslib.as_array(sa.scene.entities, shape=(36,))[:33]
    eb = np.ctypeslib.as_array(sb.scene.entities, shape=(36,))[:33]
    np.testing.assert_allclose(ea, eb, atol=1e-6)
    a["entities"] += [{"name": "NoBsdf", "shape": "Bottom", "bsdf": "missing"}, {"name": "NoShape", "shape": "missing", "bsdf": "ground"}]
    sc = LoadedScene.from_string(json.dumps(a))
    assert sc.scene.entity_count == 1 and sc.entity_name(0) == "Bottom"


def _write_png(path, width, height, ctype, depth, rows, palette=None, trns=None):
    """Minimal PNG writer for the reader's tests: `rows` are the packed sample bytes of each row (filter type 0)."""
    import struct
    import zlib

    def chunk(typ, body):
        return struct.pack(">I", len(body)) + typ + body + struct.pack(">I", zlib.crc32(typ + body) & 0xFFFFFFFF)
    raw = b"".join(b"\x00" + bytes(r) for r in rows)
    out = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", width, height, depth, ctype, 0, 0, 0))
    if palette is not None:
        out += chunk(b"PLTE", bytes(palette))
    if trns is not None:
        out += chunk(b"tRNS", bytes(trns))
    out += chunk(b"IDAT", zlib.compress(raw)) + chunk(b"IEND", b"")
    open(path, "wb").write(out)


def _texture_scene(png_name):
    return {
        "technique": {"type": "path", "max_depth": 2},
        "camera": {"type": "perspective", "fov": 40, "near_clip": 0.1, "far_clip": 100,
                   "transform": [-1, 0, 0, 0, 0, 1, 0, 0, 0, 0, -1, 3.85, 0, 0, 0, 1]},
        "film": {"size": [16, 16]},
        "textures": [{"type": "image", "name": "tex", "filename": png_name, "filter_type": "nearest", "linear": True}],
        "bsdfs": [{"type": "diffuse", "name": "mat", "reflectance": "tex"}],
        "shapes": [{"type": "rectangle", "name": "Bottom"}],
        "entities": [{"name": "Bottom", "shape": "Bottom", "bsdf": "mat"}],
        "lights": [{"type": "point", "name": "l", "position": [0, 0, 2], "intensity": [1, 1, 1]}],
    }


def _loaded_texels(tmp_path, name):
    import numpy as np
    from ignis_amd.tables import LoadedScene
    sc = LoadedScene.from_string(json.dumps(_texture_scene(name)), base_dir=str(tmp_path))
    s = sc.scene
    assert s.texture_count == 1
    t = s.textures[0]
    n = t.width * t.height * (1 if t.channels == 1 else 4)
    from types import SimpleNamespace  # (the tables die with `sc`)
    return (SimpleNamespace(width=int(t.width), height=int(t.height), channels=int(t.channels)),
            np.ctypeslib.as_array(s.texture_data, shape=(s.texture_data_size,))[t.offset:t.offset + n].copy())


def test_png_reader_palette_16bit_and_packed_gray(tmp_path):
    """What stb_image (Image.cpp:714-808) yields for the PNG variants beyond 8-bit truecolour: palette indices expanded to RGB
    (RGBA with tRNS), 16-bit samples reduced to their high byte, 1/2/4-bit gray scaled to 0..255. `linear: true` keeps the
    bytes as they are (no sRGB decode), rows are stored bottom-up."""
    import numpy as np
    # 4x2 palette image, 2 bits per index, with transparency for index 1
    pal = [10, 20, 30, 40, 50, 60, 70, 80, 90, 100, 110, 120]
    _write_png(tmp_path / "pal.png", 4, 2, 3, 2, [[0b00011011], [0b11100100]], palette=pal, trns=[255, 7])
    t, px = _loaded_texels(tmp_path, "pal.png")
    assert (t.width, t.height, t.channels) == (4, 2, 4)
    px = px.reshape(2, 4, 4)
    idx = np.array([[3, 2, 1, 0], [0, 1, 2, 3]])  # bottom row of the file first
    want = np.array(pal).reshape(4, 3)[idx]
    np.testing.assert_array_equal(px[..., :3], want)
    np.testing.assert_array_equal(px[..., 3], np.where(idx == 1, 7, 255))

    # 2x1 RGB, 16 bits per sample: the high byte survives
    _write_png(tmp_path / "rgb16.png", 2, 1, 2, 16, [[0x12, 0x34, 0x56, 0x78, 0x9A, 0xBC, 0xFF, 0x00, 0x01, 0xFF, 0x80, 0x7F]])
    t, px = _loaded_texels(tmp_path, "rgb16.png")
    assert (t.width, t.height, t.channels) == (2, 1, 4)
    np.testing.assert_array_equal(px.reshape(2, 4), [[0x12, 0x56, 0x9A, 255], [0xFF, 0x01, 0x80, 255]])

    # 8x1 gray at 1 bit and 2x1 gray at 4 bits
    _write_png(tmp_path / "g1.png", 8, 1, 0, 1, [[0b10110001]])
    t, px = _loaded_texels(tmp_path, "g1.png")
    assert t.channels == 1
    np.testing.assert_array_equal(px, np.array([1, 0, 1, 1, 0, 0, 0, 1]) * 255)
    _write_png(tmp_path / "g4.png", 2, 1, 0, 4, [[0x3C]])
    t, px = _loaded_texels(tmp_path, "g4.png")
    np.testing.assert_array_equal(px, [3 * 17, 12 * 17])


def test_png_reader_rejects_hostile_headers(tmp_path):
    """IHDR dimensions are untrusted: zero, huge and overflowing sizes fail with a message instead of a wrapped allocation."""
    import struct
    import zlib
    from ignis_amd.tables import LoadedScene

    def chunk(typ, body):
        return struct.pack(">I", len(body)) + typ + body + struct.pack(">I", zlib.crc32(typ + body) & 0xFFFFFFFF)
    for w, h in ((0, 4), (0xFFFFFFFF, 0xFFFFFFFF), (70000, 2), (0x80000001, 2)):
        data = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 6, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(b"\0" * 64)) + chunk(b"IEND", b"")
        open(tmp_path / "bad.png", "wb").write(data)
        with pytest.raises(RuntimeError, match="dimensions out of range"):
            LoadedScene.from_string(json.dumps(_texture_scene("bad.png")), base_dir=str(tmp_path))
    # palette index beyond the PLTE chunk
    _write_png(tmp_path / "bad.png", 1, 1, 3, 8, [[5]], palette=[1, 2, 3])
    with pytest.raises(RuntimeError, match="palette index"):
        LoadedScene.from_string(json.dumps(_texture_scene("bad.png")), base_dir=str(tmp_path))


def _ply_bytes(fmt, vertex_props, verts, faces, extra_element=False):
    """A PLY file with arbitrary scalar types. vertex_props: [(type, name)]; verts: rows of values in that order;
    faces: lists of corner indices; list types are (uchar, int) for little endian and (ushort, uint) for big endian."""
    import struct
    codes = {"char": "b", "uchar": "B", "short": "h", "ushort": "H", "int": "i", "uint": "I", "float": "f", "double": "d",
             "int8": "b", "uint8": "B", "int16": "h", "uint16": "H", "int32": "i", "uint32": "I", "float32": "f", "float64": "d"}
    end = {"ascii": None, "binary_little_endian": "<", "binary_big_endian": ">"}[fmt]
    count_t, idx_t = ("uchar", "int") if fmt != "binary_big_endian" else ("ushort", "uint")
    head = f"ply\nformat {fmt} 1.0\ncomment made by the tests\nelement vertex {len(verts)}\n"
    head += "".join(f"property {t} {n}\n" for t, n in vertex_props)
    if extra_element:
        head += "element edge 2\nproperty int a\nproperty int b\nproperty list uchar float weights\n"
    head += f"element face {len(faces)}\nproperty uchar flags\nproperty list {count_t} {idx_t} vertex_indices\nend_header\n"
    body = b""
    if end is None:
        lines = [" ".join(repr(float(v)) if t in ("float", "double", "float32", "float64") else str(int(v)) for (t, _), v in zip(vertex_props, row)) for row in verts]
        if extra_element:
            lines += ["0 1 2 0.5 0.25", "1 2 0"]
        lines += [" ".join(["7", str(len(f))] + [str(i) for i in f]) for f in faces]
        body = ("\n".join(lines) + "\n").encode()
    else:
        for row in verts:
            for (t, _), v in zip(vertex_props, row):
                body += struct.pack(end + codes[t], float(v) if codes[t] in "fd" else int(v))
        if extra_element:
            body += struct.pack(end + "iiBff", 0, 1, 2, 0.5, 0.25) + struct.pack(end + "iiB", 1, 2, 0)
        for f in faces:
            body += struct.pack(end + "B" + codes[count_t], 7, len(f)) + struct.pack(end + codes[idx_t] * len(f), *f)
    return head.encode() + body


def test_ply_reader_is_table_driven(tmp_path):
    """Every scalar type of the format, either byte order, ascii, extra properties / lists / elements in between, polygons:
    all load into the same tables as the plain float file (the reference reads float properties and uchar/uint lists only,
    src/runtime/mesh/PlyFile.cpp:70-290)."""
    import numpy as np
    from ignis_amd.tables import LoadedScene
    verts = [(0, 0, 0, 0.0, 0.0), (1, 0, 0, 1.0, 0.0), (1.5, 1, 0.25, 1.0, 1.0), (0.5, 2, 0, 0.5, 1.0), (-0.5, 1, 0.25, 0.0, 1.0), (3, 0, 0, 0.25, 0.75), (4, 0, 1, 0.5, 0.5), (3, 1, 0, 0.125, 0.0)]
    faces = [[0, 1, 2, 3, 4], [5, 6, 7], [5, 7, 6, 0]]  # pentagon (fan of 3), triangle, quad (fan of 2)
    plain = [("float", "x"), ("float", "y"), ("float", "z"), ("float", "s"), ("float", "t")]
    variants = {
        "plain.ply": _ply_bytes("binary_little_endian", plain, verts, faces),
        "ascii.ply": _ply_bytes("ascii", plain, verts, faces),
        "big.ply": _ply_bytes("binary_big_endian", plain, verts, faces),
        "double.ply": _ply_bytes("binary_little_endian", [("double", "x"), ("float64", "y"), ("double", "z"), ("float32", "u"), ("double", "v")], verts, faces),
        "extra.ply": _ply_bytes("binary_little_endian", [("uchar", "red"), ("float", "x"), ("short", "quality"), ("float", "y"), ("float", "z"), ("float", "s"), ("float", "t"), ("int32", "id")],
                                [(200, v[0], -3, v[1], v[2], v[3], v[4], 99) for v in verts], faces, extra_element=True),
        "extra_ascii.ply": _ply_bytes("ascii", plain, verts, faces, extra_element=True),
    }
    tables = {}
    for name, data in variants.items():
        (tmp_path / name).write_bytes(data)
        s = flat_scene()
        s["shapes"][0] = {"type": "ply", "name": s["shapes"][0]["name"], "filename": name}
        sc = LoadedScene.from_string(json.dumps(s), str(tmp_path))
        tables[name] = (np.ctypeslib.as_array(sc.scene.shape_data, shape=(sc.scene.shape_data_size,)).copy(), np.frombuffer(sc.primbvh_bytes(), np.uint8).copy())
    hdr = np.frombuffer(tables["plain.ply"][0][:16].tobytes(), np.int32)
    assert hdr[0] == 3 + 1 + 2 and hdr[1] == 8  # faces after the fans, vertices
    for name, (shape, bvh) in tables.items():
        np.testing.assert_array_equal(shape, tables["plain.ply"][0], err_msg=name)
        np.testing.assert_array_equal(bvh, tables["plain.ply"][1], err_msg=name)
    # failures are loud
    (tmp_path / "bad.ply").write_bytes(variants["plain.ply"][:-5])
    s = flat_scene()
    s["shapes"][0] = {"type": "ply", "name": s["shapes"][0]["name"], "filename": "bad.ply"}
    with pytest.raises(RuntimeError, match="truncated"):
        LoadedScene.from_string(json.dumps(s), str(tmp_path))
    (tmp_path / "bad.ply").write_bytes(_ply_bytes("ascii", plain, verts, [[0, 1, 99]]))
    with pytest.raises(RuntimeError, match="out of range"):
        LoadedScene.from_string(json.dumps(s), str(tmp_path))
    (tmp_path / "bad.ply").write_bytes(variants["plain.ply"].replace(b"property float x", b"property quaternion x"))
    with pytest.raises(RuntimeError, match="unknown property type"):
        LoadedScene.from_string(json.dumps(s), str(tmp_path))


def _area_light_on_inline_mesh(verts, faces, uvs=None):
    from ignis_amd.tables import LoadedScene
    s = flat_scene()
    shape = {"type": "inline", "name": "L", "vertices": [float(c) for v in verts for c in v], "indices": [int(i) for f in faces for i in f]}
    if uvs is not None:
        shape["texcoords"] = [float(c) for t in uvs for c in t]
    s["shapes"].append(shape)
    s["entities"].append({"name": "L", "shape": "L", "bsdf": "ground"})
    s["lights"] = [{"type": "area", "name": "A", "entity": "L", "radiance": [1, 1, 1]}]
    sc = LoadedScene.from_string(json.dumps(s))
    import numpy as np
    return sc.scene.lights[0].type, np.array(list(sc.scene.lights[0].d), np.float32)


def test_plane_detection_frames_and_rejections():
    """TriMesh::getAsPlane (src/runtime/mesh/TriMesh.cpp:521-620), beyond the two cases of src/tests/units/trimesh_plane.cpp: the
    origin is vertex 0, the axes are its two quad edges (never the diagonal) in the order whose cross product follows the face
    normal — whatever the vertex numbering and the diagonal the quad is split along; folded, non-congruent and five-vertex
    meshes are not planes."""
    import itertools
    import numpy as np
    corners = np.array([[0.5, -1, 2], [2.5, -1, 2.5], [3.0, 1, 3.0], [1.0, 1, 2.5]], np.float32)  # a parallelogram a, b, c, d (in order around)
    normal = np.cross(corners[1] - corners[0], corners[3] - corners[0])
    normal /= np.linalg.norm(normal)
    for perm in itertools.permutations(range(4)):
        v = corners[list(perm)]
        where = {p: i for i, p in enumerate(perm)}  # corner -> vertex number
        for diag in (0, 1):  # split along a-c or along b-d
            tris = [[0, 1, 2], [0, 2, 3]] if diag == 0 else [[0, 1, 3], [1, 2, 3]]
            faces = [[where[c] for c in t] for t in tris]
            kind, d = _area_light_on_inline_mesh(v, faces)
            assert kind == 0, (perm, diag)
            o, x, y, n = d[0:3], d[4:7], d[8:11], np.array([d[3], d[7], d[11]])
            np.testing.assert_allclose(o, v[0], rtol=1e-6)
            o_corner = perm[0]
            want = {tuple(np.round(corners[(o_corner + 1) % 4] - corners[o_corner], 4)), tuple(np.round(corners[(o_corner - 1) % 4] - corners[o_corner], 4))}
            assert {tuple(np.round(x, 4)), tuple(np.round(y, 4))} == want, (perm, diag)
            np.testing.assert_allclose(n, normal, atol=1e-5)
            np.testing.assert_allclose(np.cross(x, y) / np.linalg.norm(np.cross(x, y)), normal, atol=1e-5)
            np.testing.assert_allclose(d[23], np.linalg.norm(np.cross(corners[1] - corners[0], corners[3] - corners[0])), rtol=1e-5)  # area
    quad = [[0, 0, 0], [1, 0, 0], [1, 1, 0], [0, 1, 0]]
    folded = [[0, 0, 0], [1, 0, 0], [1, 1, 0.3], [0, 1, 0]]
    assert _area_light_on_inline_mesh(folded, [[0, 1, 2], [0, 2, 3]])[0] == 8          # the two triangles face differently
    kite = [[0, 0, 0], [1, -1, 0], [3, 0, 0], [1, 1, 0]]
    assert _area_light_on_inline_mesh(kite, [[0, 1, 3], [1, 2, 3]])[0] == 8            # not congruent
    assert _area_light_on_inline_mesh(quad + [[0, 0, 0]], [[0, 1, 2], [4, 2, 3]])[0] == 8  # five vertices (one duplicated): the reference rejects these too
    assert _area_light_on_inline_mesh(quad, [[0, 1, 2]])[0] == 8                       # one face
    # texture coordinates travel with the corners
    uvs = [[0.1, 0.2], [0.9, 0.2], [0.9, 0.8], [0.1, 0.8]]
    kind, d = _area_light_on_inline_mesh(quad, [[0, 1, 2], [0, 2, 3]], uvs)
    assert kind == 0


def test_constant_expressions_in_number_and_colour_properties():
    """The exporters write numbers as expressions (`"roughness": "(0.175)^2"`); constant ones are folded by the loader: PExpr's power
    operator (tighter than a sign, right-associative), Pi / E / Eps, a few scalar functions. Numbers that vary over the surface become entries of the material's number list; the few that cannot are refused."""
    import json
    from conftest import SCENES, flat_scene
    from ignis_amd.tables import LoadedScene

    def material(bsdf):
        s = flat_scene([{"type": "env", "name": "sky", "radiance": [1, 1, 1]}])
        s["bsdfs"] = [dict({"name": "ground"}, **bsdf)]
        sc = LoadedScene.from_string(json.dumps(s), SCENES, 16, 16)
        m = sc.scene.materials[0]
        return [float(x) for x in m.p]

    p = material({"type": "dielectric", "roughness": "(0.17499999701976776)^2", "int_ior": "1 + 0.55"})
    assert p[9] == pytest.approx(0.175 ** 2, rel=1e-6) and p[10] == p[9] and p[1] == pytest.approx(1.55)
    p = material({"type": "diffuse", "reflectance": "color(0.5, 0.25, 1) * -2^2 * -0.25"})  # -(2^2)
    assert p[0:3] == pytest.approx([0.5, 0.25, 1.0])
    p = material({"type": "diffuse", "reflectance": "color(cos(Pi), 2^3^2 / 512, sqrt(max(4, 1)))"})
    assert p[0:3] == pytest.approx([-1.0, 1.0, 2.0], abs=1e-6)
    with pytest.raises(RuntimeError, match="not a constant number"):
        material({"type": "conductor", "roughness": 0.2, "anisotropic": "uv.x * 0.5"})  # (roughness itself may vary: tests/test_number_expressions.py)


def test_meshes_recognised_as_spheres_like_the_reference_unit_test():
    """src/tests/units/trimesh_sphere.cpp: TriMesh::getAsSphere recognises MakeIcoSphere(0, 4, 4) — origin 0, radius 4, area 4 pi r^2 —
    and refuses the ico sphere with 0 subdivisions, MakeUVSphere(0, 4, 4, 2), a triangle and the capped / uncapped cylinder
    MakeCylinder(0, 4, +z, 4, 32, *). Observable through the loader: an area light on a recognised mesh becomes the sphere emitter
    (IG_LIGHT_SPHERE: centre, radius, area), on any other mesh the triangle-mesh emitter (AreaLight.cpp:50-62)."""
    import numpy as np
    from ignis_amd.tables import LoadedScene

    def light_of(shape):
        s = flat_scene([{"type": "area", "name": "l", "entity": "E", "radiance": [1, 1, 1]}])
        s["shapes"].append(dict(shape, name="S"))
        s["entities"].append({"name": "E", "shape": "S", "bsdf": "ground"})
        sc = LoadedScene.from_string(json.dumps(s))
        assert sc.scene.light_count == 1
        return sc.scene.lights[0].type, np.float64(list(sc.scene.lights[0].d))

    IG_LIGHT_PLANE, IG_LIGHT_MESH_AREA, IG_LIGHT_SPHERE = 0, 8, 9
    t, d = light_of({"type": "icosphere", "radius": 4, "subdivisions": 4})
    assert t == IG_LIGHT_SPHERE
    np.testing.assert_allclose(d[0:3], 0, atol=1e-6)
    assert d[3] == pytest.approx(4, rel=1e-6) and d[7] == pytest.approx(4 * np.pi * 16, rel=1e-5)
    for shape in ({"type": "icosphere", "radius": 4, "subdivisions": 0}, {"type": "uvsphere", "radius": 4, "stacks": 4, "slices": 2},
                  {"type": "cylinder", "radius": 4, "p0": [0, 0, 0], "p1": [0, 0, 4], "sections": 32, "filled": True},
                  {"type": "cylinder", "radius": 4, "p0": [0, 0, 0], "p1": [0, 0, 4], "sections": 32, "filled": False}):
        assert light_of(shape)[0] == IG_LIGHT_MESH_AREA, shape
    assert light_of({"type": "triangle", "p0": [0, 0, 0], "p1": [1, 0, 0], "p2": [0, 1, 0]})[0] in (IG_LIGHT_PLANE, IG_LIGHT_MESH_AREA)  # (a plane shape at most: never a sphere)


def test_half_edge_structure_like_the_reference_unit_test():
    """src/tests/units/trimesh_he.cpp: on MakeIcoSphere(0, 4, 4) every half edge has a twin, the twin of the twin is the edge itself, next
    and previous stay in the face (by construction here: edge 3 f + k), and the twin of the previous half edge starts at the edge's own vertex;
    a lone triangle has no twins at all. The pairing is what TriMesh::getAsSphere walks (csrc/host/mesh.cpp)."""
    import ctypes as C
    from ignis_amd.tables import host_lib
    l = host_lib()
    l.igh_test_mesh_edges.restype = C.c_int32
    l.igh_test_mesh_edges.argtypes = [C.c_int32, C.c_float, C.c_uint32, C.POINTER(C.c_uint64)]
    out = (C.c_uint64 * 6)()
    assert l.igh_test_mesh_edges(0, 4.0, 4, out) == 0
    faces, edges, distinct, with_twin, mutual, around = (int(v) for v in out)
    assert faces == 20 * 4 ** 4 and edges == 3 * faces and distinct == edges  # a closed, consistently oriented surface: no directed edge twice
    assert edges < faces * 3 * 2                                              # (the reference's first REQUIRE)
    assert with_twin == edges and mutual == edges and around == edges
    assert l.igh_test_mesh_edges(1, 0.0, 0, out) == 0
    assert [int(v) for v in out] == [1, 3, 3, 0, 0, 0]
