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
    assert lib.igd_get_abi_version() == 1


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
    bad["bsdfs"][0] = {"type": "principled", "name": "ground"}
    with pytest.raises(RuntimeError, match="not supported"):
        LoadedScene.from_string(json.dumps(bad))
    bad = flat_scene()
    bad["bsdfs"][0]["reflectance"] = "some_texture"
    with pytest.raises(RuntimeError, match="not a constant colour"):
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
    sc = LoadedScene.from_file(os.path.join(ROOT, "scenes", "many_point_lights_hip.json"), 64, 64)
    s = sc.scene
    assert s.texture_count == 1
    t = s.textures[0]
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
