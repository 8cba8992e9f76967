"""Floating-point image files either side of the hot path: the loader's OpenEXR / Radiance HDR readers
(ignis_amd/csrc/host/floatimage.h; the reference reads them through tinyexr and stb_image, src/runtime/Image.cpp:497-712) and
float textures on the device (driver/image.art:1-7).

  * every reference-held EXR (PIZ and ZIP, HALF and FLOAT) decodes bit for bit like the independent Python decoder of
    tests/golden/exr_decode.py;
  * files written here with the encoders below (NONE / RLE / ZIPS / ZIP; HALF / FLOAT / UINT; gray; odd sizes) round-trip;
  * RGBE: flat and run-length encoded scanlines against mantissa * 2^(e - 136);
  * a float environment map renders like the oracle (`-m gpu`).
"""
import json
import os
import struct
import zlib
from pathlib import Path

import numpy as np
import pytest

from conftest import GOLDEN, SCENES, flat_scene


def _read(path):
    from ignis_amd.tables import read_float_image
    return read_float_image(path)


def test_reference_held_exr_files_decode_like_the_python_decoder():
    import sys
    sys.path.insert(0, GOLDEN)
    import exr_decode
    files = sorted(Path(GOLDEN, "references").glob("*.exr")) + [Path(GOLDEN, "constant.exr")]
    assert len(files) >= 50
    methods = set()
    for p in files:
        got = _read(p)
        ref = exr_decode.read_exr(str(p))
        methods.add(exr_decode.parse_header(open(p, "rb").read())[0]["compression"])
        want = np.stack([ref["R"], ref["G"], ref["B"], ref.get("A", np.ones_like(ref["R"]))], -1)
        assert got.shape == want.shape, p
        assert np.array_equal(got, want, equal_nan=True), p
    assert {3, 4} <= methods  # ZIP and PIZ are both present among the references


# ---- a small OpenEXR writer (test infrastructure): scanline, any of NONE / RLE / ZIPS / ZIP

def _predict_and_split(raw):
    a = np.frombuffer(raw, np.uint8)
    t = np.concatenate([a[0::2], a[1::2]]).astype(np.int64)
    d = t.copy()
    d[1:] = (t[1:] - t[:-1] + 128 + 256) & 255
    return d.astype(np.uint8).tobytes()


def _rle(b):
    out, i, n = bytearray(), 0, len(b)
    while i < n:
        j = i + 1
        while j < n and b[j] == b[i] and j - i < 128:
            j += 1
        if j - i >= 3:
            out += struct.pack("bB", j - i - 1, b[i])
            i = j
            continue
        k = i
        while k < n and k - i < 127 and not (k + 2 < n and b[k] == b[k + 1] == b[k + 2]):
            k += 1
        out += struct.pack("b", -(k - i)) + bytes(b[i:k])
        i = k
    return bytes(out)


def _write_exr(path, channels, compression, width, height, y0=0):
    """channels: list of (name, pixel type 0/1/2, array [height][width]) sorted by name."""
    lines = {0: 1, 1: 1, 2: 1, 3: 16}[compression]
    head = struct.pack("<II", 20000630, 2)

    def attr(name, typ, data):
        return name.encode() + b"\0" + typ.encode() + b"\0" + struct.pack("<i", len(data)) + data
    chl = b"".join(n.encode() + b"\0" + struct.pack("<iB3xii", t, 0, 1, 1) for n, t, _ in channels) + b"\0"
    head += attr("channels", "chlist", chl) + attr("compression", "compression", bytes([compression]))
    head += attr("dataWindow", "box2i", struct.pack("<4i", 0, y0, width - 1, y0 + height - 1))
    head += attr("displayWindow", "box2i", struct.pack("<4i", 0, y0, width - 1, y0 + height - 1))
    head += attr("lineOrder", "lineOrder", b"\0") + attr("pixelAspectRatio", "float", struct.pack("<f", 1))
    head += attr("screenWindowCenter", "v2f", struct.pack("<ff", 0, 0)) + attr("screenWindowWidth", "float", struct.pack("<f", 1)) + b"\0"
    dt = {0: "<u4", 1: "<f2", 2: "<f4"}
    chunks = []
    for y in range(0, height, lines):
        raw = b"".join(np.asarray(a[r], dt[t]).tobytes() for r in range(y, min(y + lines, height)) for _, t, a in channels)
        if compression == 0:
            data = raw
        elif compression == 1:
            data = _rle(_predict_and_split(raw))
        else:
            data = zlib.compress(_predict_and_split(raw))
        if len(data) >= len(raw):
            data = raw  # writers store the chunk as is when compression does not help
        chunks.append(struct.pack("<ii", y0 + y, len(data)) + data)
    table_at = len(head)
    off, table = table_at + 8 * len(chunks), b""
    for c in chunks:
        table += struct.pack("<Q", off)
        off += len(c)
    open(path, "wb").write(head + table + b"".join(chunks))


@pytest.mark.parametrize("compression", [0, 1, 2, 3], ids=["none", "rle", "zips", "zip"])
@pytest.mark.parametrize("size", [(37, 21), (64, 33), (1, 1)])
def test_exr_round_trip(tmp_path, compression, size):
    w, h = size
    rng = np.random.default_rng(compression * 10 + w)
    r = rng.random((h, w)).astype(np.float32) * 50
    g = np.round(rng.random((h, w)) * 8) / 8  # runs of equal bytes for the RLE path
    b = rng.random((h, w)).astype(np.float16)
    a = (rng.random((h, w)) * 1000).astype(np.uint32)
    p = str(tmp_path / "t.exr")
    _write_exr(p, [("A", 0, a), ("B", 1, b), ("G", 2, g.astype(np.float32)), ("R", 2, r)], compression, w, h, y0=-3)
    got = _read(p)
    assert got.shape == (h, w, 4)
    np.testing.assert_array_equal(got[..., 0], r)
    np.testing.assert_array_equal(got[..., 1], g.astype(np.float32))
    np.testing.assert_array_equal(got[..., 2], b.astype(np.float32))
    np.testing.assert_array_equal(got[..., 3], a.astype(np.float32))
    # a lone luminance channel stays one channel; RGB without alpha gets alpha 1 (Image.cpp:617-646)
    _write_exr(p, [("Y", 1, b)], compression, w, h)
    gray = _read(p)
    assert gray.shape == (h, w, 1)
    np.testing.assert_array_equal(gray[..., 0], b.astype(np.float32))
    _write_exr(p, [("B", 2, r), ("G", 2, r), ("R", 1, b)], compression, w, h)
    rgb = _read(p)
    assert np.all(rgb[..., 3] == 1)
    np.testing.assert_array_equal(rgb[..., 0], b.astype(np.float32))


def test_half_conversion_covers_every_bit_pattern(tmp_path):
    """All 65536 half values (zeros, subnormals, infinities, NaNs) convert exactly like numpy's float16 -> float32."""
    bits = np.arange(65536, dtype=np.uint16).reshape(256, 256)
    p = str(tmp_path / "h.exr")
    _write_exr(p, [("Y", 1, bits.view(np.float16))], 0, 256, 256)
    got = _read(p)[..., 0]
    want = bits.view(np.float16).astype(np.float32)
    assert np.array_equal(got.view(np.uint32) & 0xFF800000, want.view(np.uint32) & 0xFF800000)  # sign + exponent, NaNs included
    assert np.array_equal(got, want, equal_nan=True)


def test_exr_errors_are_reported(tmp_path):
    p = tmp_path / "bad.exr"
    p.write_bytes(b"not an exr file at all")
    with pytest.raises(RuntimeError, match="not an OpenEXR"):
        _read(p)
    good = tmp_path / "g.exr"
    _write_exr(str(good), [("Y", 2, np.ones((4, 4), np.float32))], 3, 4, 4)
    data = bytearray(good.read_bytes())
    (tmp_path / "cut.exr").write_bytes(data[:len(data) - 5])
    with pytest.raises(RuntimeError):
        _read(tmp_path / "cut.exr")
    with pytest.raises(RuntimeError, match="cannot open"):
        _read(tmp_path / "missing.exr")


def test_corrupt_files_are_refused_not_crashed_on(tmp_path):
    """Byte flips, truncations and damaged headers of a PIZ and a ZIP file (the sanitizer run of tools/fuzz_float_images.cpp, here
    without the sanitizers): the reader returns an image or raises, e.g. for a Huffman table that is not a prefix code."""
    rng = np.random.default_rng(11)
    for name in ("ref-cbox-d1-4096.exr", "ref-cycles-mix-trans-trans-4096.exr"):
        base = np.frombuffer(Path(GOLDEN, "references", name).read_bytes(), np.uint8)
        raised = 0
        for it in range(120):
            b = base.copy()
            if it % 3 == 0:
                b[rng.integers(0, b.size, 6)] = rng.integers(0, 256, 6)
            elif it % 3 == 1:
                b = b[:rng.integers(1, b.size)]
            else:
                idx = 320 + rng.integers(0, b.size - 320, 16)
                b[idx] ^= (1 << rng.integers(0, 8, 16)).astype(np.uint8)
            p = tmp_path / "m.exr"
            p.write_bytes(b.tobytes())
            try:
                img = _read(p)
                assert img.ndim == 3
            except RuntimeError:
                raised += 1
        assert raised > 20


# ---- Radiance RGBE

def _write_hdr(path, rgbe, rle):
    h, w, _ = rgbe.shape
    out = bytearray(b"#?RADIANCE\nSOFTWARE=test\nFORMAT=32-bit_rle_rgbe\n\n" + f"-Y {h} +X {w}\n".encode())
    for y in range(h):
        if not rle:
            out += rgbe[y].tobytes()
            continue
        out += bytes([2, 2, w >> 8, w & 255])
        for c in range(4):
            row, x = rgbe[y, :, c], 0
            while x < w:
                run = 1
                while x + run < w and run < 127 and row[x + run] == row[x]:
                    run += 1
                if run >= 3:
                    out += bytes([128 + run, int(row[x])])
                    x += run
                else:
                    k = x
                    while k < w and k - x < 128 and not (k + 2 < w and row[k] == row[k + 1] == row[k + 2]):
                        k += 1
                    out += bytes([k - x]) + row[x:k].tobytes()
                    x = k
    open(path, "wb").write(out)


@pytest.mark.parametrize("rle", [False, True], ids=["flat", "rle"])
def test_radiance_hdr(tmp_path, rle):
    rng = np.random.default_rng(5)
    w, h = 40, 9
    rgbe = rng.integers(0, 256, (h, w, 4), dtype=np.uint8)
    rgbe[:, :, 3] = rng.integers(120, 140, (h, w))
    rgbe[2, 5:30] = (10, 20, 30, 129)  # a run
    rgbe[3, 3] = (200, 100, 50, 0)     # zero exponent: black
    p = str(tmp_path / "t.hdr")
    _write_hdr(p, rgbe, rle)
    got = _read(p)
    assert got.shape == (h, w, 4)
    scale = np.ldexp(np.float32(1), rgbe[..., 3].astype(np.int32) - 136).astype(np.float32)
    want = rgbe[..., :3].astype(np.float32) * scale[..., None]
    want[rgbe[..., 3] == 0] = 0
    np.testing.assert_array_equal(got[..., :3], want)
    assert np.all(got[..., 3] == 1)


def test_reference_constant_environment_files():
    """scenes/textures/environment/constant.{exr,hdr} of the reference (copied as data): 60 x 40, every texel white."""
    for name in ("constant.exr", "constant.hdr"):
        img = _read(os.path.join(GOLDEN, name))
        assert img.shape == (40, 60, 4) and np.all(img == 1)


# ---- float textures through the loader, the oracle and the device

def _env_scene(tmp_path, img, light=None, size=(48, 32), max_depth=2):
    """A float environment map (rows top to bottom) over the integrator test plane."""
    h, w, _ = img.shape
    z = np.zeros((h, w), np.float32)
    _write_exr(str(tmp_path / "env.exr"), [("B", 2, img[..., 2]), ("G", 1, img[..., 1]), ("R", 2, img[..., 0])], 3, w, h)
    s = flat_scene([dict({"type": "env", "name": "sky", "radiance": "envtex"}, **(light or {}))], max_depth=max_depth, size=size)
    s["textures"] = [{"type": "image", "name": "envtex", "filename": "env.exr", "filter_type": "bilinear"}]
    from ignis_amd.tables import LoadedScene
    return LoadedScene.from_string(json.dumps(s), str(tmp_path), *size)


def test_float_texture_is_stored_unconverted_and_flipped(tmp_path):
    img = np.zeros((4, 6, 3), np.float32)
    img[0, :, 0] = 1000.5  # top row
    img[3, :, 1] = 0.25    # bottom row
    sc = _env_scene(tmp_path, img)
    t = sc.scene.textures[0]
    assert (t.width, t.height) == (6, 4) and t.channels == 0x104
    raw = np.ctypeslib.as_array(sc.scene.texture_data, shape=(sc.scene.texture_data_size,))
    tex = np.frombuffer(raw[t.offset:t.offset + 6 * 4 * 16].tobytes(), np.float32).reshape(4, 6, 4)
    assert np.all(tex[0, :, 1] == 0.25) and np.all(tex[3, :, 0] == 1000.5) and np.all(tex[..., 3] == 1)  # row 0 = bottom (Image::flipY)


def test_constant_float_environment_equals_constant_colour(tmp_path):
    """A one-colour EXR environment lights the plane like the same constant radiance (values above 1 survive: not packed)."""
    import oracle
    img = np.empty((8, 16, 3), np.float32)
    img[...] = (3.5, 2.0, 0.5)
    sc = _env_scene(tmp_path, img, size=(32, 32))
    fb, _ = oracle.render(sc, 64, 32, 32, seed=3)
    s = flat_scene([{"type": "env", "name": "sky", "radiance": [3.5, 2.0, 0.5]}], size=(32, 32))
    from ignis_amd.tables import LoadedScene
    ref, _ = oracle.render(LoadedScene.from_string(json.dumps(s), SCENES, 32, 32), 64, 32, 32, seed=3)
    # the white plane fills the view: under a constant environment it reflects exactly that radiance
    lit, lit_ref = fb.mean(axis=(0, 1)), ref.mean(axis=(0, 1))
    np.testing.assert_allclose(lit, lit_ref, rtol=0.02)
    np.testing.assert_allclose(lit, (3.5, 2.0, 0.5), rtol=0.02)


def test_environment_without_cdf_is_the_uniformly_sampled_function_environment(tmp_path):
    """"cdf": "none" (EnvironmentLight.cpp:60,89-96): make_environment_light — uniform directions, scale * texture. With a
    one-colour float map this is the constant environment light, random number for random number."""
    import oracle
    from ignis_amd.tables import LoadedScene
    img = np.empty((8, 16, 3), np.float32)
    img[...] = (3.5, 2.0, 0.5)
    sc = _env_scene(tmp_path, img, light={"cdf": "None", "scale": [0.5, 2, 1]}, size=(32, 32), max_depth=3)
    assert tuple(np.array(list(sc.scene.lights[0].d)[12:16], np.float32).view(np.uint32)[2:]) == (0, 0)
    fb, st = oracle.render(sc, 8, 32, 32, seed=3)
    s = flat_scene([{"type": "env", "name": "sky", "radiance": [1.75, 4.0, 0.5]}], max_depth=3, size=(32, 32))
    ref, st_ref = oracle.render(LoadedScene.from_string(json.dumps(s), SCENES, 32, 32), 8, 32, 32, seed=3)
    assert st == st_ref
    np.testing.assert_array_equal(fb, ref)


@pytest.mark.gpu
def test_environment_without_cdf_vs_oracle(gpu_device, tmp_path):
    import oracle
    rng = np.random.default_rng(8)
    img = (rng.random((16, 32, 3)) * 2).astype(np.float32)
    sc = _env_scene(tmp_path, img, light={"cdf": "none", "scale": [1, 0.5, 2], "transform": [{"rotate": [10, 20, 30]}]}, size=(64, 48), max_depth=3)
    gpu_device.assign_scene(sc)
    gpu_device.resize(64, 48)
    gpu_device.clear_framebuffer()
    gpu_device.render(8, 64, 48, iteration=0, seed=4)
    ref, _ = oracle.render(sc, 8, 64, 48, iteration=0, seed=4)
    fb = gpu_device.framebuffer()
    assert np.linalg.norm(fb - ref) / np.linalg.norm(ref) <= 1e-4


@pytest.mark.gpu
@pytest.mark.parametrize("gray", [False, True])
def test_float_environment_map_vs_oracle(gpu_device, tmp_path, gray):
    """An HDR environment with bright spots (radiance 5000, far beyond what 8 bits hold), CDF-sampled: HIP == oracle."""
    import oracle
    rng = np.random.default_rng(2)
    img = (rng.random((32, 64, 3)) * 0.5).astype(np.float32)
    for r, c in ((6, 40), (24, 10), (15, 25), (20, 55)):
        img[r:r + 3, c:c + 4] = (5000, 4000, 3000)
    if gray:
        from ignis_amd.tables import LoadedScene
        _write_exr(str(tmp_path / "env.exr"), [("Y", 1, img[..., 0].astype(np.float16))], 2, 64, 32)
        s = flat_scene([{"type": "env", "name": "sky", "radiance": "envtex"}], max_depth=3, size=(64, 48))
        s["textures"] = [{"type": "image", "name": "envtex", "filename": "env.exr"}]
        sc = LoadedScene.from_string(json.dumps(s), str(tmp_path), 64, 48)
        assert sc.scene.textures[0].channels == 0x101
    else:
        sc = _env_scene(tmp_path, img, size=(64, 48), max_depth=3)
    gpu_device.assign_scene(sc)
    gpu_device.resize(64, 48)
    gpu_device.clear_framebuffer()
    ref = np.zeros((48, 64, 3), np.float32)
    for it in range(2):
        gpu_device.render(8, 64, 48, iteration=it, seed=4)
        oracle.render(sc, 8, 64, 48, iteration=it, seed=4, fb=ref)
    fb = gpu_device.framebuffer()
    assert fb.mean() > 20  # the spots light the plane: far more than 8-bit texels could carry
    assert np.linalg.norm(fb - ref) / np.linalg.norm(ref) <= 1e-4
