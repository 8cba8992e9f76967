"""The loader's JPEG reader (ignis_amd/csrc/host/jpeg.h, written from ITU-T T.81; the reference reads JPEG textures through
stb_image) against libjpeg as Pillow drives it: baseline and progressive files, 4:4:4 / 4:2:2 / 4:2:0 sampling, optimised Huffman
tables, restart intervals, gray images, sizes that are not multiples of the MCU. Samples may differ by a few levels (floating-point
inverse DCT here, fixed-point there; the chroma filters agree)."""
import json
import os

import numpy as np
import pytest

from conftest import SCENES, flat_scene

Image = pytest.importorskip("PIL.Image")


def _picture(h, w):
    yy, xx = np.mgrid[0:h, 0:w]
    return np.stack([127 + 100 * np.sin(xx / 9.0) * np.cos(yy / 7.0), 127 + 90 * np.cos(xx / 5.0 + yy / 11.0), (xx * 2 + yy) % 256], -1).clip(0, 255).astype(np.uint8)


@pytest.mark.parametrize("kw", [
    dict(subsampling=0), dict(subsampling=1), dict(subsampling=2), dict(subsampling=0, progressive=True), dict(subsampling=2, progressive=True),
    dict(subsampling=2, optimize=True), dict(subsampling=2, progressive=True, quality=30), dict(subsampling=0, quality=100),
    dict(subsampling=2, restart_marker_blocks=3), dict(subsampling=1, progressive=True, restart_marker_rows=1),
], ids=["444", "422", "420", "progressive-444", "progressive-420", "optimised", "progressive-q30", "q100", "restart-blocks", "progressive-restart-rows"])
@pytest.mark.parametrize("size", [(77, 101), (8, 8), (33, 16)])
def test_jpeg_reader_agrees_with_libjpeg(tmp_path, kw, size):
    from ignis_amd.tables import read_image8
    kw = dict(kw)
    img = _picture(*size)
    p = str(tmp_path / "t.jpg")
    Image.fromarray(img).save(p, quality=kw.pop("quality", 90), **kw)
    ref = np.asarray(Image.open(p).convert("RGB")).astype(int)
    got = read_image8(p)
    assert got.shape == ref.shape
    d = np.abs(got.astype(int) - ref)
    assert d.max() <= 4 and d.mean() < 0.5


def test_gray_jpeg_and_errors(tmp_path):
    from ignis_amd.tables import read_image8
    img = _picture(40, 56)[..., 0]
    p = str(tmp_path / "g.jpg")
    Image.fromarray(img).save(p, quality=85)
    got = read_image8(p)
    assert got.shape == (40, 56, 1)
    assert np.abs(got[..., 0].astype(int) - np.asarray(Image.open(p)).astype(int)).max() <= 2
    Image.fromarray(_picture(16, 16)).convert("CMYK").save(str(tmp_path / "c.jpg"))
    with pytest.raises(RuntimeError, match="gray and YCbCr|CMYK"):
        read_image8(str(tmp_path / "c.jpg"))
    (tmp_path / "bad.jpg").write_bytes(b"\xff\xd8\xff\xe0 nothing")
    with pytest.raises(RuntimeError):
        read_image8(str(tmp_path / "bad.jpg"))
    data = open(p, "rb").read()
    rng = np.random.default_rng(4)
    for it in range(150):  # damaged files are refused or decoded, never crashed on (tools/fuzz_* run the readers under sanitizers)
        b = bytearray(data)
        if it % 2:
            b = b[:rng.integers(2, len(b))]
        else:
            for k in rng.integers(0, len(b), 4):
                b[k] = int(rng.integers(0, 256))
        (tmp_path / "m.jpg").write_bytes(bytes(b))
        try:
            read_image8(str(tmp_path / "m.jpg"))
        except RuntimeError:
            pass


def test_reference_jpeg_texture_decodes_like_libjpeg():
    """scenes/textures/boats.jpg (the texture of the reference's cycles-tex evaluation scene): 2415 x 2415, progressive."""
    from ignis_amd.tables import read_image8
    p = os.path.join(SCENES, "textures", "boats.jpg")
    got = read_image8(p)
    ref = np.asarray(Image.open(p).convert("RGB")).astype(int)
    assert got.shape == (2415, 2415, 3)
    d = np.abs(got.astype(int) - ref)
    assert d.max() <= 4 and d.mean() < 0.1


def test_jpeg_texture_goes_through_the_packed_path(tmp_path):
    """A JPEG reflectance texture is packed like a PNG one (sRGB -> linear 8 bit, rows bottom to top, alpha 255): the same picture
    as PNG gives the same texels up to the decoder's few levels."""
    from ignis_amd.tables import LoadedScene
    pic = _picture(24, 32)
    Image.fromarray(pic).save(str(tmp_path / "t.jpg"), quality=95, subsampling=0)
    Image.fromarray(np.asarray(Image.open(str(tmp_path / "t.jpg")).convert("RGB"))).save(str(tmp_path / "t.png"))
    texels = {}
    for ext in ("jpg", "png"):
        s = flat_scene([{"type": "env", "name": "sky", "radiance": [1, 1, 1]}])
        s["textures"] = [{"type": "image", "name": "tex", "filename": f"t.{ext}"}]
        s["bsdfs"] = [{"type": "diffuse", "name": "ground", "reflectance": "tex"}]
        sc = LoadedScene.from_string(json.dumps(s), str(tmp_path), 16, 16)
        t = sc.scene.textures[0]
        assert (t.width, t.height, t.channels) == (32, 24, 4)
        raw = np.ctypeslib.as_array(sc.scene.texture_data, shape=(sc.scene.texture_data_size,))
        texels[ext] = raw[t.offset:t.offset + 32 * 24 * 4].reshape(24, 32, 4).astype(int)
    assert np.all(texels["jpg"][..., 3] == 255)
    assert np.abs(texels["jpg"] - texels["png"]).max() <= 4
