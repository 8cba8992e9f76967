"""The "sky" light (src/runtime/light/SkyLight.cpp, skysun/SkyModel.cpp): the Hosek-Wilkie model restated in ignis_amd/csrc/host/hosek.h
against vectors produced by the sample implementation the reference ships (tests/golden/make_hosek_golden.py compiles it from
/root/reference through oracle/Makefile), and the image / CDF the loader turns it into."""
import ctypes as C
import os

import numpy as np

from conftest import ROOT, SCENES
from ignis_amd import LoadedScene, tables


def _sky(channel, turbidity, albedo, elevation, theta, gamma):
    l = tables.host_lib()
    l.igh_eval_sky.restype = C.c_double
    l.igh_eval_sky.argtypes = [C.c_int32] + [C.c_double] * 5
    return l.igh_eval_sky(int(channel), float(turbidity), float(albedo), float(elevation), float(theta), float(gamma))


def test_hosek_model_matches_the_reference_implementation_bit_for_bit():
    g = np.load(os.path.join(ROOT, "tests", "golden", "hosek_golden.npz"))
    got = np.array([_sky(c, t, a, e, th, ga) for c, t, a, e, th, ga in
                    zip(g["channel"], g["turbidity"], g["albedo"], g["elevation"], g["theta"], g["gamma"])])
    assert np.array_equal(got, g["radiance"])  # double precision, same operations in the same order
    assert _sky(0, 0.5, 0.5, 1.0, 0.3, 0.3) == 0 and _sky(3, 3, 0.5, 1.0, 0.3, 0.3) == 0  # outside the fitted range: refused


def test_sky_light_image_and_cdf_follow_skymodel_cpp():
    """scenes/many_point_lights.json names {"type": "sky"} with every default: sun position from the default date and place, ground
    0.8, turbidity 3. The loader's texture must be SkyModel::SkyModel's image (row = angle from the zenith, column = azimuth
    shifted by pi / 4, /106.856980, clamped at 0) with the rows flipped by the EXR round trip, alpha 1; the light is a textured
    environment light over it with the CDF of the image itself."""
    sc = LoadedScene.from_file(os.path.join(SCENES, "many_point_lights.json"), 64, 64)
    s = sc.scene
    assert s.infinite_light_count == 1 and s.lights[0].type == 5  # IG_LIGHT_ENV_TEXTURED
    ints = np.frombuffer(bytes(bytearray(np.ctypeslib.as_array(s.lights[0].d)[12:16].view(np.uint8))), np.uint32)
    tex_id, cdf_off, w, h = (int(v) for v in ints)
    assert (w, h) == (512, 256)
    t = s.textures[tex_id]
    assert (t.width, t.height, t.channels, t.filter, t.wrap_u) == (512, 256, 0x104, 1, 0)
    addr = C.cast(s.texture_data, C.c_void_p).value + t.offset
    img = np.frombuffer((C.c_float * (w * h * 4)).from_address(addr), np.float32).reshape(h, w, 4).copy()
    assert (img[..., 3] == 1).all() and (img[..., :3] >= 0).all()

    # default sun: 6 May 2020 12:00, Saarbruecken (LoaderUtils::getTimePoint / getLocation), through the loader's own PSA restatement
    sun = LoadedScene.from_string('{"technique":{"type":"path"},"camera":{"type":"perspective"},"film":{"size":[8,8]},"bsdfs":[],"shapes":[],'
                                  '"entities":[],"lights":[{"type":"sun","name":"s"}]}', "", 8, 8).scene.lights[0]
    d = np.array(list(sun.d)[:3], np.float64)
    elevation, azimuth = np.pi / 2 - np.arccos(d[1]), np.arctan2(-d[0], -d[2]) % (2 * np.pi)
    zen = np.float32(np.pi / 2) - np.float32(elevation)  # what SkyModel.cpp calls solar_elevation
    rng = np.random.default_rng(4)
    for y, x in zip(rng.integers(0, h, 24), rng.integers(0, w, 24)):
        theta = np.float32(np.pi / 2) * np.float32(y) / np.float32(h)
        az = np.float32(2 * np.pi) * np.float32(x) / np.float32(w) - np.float32(np.pi / 4)
        if az < 0:
            az += np.float32(2 * np.pi)
        cg = np.cos(theta) * np.cos(zen) + np.sin(theta) * np.sin(zen) * np.cos(az - np.float32(azimuth))
        gamma = np.arccos(np.clip(cg, -1, 1))
        want = [max(0.0, _sky(k, 3.0, 0.8, zen, theta, gamma) / 106.856980) for k in range(3)]
        np.testing.assert_allclose(img[h - 1 - y, x, :3], want, rtol=2e-4, atol=1e-6)
    # zenith bluer and darker than the horizon
    assert img[-1, :, 2].mean() > 2 * img[-1, :, 0].mean() and img[0, :, :3].mean() > 2 * img[-1, :, :3].mean()

    # CDF::computeForImage over the image as loaded (sin-weighted rows, no compensation): marginal + conditional, both ending at 1
    cdf = np.ctypeslib.as_array(s.cdf_data, shape=(s.cdf_data_count,))[cdf_off:cdf_off + h + w * h].copy()
    marginal, cond = cdf[:h], cdf[h:].reshape(h, w)
    resp = img[..., :3].astype(np.float32).sum(axis=2) / 3
    rows = resp.astype(np.float64).sum(axis=1) * np.sin(np.pi * (np.arange(h) + 0.5) / h)
    np.testing.assert_allclose(marginal, np.cumsum(rows) / rows.sum(), rtol=2e-3, atol=1e-5)
    np.testing.assert_allclose(cond[37], np.cumsum(resp[37].astype(np.float64)) / resp[37].sum(), rtol=2e-3, atol=1e-5)
    assert marginal[-1] == 1 and (cond[:, -1] == 1).all()
