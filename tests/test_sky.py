"""The "sky" light (src/runtime/light/SkyLight.cpp, skysun/SkyModel.cpp): the Hosek-Wilkie model restated in ignis_amd/csrc/host/hosek.h
against vectors produced by the sample implementation the reference ships (tests/golden/make_hosek_golden.py compiles it from
/root/reference through oracle/Makefile), and the image / CDF the loader turns it into."""
import ctypes as C
import os

import numpy as np
import pytest

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


# ---- reference-held known answers of the sky / sun helpers (src/tests/units/{sun,perez,elevation_azimuth}.cpp), at the reference's own tolerances

def _sky_light(**kw):
    """The loader's record of a function sky whose sun comes from LoaderUtils::getEA: d[9:12] = ElevationAzimuth::toDirectionYUp."""
    import json
    s = {"technique": {"type": "path", "max_depth": 2},
         "camera": {"type": "perspective", "fov": 90, "transform": [1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, -1]}, "film": {"size": [16, 16]},
         "bsdfs": [{"type": "diffuse", "name": "g"}], "shapes": [{"type": "rectangle", "name": "r"}],
         "entities": [{"name": "r", "shape": "r", "bsdf": "g"}], "lights": [dict({"type": "cie_clear", "name": "sky"}, **kw)]}
    sc = LoadedScene.from_string(json.dumps(s))
    return sc, np.float64(list(sc.scene.lights[0].d))


def _ea_of_yup(d):
    """ElevationAzimuth::fromDirectionYUp (skysun/ElevationAzimuth.h:16-21)"""
    phi = np.arctan2(-d[0], -d[2])
    return np.pi / 2 - np.arccos(d[1]), phi + 2 * np.pi if phi < 0 else phi


def test_sun_position_known_answer_of_the_reference_unit_test():
    """src/tests/units/sun.cpp:8-37: computeSunEA(2022-11-18 13:00:00, 49.235422 N, -6.9965744 (degrees west), timezone -1) = elevation
    20.86 deg, azimuth 10.81 deg west of south (relative 1e-2), (20.8, 10.8) within 1e-3 rad, and the Z-up direction
    (-0.175382, -0.918072, 0.355506) of the Radiance cross-check (relative 1e-3). The loader stores the Y-up direction (x, z, y swapped)."""
    _, d = _sky_light(year=2022, month=11, day=18, hour=13, minute=0, seconds=0, latitude=49.235422, longitude=-6.9965744, timezone=-1)
    yup = d[9:12]
    el, az = _ea_of_yup(yup)
    assert el == pytest.approx(np.radians(20.86), rel=1e-2) and az == pytest.approx(np.radians(10.81), rel=1e-2)
    assert el == pytest.approx(np.radians(20.8), abs=1e-3) and az == pytest.approx(np.radians(10.8), abs=1e-3)
    zup = np.array([yup[0], yup[2], yup[1]])  # toDirectionZUp = (-cosE sinA, -cosE cosA, sinE), toDirectionYUp = (-cosE sinA, sinE, -cosE cosA)
    np.testing.assert_allclose(zup, [-0.175382, -0.918072, 0.355506], rtol=1e-3)


def test_elevation_azimuth_mappings_of_the_reference_unit_test():
    """src/tests/units/elevation_azimuth.cpp: the nominal directions ([90 deg, 0] -> +up, [0, 0] -> south, [0, 90 deg] -> west,
    [0, 180 deg] -> north, [0, 270 deg] -> east; absolute 1e-4), the Radiance cross-check (12.5 deg, 39.1 deg) ->
    (-0.615494, -0.757725, 0.216842) Z-up (relative 1e-2), and direction -> (elevation, azimuth) -> direction round trips through the
    `direction` property (fromDirectionYUp then toDirectionYUp), (0, -pi) coming back as azimuth +pi."""
    def zup(**kw):
        y = _sky_light(**kw)[1][9:12]
        return np.array([y[0], y[2], y[1]])
    for (el, az), exp in (((90, 0), (0, 0, 1)), ((0, 0), (0, -1, 0)), ((0, 90), (-1, 0, 0)), ((0, 180), (0, 1, 0)), ((0, 270), (1, 0, 0))):
        np.testing.assert_allclose(zup(elevation=float(np.radians(el)), azimuth=float(np.radians(az))), exp, atol=1e-4)
    np.testing.assert_allclose(zup(elevation=float(np.radians(12.5)), azimuth=float(np.radians(39.1))), [-0.615494, -0.757725, 0.216842], rtol=1e-2)
    for el, az in ((0.0, 0.0), (1.0, 1.0), (0.0, -np.pi)):
        want = np.array([-np.cos(el) * np.sin(az), np.sin(el), -np.cos(el) * np.cos(az)])
        got = _sky_light(direction=[float(v) for v in want])[1][9:12]  # fromDirectionYUp, then toDirectionYUp
        np.testing.assert_allclose(got, want, atol=2e-6)
        e2, a2 = _ea_of_yup(got)
        assert e2 == pytest.approx(el, abs=2e-6) and a2 == pytest.approx(az + 2 * np.pi if az < 0 else az, abs=2e-6)


def test_perez_model_known_answer_of_the_reference_unit_test():
    """src/tests/units/perez.cpp:9-40, the cross-check with Radiance's gendaylit (11 18 13 -y 2022 ... -W 0.39 57.03 -O 1): from diffuse
    irradiance 57.03, direct irradiance 0.39, the sun at (-0.615494, -0.757725, 0.216842) Z-up (zenith angle 77.5 deg) on day 322
    the model's coefficients are a..e = 0.597123, -0.562370, 0.828195, -0.625727, 0.009207 (relative 1e-4) and the diffuse
    normalisation diffirrad / integrate = 10.64 (relative 1e-2). (The sky clearness 1.0019 and brightness 0.1841 the test also names
    select and weight those coefficients: Perez' table is binned by the first and linear in the second.) `output: solarradiance`
    makes the light's sky colour that normalisation (PerezLight.cpp:10-136)."""
    _, d = _sky_light(type="perez", direction=[-0.615494, 0.216842, -0.757725], year=2022, month=11, day=18, diffuse_irradiance=57.029998779296875,
                      direct_irradiance=0.38999998569488525, output="solarradiance", has_sun=False, color=[1, 1, 1])
    zenith = np.arccos(d[10])
    assert zenith == pytest.approx(np.radians(77.5), rel=1e-3)
    np.testing.assert_allclose([d[6], d[7], d[8], d[12], d[13]], [0.597123, -0.562370, 0.828195, -0.625727, 0.009207], rtol=1e-4)
    assert d[0] == pytest.approx(10.64, rel=1e-2) and d[0] == d[1] == d[2]
