"""The photon mapper (src/artic/technique/photonmapper.art, PhotonMappingTechnique.cpp): loader, the photon encodings and grid of
include/ig_photon.h against independent numpy restatements, the oracle against the path tracer, and — marked gpu — the HIP path
against the oracle."""
import ctypes as C
import json
import os

import numpy as np
import pytest

import oracle
from conftest import SCENES
from ignis_amd import LoadedScene

EVAL = os.path.join(SCENES, "evaluation")


def _scene(technique, w=64, h=64):
    return LoadedScene.from_string(json.dumps({"technique": technique, "externals": [{"filename": "cycles-lights.json"}]}), EVAL, w, h)


def _lib():
    l = oracle.lib()
    l.oracle_photon_codec.argtypes = [C.POINTER(C.c_float)] * 2 + [C.POINTER(C.c_int32)] * 2 + [C.POINTER(C.c_float)] * 2
    l.oracle_photon_cell.restype = C.c_int32
    l.oracle_photon_cell.argtypes = [C.POINTER(C.c_float)] * 3
    l.oracle_photon_radius.restype = C.c_float
    l.oracle_photon_radius.argtypes = [C.c_float, C.c_int32]
    return l


def test_loader_lowers_the_technique():
    sc = LoadedScene.from_file(os.path.join(EVAL, "cycles-lights-ppm.json"), 64, 64)
    t = sc.scene.technique
    s = sc.scene
    diameter = float(np.linalg.norm(np.array(list(s.bbox_max)) - np.array(list(s.bbox_min))))
    assert (t.type, t.max_depth, t.min_depth, t.photon_count, t.max_light_depth) == (6, 16, 2, 1000000, 8)
    assert t.merge_radius == pytest.approx(0.01 * diameter, rel=1e-6)  # __tech_radius = radius * SceneDiameter
    other = _scene({"type": "photonmapper", "photons": 7, "max_camera_depth": 5, "min_camera_depth": 3, "max_light_depth": 4, "radius": 0.5})
    t = other.scene.technique
    assert (t.photon_count, t.max_depth, t.min_depth, t.max_light_depth) == (100, 5, 3, 4)  # at least 100 photons
    sky = LoadedScene.from_string(json.dumps({"technique": {"type": "ppm"}, "lights": [{"type": "cie_cloudy", "name": "s"}]}), "", 32, 32)
    assert sky.scene.lights[0].type == 7  # (sky models and textured environments have their sample_emission since round 4)


def _snorm16(f):
    """encode_signed_norm_16 as the generated code performs it: round half away from zero, f32 -> i32, low 16 bits as i16."""
    v = np.float32(f) * np.float32(65535)
    r = int(np.sign(v) * np.floor(np.abs(np.float64(v)) + 0.5))
    r &= 0xFFFF
    return r - 0x10000 if r >= 0x8000 else r


def test_photon_encodings_and_grid_against_numpy():
    l = _lib()
    rng = np.random.default_rng(11)
    f3 = lambda a: (C.c_float * 3)(*[float(x) for x in a])
    for _ in range(400):
        d = rng.normal(size=3).astype(np.float32)
        d /= np.float32(np.linalg.norm(d))
        p = (rng.uniform(0, 1, 3) * 10.0 ** rng.uniform(-3, 3)).astype(np.float32)
        ed, ep = C.c_int32(), C.c_int32()
        od, op = (C.c_float * 3)(), (C.c_float * 3)()
        l.oracle_photon_codec(f3(d), f3(p), C.byref(ed), C.byref(ep), od, op)
        # encode_normal_32 (core/common.art:155-197)
        a = np.float32(abs(d[0])) + np.float32(abs(d[1])) + np.float32(abs(d[2]))
        ox, oy = np.float32(d[0] / a), np.float32(d[1] / a)
        if d[2] < 0:
            ox, oy = np.float32((1 - abs(oy)) * (1 if ox >= 0 else -1)), np.float32((1 - abs(ox)) * (1 if oy >= 0 else -1))
        ex, ey = _snorm16(ox), _snorm16(oy)
        want = ((ex << 16) | (ey & 0xFFFFFFFF if ey >= 0 else ey + (1 << 32))) & 0xFFFFFFFF
        assert (ed.value & 0xFFFFFFFF) == want
        # decode_normal_32: (val >> 16) as i16, val as i16, reflected branch, normalise
        val = ed.value
        hi, lo = (val >> 16) & 0xFFFF, val & 0xFFFF
        hi, lo = (hi - 0x10000 if hi >= 0x8000 else hi), (lo - 0x10000 if lo >= 0x8000 else lo)
        dx, dy = np.clip(np.float32(hi) / np.float32(65535), -1, 1), np.clip(np.float32(lo) / np.float32(65535), -1, 1)
        oz = np.float32(1) - abs(dx) - abs(dy)
        vx, vy = (dx, dy) if oz >= 0 else (np.float32(1 - abs(dy) * (1 if dx >= 0 else -1)), np.float32(1 - abs(dx) * (1 if dy >= 0 else -1)))
        v = np.array([vx, vy, oz], np.float64)
        np.testing.assert_allclose(np.array(list(od)), v / np.linalg.norm(v), rtol=2e-6, atol=2e-7)
        # a component of the projection within +-0.5 survives; beyond it the i16 wraps (as written)
        if abs(ox) < 0.49 and abs(oy) < 0.49 and oy >= 0:
            np.testing.assert_allclose(np.array(list(od)), d, atol=3e-5)
        # encode_rgbe / decode_rgbe (core/color.art:341-368): 8-bit mantissas under the largest component's exponent
        m, e = np.frexp(np.float32(p.max()))
        val2 = np.float32(np.float32(m) * np.float32(256)) / np.float32(p.max())
        q = [int(np.float32(c) * val2) & 0xFF for c in p]
        assert (ep.value & 0xFFFFFFFF) == ((q[0] << 24) | (q[1] << 16) | (q[2] << 8) | (int(e) + 128))
        np.testing.assert_allclose(np.array(list(op)), np.array(q, np.float64) * 2.0 ** (int(e) - 8), rtol=1e-7)
        assert np.all(np.array(list(op)) <= p * (1 + 1e-6)) and np.all(np.array(list(op)) >= p - p.max() / 128)
    # grid: 128^3 cells over the bounding box, Morton order (x lowest)
    bmin, bmax = f3([-1, -2, 0]), f3([3, 2, 8])
    for _ in range(200):
        pos = rng.uniform([-1.5, -2.5, -0.5], [3.5, 2.5, 8.5]).astype(np.float32)
        n = np.clip((pos - np.array([-1, -2, 0], np.float32)) / np.array([4, 4, 8], np.float32) * np.float32(0.99), 0, 1)
        c = np.minimum((n * np.float32(128)).astype(np.int64), 127)
        morton = 0
        for bit in range(7):
            morton |= ((int(c[0]) >> bit) & 1) << (3 * bit) | ((int(c[1]) >> bit) & 1) << (3 * bit + 1) | ((int(c[2]) >> bit) & 1) << (3 * bit + 2)
        assert l.oracle_photon_cell(f3(pos), bmin, bmax) == morton
    # ppm_compute_radius: r_i = r_0 * prod (k + 1.8) / (k + 2), never below 1e-5
    assert l.oracle_photon_radius(0.5, 0) == 0.5
    assert l.oracle_photon_radius(0.5, 3) == pytest.approx(0.5 * (1.8 / 2) * (2.8 / 3) * (3.8 / 4), rel=1e-6)
    assert l.oracle_photon_radius(1e-9, 5) == pytest.approx(1e-5)


def test_oracle_photon_mapper_agrees_with_the_path_tracer_up_to_the_spot_lights_emission():
    """cycles-lights (blue point, green spot and red area light over a diffuse plane): direct light gathered from 300 k photons per
    iteration against next-event estimation. Point and area light agree to a per cent. The spot light is darker by exactly the factor its
    sample_emission carries and its sample_direct does not: 1 / spot_area = 1 / (pi tan^2 cutoff) (light/spot.art:13-14,41-47) — the
    relation tests/test_lighttracer.py asserts for the light tracer. That pins the technique's light pass, grid, gather, radius and
    kernel for all three light types; round 3 had blamed the green deficit on encode_signed_norm_16, which the second half refutes:
    with photon directions stored through a sound 16-bit encoding (oracle-only switch) the image is the same to 1e-6."""
    a, b = _scene({"type": "path", "max_depth": 16}), _scene({"type": "ppm", "max_depth": 16, "photons": 300000})

    def run():
        fa, fb = np.zeros((64, 64, 3), np.float32), np.zeros((64, 64, 3), np.float32)
        for it in range(6):
            oracle.render(a, 8, 64, 64, iteration=it, seed=3, fb=fa)
            _, st = oracle.render(b, 8, 64, 64, iteration=it, seed=3, fb=fb)
        return fa, fb, st

    fa, fb, st = run()
    ma, mb = fa.mean(axis=(0, 1)), fb.mean(axis=(0, 1))
    spot = next(l for l in json.load(open(os.path.join(EVAL, "cycles-lights.json")))["lights"] if l["type"] == "spot")
    spot_area = np.pi * np.tan(np.radians(spot["cutoff"])) ** 2
    assert mb[0] == pytest.approx(ma[0], rel=0.02) and mb[2] == pytest.approx(ma[2], rel=0.02)  # area light, point light
    assert mb[1] * spot_area == pytest.approx(ma[1], rel=0.02)                                  # spot light: 0.54 x as written
    assert st["shadow_rays"] == 0 and st["camera_rays"] == 300000 + 64 * 64 * 8  # emitter rays + camera rays; no shadow rays at all
    oracle.set_ppm_sound_directions(True)
    try:
        _, fs, _ = run()
    finally:
        oracle.set_ppm_sound_directions(False)
    assert float(np.linalg.norm(fs - fb) / np.linalg.norm(fb)) < 1e-5


@pytest.mark.gpu
def test_photon_mapper_with_sky_lights_vs_oracle():
    """The light pass from a hemisphere-sampled CIE sky, the Perez sky with its sun and a CDF-sampled textured environment."""
    from ignis_amd import Device
    s = json.load(open(os.path.join(SCENES, "diamond_scene.json")))
    s["technique"] = {"type": "ppm", "max_depth": 8, "photons": 40000, "radius": 0.03, "light_selector": "uniform"}
    s["textures"] = [{"type": "image", "name": "sky", "filename": "textures/sky_gradient.png"}]
    s["lights"] = s["lights"][:1] + [
        {"type": "cie_cloudy", "name": "c1", "zenith": [0.5, 0.6, 0.9], "ground": [0.4, 0.3, 0.2], "has_ground": False},
        {"type": "perez", "name": "pz", "clearness": 8, "brightness": 0.1},
        {"type": "env", "name": "e1", "radiance": "sky", "scale": [1, 0.8, 0.6]}]
    s["entities"] = [e for e in s["entities"] if e["name"] not in ("Back", "Top")]
    sc = LoadedScene.from_string(json.dumps(s), SCENES, 96, 72)
    dev = Device(0, acquire_stats=True)
    dev.assign_scene(sc)
    ref = np.zeros((72, 96, 3), np.float32)
    cam = bounce = 0
    for it in range(2):
        dev.render(4, 96, 72, iteration=it, seed=5)
        _, st = oracle.render(sc, 4, 96, 72, iteration=it, seed=5, fb=ref)
        cam += st["camera_rays"]
        bounce += st["bounce_rays"]
    got, ds = dev.framebuffer(), dev.stats()
    dev.close()
    assert ref.max() > 0
    assert float(np.linalg.norm(got - ref) / np.linalg.norm(ref)) <= 1e-4
    assert (ds["camera_rays"], ds["bounce_rays"], ds["shadow_rays"]) == (cam, bounce, 0)


@pytest.mark.gpu
@pytest.mark.parametrize("photons,spi", [(50000, 4), (400, 2)])
def test_photon_mapper_vs_oracle(photons, spi):
    """Light pass, grid and camera pass on the GPU: photons land in the slot of their light path and are gathered in (cell, index)
    order, as in the oracle, so the image agrees like any other technique's."""
    from ignis_amd import Device
    sc = _scene({"type": "ppm", "max_depth": 8, "photons": photons, "radius": 0.02}, 96, 72)
    dev = Device(0, acquire_stats=True)
    dev.assign_scene(sc)
    ref = np.zeros((72, 96, 3), np.float32)
    cam = bounce = 0
    for it in range(3):
        dev.render(spi, 96, 72, iteration=it, seed=9)
        _, st = oracle.render(sc, spi, 96, 72, iteration=it, seed=9, fb=ref)
        cam += st["camera_rays"]
        bounce += st["bounce_rays"]
    got = dev.framebuffer()
    ds = dev.stats()
    dev.close()
    assert ref.max() > 0
    assert float(np.linalg.norm(got - ref) / np.linalg.norm(ref)) <= 1e-4
    assert (ds["camera_rays"], ds["bounce_rays"], ds["shadow_rays"]) == (cam, bounce, 0)


@pytest.mark.gpu
def test_photon_mapper_chunks_end_at_iteration_boundaries():
    """A stream capacity that does not divide an iteration (96 x 72 x 2 = 13 824 rays, capacity 5 000) and three iterations in one
    call: every iteration must get its own light pass and merge radius, i.e. the image of three single-iteration calls on a
    device whose streams hold a whole iteration (chunks used to start at 0, 5 000, 10 000, 15 000 ... so no chunk started
    at 13 824 and iterations 1 and 2 were rendered with iteration 0's photon map)."""
    from ignis_amd import Device
    sc = _scene({"type": "ppm", "max_depth": 8, "photons": 20000, "radius": 0.02}, 96, 72)
    whole = Device(0, acquire_stats=True)
    whole.assign_scene(sc)
    for it in range(3):
        whole.render(2, 96, 72, iteration=it, seed=4)
    want, want_st = whole.framebuffer(), whole.stats()
    whole.close()
    small = Device(0, acquire_stats=True, stream_capacity=5000)
    small.assign_scene(sc)
    small.render(2, 96, 72, iteration=0, seed=4, iterations=3)
    got, got_st = small.framebuffer(), small.stats()
    small.close()
    np.testing.assert_array_equal(got, want)
    for k in ("camera_rays", "bounce_rays", "shadow_rays"):
        assert got_st[k] == want_st[k], k
