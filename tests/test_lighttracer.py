"""The light tracer (src/artic/technique/lighttracer.art): loader, oracle, and — marked gpu — the HIP path against the oracle."""
import json
import math
import os

import numpy as np
import pytest

from conftest import SCENES
from ignis_amd.tables import LoadedScene


def _plane_scene(technique, fov=10.0, lights=None, bsdf=None):
    return {"technique": technique,
            "camera": {"type": "perspective", "fov": fov, "near_clip": 0.01, "far_clip": 100,
                       "transform": [{"lookat": {"origin": [0, 0, 6], "target": [0, 0, 0], "up": [0, 1, 0]}}]},
            "film": {"size": [64, 64]}, "bsdfs": [bsdf or {"type": "diffuse", "name": "m", "reflectance": [0.7, 0.6, 0.5]}],
            "shapes": [{"type": "rectangle", "name": "quad", "width": 4, "height": 4}],
            "entities": [{"name": "quad", "shape": "quad", "bsdf": "m"}],
            "lights": lights or [{"type": "point", "name": "p", "position": [0.3, -0.2, 2], "intensity": [5, 5, 5]}]}


def test_loader_lowers_the_technique_and_refuses_what_it_cannot_connect():
    sc = LoadedScene.from_string(json.dumps(_plane_scene({"type": "lt", "max_light_depth": 7, "min_depth": 3, "clamp": 2.5})), SCENES, 64, 64)
    t = sc.scene.technique
    assert (t.type, t.max_depth, t.min_depth, t.clamp) == (4, 7, 3, 2.5)
    other = LoadedScene.from_string(json.dumps(_plane_scene({"type": "lighttracer"})), SCENES, 64, 64)  # (keep the owner of the tables alive)
    assert other.scene.technique.max_depth == 64
    # (every light type has its sample_emission since round 4: the sky models and textured environments too)
    sky = LoadedScene.from_string(json.dumps(_plane_scene({"type": "lt"}, lights=[{"type": "cie_cloudy", "name": "s"}])), SCENES, 64, 64)
    assert sky.scene.lights[0].type == 7
    bad = _plane_scene({"type": "lt"})
    bad["camera"]["type"] = "fishlens"
    with pytest.raises(RuntimeError, match="perspective camera"):
        LoadedScene.from_string(json.dumps(bad), SCENES, 64, 64)


@pytest.mark.parametrize("lights", [
    [{"type": "point", "name": "p", "position": [0.3, -0.2, 2], "intensity": [5, 5, 5]}],
    [{"type": "spot", "name": "s", "position": [0.2, 0.1, 3], "direction": [0, 0, -1], "cutoff": 40, "falloff": 30, "intensity": [8, 8, 8]}],
    [{"type": "directional", "name": "d", "direction": [0.2, 0.1, -1], "irradiance": [2, 2, 2]}],
    [{"type": "env", "name": "e", "radiance": [1, 1, 1]}],
    [{"type": "sun", "name": "s", "direction": [-0.1, -0.2, 1], "irradiance": [2, 2, 2], "angle": 4}],
    [{"type": "cie_cloudy", "name": "sky", "zenith": [0.5, 0.6, 0.9], "ground": [0.4, 0.3, 0.2], "has_ground": False, "transform": [{"rotate": [70, 0, 0]}]}],
    [{"type": "cie_clear", "name": "sky", "zenith": [0.5, 0.6, 0.9], "ground": [0.4, 0.3, 0.2], "direction": [0.3, 0.4, 0.8], "turbidity": 3.0}],
    [{"type": "env", "name": "e", "radiance": "sky"}],  # (no `scale`: the CDF-sampled environment drops it in sample_dir, light/env.art:112-113)
    [{"type": "env", "name": "e", "radiance": "sky", "cdf": "none", "transform": [{"rotate": [10, 20, 30]}]}],
], ids=["point", "spot", "directional", "env", "sun", "cie-hemisphere", "cie-sphere", "env-cdf", "env-uniform"])
def test_oracle_light_tracer_agrees_with_the_path_tracer_up_to_the_pixel_measure(lights):
    """As written the camera connection weighs a vertex with image_area = 1 (camera/perspective.art:36,47-51) instead of the
    pixel's importance 1 / (A cos^3), A = 4 sx sy the area of the image plane at distance 1: with a narrow field of view
    (cos^3 > 0.988 at 10 degrees) a light-tracer image is the path tracer's direct lighting times A. (Not in the list: the Perez sky
    with its sun. Its sample_emission, light/perez.art:309-313, draws directions inside the sun's cone only and adds the sky seen there, so a
    light tracer never sees the rest of the sky the path tracer reaches through BSDF-sampled misses: 0.67 x on this scene, as written.)"""
    import oracle
    fov = 10.0
    def scene(technique):
        s = _plane_scene(technique, fov, lights)
        if lights[0].get("radiance") == "sky":
            s["textures"] = [{"type": "image", "name": "sky", "filename": "textures/sky_gradient.png"}]
        return LoadedScene.from_string(json.dumps(s), SCENES, 64, 64)

    a, b = scene({"type": "path", "max_depth": 2}), scene({"type": "lt", "max_depth": 2})
    pt = np.zeros((64, 64, 3), np.float32)
    lt = np.zeros((64, 64, 3), np.float32)
    n_lt = 16 if lights[0]["type"] == "spot" else 64
    for it in range(4):
        oracle.render(a, 8, 64, 64, iteration=it, seed=2, fb=pt)
    for it in range(n_lt):
        oracle.render(b, 16, 64, 64, iteration=it, seed=2, fb=lt)
    pt /= 4
    lt /= n_lt
    sx = math.tan(math.radians(fov) / 2)
    area = 4 * sx * sx  # square film
    c = slice(8, 56)
    ratio = lt[c, c].mean() / (pt[c, c].mean() * area)
    if lights[0]["type"] == "spot":
        # make_spot_light.sample_emission (light/spot.art:41-47) divides the intensity by spot_area * pdf and the emitter multiplies
        # by the sample's cosine, where sample_direct of the same light has neither: as written, a spot light's light-tracer image
        # is 1 / spot_area = 1 / (pi tan^2(cutoff)) of its path-tracer image near the axis. Restated as written.
        ratio *= math.pi * math.tan(math.radians(lights[0]["cutoff"])) ** 2
    assert pt[c, c].mean() > 1e-3 and abs(ratio - 1) < 0.05, ratio


def test_oracle_light_tracer_only_counts_connections_it_traces():
    import oracle
    sc = LoadedScene.from_file(os.path.join(SCENES, "evaluation", "cycles-lights-lt.json"), 64, 64)
    fb, st = oracle.render(sc, 4, 64, 64, iteration=0, seed=1)
    assert np.isfinite(fb).all() and fb.mean() > 0.01
    assert st["camera_rays"] == 64 * 64 * 4 and 0 < st["unoccluded"] <= st["shadow_rays"] and st["bounce_rays"] > 0


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["cycles-lights", "sphere-light", "diamond", "diamond-principled-bump", "diamond-skies"])
def test_light_tracer_vs_oracle(gpu_device, case):
    """The path set (counters) is the oracle's exactly; the pixel sums agree to the rounding of their summation order. Since round 5 the
    connections of a round are added in (pixel slot, light path) order (K7 / K8: launch_lt_splat, photon.hip) instead of by float
    atomics: rendering the same iterations again gives the same bits."""
    import oracle
    if case == "cycles-lights":
        sc = LoadedScene.from_file(os.path.join(SCENES, "evaluation", "cycles-lights-lt.json"), 96, 96)
        w, h = 96, 96
    elif case == "sphere-light":  # an icosphere mesh recognised as a sphere: make_sphere_area_emitter.sample_emission
        s = json.load(open(os.path.join(SCENES, "evaluation", "sphere-light-ico.json")))
        s["technique"] = {"type": "lt", "max_depth": 6}
        sc = LoadedScene.from_string(json.dumps(s), os.path.join(SCENES, "evaluation"), 96, 96)
        assert any(sc.scene.lights[i].type == 9 for i in range(sc.scene.light_count))
        w, h = 96, 96
    else:
        s = json.load(open(os.path.join(SCENES, "diamond_scene.json")))
        s["technique"] = {"type": "lt", "max_depth": 8, "light_selector": "uniform"}
        s["lights"] += [{"type": "env", "name": "sky", "radiance": [0.3, 0.3, 0.4]}, {"type": "directional", "name": "d", "direction": [0.3, -1, 0.2], "irradiance": [1, 1, 1]},
                        {"type": "sun", "name": "sun", "direction": [0.2, 1, -0.1], "irradiance": [1, 1, 1], "angle": 3},
                        {"type": "point", "name": "p", "position": [0, 1.2, 0], "intensity": [1, 1, 1]}]
        if "skies" in case:  # emission sampling of the sky models and textured environments (light/env.art:38-46,87-93,141-145, perez.art:309-313)
            s["textures"] = [{"type": "image", "name": "sky", "filename": "textures/sky_gradient.png"}]
            s["lights"] = s["lights"][:1] + [
                {"type": "cie_cloudy", "name": "c1", "zenith": [0.5, 0.6, 0.9], "ground": [0.4, 0.3, 0.2], "has_ground": False, "transform": [{"rotate": [0, 0, 25]}]},
                {"type": "cie_clear", "name": "c2", "zenith": [0.5, 0.6, 0.9], "ground": [0.4, 0.3, 0.2], "direction": [0.3, 0.7, -0.5], "turbidity": 3.0},
                {"type": "perez", "name": "pz", "clearness": 8, "brightness": 0.1},
                {"type": "env", "name": "e1", "radiance": "sky", "scale": [1, 0.8, 0.6]},
                {"type": "env", "name": "e2", "radiance": "sky", "cdf": "none", "transform": [{"rotate": [10, 20, 30]}]}]
            s["entities"] = [e for e in s["entities"] if e["name"] not in ("Back", "Top")]
        if "bump" in case:
            s["textures"] = [{"type": "image", "name": "bumps", "filename": "textures/bumpmap.png"}]
            for b in s["bsdfs"]:
                if b["name"] == "mat-GrayWall":
                    b["name"] = "wall-inner"
                if b["name"] == "mat-Diamond":  # refracting principled BSDF: the 1 / eta^2 of light paths (principled.art:471)
                    b.clear()
                    b.update({"type": "principled", "name": "mat-Diamond", "base_color": [0.9, 0.95, 1.0], "roughness": 0.15, "specular_transmission": 0.9, "ior": 1.5})
                if b["name"] == "mat-ColoredWall":  # rough dielectric: dielectric.art:181
                    b.clear()
                    b.update({"type": "roughdielectric", "name": "mat-ColoredWall", "roughness": 0.3, "int_ior": 1.4, "ext_ior": 1.0})
            s["bsdfs"].append({"type": "bumpmap", "name": "mat-GrayWall", "bsdf": "wall-inner", "map": "bumps", "strength": 0.5})
        sc = LoadedScene.from_string(json.dumps(s), SCENES, 96, 72)
        w, h = 96, 72
    gpu_device.assign_scene(sc)
    gpu_device.resize(w, h)
    gpu_device.clear_framebuffer()
    gpu_device.reset_stats()
    ref = np.zeros((h, w, 3), np.float32)
    tot = {}
    for it in range(2):
        gpu_device.render(4, w, h, iteration=it, seed=17)
        _, st = oracle.render(sc, 4, w, h, iteration=it, seed=17, fb=ref)
        for k, v in st.items():
            tot[k] = tot.get(k, 0) + v
    fb = gpu_device.framebuffer()
    gst = gpu_device.stats()
    for k in ("camera_rays", "bounce_rays", "shadow_rays", "unoccluded"):
        assert gst[k] == tot[k], k
    assert ref.mean() > 1e-3
    assert np.linalg.norm(fb - ref) / np.linalg.norm(ref) < 1e-5
    first = fb.copy()
    gpu_device.clear_framebuffer()
    for it in range(2):
        gpu_device.render(4, w, h, iteration=it, seed=17)
    again = gpu_device.framebuffer()
    assert np.array_equal(first.view(np.uint32), again.view(np.uint32))  # bit-reproducible: no float atomics on the way


# ---- the wireframe technique (src/artic/technique/wireframe.art)

def test_oracle_wireframe_shows_triangle_edges_and_nothing_else():
    import oracle
    s = _plane_scene({"type": "wireframe"}, fov=40.0)
    sc = LoadedScene.from_string(json.dumps(s), SCENES, 64, 64)
    assert sc.scene.technique.type == 5
    fb, st = oracle.render(sc, 8, 64, 64, iteration=0, seed=3)
    g = fb.mean(axis=2)
    # the quad is two triangles: its diagonal and its outline are lit, the faces stay black (the ray goes on and misses)
    assert g.max() > 0.5 and (g < 1e-3).mean() > 0.6 and (g > 0.05).mean() > 0.02
    assert g[32, 32] > 0.05 or g[31, 32] > 0.05 or g[32, 31] > 0.05  # the diagonal passes the centre
    assert st["shadow_rays"] == 0 and st["bounce_rays"] > 0
    bad = _plane_scene({"type": "wireframe"})
    bad["camera"]["type"] = "fishlens"
    with pytest.raises(RuntimeError, match="camera differential"):
        LoadedScene.from_string(json.dumps(bad), SCENES, 64, 64)


@pytest.mark.gpu
@pytest.mark.parametrize("camera", ["perspective", "orthogonal"])
def test_wireframe_vs_oracle(gpu_device, camera):
    import oracle
    s = json.load(open(os.path.join(SCENES, "diamond_scene.json")))
    s["technique"] = {"type": "wireframe"}
    if camera == "orthogonal":
        s["camera"]["type"] = "orthogonal"
        s["camera"]["scale"] = 1.2
    sc = LoadedScene.from_string(json.dumps(s), SCENES, 128, 96)
    gpu_device.assign_scene(sc)
    gpu_device.resize(128, 96)
    gpu_device.clear_framebuffer()
    gpu_device.reset_stats()
    gpu_device.render(4, 128, 96, iteration=0, seed=5)
    fb = gpu_device.framebuffer()
    ref, st = oracle.render(sc, 4, 128, 96, iteration=0, seed=5)
    gst = gpu_device.stats()
    for k in ("camera_rays", "bounce_rays", "shadow_rays"):
        assert gst[k] == st[k], k
    assert ref.max() > 0.5 and np.linalg.norm(fb - ref) / np.linalg.norm(ref) < 1e-6  # (a pixel's samples are summed in another order)
