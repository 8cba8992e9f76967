"""Number properties of BSDFs given as shading expressions / textures (ShadingTree::addNumber with a PExpr string,
src/runtime/loader/ShadingTree.cpp:211-251,811-840): the loader's number list (IG_MAT_EXPR_NUMBERS, include/ig_tables.h), the
oracle evaluating it per hit, and — marked gpu — the HIP path against the oracle."""
import json
import os

import numpy as np
import pytest

import oracle
from conftest import SCENES
from ignis_amd import LoadedScene

W, H = 72, 56


def _scene(bsdf, extra_textures=(), extra_bsdfs=()):
    s = json.load(open(os.path.join(SCENES, "diamond_scene.json")))
    s["technique"] = {"type": "path", "max_depth": 6}
    s["textures"] = list(s.get("textures", [])) + list(extra_textures)
    for b in s["bsdfs"]:
        if b["name"] == "mat-GrayWall":  # bottom, top and back walls
            b.clear()
            b.update(dict(bsdf, name="mat-GrayWall"))
    s["bsdfs"] += list(extra_bsdfs)
    return LoadedScene.from_string(json.dumps(s), SCENES, W, H)


def _render(sc, its=2, spi=4):
    fb = np.zeros((H, W, 3), np.float32)
    for it in range(its):
        oracle.render(sc, spi, W, H, iteration=it, seed=6, fb=fb)
    return fb


CASES = [
    # (bsdf with constant numbers, the same with expressions that evaluate to those numbers at every hit)
    ({"type": "conductor", "eta": [0.2, 0.9, 1.1], "k": [3.9, 2.4, 2.2], "roughness": 0.3, "anisotropic": 0.4},
     {"type": "conductor", "eta": [0.2, 0.9, 1.1], "k": [3.9, 2.4, 2.2], "roughness": "0.3 + 0 * uv.x", "anisotropic": 0.4}),
    ({"type": "plastic", "diffuse_reflectance": [0.5, 0.3, 0.2], "roughness": 0.2},
     {"type": "plastic", "diffuse_reflectance": [0.5, 0.3, 0.2], "roughness": "select(uv.x < 2, 0.2, 0.9)", "int_ior": "1.49 + 0 * uv.y"}),
    ({"type": "dielectric", "int_ior": 1.5, "roughness": 0.25},
     {"type": "dielectric", "int_ior": "1.5 + 0 * uv.x", "roughness": "0.25 * (1 + 0 * P.y)"}),
    ({"type": "principled", "base_color": [0.7, 0.6, 0.5], "metallic": 0.6, "roughness": 0.35, "anisotropic": 0.2, "clearcoat": 0.5, "sheen": 0.3, "specular_tint": 0.4},
     {"type": "principled", "base_color": [0.7, 0.6, 0.5], "metallic": "0.6 + 0 * uv.y", "roughness": "0.35 + 0 * uv.x", "anisotropic": 0.2,
      "clearcoat": "color(0.5, 0.5, 0.5) * (1 + 0 * uv.x)", "sheen": "0.3 * (uv.x * 0 + 1)", "specular_tint": "0.4 + 0 * N.x"}),
    ({"type": "diffuse", "reflectance": [0.6, 0.6, 0.6], "roughness": 0.4},
     {"type": "diffuse", "reflectance": [0.6, 0.6, 0.6], "roughness": "0.4 + 0 * uv.x"}),
    ({"type": "phong", "specular_reflectance": [0.7, 0.7, 0.7], "exponent": 12},
     {"type": "phong", "specular_reflectance": [0.7, 0.7, 0.7], "exponent": "12 + 0 * uv.x"}),
    ({"type": "dielectric", "int_ior": 1.33, "ext_ior": 1.0},
     {"type": "dielectric", "int_ior": "1.33 + 0 * uv.x", "ext_ior": "1 + 0 * uv.y"}),
]


def test_loader_builds_the_number_list_and_folds_constants():
    const, expr = CASES[3]
    a, b = _scene(const), _scene(expr)
    fa = [a.scene.materials[i].flags for i in range(a.scene.material_count)]
    fb = [b.scene.materials[i].flags for i in range(b.scene.material_count)]
    assert not any(f & (1 << 11) for f in fa) and sum(1 for f in fb if f & (1 << 11)) == 1
    m = next(b.scene.materials[i] for i in range(b.scene.material_count) if b.scene.materials[i].flags & (1 << 11))
    at = int(np.float32(m.r[7]).view(np.uint32))
    code = np.ctypeslib.as_array(b.scene.expr_code, shape=(b.scene.expr_code_count,))
    n = int(code[at])
    heads = [int(code[at + 1 + 3 * i]) for i in range(n)]
    assert n == 5
    assert sorted((h & 0xFF, (h >> 8) & 0xFF, (h >> 16) & 0xFF) for h in heads) == sorted([(0, 20, 0), (1, 8, 9), (0, 23, 0), (0, 21, 0), (0, 7, 0)])
    aspect = float(code[at + 2 + 3 * heads.index(next(h for h in heads if h & 0xFF == 1))].view(np.float32))
    assert aspect == pytest.approx(np.sqrt(1 - 0.2 * 0.99), rel=1e-6)
    # "(0.175)^2" and friends still fold at load time: no list
    c = _scene({"type": "conductor", "roughness": "(0.175)^2"})
    assert not any(c.scene.materials[i].flags & (1 << 11) for i in range(c.scene.material_count))
    with pytest.raises(RuntimeError, match="not a number"):
        _scene({"type": "conductor", "roughness": "uv"})
    with pytest.raises(RuntimeError, match="blend"):
        _scene({"type": "blend", "first": "inner_a", "second": "mat-ColoredWall", "weight": 0.5}, extra_bsdfs=[{"type": "conductor", "name": "inner_a", "roughness": "0.1 + uv.x"}])


@pytest.mark.parametrize("case", range(len(CASES)))
def test_oracle_expression_numbers_equal_their_constant_values(case):
    """An expression that evaluates to the same number at every hit must give the constant's image bit for bit: slot mapping,
    the roughness -> (alpha_u, alpha_v) arithmetic, the colour -> average rule."""
    const, expr = CASES[case]
    np.testing.assert_array_equal(_render(_scene(const)), _render(_scene(expr)))


def test_oracle_expression_numbers_vary_over_the_surface():
    """roughness = a checkerboard over the uv: the image lies between the two constant renders and differs from both."""
    tex = [{"type": "checkerboard", "name": "chk", "scale_x": 4, "scale_y": 4, "color0": [0.05, 0.05, 0.05], "color1": [0.6, 0.6, 0.6]}]
    lo = _render(_scene({"type": "conductor", "roughness": 0.05}), 3, 8)
    hi = _render(_scene({"type": "conductor", "roughness": 0.6}), 3, 8)
    mix = _render(_scene({"type": "conductor", "roughness": "select(checkerboard(uv * 4) > 0.5, 0.6, 0.05)"}, tex), 3, 8)
    assert not np.array_equal(mix, lo) and not np.array_equal(mix, hi)
    d_lo, d_hi, d = np.abs(mix - lo).mean(), np.abs(mix - hi).mean(), np.abs(hi - lo).mean()
    assert d > 0 and d_lo < d and d_hi < d


@pytest.mark.gpu
@pytest.mark.parametrize("case", [0, 1, 2, 3, 5, 6])
def test_expression_numbers_vs_oracle(gpu_device, case):
    sc = _scene(CASES[case][1])
    gpu_device.assign_scene(sc)
    gpu_device.resize(W, H)
    gpu_device.clear_framebuffer()
    for it in range(2):
        gpu_device.render(4, W, H, iteration=it, seed=6)
    got, ref = gpu_device.framebuffer(), _render(sc)
    assert float(np.linalg.norm(got - ref) / np.linalg.norm(ref)) <= 1e-4


@pytest.mark.gpu
def test_varying_roughness_vs_oracle(gpu_device):
    sc = _scene({"type": "principled", "base_color": [0.8, 0.7, 0.6], "metallic": "clamp(uv.x, 0, 1)", "roughness": "0.05 + 0.5 * clamp(uv.y, 0, 1)"})
    gpu_device.assign_scene(sc)
    gpu_device.resize(W, H)
    gpu_device.clear_framebuffer()
    for it in range(2):
        gpu_device.render(4, W, H, iteration=it, seed=6)
    got, ref = gpu_device.framebuffer(), _render(sc)
    assert float(np.linalg.norm(got - ref) / np.linalg.norm(ref)) <= 1e-4


@pytest.mark.gpu
def test_a_number_list_alone_selects_the_expression_variant(gpu_device):
    """A point light and a conductor: nothing but the number list asks for the full shading variant (igd_assign_scene)."""
    from conftest import flat_scene

    s = flat_scene([{"type": "point", "name": "l", "position": [0.3, 0.2, -0.6], "intensity": [3, 3, 3]}], max_depth=4)
    s["bsdfs"] = [{"type": "conductor", "name": "ground", "roughness": "0.05 + 0.4 * clamp(uv.x, 0, 1)"}]
    sc = LoadedScene.from_string(json.dumps(s), SCENES, 64, 64)
    gpu_device.assign_scene(sc)
    gpu_device.resize(64, 64)
    gpu_device.clear_framebuffer()
    gpu_device.render(8, 64, 64, iteration=0, seed=3)
    ref = np.zeros((64, 64, 3), np.float32)
    oracle.render(sc, 8, 64, 64, iteration=0, seed=3, fb=ref)
    smooth = dict(s, bsdfs=[{"type": "conductor", "name": "ground", "roughness": 0.1}])
    ref_const = np.zeros((64, 64, 3), np.float32)
    oracle.render(LoadedScene.from_string(json.dumps(smooth), SCENES, 64, 64), 8, 64, 64, iteration=0, seed=3, fb=ref_const)
    assert not np.array_equal(ref, ref_const)
    assert float(np.linalg.norm(gpu_device.framebuffer() - ref) / np.linalg.norm(ref)) <= 1e-4
