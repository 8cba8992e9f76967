"""Shading expressions: the PExpr compiler of the loader (ignis_amd/csrc/host/pexpr.h) and the interpreter shared by the shading
kernel and the oracle (include/ig_expr.h), through the C ABI (igh_eval_expression) against numpy float32 restatements of the
functions the reference's transpiler maps the names to (src/runtime/loader/Transpiler.cpp:602-922), and through the loader.
"""
import json
import math
import os

import numpy as np
import pytest

from conftest import SCENES
from ignis_amd.tables import LoadedScene, eval_expression as ev

F = np.float32


def near(a, b, tol=2e-6):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.all(np.abs(a - b) <= tol * np.maximum(1, np.abs(b)))


@pytest.mark.parametrize("src,ty,val", [
    ("1+2*3", "int", 7), ("(1+2)*3", "int", 9), ("7 % 3 + 7/2", "int", 4), ("7/2.0", "num", 3.5), ("-2^2", "int", -4), ("2^3^2", "int", 512),
    ("(0.175)^2", "num", float(F(0.175) * F(0.175))), ("1 + 2.5", "num", 3.5), ("1e2 + .5", "num", 100.5), ("--3", "int", 3), ("+3", "int", 3),
    ("1 < 2", "bool", True), ("2 <= 1", "bool", False), ("1 == 1.0", "bool", True), ("1 != 2", "bool", True), ("!(1 > 2) && true || false", "bool", True),
    ("vec3(1,2,3) == vec3(1,2,3)", "bool", True), ("vec3(1,2,3) != vec3(1,2,4)", "bool", True),
    ("Pi", "num", float(F(3.14159265359))), ("Eps", "num", float(np.finfo(F).eps)),
])
def test_scalars_and_operators(src, ty, val):
    t, v = ev(src)
    assert t == ty
    assert near(v, val) if ty != "bool" else v == val


def test_vectors_swizzles_and_constructors():
    assert ev("vec3(1,2,3).zyxx") == ("vec4", (3, 2, 1, 1))
    assert ev("vec2(4).yx") == ("vec2", (4, 4))
    assert ev("color(0.1, 0.2, 0.3)") == ("vec4", tuple(float(F(x)) for x in (0.1, 0.2, 0.3, 1)))  # make_vec4(r, g, b, 1)
    assert ev("color(0.5).a") == ("num", 0.5)
    assert ev("vec4(1,2,3,4).rgb + vec3(1)") == ("vec3", (2, 3, 4))
    assert ev("2 * vec3(1,2,3) / 4") == ("vec3", (0.5, 1, 1.5))           # onScale both ways, int -> num
    assert ev("vec3(1,2,3) * vec3(2)") == ("vec3", (2, 4, 6))
    assert ev("-vec2(1,-2)") == ("vec2", (-1, 2))
    assert ev("uv", uvw=(0.25, 0.75, 0)) == ("vec2", (0.25, 0.75))
    assert ev("uvw.zy", uvw=(0.25, 0.75, 0)) == ("vec2", (0, 0.75))
    assert ev("(N + Nx * 2 - Ny).xzy", N=(0, 0, 1), Nx=(1, 0, 0), Ny=(0, 1, 0)) == ("vec3", (2, 1, -1))
    assert ev("select(frontside, 1, 2.5)", frontside=1) == ("num", 1.0)
    assert ev("select(frontside, vec2(1), vec2(2))", frontside=0) == ("vec2", (2, 2))


@pytest.mark.parametrize("name,fn", [
    ("sin", np.sin), ("cos", np.cos), ("tan", np.tan), ("asin", np.arcsin), ("acos", np.arccos), ("atan", np.arctan), ("exp", np.exp),
    ("exp2", np.exp2), ("log", np.log), ("log2", np.log2), ("log10", np.log10), ("sqrt", np.sqrt), ("floor", np.floor), ("ceil", np.ceil),
    ("abs", np.abs), ("fract", lambda x: x - np.floor(x)), ("trunc", np.trunc), ("sign", np.sign), ("rad", np.radians), ("deg", np.degrees),
    ("smoothstep", lambda x: x * x * (3 - 2 * x)), ("smootherstep", lambda x: x * x * x * (x * (x * 6 - 15) + 10)),
])
def test_lanewise_functions(name, fn):
    for x in (0.1, 0.37, 0.5, 0.93):
        t, v = ev(f"{name}(P.x)", P=(x, 0, 0))
        assert t == "num" and near(v, fn(np.float64(F(x))), 3e-6), (name, x, v)
    if name not in ("smoothstep", "smootherstep"):
        t, v = ev(f"{name}(P)", P=(0.2, 0.4, 0.8))
        assert t == "vec3" and near(v, fn(np.array([0.2, 0.4, 0.8], F).astype(np.float64)), 3e-6)


def test_round_negative_floor_and_int_functions():
    assert ev("round(P.x)", P=(2.5, 0, 0))[1] == 3 and ev("round(P.x)", P=(-2.5, 0, 0))[1] == -3
    assert ev("floor(P.x)", P=(-0.5, 0, 0))[1] == -1 and ev("ceil(P.x)", P=(-0.5, 0, 0))[1] == 0
    assert ev("abs(-3)") == ("int", 3) and ev("sign(-3)") == ("int", -1) and ev("int(3.9)") == ("int", 3) and ev("num(3) / 2") == ("num", 1.5)
    assert ev("min(3, 5)") == ("int", 3) and ev("max(3, 5.5)") == ("num", 5.5) and ev("clamp(7, 0, 5)") == ("int", 5)


def test_binary_and_ternary_functions():
    p, q = np.array([0.3, -1.2, 2.0], F), np.array([1.5, 0.4, -0.7], F)
    kw = dict(P=p, V=q)
    assert near(ev("min(P, V)", **kw)[1], np.minimum(p, q)) and near(ev("max(P, V)", **kw)[1], np.maximum(p, q))
    assert near(ev("dot(P, V)", **kw)[1], np.dot(p.astype(np.float64), q)) and near(ev("cross(P, V)", **kw)[1], np.cross(p, q))
    assert near(ev("length(P)", **kw)[1], np.linalg.norm(p)) and near(ev("norm(P)", **kw)[1], p / np.linalg.norm(p))
    assert near(ev("dist(P, V)", **kw)[1], np.linalg.norm(p - q)) and near(ev("sum(P)", **kw)[1], p.sum()) and near(ev("avg(P)", **kw)[1], p.mean())
    n = q / np.linalg.norm(q)
    assert near(ev("reflect(P, norm(V))", **kw)[1], n * 2 * np.dot(n, p) - p, 1e-5)  # vec3_reflect(v, n) (core/vector.art:124)
    assert near(ev("mix(P, V, 0.25)", **kw)[1], 0.75 * p + 0.25 * q) and near(ev("mix(1, 3, 0.5)")[1], 2.0)
    assert near(ev("clamp(P, vec3(0), vec3(1))", **kw)[1], np.clip(p, 0, 1))
    assert near(ev("pow(abs(P), vec3(2.5))", **kw)[1], np.abs(p).astype(np.float64) ** 2.5, 5e-6) and near(ev("P.x ^ 3", **kw)[1], float(p[0]) ** 3, 5e-6)
    assert near(ev("P.y ^ 3", **kw)[1], float(p[1]) ** 3, 5e-6) and near(ev("P.y ^ 2", **kw)[1], float(p[1]) ** 2, 5e-6)  # negative base, integral exponent
    assert near(ev("atan2(P.x, V.x)", **kw)[1], math.atan2(p[0], q[0]), 5e-6)
    assert near(ev("fmod(P.z, 0.75)", **kw)[1], math.fmod(2.0, 0.75)) and near(ev("wrap(P.y, 0, 1)", **kw)[1], float(p[1]) - math.floor(p[1]))
    assert near(ev("luminance(color(0.2, 0.5, 0.9))")[1], 0.2 * 0.2126 + 0.5 * 0.7152 + 0.9 * 0.0722)  # color_luminance (core/color.art:29)


def _parity(v):
    rng = 2.0
    return int(v - rng * math.floor(v / rng)) % 2


def test_checkerboard_matches_the_texture_node():
    """node_checkerboard2 / 3 (src/artic/texture/checkerboard.art:1-2) over a grid of coordinates, negative ones included."""
    for x in np.linspace(-2.3, 2.3, 13):
        for y in np.linspace(-1.7, 2.9, 11):
            xy = _parity(float(F(x))) == _parity(float(F(y)))
            assert ev("checkerboard(P.xy)", P=(x, y, 0)) == ("int", int(xy))
            assert ev("checkerboard(P)", P=(x, y, 1.5)) == ("int", int(xy == (_parity(1.5) == 1)))
            assert ev("checkerboard(P)", P=(x, y, 0)) == ("int", int(not xy))  # w = 0: 1 where the parities differ


def test_bump_node():
    """node_bump (src/artic/texture/bump.art:3-11) restated in float64."""
    n, nx, ny = np.array([0, 0, 1.0]), np.array([1.0, 0, 0]), np.array([0, 1.0, 0])
    for dist, dx, dy in ((1.0, 0.3, -0.2), (0.5, -1.5, 0.7)):
        rx, ry = np.cross(ny, n), np.cross(n, nx)
        det = np.dot(nx, rx)
        grad = rx * dx + ry * dy
        want = n * abs(det) - grad * np.sign(det) * dist
        want /= np.linalg.norm(want)
        got = ev(f"bump(N, Nx, Ny, {dist}, {dx}, {dy})", N=n, Nx=nx, Ny=ny)
        assert got[0] == "vec3" and near(got[1], want, 1e-6)


@pytest.mark.parametrize("src,what", [
    ("foo", "unknown variable 'foo'"), ("cbrt(P.x)", "'cbrt' is not supported"), ("entity_id", "'entity_id' is not supported"),
    ("uv.z", "outside of vec2"), ("1 +", "end of expression"), ("vec3(1,2) ", "no function vec3(int, int)"), ("1 && 2", "expects bool"),
    ("vec3(1) < vec3(2)", "expects int or num"), ("vec2(1) + vec3(1)", "cannot add vec2 and vec3"), ("3 % 2.0", "'%' expects int"),
    ("2 / vec3(1)", "cannot divide int and vec3"), ("'text'", "string"), ("(1", "expected ')'"), ("1 $ 2", "unexpected character"),
    ("P.x" + "".join(" * (1 + P.x" for _ in range(14)) + ")" * 14, "too deeply nested"),
])
def test_errors_name_the_problem(src, what):
    with pytest.raises(RuntimeError, match=what.replace("(", r"\(").replace(")", r"\)").replace("$", r"\$").replace("+", r"\+")):
        ev(src)


def _scene(bsdf, extra=None):
    s = {"technique": {"type": "path", "max_depth": 4}, "camera": {"type": "perspective", "fov": 40, "near_clip": 0.1, "far_clip": 100,
                                                                    "transform": [{"lookat": {"origin": [0, 0, 3], "target": [0, 0, 0], "up": [0, 1, 0]}}]},
         "film": {"size": [64, 64]}, "bsdfs": [bsdf], "shapes": [{"type": "rectangle", "name": "quad", "width": 2, "height": 2}],
         "entities": [{"name": "quad", "shape": "quad", "bsdf": bsdf["name"]}], "lights": [{"type": "env", "name": "sky", "radiance": [1, 1, 1]}]}
    s.update(extra or {})
    return s


def test_loader_folds_constant_expressions_and_keeps_the_rest_as_programs():
    sc = LoadedScene.from_string(json.dumps(_scene({"type": "diffuse", "name": "m", "reflectance": "mix(color(1,0,0), color(0,0,1), 0.25) * k"},
                                                   {"parameters": [{"name": "k", "type": "number", "value": 0.5}]})), SCENES, 64, 64)
    m = sc.scene.materials[0]
    assert m.flags & (1 << 8) == 0 and sc.scene.expr_code_count == 0 and near(list(m.p[0:3]), [0.375, 0, 0.125])
    sc = LoadedScene.from_string(json.dumps(_scene({"type": "diffuse", "name": "m", "reflectance": "color(uv.x, uv.y, 0.5) * tint"},
                                                   {"parameters": [{"name": "tint", "type": "color", "value": [1, 0.5, 0.25]}]})), SCENES, 64, 64)
    m = sc.scene.materials[0]
    assert m.flags & (1 << 8) and m.tex_refl == 0 and sc.scene.expr_code_count > 4
    assert sc.scene.expr_code[sc.scene.expr_code_count - 1] & 0xFF == 0  # IGE_END closes the program
    for bad, what in (("cbrt(P.x) * color(1)", "not supported"), ("uv", "vec2, not a number or colour"), ("nosuchtex", "unknown variable")):
        with pytest.raises(RuntimeError, match=what):
            LoadedScene.from_string(json.dumps(_scene({"type": "diffuse", "name": "m", "reflectance": bad})), SCENES, 64, 64)
    with pytest.raises(RuntimeError, match="inside a blend"):
        s = _scene({"type": "blend", "name": "m", "first": "a", "second": "b", "weight": 0.5})
        s["bsdfs"] += [{"type": "diffuse", "name": "a", "reflectance": "color(uv.x)"}, {"type": "diffuse", "name": "b"}]
        LoadedScene.from_string(json.dumps(s), SCENES, 64, 64)


def test_oracle_expression_checkerboard_equals_the_lowered_checkerboard():
    """The exporters' checkerboard idiom is lowered into the material record; written the other way round it runs through the
    interpreter. Same pixels."""
    import oracle
    a = LoadedScene.from_string(json.dumps(_scene({"type": "diffuse", "name": "m", "reflectance": "select(checkerboard(uvw * 4.0) == 1, color(0.8,0.7,0.1,1), color(0.2,0.2,0.2,1))"})), SCENES, 64, 64)
    b = LoadedScene.from_string(json.dumps(_scene({"type": "diffuse", "name": "m", "reflectance": "select(1 == checkerboard(uvw * 4.0), color(0.8,0.7,0.1,1), color(0.2,0.2,0.2,1))"})), SCENES, 64, 64)
    assert a.scene.materials[0].flags & (1 << 2) and b.scene.materials[0].flags & (1 << 8)
    fa, _ = oracle.render(a, 4, 64, 64, iteration=0, seed=3)
    fb, _ = oracle.render(b, 4, 64, 64, iteration=0, seed=3)
    assert np.array_equal(fa, fb) and fa.max() > 0.5


def _node_brick(u, v, sx, sy, gx, gy):
    """src/artic/texture/brick.art:1-12 in float32 (math::fract = x - floor(x); step(edge, x) = select(x < edge, 0, 1))."""
    su, sv = F(u) * F(sx), F(v) * F(sy)
    fr = lambda a: F(a) - F(np.floor(F(a)))
    x = fr(su + F(0.5) if fr(sv * F(0.5)) > F(0.5) else su)
    y = fr(sv)
    step = lambda edge, a: F(0) if a < edge else F(1)
    return step(x, F(1) - F(gx)) * step(y, F(1) - F(gy))


def _brick_expression(c0, c1, sx, sy, gx, gy):
    col = lambda c: "color(%r, %r, %r)" % tuple(float(F(x)) for x in c)
    su, sv = f"uv.x * {float(F(sx))!r}", f"uv.y * {float(F(sy))!r}"
    return (f"mix({col(c0)}, {col(c1)}, select((1 - {float(F(gx))!r}) < fract(select(fract({sv} * 0.5) > 0.5, {su} + 0.5, {su})), 0.0, 1.0)"
            f" * select((1 - {float(F(gy))!r}) < fract({sv}), 0.0, 1.0))")


def test_brick_texture_is_the_reference_node_and_the_loader_lowers_it_to_that_expression():
    """A "brick" texture (src/runtime/pattern/BrickPattern.cpp:13-38; make_brick_texture, src/artic/texture/brick.art) named by a colour
    property: the expression the loader writes for it equals a float32 restatement of node_brick over a grid of coordinates (negative and
    beyond 1 included), and the scene with the texture renders the very pixels of the scene with that expression written out."""
    import oracle
    c0, c1, sx, sy, gx, gy = (0.1, 0.2, 0.3), (0.9, 0.8, 0.7), 3.0, 6.0, 0.05, 0.1
    src = _brick_expression(c0, c1, sx, sy, gx, gy)
    seen = set()
    for u in np.linspace(-1.3, 2.1, 41):
        for v in np.linspace(-0.7, 1.9, 37):
            t = _node_brick(u, v, sx, sy, gx, gy)
            seen.add(float(t))
            want = tuple(float(F(a) * (F(1) - t) + F(b) * t) for a, b in zip(c0, c1)) + (1.0,)
            ty, got = ev(src, uvw=(float(F(u)), float(F(v)), 0))
            assert ty == "vec4" and near(got, want, 1e-7), (u, v, got, want)
    assert seen == {0.0, 1.0}
    tex = {"textures": [{"type": "brick", "name": "wall", "color0": list(c0), "color1": list(c1)}]}  # the pattern's defaults: scale (3, 6), gap (0.05, 0.1)
    a = LoadedScene.from_string(json.dumps(_scene({"type": "diffuse", "name": "m", "reflectance": "wall"}, tex)), SCENES, 64, 64)
    b = LoadedScene.from_string(json.dumps(_scene({"type": "diffuse", "name": "m", "reflectance": src})), SCENES, 64, 64)
    assert a.scene.materials[0].flags & (1 << 8) and b.scene.materials[0].flags & (1 << 8)
    fa, _ = oracle.render(a, 4, 64, 64, iteration=0, seed=3)
    fb, _ = oracle.render(b, 4, 64, 64, iteration=0, seed=3)
    assert np.array_equal(fa, fb) and 0.05 < fa.min() and fa.max() > 0.5 and len(np.unique(fa.round(3))) > 2


_M2D = ((1.5, 0.25, 0.125), (-0.5, 2.0, 0.75))  # upper-left 2 x 2 block and x / y translation of the 4 x 4 below
_T16 = [1.5, 0.25, 0, 0.125, -0.5, 2.0, 0, 0.75, 0, 0, 1, 0, 0, 0, 0, 1]


def _affine(u, v):
    """mat3x3_transform_point_affine (core/matrix.art:237-240): vec3_dot(row, (u, v, 1)) = fmaf(x, u, fmaf(y, v, z * 1)) per coordinate."""
    fma = lambda a, b, c: F(np.float64(F(a)) * np.float64(F(b)) + np.float64(F(c)))
    return tuple(fma(r[0], u, fma(r[1], v, F(r[2]) * F(1))) for r in _M2D)


def _uv_rows():
    return tuple("dot(vec3(%r, %r, %r), vec3(uv.x, uv.y, 1.0))" % r for r in _M2D)


def test_procedural_textures_under_a_transform():
    """The "transform" of a brick / checkerboard texture (LoaderUtils::inlineTransformAs2d: the 2 x 2 block and x / y translation of the 3D
    transform; mat3x3_transform_point_affine in front of the scale): the loader writes the affine map into the expression. Hand-written
    expressions with the same map against float32 restatements over a grid, and the scenes with the textures against the scenes with those
    expressions, pixel for pixel."""
    import oracle
    c0, c1 = (0.1, 0.2, 0.3), (0.9, 0.8, 0.7)
    tu, tv = _uv_rows()
    col = lambda c: "color(%r, %r, %r)" % tuple(float(F(x)) for x in c)
    brick = (f"mix({col(c0)}, {col(c1)}, select((1 - {float(F(0.05))!r}) < fract(select(fract({tv} * 6.0 * 0.5) > 0.5, {tu} * 3.0 + 0.5, {tu} * 3.0)), 0.0, 1.0)"
             f" * select((1 - {float(F(0.1))!r}) < fract({tv} * 6.0), 0.0, 1.0))")
    check = f"select(checkerboard(vec2({tu} * 4.0, {tv} * 2.0)) == 1, {col(c1)}, {col(c0)})"
    wrap2 = lambda x: F(x) - F(2) * F(np.floor(F(x) / F(2)))  # math::wrap(x, 0, 2) (core/math.art:88-91) for these magnitudes
    for u in np.linspace(-0.9, 1.7, 23):
        for v in np.linspace(-0.4, 1.3, 19):
            U, V = _affine(u, v)
            t = _node_brick(U, V, 3.0, 6.0, 0.05, 0.1)
            want = tuple(float(F(a) * (F(1) - t) + F(b) * t) for a, b in zip(c0, c1)) + (1.0,)
            assert near(ev(brick, uvw=(float(F(u)), float(F(v)), 0))[1], want, 1e-7), (u, v)
            px, py = int(wrap2(U * F(4))) % 2 == 0, int(wrap2(V * F(2))) % 2 == 0  # make_checkerboard_texture (texture/checkerboard.art:4-13)
            want = tuple(float(F(x)) for x in (c0 if px != py else c1)) + (1.0,)
            assert near(ev(check, uvw=(float(F(u)), float(F(v)), 0))[1], want, 1e-7), (u, v)
    for tex, src in (({"type": "brick", "name": "t", "color0": list(c0), "color1": list(c1), "transform": _T16}, brick),
                     ({"type": "checkerboard", "name": "t", "color0": list(c0), "color1": list(c1), "scale_x": 4, "scale_y": 2, "transform": _T16}, check)):
        a = LoadedScene.from_string(json.dumps(_scene({"type": "diffuse", "name": "m", "reflectance": "t"}, {"textures": [tex]})), SCENES, 64, 64)
        b = LoadedScene.from_string(json.dumps(_scene({"type": "diffuse", "name": "m", "reflectance": src})), SCENES, 64, 64)
        assert a.scene.materials[0].flags & (1 << 8) and b.scene.materials[0].flags & (1 << 8)
        fa, _ = oracle.render(a, 4, 64, 64, iteration=0, seed=3)
        fb, _ = oracle.render(b, 4, 64, 64, iteration=0, seed=3)
        assert np.array_equal(fa, fb) and len(np.unique(fa.round(3))) > 2
    # an identity transform leaves the checkerboard in its record form
    ident = {"type": "checkerboard", "name": "t", "color0": list(c0), "color1": list(c1), "transform": [1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1]}
    sc = LoadedScene.from_string(json.dumps(_scene({"type": "diffuse", "name": "m", "reflectance": "t"}, {"textures": [ident]})), SCENES, 64, 64)
    assert sc.scene.materials[0].flags & (1 << 2) and not sc.scene.materials[0].flags & (1 << 8)


def _u32(x):
    return int(x) & 0xFFFFFFFF


def _hash_combine(h, d):
    """core/random.art:7-13 (FNV over the four bytes)"""
    for k in range(4):
        h = _u32(h * 16777619) ^ ((d >> (8 * k)) & 0xFF)
    return h


def _tea(v0, v1):
    """sample_tea_u32, core/random.art:15-24"""
    s = 0
    for _ in range(4):
        s = _u32(s + 0x9e3779b9)
        v0 = _u32(v0 + ((_u32(v1 << 4) + 0xa341316c) ^ _u32(v1 + s) ^ ((v1 >> 5) + 0xc8013ea4)))
        v1 = _u32(v1 + ((_u32(v0 << 4) + 0xad90777d) ^ _u32(v0 + s) ^ ((v0 >> 5) + 0x7e95761e)))
    return v1


def _bits(f):
    return int(np.array(f, F).view(np.uint32))


def _noise2_bits(ub, vb, seed):
    """noise2[T] (texture/noise.art:35-37): the first next_f32 of create_random_generator(hash of seed, u, v) (core/random.art:65-70,82-87)"""
    x = _tea(_hash_combine(_hash_combine(_hash_combine(0x811C9DC5, _bits(seed)), ub), vb), 1)
    return F(np.array((x & 0x7FFFFF) | 0x3F800000, np.uint32).view(F)) - F(1)


def _noise2(kind, u, v, seed):
    u, v, seed = F(u), F(v), F(seed)
    if kind == "cellnoise":  # (:44) integer bits of the truncated coordinates
        return _noise2_bits(_u32(int(u)), _u32(int(v)), seed)
    if kind == "pnoise":  # (:47-59)
        ix, iy = F(int(u)), F(int(v))  # math::trunc = (x as i32) as f32 (core/math.art:73): +0 for -0.18
        sm = lambda x: F(abs(F(F(x * x) * F(F(3) - F(F(2) * x)))))
        kx, ky = sm(F(u - ix)), sm(F(v - iy))
        p = lambda a, b: _noise2_bits(_bits(a), _bits(b), seed)
        lerp = lambda a, b, k: F(F(F(F(1) - k) * a) + F(k * b))
        return lerp(lerp(p(ix, iy), p(F(ix + 1), iy), kx), lerp(p(ix, F(iy + 1)), p(F(ix + 1), F(iy + 1)), kx), ky)
    return _noise2_bits(_bits(u), _bits(v), seed)


def _cnoise2(kind, u, v, seed):
    """cnoise2 / ccellnoise2 / cpnoise2 (:42,45,61-75); the interpolation of cpnoise2 is per channel the one of pnoise2"""
    if kind == "cellnoise":
        u, v, kind = F(int(F(u))), F(int(F(v))), "noise"
    return tuple(float(_noise2(kind, u, v, F(F(seed) + F(o)))) for o in (0, 1234, 5678)) + (1.0,)


def _sperlin2(u, v, seed):
    """sperlin2 (src/artic/texture/noise.art:79-127) in float32, operation by operation; fmaf as one rounding of the exact value."""
    fma = lambda a, b, c: F(np.float64(a) * np.float64(b) + np.float64(c))
    fl = lambda x: F(np.floor(x))
    mod289 = lambda x: F(x - F(fl(F(x / F(289))) * F(289)))
    permute = lambda x: mod289(F(F(F(x * F(34)) * x) + x))
    x = _tea(_hash_combine(_hash_combine(0x811C9DC5, _bits(seed)), 1234), 1)  # noise1(1234, seed): an integer coordinate
    shift = F(np.array((x & 0x7FFFFF) | 0x3F800000, np.uint32).view(F)) - F(1)
    px, py = F(F(u) + shift), F(F(v) + shift)
    pix_, piy_ = fl(px), fl(py)
    piz_, piw_ = F(pix_ + 1), F(piy_ + 1)
    pfx, pfy = F(px - pix_), F(py - piy_)
    pfz, pfw = F(pfx - 1), F(pfy - 1)
    pix, piy, piz, piw = mod289(pix_), mod289(piy_), mod289(piz_), mod289(piw_)
    vix, viy, vfx, vfy = (pix, piz, pix, piz), (piy, piy, piw, piw), (pfx, pfz, pfx, pfz), (pfy, pfy, pfw, pfw)
    inv41 = F(F(1) / F(41))
    gx2, gy = [], []
    for i in range(4):
        vi = permute(F(permute(vix[i]) + viy[i]))
        q = F(vi * inv41)
        gx = F(F(F(q - fl(q)) * F(2)) - F(1))
        gy.append(F(F(abs(gx)) - F(0.5)))
        gx2.append(F(gx - fl(F(gx + F(0.5)))))
    len2 = [fma(gx2[i], gx2[i], F(gy[i] * gy[i])) for i in range(4)]  # lanes: g00, g10, g01, g11
    c0, c1 = F(1.79284291400159), F(0.85373472095314)
    norm = [F(c0 - F(len2[0] * c1)), F(c0 - F(len2[2] * c1)), F(c0 - F(len2[1] * c1)), F(c0 - F(len2[3] * c1))]  # (g00, g01, g10, g11)
    n = [F(fma(gx2[i], vfx[i], F(gy[i] * vfy[i])) * norm[i]) for i in range(4)]  # n00 * norm.x, n10 * norm.y, n01 * norm.z, n11 * norm.w
    sm = lambda t: F(F(F(t * t) * t) * F(F(t * F(F(t * F(6)) - F(15))) + F(10)))
    lerp = lambda a, b, k: F(F(F(F(1) - k) * a) + F(k * b))
    fx, fy = sm(pfx), sm(pfy)
    return F(F(2.3) * lerp(lerp(n[0], n[1], fx), lerp(n[2], n[3], fx), fy))


def test_perlin_noise_equals_the_reference_function():
    """perlin / sperlin / cperlin over a vec2 (Transpiler.cpp:758-761 -> sperlin2, perlin2 = (s + 1) / 2, cperlin2 = cpnoise2 * perlin2,
    src/artic/texture/noise.art:79-128,211-214) against a float32 restatement, and the "perlin" texture against its expression."""
    import oracle
    lo, hi = 1.0, -1.0
    for u in np.linspace(-2.3, 6.1, 15):
        for v in np.linspace(-1.1, 4.7, 13):
            uvw = (float(F(u)), float(F(v)), 0)
            s7 = _sperlin2(u, v, 7.0)
            lo, hi = min(lo, float(s7)), max(hi, float(s7))
            assert near(ev("sperlin(uv, 7)", uvw=uvw)[1], float(s7), 2e-6), (u, v)
            assert near(ev("perlin(uv, 7)", uvw=uvw)[1], float(F(F(s7 + F(1)) / F(2))), 2e-6), (u, v)
            pd = F(F(_sperlin2(u, v, 36326639.0) + F(1)) / F(2))
            assert near(ev("perlin(uv)", uvw=uvw)[1], float(pd), 2e-6), (u, v)
            want = tuple(float(F(F(c) * pd)) for c in _cnoise2("pnoise", u, v, 36326639.0))
            assert near(ev("cperlin(uv)", uvw=uvw)[1], want, 2e-6), (u, v)
    assert lo < -0.3 and hi > 0.3
    tex = {"type": "perlin", "name": "t", "color": [0.9, 0.7, 0.5], "scale_x": 7, "scale_y": 5}
    a = LoadedScene.from_string(json.dumps(_scene({"type": "diffuse", "name": "m", "reflectance": "t"}, {"textures": [tex]})), SCENES, 64, 64)
    b = LoadedScene.from_string(json.dumps(_scene({"type": "diffuse", "name": "m", "reflectance": "color(0.9, 0.7, 0.5) * perlin(vec2(uv.x * 7.0, uv.y * 5.0), 36326639.0)"})), SCENES, 64, 64)
    fa, _ = oracle.render(a, 4, 64, 64, iteration=0, seed=3)
    fb, _ = oracle.render(b, 4, 64, 64, iteration=0, seed=3)
    assert np.array_equal(fa, fb) and len(np.unique(fa.round(3))) > 8 and np.isfinite(fa).all()


def _voronoi2(u, v, seed):
    """voronoi2_f1_gen, Euclidean, randomness 1 (src/artic/texture/voronoi.art:100-119): (distance, cnoise2(nearest cell, seed))"""
    u, v, seed = F(u), F(v), F(seed)
    ip = (F(np.floor(u)), F(np.floor(v)))
    fp = (F(u - ip[0]), F(v - ip[1]))
    fma = lambda a, b, c: F(np.float64(a) * np.float64(b) + np.float64(c))
    dist, target = F(8), None
    for j in (-1, 0, 1):
        for i in (-1, 0, 1):
            k = (F(ip[0] + F(i)), F(ip[1] + F(j)))
            r = (F(_noise2("noise", k[0], k[1], seed) * F(1)), F(_noise2("noise", k[0], k[1], F(seed + F(175391))) * F(1)))
            dx, dy = F(F(F(i) + r[0]) - fp[0]), F(F(F(j) + r[1]) - fp[1])
            d = F(np.sqrt(fma(dx, dx, F(dy * dy))))
            if d < dist:
                dist, target = d, k
    return dist, tuple(_noise2("noise", target[0], target[1], F(seed + F(o))) for o in (0, 1234, 5678)) + (F(1),)


def _fbm2(u, v, seed):
    """fbm2_gen(uv, seed, 6, 2, 0.5, F1, Euclidean) (:221-238): the colour is weighted with the amplitude after its update"""
    s, m, a, p, b = F(0), F(0), F(0.5), (F(u), F(v)), [F(0), F(0), F(0), F(1)]
    for _ in range(6):
        f, c = _voronoi2(p[0], p[1], seed)
        s, m = F(s + F(a * f)), F(m + a)
        a = F(a * F(0.5))
        p = (F(p[0] * F(2)), F(p[1] * F(2)))
        b = [F(b[i] + F(c[i] * a)) for i in range(3)] + [min(F(1), F(b[3] + F(c[3] * a)))]
    return F(s / m), tuple(F(x / m) for x in b)


def test_voronoi_and_fbm_over_one_and_three_coordinates():
    """voronoi1_f1_gen / voronoi3_f1_gen and fbm over them (src/artic/texture/voronoi.art:46-63,158-181,240-257): three / twenty-seven cells, the
    feature point's coordinates drawn with the seed shifted by 0 / 175391 / 822167, |d| / vec3_len as the distance."""
    import functools
    hashed = lambda cbs, seed: F(np.array((_tea(functools.reduce(_hash_combine, cbs, _hash_combine(0x811C9DC5, _bits(seed))), 1) & 0x7FFFFF) | 0x3F800000, np.uint32).view(F)) - F(1)
    white = lambda xs, seed: hashed([_bits(x) for x in xs], F(seed))
    fma = lambda a, b, c: F(np.float64(a) * np.float64(b) + np.float64(c))

    def voronoi(xs, seed):
        xs, seed = [F(x) for x in xs], F(seed)
        n = len(xs)
        ip = [F(np.floor(x)) for x in xs]
        fp = [F(x - i) for x, i in zip(xs, ip)]
        dist, target = F(8), [F(0)] * n
        for c in range(3 ** n):
            g = [F((c // 3 ** i) % 3 - 1) for i in range(n)]
            k = [F(ip[i] + g[i]) for i in range(n)]
            dd = [F(F(g[i] + F(white(k, F(seed + F((0, 175391, 822167)[i]))) * F(1))) - fp[i]) for i in range(n)]
            d = F(abs(dd[0])) if n == 1 else F(np.sqrt(fma(dd[0], dd[0], fma(dd[1], dd[1], F(dd[2] * dd[2])))))
            if d < dist:
                dist, target = d, k
        return dist, tuple(white(target, F(seed + F(o))) for o in (0, 1234, 5678)) + (F(1),)

    def fbm(xs, seed):
        s, m, a, p, b = F(0), F(0), F(0.5), [F(x) for x in xs], [F(0), F(0), F(0), F(1)]
        for _ in range(6):
            f, c = voronoi(p, seed)
            s, m = F(s + F(a * f)), F(m + a)
            a = F(a * F(0.5))
            p = [F(x * F(2)) for x in p]
            b = [F(b[i] + F(c[i] * a)) for i in range(3)] + [min(F(1), F(b[3] + F(c[3] * a)))]
        return F(s / m), tuple(F(x / m) for x in b)
    for x in np.linspace(-2.4, 3.8, 7):
        for y in (-1.3, 0.45, 2.6):
            P = (float(F(x)), float(F(y)), float(F(0.7 * x + 0.2 * y)))
            d, c = voronoi(P, 3.0)
            assert near(ev("voronoi(P, 3)", P=P)[1], float(d), 1e-6) and near(ev("cvoronoi(P, 3)", P=P)[1], tuple(float(v) for v in c), 1e-7), P
            d1, c1 = voronoi(P[:1], 36326639.0)
            assert near(ev("voronoi(P.x)", P=P)[1], float(d1), 1e-6) and near(ev("cvoronoi(P.x)", P=P)[1], tuple(float(v) for v in c1), 1e-7), P
            f, fc = fbm(P, 2.0)
            assert near(ev("fbm(P, 2)", P=P)[1], float(f), 2e-6) and near(ev("cfbm(P, 2)", P=P)[1], tuple(float(v) for v in fc), 2e-6), P
    with pytest.raises(RuntimeError, match="not supported"):
        ev("gabor(P)", P=(1, 2, 3))  # (the reference has it over a vec2 only)
    # the reference has no fbm over one coordinate (Transpiler.cpp:790-795: fbm2 / fbm3 only), and its argument-list forms are refused by name
    for bad in ("fbm(P.x)", "cfbm(P.x, 3)"):
        with pytest.raises(RuntimeError, match="not supported|no function"):
            ev(bad, P=(1, 2, 3))
    for bad, what in (("fbm(uv, 1, 4, 2.0, 0.5)", "fbm with its parameters as arguments"), ("gabor(uv, 1, 32, 0.05, 0.5, 0.0)", "gabor with its parameters as arguments"),
                      ("voronoi(P, 1, 1.0, 'euclidean', 'f1')", "voronoi with its parameters as arguments")):
        with pytest.raises(RuntimeError, match=what):
            ev(bad, P=(1, 2, 3))
    for x in (0.0, 1.0, -3.25, 36326639.0, 0.1):  # hash(x) = hash_rndf (core/random.art:91-93): the generator's first float for the seed hash(bits(x))
        want = F(np.array((_tea(_hash_combine(0x811C9DC5, _bits(F(x))), 1) & 0x7FFFFF) | 0x3F800000, np.uint32).view(F)) - F(1)
        assert near(ev("hash(P.x)", P=(x, 0, 0))[1], float(want), 1e-7), x


def test_voronoi_and_fbm_equal_the_reference_functions():
    """voronoi / cvoronoi / fbm / cfbm over a vec2 (Transpiler.cpp:769-792 -> voronoi2 / cvoronoi2 / fbm2 / cfbm2, src/artic/texture/voronoi.art)
    against a float32 restatement, and the "voronoi" / "fbm" textures (NoisePattern.cpp, make_[c]voronoi_texture / make_[c]fbm_texture) against
    their expressions."""
    import oracle
    for u in np.linspace(-2.7, 4.9, 9):
        for v in np.linspace(-1.6, 3.3, 7):
            uvw = (float(F(u)), float(F(v)), 0)
            d, c = _voronoi2(u, v, 5.0)
            assert near(ev("voronoi(uv, 5)", uvw=uvw)[1], float(d), 1e-6) and near(ev("cvoronoi(uv, 5)", uvw=uvw)[1], tuple(float(x) for x in c), 1e-7), (u, v)
            f, fc = _fbm2(u, v, 36326639.0)
            assert near(ev("fbm(uv)", uvw=uvw)[1], float(f), 2e-6) and near(ev("cfbm(uv)", uvw=uvw)[1], tuple(float(x) for x in fc), 2e-6), (u, v)
    for tex, src in (({"type": "voronoi", "name": "t", "color": [0.9, 0.8, 0.6], "scale_x": 6, "scale_y": 4}, "color(0.9, 0.8, 0.6) * voronoi(vec2(uv.x * 6.0, uv.y * 4.0), 36326639.0)"),
                     ({"type": "fbm", "name": "t", "colored": True, "scale_x": 3, "scale_y": 3, "seed": 9}, "color(1, 1, 1) * cfbm(vec2(uv.x * 3.0, uv.y * 3.0), 9.0)")):
        a = LoadedScene.from_string(json.dumps(_scene({"type": "diffuse", "name": "m", "reflectance": "t"}, {"textures": [tex]})), SCENES, 64, 64)
        b = LoadedScene.from_string(json.dumps(_scene({"type": "diffuse", "name": "m", "reflectance": src})), SCENES, 64, 64)
        fa, _ = oracle.render(a, 4, 64, 64, iteration=0, seed=3)
        fb, _ = oracle.render(b, 4, 64, 64, iteration=0, seed=3)
        assert np.array_equal(fa, fb) and len(np.unique(fa.round(3))) > 8 and np.isfinite(fa).all(), tex["type"]


def test_gabor_noise_equals_the_reference_function():
    """gabor(vec2[, seed]) = gabor2_gen(uv, seed, 100, 20, 5, 0.01) (Transpiler.cpp:796-797, src/artic/texture/noise.art:131-150) against a
    restatement with numpy's exp / cos / sin / arctan2 (the backend's deterministic ones differ from libm in the last places: 100 terms)."""
    pi = F(3.14159265359)
    for seed, src in ((36326639.0, "gabor(uv)"), (4.0, "gabor(uv, 4)")):
        for u, v in ((0.3, 0.4), (1.7, -0.6), (-2.2, 3.1), (5.5, 0.25)):
            acc = F(0)
            for i in range(100):
                n = [_noise2_bits(i, k, F(seed)) for k in range(4)]  # noise2(i, k, seed): integer coordinates
                od = F(F(np.arctan2(n[2], n[3])) * F(20))
                ox, oy = F(np.cos(od)), F(np.sin(od))
                ln = F(np.sqrt(F(F(n[0] * n[0]) + F(n[1] * n[1]))))
                kk = F(np.exp(F(F(F(-ln) * F(0.01)) * pi)))
                dot = F(F(F(F(u) - n[0]) * ox) + F(F(F(v) - n[1]) * oy))
                acc = F(acc + F(kk * F(np.cos(F(F(F(F(2) * pi) * F(5)) * dot)))))
            want = float(F(acc / F(np.sqrt(F(100)))))
            got = ev(src, uvw=(float(F(u)), float(F(v)), 0))[1]
            assert abs(got - want) < 2e-4, (src, u, v, got, want)


def test_expr_texture_with_custom_variables():
    """An "expr" texture (ExprPattern.cpp:14-75): its expression with the num_ / color_ / vec_ / bool_ properties as variables, named by a colour
    property — the same pixels as the expression written out with the values in place."""
    import oracle
    tex = {"type": "expr", "name": "t", "expr": "mix(base, color(shift.x, shift.y, shift.z) * pnoise(uv * freq), select(flip, 0.25, 0.75))",
           "num_freq": 6, "color_base": [0.9, 0.2, 0.1], "vec_shift": [0.1, 0.5, 0.9], "bool_flip": True}
    src = "mix(color(0.9, 0.2, 0.1), color(0.1, 0.5, 0.9) * pnoise(uv * 6.0), select(true, 0.25, 0.75))"
    a = LoadedScene.from_string(json.dumps(_scene({"type": "diffuse", "name": "m", "reflectance": "t"}, {"textures": [tex]})), SCENES, 64, 64)
    b = LoadedScene.from_string(json.dumps(_scene({"type": "diffuse", "name": "m", "reflectance": src})), SCENES, 64, 64)
    assert a.scene.materials[0].flags & (1 << 8) and b.scene.materials[0].flags & (1 << 8)
    fa, _ = oracle.render(a, 4, 64, 64, iteration=0, seed=3)
    fb, _ = oracle.render(b, 4, 64, 64, iteration=0, seed=3)
    assert np.array_equal(fa, fb) and len(np.unique(fa.round(3))) > 8
    with pytest.raises(RuntimeError, match="requires an expression"):
        LoadedScene.from_string(json.dumps(_scene({"type": "diffuse", "name": "m", "reflectance": "t"}, {"textures": [{"type": "expr", "name": "t"}]})), SCENES, 64, 64)
    with pytest.raises(RuntimeError, match="unknown variable 'freq'"):  # (the variables are the texture's own)
        LoadedScene.from_string(json.dumps(_scene({"type": "diffuse", "name": "m", "reflectance": "pnoise(uv * freq)"}, {"textures": [tex]})), SCENES, 64, 64)


def test_hash_noises_equal_the_reference_functions_and_the_textures_are_lowered_to_them():
    """noise / cellnoise / pnoise and their colour forms over a number, vec2 or vec3 (Transpiler.cpp:734-789 -> src/artic/texture/noise.art:2-75,152-244)
    against a Python restatement of hash_combine, sample_tea_u32 and the generator's first float, on a grid with negative coordinates, with
    and without a seed; then "noise" / "cellnoise" / "pnoise" textures (NoisePattern.cpp:33-57: color * func(uv * scale, seed), "colored")
    against the scenes with those expressions written out."""
    import oracle
    default = 36326639.0
    for u in np.linspace(-3.7, 9.3, 14):
        for v in np.linspace(-2.2, 7.9, 11):
            uvw = (float(F(u)), float(F(v)), 0)
            for kind in ("noise", "cellnoise", "pnoise"):
                assert near(ev(f"{kind}(uv)", uvw=uvw)[1], float(_noise2(kind, u, v, default)), 1e-7), (kind, u, v)
                assert near(ev(f"{kind}(uv, 7.5)", uvw=uvw)[1], float(_noise2(kind, u, v, 7.5)), 1e-7), (kind, u, v)
                assert near(ev(f"c{kind}(uv, 3)", uvw=uvw)[1], _cnoise2(kind, u, v, 3.0), 1e-7), (kind, u, v)
    assert near(ev("snoise(uv, 2)", uvw=(0.3, 0.7, 0))[1], float(_noise2("noise", 0.3, 0.7, 2.0) * F(2) - F(1)), 1e-7)  # snoise2 (:40)
    vals = [ev("noise(uv)", uvw=(float(x), 0.5, 0))[1] for x in np.linspace(0, 1, 200)]
    assert 0 <= min(vals) < 0.05 and 0.95 < max(vals) < 1 and 0.4 < np.mean(vals) < 0.6
    with pytest.raises(RuntimeError, match="not supported"):
        ev("perlin(P)", P=(1, 2, 3))  # (the reference has the gradient noise over a vec2 only)
    # the forms over one and three coordinates (noise1 / noise3, cellnoise, pnoise: src/artic/texture/noise.art:2-33,152-206)
    def noise_n(kind, xs, seed):
        xs, seed = [F(x) for x in xs], F(seed)
        hashed = lambda cbs: F(np.array((_tea(__import__("functools").reduce(_hash_combine, cbs, _hash_combine(0x811C9DC5, _bits(seed))), 1) & 0x7FFFFF) | 0x3F800000, np.uint32).view(F)) - F(1)
        if kind == "cellnoise":
            return hashed([_u32(int(x)) for x in xs])
        if kind == "pnoise":
            ip = [F(int(x)) for x in xs]
            sm = lambda x: F(abs(F(F(x * x) * F(F(3) - F(F(2) * x)))))
            k = [sm(F(x - i)) for x, i in zip(xs, ip)]
            lerp = lambda a, b, t: F(F(F(F(1) - t) * a) + F(t * b))
            p = [hashed([_bits(F(ip[i] + 1) if (c >> i) & 1 else ip[i]) for i in range(len(xs))]) for c in range(1 << len(xs))]
            for i in range(len(xs)):
                p = [lerp(p[2 * c], p[2 * c + 1], k[i]) for c in range(len(p) // 2)]
            return p[0]
        return hashed([_bits(x) for x in xs])
    for x in np.linspace(-2.6, 5.3, 9):
        for y in np.linspace(-1.4, 3.9, 5):
            z = 0.37 * x - 1.1 * y
            P = (float(F(x)), float(F(y)), float(F(z)))
            for kind in ("noise", "cellnoise", "pnoise"):
                assert near(ev(f"{kind}(P, 4)", P=P)[1], float(noise_n(kind, P, 4.0)), 1e-7), (kind, P)
                assert near(ev(f"{kind}(P.x)", P=P)[1], float(noise_n(kind, P[:1], default)), 1e-7), (kind, P)
                assert near(ev(f"{kind}(uv)", uvw=P)[1], float(_noise2(kind, P[0], P[1], default)), 1e-7), (kind, P)  # (and the two-coordinate form again)
            want = tuple(float(noise_n("pnoise", P, F(F(2.0) + F(o)))) for o in (0, 1234, 5678)) + (1.0,)
            assert near(ev("cpnoise(P, 2)", P=P)[1], want, 1e-7), P
            assert near(ev("snoise(P)", P=P)[1], float(noise_n("noise", P, default) * F(2) - F(1)), 1e-7), P
    for tex, src in (({"type": "noise", "name": "t", "color": [0.9, 0.5, 0.3], "scale_x": 40, "scale_y": 30}, "color(0.9, 0.5, 0.3) * noise(vec2(uv.x * 40.0, uv.y * 30.0), 36326639.0)"),
                     ({"type": "cellnoise", "name": "t", "seed": 11, "colored": True}, "color(1, 1, 1) * ccellnoise(vec2(uv.x * 10.0, uv.y * 10.0), 11.0)"),
                     ({"type": "pnoise", "name": "t", "scale_x": 6, "scale_y": 6, "transform": _T16}, "color(1, 1, 1) * pnoise(vec2(%s * 6.0, %s * 6.0), 36326639.0)" % _uv_rows())):
        a = LoadedScene.from_string(json.dumps(_scene({"type": "diffuse", "name": "m", "reflectance": "t"}, {"textures": [tex]})), SCENES, 64, 64)
        b = LoadedScene.from_string(json.dumps(_scene({"type": "diffuse", "name": "m", "reflectance": src})), SCENES, 64, 64)
        assert a.scene.materials[0].flags & (1 << 8) and b.scene.materials[0].flags & (1 << 8)
        fa, _ = oracle.render(a, 4, 64, 64, iteration=0, seed=3)
        fb, _ = oracle.render(b, 4, 64, 64, iteration=0, seed=3)
        assert np.array_equal(fa, fb) and len(np.unique(fa.round(3))) > 8, tex["type"]


def test_oracle_transform_bsdf_with_the_plain_normal_changes_nothing_and_a_tilted_one_does():
    """make_normal_set (bsdf/map.art:36-42) with normal = N aligns the frame with itself; tilting the normal re-weights the cosine."""
    import oracle
    base = {"type": "diffuse", "name": "inner", "reflectance": [0.8, 0.8, 0.8]}
    light = {"lights": [{"type": "directional", "name": "d", "direction": [0, 0, -1], "irradiance": [3, 3, 3]}]}

    def render(normal):
        s = _scene({"type": "transform", "name": "m", "bsdf": "inner", "normal": normal}, light)
        s["bsdfs"].append(base)
        sc = LoadedScene.from_string(json.dumps(s), SCENES, 64, 64)
        assert sc.scene.materials[0].flags & (1 << 9)
        return oracle.render(sc, 4, 64, 64, iteration=0, seed=5)[0]
    s0 = _scene(dict(base, name="m"), light)
    plain = oracle.render(LoadedScene.from_string(json.dumps(s0), SCENES, 64, 64), 4, 64, 64, iteration=0, seed=5)[0]
    same = render("N")
    assert np.allclose(same, plain, rtol=1e-5, atol=1e-6)
    tilted = render("norm(N + Nx)")  # 45 degrees: the directional light's cosine drops to 1 / sqrt(2)
    c = slice(24, 40)
    assert abs(tilted[c, c].mean() / plain[c, c].mean() - math.sqrt(0.5)) < 0.01
    const = render([1, 0, 1])  # a constant world-space vector works as well
    assert np.allclose(const[c, c].mean(), tilted[c, c].mean(), rtol=1e-3)


def test_blend_and_mask_weights_can_be_expressions():
    """BlendBSDF.cpp:40 / MaskBSDF.cpp:30-55: "weight" goes through ShadingTree::addNumber; "cutoff" compares it with a threshold."""
    import oracle
    s = _scene({"type": "blend", "name": "m", "first": "a", "second": "b", "weight": "uv.x"})
    s["bsdfs"] += [{"type": "diffuse", "name": "a", "reflectance": [0, 0, 0]}, {"type": "diffuse", "name": "b", "reflectance": [1, 1, 1]}]
    sc = LoadedScene.from_string(json.dumps(s), SCENES, 64, 64)
    m = sc.scene.materials[0]
    assert m.bsdf_type == 6 and m.flags & (1 << 10) and m.tex_id >= 0
    fb, _ = oracle.render(sc, 16, 64, 64, iteration=0, seed=3)
    g = fb.mean(axis=2)
    left, right = g[24:40, 12:20].mean(), g[24:40, 44:52].mean()
    assert right > left * 2 > 0  # the white side of the blend grows with u
    s["bsdfs"][0]["weight"] = "0.25 * 2"
    folded = LoadedScene.from_string(json.dumps(s), SCENES, 64, 64)
    assert folded.scene.materials[0].flags & (1 << 10) == 0 and folded.scene.materials[0].p[0] == 0.5
    s["bsdfs"][0] = {"type": "cutoff", "name": "m", "bsdf": "a", "weight": "fract(uv.y * 4)", "cutoff": 0.5}
    cut = LoadedScene.from_string(json.dumps(s), SCENES, 64, 64)
    assert cut.scene.materials[0].flags & (1 << 10)
    fb, _ = oracle.render(cut, 8, 64, 64, iteration=0, seed=3)
    col = fb.mean(axis=(1, 2))[12:52]
    assert (col < 0.2).mean() > 0.3 and (col > 0.9).mean() > 0.3  # stripes: the black masked BSDF where the weight is below the threshold, see-through to the white environment elsewhere
    s["bsdfs"][0] = {"type": "mask", "name": "m", "bsdf": "a", "weight": "uv"}
    with pytest.raises(RuntimeError, match="not a number"):
        LoadedScene.from_string(json.dumps(s), SCENES, 64, 64)
