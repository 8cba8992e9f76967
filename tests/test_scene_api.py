"""Scene / SceneObject / SceneProperty / SceneParser and loadFromScene of the reference's Python API
(src/frontend/python/scene.cpp:21-145, runtime.cpp:340-350) on the host side: typing rules, getters, and that a scene lowered
through the objects gives the native loader's tables of the file itself, byte for byte. Rendering one is a GPU test."""
import ctypes as C
import json
import os

import numpy as np
import pytest

import ignis_amd
from ignis_amd import Scene, SceneObject, SceneParser, SceneProperty
from ignis_amd.tables import LoadedScene

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCENES = os.path.join(ROOT, "scenes")


def _table_bytes(ls):
    """Every table of igd_scene as bytes, keyed by field name (scalars as their value)."""
    t, out = ls.scene, {}
    sizes = {"entities": t.entity_count * 36 * 4, "shape_data": t.shape_data_size, "primbvh": t.primbvh_size,
             "scene_nodes": t.scene_node_count * C.sizeof(t.scene_nodes._type_), "scene_leaves": t.scene_leaf_count * C.sizeof(t.scene_leaves._type_),
             "materials": t.material_count * C.sizeof(t.materials._type_), "lights": t.light_count * C.sizeof(t.lights._type_),
             "textures": t.texture_count * C.sizeof(t.textures._type_), "texture_data": t.texture_data_size, "cdf_data": t.cdf_data_count * 4,
             "expr_code": t.expr_code_count * 4, "light_cdf": t.light_cdf_count * 4}
    for name, n in sizes.items():
        p = getattr(t, name)
        out[name] = C.string_at(p, n) if n and p else b""
    out["camera"] = bytes(t.camera)
    out["technique"] = bytes(t.technique)
    out["film"] = (t.film_width, t.film_height)
    return out


def test_property_typing_follows_the_parser():
    """getProperty (loader/Parser.cpp:284-322) and the getters' defaults (SceneProperty.h:45-60: an Integer reads as a Number)."""
    T = SceneProperty.Type
    cases = [(True, T.Bool), ("x", T.String), (3, T.Integer), (3.5, T.Number), ([1, 2], T.Vector2), ([1, 2, 3], T.Vector3),
             ([1, 0, 0, 0, 1, 0, 0, 0, 1], T.Transform), (list(range(12)), T.Transform), (list(range(16)), T.Transform),
             ([{"translate": [1, 2, 3]}], T.Transform), ({"scale": 2}, T.Transform), ({"type": "integer", "values": [1, 2]}, T.IntegerArray),
             ({"values": [1.5]}, T.NumberArray), ([1, 2, 3, 4], T["None"]), (None, T["None"])]
    for v, want in cases:
        assert SceneProperty.fromJSON(v).type == want, v
    assert SceneProperty.fromInteger(4).getNumber(9.0) == 4.0 and SceneProperty.fromNumber(4.5).getInteger(7) == 7
    assert SceneProperty.fromString("a").getNumber(2.0) == 2.0 and not SceneProperty().isValid() and SceneProperty.fromInteger(1).canBeNumber()
    assert SceneProperty.fromVector3([1, 2, 3]).getVector3() == (1.0, 2.0, 3.0) and SceneProperty.fromBool(1).getBool() is True
    m = SceneProperty.fromJSON([{"translate": [0.5, 0, 0]}, {"scale": 2}]).getTransform()
    np.testing.assert_array_equal(m, np.array([[2, 0, 0, 0.5], [0, 2, 0, 0], [0, 0, 2, 0], [0, 0, 0, 1]], np.float32))
    look = SceneProperty.fromJSON([{"lookat": {"origin": [0, 0, 4], "target": [0, 0, 0], "up": [0, 1, 0]}}]).getTransform()
    np.testing.assert_allclose(look[:3, 2], [0, 0, -1], atol=1e-7)  # the view direction is the third column (Parser.cpp:142-170)
    np.testing.assert_array_equal(look[:3, 3], [0, 0, 4])
    rot = SceneProperty.fromJSON([{"rotate": [0, 0, 90]}]).getTransform()
    np.testing.assert_allclose(rot[:3, :3] @ [1, 0, 0], [0, 1, 0], atol=1e-6)
    np.testing.assert_array_equal(SceneProperty.fromTransform(np.eye(4)).toJSON(), np.eye(4).reshape(-1))
    with pytest.raises(ValueError):
        SceneProperty.fromJSON([{"shear": 1}])


def test_scene_objects_and_containers():
    s = Scene()
    assert s.camera is None and s.bsdfs == {}
    o = SceneObject(SceneObject.Type.Bsdf, "diffuse", "/tmp")
    o["reflectance"] = SceneProperty.fromVector3([0.1, 0.2, 0.3])
    assert "reflectance" in o and o.hasProperty("reflectance") and not o["missing"].isValid()
    assert (o.type, o.pluginType, o.baseDir) == (SceneObject.Type.Bsdf, "diffuse", "/tmp") and list(o.properties) == ["reflectance"]
    with pytest.raises(TypeError):
        o.setProperty("x", 3)
    s.addBSDF("a", o)
    s.addConstantEnvLight()
    s.addConstantEnvLight()
    assert list(s.lights) == ["__env"] and s.light("__env")["radiance"].getNumber() == 1 and s.bsdf("a") is o and s.bsdf("b") is None
    other = Scene()
    other.camera = SceneObject(SceneObject.Type.Camera, "perspective")
    other.addShape("sh", SceneObject(SceneObject.Type.Shape, "rectangle"))
    s.addFrom(other)  # Scene.cpp:5-28
    assert s.camera is other.camera and list(s.shapes) == ["sh"] and list(s.bsdfs) == ["a"]
    d = s.toJSON()
    assert d["bsdfs"] == [{"type": "diffuse", "name": "a", "reflectance": [0.1, 0.2, 0.3]}] and d["camera"] == {"type": "perspective"}
    mesh = SceneObject(SceneObject.Type.Shape, "ply", "/data")
    mesh["filename"] = SceneProperty.fromString("m/x.ply")
    assert mesh.toJSON("m")["filename"] == "/data/m/x.ply"


@pytest.mark.parametrize("name", ["diamond_scene.json", "many_point_lights.json", "diamond_scene_principled.json"])
def test_parsed_scene_lowers_to_the_tables_of_the_file(name):
    path = os.path.join(SCENES, name)
    s = Scene.loadFromFile(path)
    doc = json.load(open(path))
    assert len(s.entities) == len(doc["entities"]) and len(s.bsdfs) == len(doc["bsdfs"]) and s.technique.pluginType == doc["technique"]["type"]
    want = _table_bytes(LoadedScene.from_file(path, 64, 48))
    got = _table_bytes(LoadedScene.from_string(json.dumps(s.toJSON()), "", 64, 48))  # (file names were made absolute by the objects' baseDir)
    assert got == want


def test_parser_flags_and_programmatic_edit():
    path = os.path.join(SCENES, "diamond_scene.json")
    F = SceneParser.Flags
    part = SceneParser().loadFromFile(path, F.F_LoadCamera | F.F_LoadFilm)
    assert part.camera is not None and part.film is not None and part.technique is None and not part.entities and not part.bsdfs
    assert SceneParser.F_LoadAll == F.F_LoadAll
    # a scene edited through the objects equals the same edit in the JSON
    s = Scene.loadFromFile(path)
    name = next(iter(s.bsdfs))
    plugin = s.bsdf(name).pluginType
    repl = SceneObject(SceneObject.Type.Bsdf, "diffuse", "")
    repl["reflectance"] = SceneProperty.fromVector3([0.25, 0.5, 0.75])
    s.addBSDF(name, repl)
    doc = json.load(open(path))
    for b in doc["bsdfs"]:
        if b["name"] == name:
            assert b["type"] == plugin
            b.clear()
            b.update({"type": "diffuse", "name": name, "reflectance": [0.25, 0.5, 0.75]})
    want = _table_bytes(LoadedScene.from_string(json.dumps(doc), SCENES, 64, 48))
    got = _table_bytes(LoadedScene.from_string(json.dumps(s.toJSON()), "", 64, 48))
    assert got == want
    with pytest.raises(TypeError):
        ignis_amd.loadFromScene({"camera": {}})
