"""The Mitsuba XML -> Ignis JSON converter (ignis_amd/mts.py), the `.xml` leg of the reference's `igutil convert`
(src/tools/util/MtsConverter.cpp): what is written follows export_scene there; the result loads and renders."""
import json
import os
import shutil

import numpy as np
import pytest

from conftest import SCENES

SCENE_V2 = """<?xml version="1.0" encoding="utf-8"?>
<scene version="2.1.0">
  <default name="spp" value="16"/>
  <default name="res" value="64"/>
  <integrator type="path"><integer name="max_depth" value="6"/></integrator>
  <sensor type="perspective">
    <float name="fov" value="60"/>
    <transform name="to_world"><lookat origin="2.5, 0, 0" target="0, 0, 0" up="0, 0, -1"/></transform>
    <sampler type="independent"><integer name="sample_count" value="$spp"/></sampler>
    <film type="hdrfilm"><integer name="width" value="$res"/><integer name="height" value="$res"/></film>
  </sensor>
  <bsdf type="twosided" id="wall"><bsdf type="diffuse"><rgb name="reflectance" value="0.8, 0.7, 0.6"/></bsdf></bsdf>
  <bsdf type="roughplastic" id="plastic"><rgb name="diffuse_reflectance" value="0.2 0.4 0.8"/><float name="alpha" value="0.15"/>
    <string name="int_ior" value="polypropylene"/></bsdf>
  <bsdf type="conductor" id="metal"><string name="material" value="Cu"/></bsdf>
  <shape type="obj"><string name="filename" value="Room.obj"/><ref id="wall"/></shape>
  <shape type="sphere"><point name="center" x="0" y="0.3" z="0.5"/><float name="radius" value="0.3"/><ref id="plastic"/></shape>
  <shape type="cube"><transform name="to_world"><scale value="0.2"/><translate x="0" y="-0.4" z="0.6"/></transform><ref id="metal"/></shape>
  <shape type="rectangle">
    <transform name="to_world"><scale x="0.3" y="0.3" z="1"/><rotate x="1" y="0" z="0" angle="180"/><translate x="0" y="0" z="-0.9"/></transform>
    <emitter type="area"><rgb name="radiance" value="20, 18, 15"/></emitter>
  </shape>
  <emitter type="constant"><rgb name="radiance" value="0.3"/></emitter>
</scene>
"""

SCENE_V0 = """<scene version="0.6.0">
  <integrator type="volpath"><integer name="maxDepth" value="12"/></integrator>
  <sensor type="perspective"><float name="fov" value="45"/><string name="fovAxis" value="x"/>
    <transform name="toWorld"><matrix value="1 0 0 0  0 1 0 0  0 0 1 -4  0 0 0 1"/></transform>
    <film type="hdrfilm"><integer name="width" value="32"/><integer name="height" value="24"/></film></sensor>
  <texture type="bitmap" id="tex"><string name="filename" value="wood.png"/></texture>
  <medium type="homogeneous" id="fog"><rgb name="sigmaA" value="0.1"/><spectrum name="sigmaS" value="0.4"/></medium>
  <bsdf type="diffuse" id="d"><ref name="reflectance" id="tex"/></bsdf>
  <bsdf type="dielectric" id="glass"><string name="intIOR" value="bk7"/><float name="extIOR" value="1.0"/></bsdf>
  <shape type="shapegroup" id="group">
    <shape type="serialized"><string name="filename" value="m.serialized"/><integer name="shapeIndex" value="2"/><ref id="d"/></shape>
    <shape type="ply"><string name="filename" value="b.ply"/><ref id="glass"/></shape>
  </shape>
  <shape type="instance"><ref id="group"/><transform name="toWorld"><translate x="1" y="2" z="3"/></transform></shape>
  <shape type="sphere"><float name="radius" value="2"/><ref name="interior" id="fog"/></shape>
  <emitter type="sunsky"><float name="turbidity" value="3"/></emitter>
  <emitter type="point"><point name="position" x="0" y="5" z="0"/><blackbody name="intensity" temperature="5000" scale="2"/></emitter>
</scene>
"""


def _convert(tmp_path, text, **defines):
    from ignis_amd import mts
    p = tmp_path / "scene.xml"
    p.write_text(text)
    return mts.convert_file(str(p), defines)


def test_converted_scene_follows_the_reference_export(tmp_path):
    d = _convert(tmp_path, SCENE_V2, res="48")
    assert d["technique"] == {"type": "path", "max_depth": 6}
    assert d["film"] == {"size": [48, 48]}  # -D overrides <default>
    cam = d["camera"]
    assert cam["type"] == "perspective" and cam["fov"] == 60 and "to_world" not in cam
    T = np.float64(cam["transform"]).reshape(4, 4)  # Mitsuba lookat: columns left, up, direction, origin
    np.testing.assert_allclose(T[:3, 2], [-1, 0, 0], atol=1e-12)
    np.testing.assert_allclose(T[:3, 3], [2.5, 0, 0], atol=1e-12)
    np.testing.assert_allclose(T[:3, 1], [0, 0, -1], atol=1e-12)
    names = [b["name"] for b in d["bsdfs"]]
    assert names[:2] == ["__black", "__pass"] and d["bsdfs"][1]["type"] == "passthrough"
    two = d["bsdfs"][2]
    assert two["type"] == "twosided" and two["bsdf"] == "__bsdf_1" and d["bsdfs"][3]["reflectance"] == [0.8, 0.7, 0.6]
    plastic = next(b for b in d["bsdfs"] if b["type"] == "roughplastic")
    assert plastic["int_ior"] == 1.49 and plastic["alpha"] == 0.15  # named IOR looked up (MtsConverter.cpp:30-57)
    metal = next(b for b in d["bsdfs"] if b["type"] == "conductor")
    assert (metal["eta"], metal["k"]) == (1.040, 2.583) and "material" not in metal
    assert [s["type"] for s in d["shapes"]] == ["obj", "sphere", "cube", "rectangle"]
    np.testing.assert_allclose(np.float64(d["shapes"][2]["transform"]).reshape(4, 4)[:3, 3], [0, -0.4, 0.6])  # scale, then translate
    ents = d["entities"]
    assert [e["bsdf"] for e in ents] == ["__bsdf_0", "__bsdf_2", "__bsdf_3", "__black"]  # the emitter's shape has no bsdf
    assert d["lights"][0] == {"name": "__light_0", "type": "area", "entity": "__entity_3", "radiance": [20, 18, 15]}
    assert d["lights"][1]["type"] == "constant" and d["lights"][1]["radiance"] == [0.3, 0.3, 0.3]


def test_old_dialect_groups_media_and_special_emitters(tmp_path):
    d = _convert(tmp_path, SCENE_V0)
    assert d["technique"] == {"type": "volpath", "max_depth": 12}  # camelCase of Mitsuba 0.6 -> snake_case
    assert d["camera"]["fov_axis"] == "x" and d["camera"]["transform"][11] == -4
    assert d["textures"] == [{"name": "__texture_0", "type": "bitmap", "filename": "wood.png"}]
    assert next(b for b in d["bsdfs"] if b["type"] == "diffuse" and b["name"] != "__black")["reflectance"] == "__texture_0"
    glass = next(b for b in d["bsdfs"] if b["type"] == "dielectric")
    assert glass["int_ior"] == 1.5046 and glass["ext_ior"] == 1.0
    assert d["media"] == [{"name": "__medium_0", "type": "homogeneous", "sigma_a": [0.1] * 3, "sigma_s": [0.4] * 3}]
    assert [s["type"] for s in d["shapes"]] == ["mitsuba", "ply", "sphere"] and d["shapes"][0]["shape_index"] == 2
    ents = {e["name"]: e for e in d["entities"]}
    assert set(ents) == {"__entity_0_0", "__entity_0_1", "__entity_1"}  # the instanced group's two members, then the sphere
    assert ents["__entity_0_0"]["transform"][3::4][:3] == [1, 2, 3] and ents["__entity_0_1"]["shape"] == "__shape_1"
    assert ents["__entity_1"]["bsdf"] == "__pass" and ents["__entity_1"]["inner_medium"] == "__medium_0"
    lights = d["lights"]
    assert [(l["name"], l["type"]) for l in lights[:2]] == [("__light_0_sun", "sun"), ("__light_0_sky", "sky")] and lights[0]["turbidity"] == 3
    assert lights[2]["type"] == "point" and lights[2]["intensity"] == "blackbody(5000)*2"


def test_converter_errors(tmp_path):
    from ignis_amd import mts
    with pytest.raises(mts.MtsError, match="undefined parameter"):
        _convert(tmp_path, '<scene version="2.0.0"><integrator type="path"><integer name="max_depth" value="$depth"/></integrator></scene>')
    with pytest.raises(mts.MtsError, match="uniform spectra"):
        _convert(tmp_path, '<scene version="2.0.0"><bsdf type="diffuse"><spectrum name="reflectance" value="400:0.1, 700:0.9"/></bsdf></scene>')
    with pytest.raises(mts.MtsError, match="names nothing"):
        _convert(tmp_path, '<scene version="2.0.0"><shape type="cube"><ref id="nope"/></shape></scene>')
    assert mts.main([str(tmp_path / "missing.xml")]) == 1


def test_converted_scene_loads_and_renders(tmp_path):
    """The JSON the converter writes for a small Mitsuba scene is a scene the loader accepts as it is; rendered through the
    oracle it shows the lit room (and, command line: the same file through `python -m ignis_amd.mts`)."""
    import oracle
    from ignis_amd import mts
    from ignis_amd.tables import LoadedScene
    shutil.copy(os.path.join(SCENES, "meshes", "Room.obj"), tmp_path / "Room.obj")
    (tmp_path / "scene.xml").write_text(SCENE_V2)
    assert mts.main([str(tmp_path / "scene.xml"), "-D", "res=40"]) == 0
    data = json.load(open(tmp_path / "scene.json"))
    assert data == mts.convert_file(str(tmp_path / "scene.xml"), {"res": "40"})
    sc = LoadedScene.from_file(str(tmp_path / "scene.json"))
    assert (sc.scene.film_width, sc.scene.film_height) == (40, 40) and sc.scene.entity_count == 4 and sc.scene.light_count == 2
    types = sorted(sc.scene.materials[i].bsdf_type for i in range(sc.scene.material_count))
    assert types == [0, 0, 2, 4]  # wall (two-sided diffuse), the emitter's black, conductor, plastic
    fb = np.mean([oracle.render(sc, 16, 40, 40, iteration=i, seed=2)[0] for i in range(4)], axis=0)
    assert np.isfinite(fb).all() and 0.05 < fb.mean() < 5
    assert fb[18:22, 18:22].mean() > fb[:4, :4].mean() * 0.2  # something is lit in the middle of the room
