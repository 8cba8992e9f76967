"""Scene description objects of the reference's Python API, for `loadFromScene`.

Mirrors `Scene`, `SceneObject`, `SceneProperty` and `SceneParser` of the nanobind module (src/frontend/python/scene.cpp:21-145 over
src/runtime/Scene.cpp, SceneObject.h, SceneProperty.h, loader/Parser.cpp:284-337): a scene assembled object by object (or parsed,
then edited) and handed to `ignis_amd.loadFromScene(scene[, dir][, opts])` (runtime.cpp:340-350, Runtime::loadFromScene,
Runtime.cpp:219). There is one loader in this backend -- the native host library behind include/igh_host.h -- so a `Scene` is
lowered to the JSON text that loader reads (`Scene.toJSON`), property for property; what was parsed from JSON is written back as it
was read, so parse -> toJSON -> load gives the tables of loading the file directly.
"""
import enum
import json
import math
import os

import numpy as np


def _mat4(values):
    a = np.asarray(values, dtype=np.float64).reshape(-1)
    m = np.eye(4)
    if a.size == 9:
        m[:3, :3] = a.reshape(3, 3)
    elif a.size == 12:
        m[:3, :4] = a.reshape(3, 4)
    elif a.size == 16:
        m[:, :] = a.reshape(4, 4)
    else:
        raise ValueError("a transform matrix has 9, 12 or 16 entries")
    return m


def _angle_axis(deg, axis):
    t = math.radians(deg)
    c, s = math.cos(t), math.sin(t)
    x, y, z = axis
    r = np.eye(4)
    r[:3, :3] = [[c + x * x * (1 - c), x * y * (1 - c) - z * s, x * z * (1 - c) + y * s],
                 [y * x * (1 - c) + z * s, c + y * y * (1 - c), y * z * (1 - c) - x * s],
                 [z * x * (1 - c) - y * s, z * y * (1 - c) + x * s, c + z * z * (1 - c)]]
    return r


def _look_at(eye, center, up):
    """Parser.cpp:142-170."""
    eye, center, up = (np.asarray(v, dtype=np.float64) for v in (eye, center, up))
    f = center - eye
    f = f / np.linalg.norm(f) if np.dot(f, f) > 1.1920928955e-07 else np.array([0.0, 0.0, 1.0])
    u = up / np.linalg.norm(up)
    s = np.cross(f, u)
    s = s / np.linalg.norm(s)
    u = np.cross(s, f)
    m = np.eye(4)
    m[:3, 0], m[:3, 1], m[:3, 2], m[:3, 3] = s, u, f, eye
    return m


def _transform_ops(ops):
    """Parser.cpp:172-282: a list (or, deprecated, one object) of translate / scale / rotate / qrotate / lookat / matrix entries,
    applied left to right."""
    m = np.eye(4)
    for op in ([ops] if isinstance(ops, dict) else ops):
        for name, val in op.items():
            t = np.eye(4)
            if name == "translate":
                t[:3, 3] = val
            elif name == "scale":
                t[:3, :3] = np.diag([val] * 3 if isinstance(val, (int, float)) else list(val))
            elif name == "rotate":
                t = _angle_axis(val[0], (1, 0, 0)) @ _angle_axis(val[1], (0, 1, 0)) @ _angle_axis(val[2], (0, 0, 1))
            elif name == "qrotate":
                w, x, y, z = val
                t[:3, :3] = [[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                             [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                             [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]]
            elif name == "lookat":
                origin = val.get("origin", [0, 0, 0])
                target = np.add(val["direction"], origin) if "direction" in val else val.get("target", [0, 1, 0])
                t = _look_at(origin, target, val.get("up", [0, 0, 1]))
            elif name == "matrix":
                t = _mat4(val)
            else:
                raise ValueError(f"Transform property got unknown entry type '{name}'")
            m = m @ t
    return m


class SceneProperty:
    """SceneProperty.h:14-273: a typed value; the getters return their default when the type does not match (an Integer reads as a
    Number, :45-60)."""

    Type = enum.IntEnum("Type", ["None", "Bool", "Integer", "Number", "String", "Transform", "Vector2", "Vector3", "IntegerArray",
                                 "NumberArray"], start=0)

    def __init__(self, type=None, value=None, raw=None):
        self._type = SceneProperty.Type["None"] if type is None else type
        self._value = value
        self._raw = raw  # the JSON this was parsed from, written back as it was (transform op lists, typed arrays)

    type = property(lambda self: self._type)

    def isValid(self):
        return self._type != SceneProperty.Type["None"]

    def canBeNumber(self):
        return self._type in (SceneProperty.Type.Number, SceneProperty.Type.Integer)

    def _get(self, type, default):
        return self._value if self._type == type else default

    def getBool(self, default=False):
        return self._get(SceneProperty.Type.Bool, default)

    def getInteger(self, default=0):
        return self._get(SceneProperty.Type.Integer, default)

    def getNumber(self, default=0.0):
        return float(self._value) if self.canBeNumber() else default

    def getString(self, default=""):
        return self._get(SceneProperty.Type.String, default)

    def getVector2(self, default=(0.0, 0.0)):
        return self._get(SceneProperty.Type.Vector2, tuple(default))

    def getVector3(self, default=(0.0, 0.0, 0.0)):
        return self._get(SceneProperty.Type.Vector3, tuple(default))

    def getTransform(self, default=None):
        if self._type != SceneProperty.Type.Transform:
            return np.eye(4, dtype=np.float32) if default is None else np.asarray(default, dtype=np.float32)
        return self._value.astype(np.float32)

    def getIntegerArray(self):
        return list(self._get(SceneProperty.Type.IntegerArray, []))

    def getNumberArray(self):
        return list(self._get(SceneProperty.Type.NumberArray, []))

    fromBool = staticmethod(lambda v: SceneProperty(SceneProperty.Type.Bool, bool(v)))
    fromInteger = staticmethod(lambda v: SceneProperty(SceneProperty.Type.Integer, int(v)))
    fromNumber = staticmethod(lambda v: SceneProperty(SceneProperty.Type.Number, float(v)))
    fromString = staticmethod(lambda v: SceneProperty(SceneProperty.Type.String, str(v)))
    fromVector2 = staticmethod(lambda v: SceneProperty(SceneProperty.Type.Vector2, tuple(float(x) for x in np.asarray(v).reshape(2))))
    fromVector3 = staticmethod(lambda v: SceneProperty(SceneProperty.Type.Vector3, tuple(float(x) for x in np.asarray(v).reshape(3))))
    fromTransform = staticmethod(lambda m: SceneProperty(SceneProperty.Type.Transform, _mat4(m)))
    fromIntegerArray = staticmethod(lambda v: SceneProperty(SceneProperty.Type.IntegerArray, [int(x) for x in v]))
    fromNumberArray = staticmethod(lambda v: SceneProperty(SceneProperty.Type.NumberArray, [float(x) for x in v]))

    @staticmethod
    def fromJSON(v):
        """getProperty, Parser.cpp:284-322; an invalid property for what the reference does not type either."""
        T = SceneProperty.Type
        if isinstance(v, bool):
            return SceneProperty(T.Bool, v)
        if isinstance(v, str):
            return SceneProperty(T.String, v)
        if isinstance(v, int):
            return SceneProperty(T.Integer, v)
        if isinstance(v, float):
            return SceneProperty(T.Number, v)
        if isinstance(v, list):
            if v and isinstance(v[0], dict):
                return SceneProperty(T.Transform, _transform_ops(v), raw=v)
            if len(v) == 2:
                return SceneProperty(T.Vector2, tuple(float(x) for x in v), raw=v)
            if len(v) == 3:
                return SceneProperty(T.Vector3, tuple(float(x) for x in v), raw=v)
            if len(v) in (9, 12, 16):
                return SceneProperty(T.Transform, _mat4(v), raw=v)
        elif isinstance(v, dict):
            if "values" in v:  # handleArrayProperty: {"type": "integer" | "number", "values": [...]}
                integer = str(v.get("type", "number")).lower().startswith("int")
                return SceneProperty(T.IntegerArray if integer else T.NumberArray, list(v["values"]), raw=v)
            return SceneProperty(T.Transform, _transform_ops(v), raw=v)  # the deprecated object form of a transform
        return SceneProperty()

    def toJSON(self):
        T = SceneProperty.Type
        if self._raw is not None:
            return self._raw
        if self._type == T.Transform:
            return [float(x) for x in self._value.reshape(-1)]  # 16 entries, row-major (Parser.cpp:306-307)
        if self._type in (T.Vector2, T.Vector3):
            return list(self._value)
        if self._type == T.IntegerArray:
            return {"type": "integer", "values": list(self._value)}
        if self._type == T.NumberArray:
            return {"type": "number", "values": list(self._value)}
        return self._value

    def __repr__(self):
        return f"SceneProperty({self._type.name}, {self._value!r})"


class SceneObject:
    """SceneObject.h: a plugin type, the directory its relative paths start from, named properties."""

    Type = enum.IntEnum("Type", ["Bsdf", "Camera", "Entity", "Film", "Light", "Medium", "Shape", "Technique", "Texture", "Parameter"], start=0)

    def __init__(self, type, pluginType="", baseDir=""):
        self._type = SceneObject.Type(type)
        self._plugin = str(pluginType)
        self._base = str(baseDir)
        self._props = {}

    type = property(lambda self: self._type)
    pluginType = property(lambda self: self._plugin)
    baseDir = property(lambda self: self._base)
    properties = property(lambda self: dict(self._props))

    def property(self, name):
        return self._props.get(name, SceneProperty())

    def setProperty(self, name, prop):
        if not isinstance(prop, SceneProperty):
            raise TypeError("setProperty expects a SceneProperty")
        self._props[str(name)] = prop

    def hasProperty(self, name):
        return name in self._props

    __getitem__ = property
    __setitem__ = setProperty
    __contains__ = hasProperty

    @staticmethod
    def fromJSON(type, obj, baseDir=""):
        """handleAnonymousObject / populateObject (Parser.cpp:324-351): "type" and "name" are not properties."""
        if "type" in obj and not isinstance(obj["type"], str):
            raise ValueError("Expected type to be a string")
        o = SceneObject(type, obj.get("type", ""), baseDir)
        for k, v in obj.items():
            if k in ("name", "type"):
                continue
            p = SceneProperty.fromJSON(v)
            if p.isValid():
                o._props[k] = p
        return o

    def toJSON(self, name=None):
        d = {}
        if self._plugin:
            d["type"] = self._plugin
        if name is not None:
            d["name"] = name
        for k, p in self._props.items():
            v = p.toJSON()
            # file names are relative to the object's directory (SceneObject::baseDir), the loader's to the one directory it is given
            if self._base and p.type == SceneProperty.Type.String and k in ("filename", "file") and not os.path.isabs(v):
                v = os.path.join(self._base, v)
            d[k] = v
        return d


class SceneParser:
    """loader/Parser.h: which parts of a description `loadFromFile` / `loadFromString` keep."""

    class Flags(enum.IntFlag):
        F_LoadCamera = 0x1
        F_LoadFilm = 0x2
        F_LoadTechnique = 0x4
        F_LoadBSDFs = 0x8
        F_LoadMedia = 0x10
        F_LoadLights = 0x20
        F_LoadTextures = 0x40
        F_LoadShapes = 0x80
        F_LoadEntities = 0x100
        F_LoadExternals = 0x200
        F_LoadAll = 0x3FF

    locals().update(Flags.__members__)

    def loadFromFile(self, path, flags=Flags.F_LoadAll):
        with open(path, "r") as f:
            return self.loadFromString(f.read(), os.path.dirname(os.path.abspath(str(path))), flags)

    def loadFromString(self, text, dir="", flags=Flags.F_LoadAll):
        doc = json.loads(text)
        if not isinstance(doc, dict):
            raise ValueError("Expected root element to be an object")
        F = SceneParser.Flags
        s = Scene()
        base = str(dir)
        for key, flag, typ, setter in (("camera", F.F_LoadCamera, SceneObject.Type.Camera, "setCamera"),
                                       ("technique", F.F_LoadTechnique, SceneObject.Type.Technique, "setTechnique"),
                                       ("film", F.F_LoadFilm, SceneObject.Type.Film, "setFilm")):
            if key in doc and flags & flag:
                getattr(s, setter)(SceneObject.fromJSON(typ, doc[key], base))
        for key, flag, typ, adder in _LISTS:
            if not flags & flag:
                continue
            for obj in doc.get(key, []):
                if not isinstance(obj.get("name"), str):
                    raise ValueError("Expected name to be a string")
                getattr(s, adder)(obj["name"], SceneObject.fromJSON(typ, obj, base))
        if flags & F.F_LoadExternals:
            # the native loader resolves externals itself (merge + override, csrc/host/loader.cpp); they travel as they are
            s._externals = [dict(e, filename=e["filename"] if os.path.isabs(e.get("filename", "")) or not base else os.path.join(base, e["filename"]))
                            if isinstance(e, dict) and "filename" in e else e for e in doc.get("externals", [])]
        return s


_LISTS = (("textures", SceneParser.Flags.F_LoadTextures, SceneObject.Type.Texture, "addTexture"),
          ("bsdfs", SceneParser.Flags.F_LoadBSDFs, SceneObject.Type.Bsdf, "addBSDF"),
          ("shapes", SceneParser.Flags.F_LoadShapes, SceneObject.Type.Shape, "addShape"),
          ("lights", SceneParser.Flags.F_LoadLights, SceneObject.Type.Light, "addLight"),
          ("media", SceneParser.Flags.F_LoadMedia, SceneObject.Type.Medium, "addMedium"),
          ("entities", SceneParser.Flags.F_LoadEntities, SceneObject.Type.Entity, "addEntity"),
          ("parameters", SceneParser.Flags.F_LoadAll, SceneObject.Type.Parameter, "addParameter"))


class Scene:
    """Scene.h / Scene.cpp: the three singular objects and the named ones, in insertion order."""

    def __init__(self):
        self._camera = self._technique = self._film = None
        self._maps = {k: {} for k, _, _, _ in _LISTS}
        self._externals = []

    camera = property(lambda self: self._camera, lambda self, o: self.setCamera(o))
    technique = property(lambda self: self._technique, lambda self, o: self.setTechnique(o))
    film = property(lambda self: self._film, lambda self, o: self.setFilm(o))

    def setCamera(self, o):
        self._camera = o

    def setTechnique(self, o):
        self._technique = o

    def setFilm(self, o):
        self._film = o

    def _add(self, kind, name, obj):
        if not isinstance(obj, SceneObject):
            raise TypeError("expected a SceneObject")
        self._maps[kind][str(name)] = obj

    def addTexture(self, name, obj):
        self._add("textures", name, obj)

    def addBSDF(self, name, obj):
        self._add("bsdfs", name, obj)

    def addShape(self, name, obj):
        self._add("shapes", name, obj)

    def addLight(self, name, obj):
        self._add("lights", name, obj)

    def addMedium(self, name, obj):
        self._add("media", name, obj)

    def addEntity(self, name, obj):
        self._add("entities", name, obj)

    def addParameter(self, name, obj):
        self._add("parameters", name, obj)

    texture = lambda self, name: self._maps["textures"].get(name)    # noqa: E731
    bsdf = lambda self, name: self._maps["bsdfs"].get(name)          # noqa: E731
    shape = lambda self, name: self._maps["shapes"].get(name)        # noqa: E731
    light = lambda self, name: self._maps["lights"].get(name)        # noqa: E731
    medium = lambda self, name: self._maps["media"].get(name)        # noqa: E731
    entity = lambda self, name: self._maps["entities"].get(name)     # noqa: E731
    textures = property(lambda self: dict(self._maps["textures"]))
    bsdfs = property(lambda self: dict(self._maps["bsdfs"]))
    shapes = property(lambda self: dict(self._maps["shapes"]))
    lights = property(lambda self: dict(self._maps["lights"]))
    media = property(lambda self: dict(self._maps["media"]))
    entities = property(lambda self: dict(self._maps["entities"]))
    parameters = property(lambda self: dict(self._maps["parameters"]))

    def addFrom(self, other):
        """Scene.cpp:5-28: everything named, and the singular objects the other scene has."""
        for kind, m in other._maps.items():
            self._maps[kind].update(m)
        self._externals += other._externals
        self._technique = other._technique or self._technique
        self._camera = other._camera or self._camera
        self._film = other._film or self._film

    def addConstantEnvLight(self):
        """Scene.cpp:30-37."""
        if "__env" not in self._maps["lights"]:
            env = SceneObject(SceneObject.Type.Light, "constant", "")
            env.setProperty("radiance", SceneProperty.fromNumber(1))
            self.addLight("__env", env)

    @staticmethod
    def loadFromFile(path, flags=SceneParser.Flags.F_LoadAll):
        return SceneParser().loadFromFile(path, flags)

    @staticmethod
    def loadFromString(text, dir="", flags=SceneParser.Flags.F_LoadAll):
        return SceneParser().loadFromString(text, dir, flags)

    def toJSON(self):
        """The description the native loader reads (igh_load_string)."""
        d = {}
        for key, o in (("technique", self._technique), ("camera", self._camera), ("film", self._film)):
            if o is not None:
                d[key] = o.toJSON()
        if self._externals:
            d["externals"] = list(self._externals)
        for kind, m in self._maps.items():
            if m:
                d[kind] = [o.toJSON(name) for name, o in m.items()]
        return d
