"""Mitsuba scene description (XML, 0.6 and 2 / 3 dialects) -> Ignis scene JSON: the counterpart of `igutil convert` for `.xml`
input (src/tools/util/MtsConverter.cpp:128-960; the reference parses with the un-vendored TinyParser-Mitsuba library, this module
with the standard library's ElementTree).

What is written follows export_scene (MtsConverter.cpp:796-955) line by line: the sensor becomes "camera" (+ "film" from its nested
film), the integrator "technique", every texture / bsdf / medium / shape a named entry `__texture_i`, `__bsdf_i`, `__medium_i`,
`__shape_i` with its plugin type and its properties copied under their own names (`to_world` -> `transform`, :23-28; named IORs
and conductor materials looked up, :30-126,305-333; `serialized` -> `mitsuba`, :416-418), every shape an entity `__entity_i` with
its bsdf (or `__black` / `__pass`, :607-613) and media, shape groups instanced, area emitters lights of type "area" on their
entity, other emitters lights under their plugin type (`sunsky` split into sun and sky, :482-499).

    python -m ignis_amd.mts scene.xml -o scene.json [-D name=value ...]

Not handled (an error names the element): non-uniform spectra (the reference maps them through CIE tables), animations.
"""
import json
import math
import os
import re
import sys
import xml.etree.ElementTree as ET

OBJECT_TAGS = {"bsdf", "emitter", "shape", "texture", "medium", "sensor", "film", "integrator", "sampler", "rfilter", "phase", "volume", "subsurface"}

# MtsConverter.cpp:30-73
IOR = {"vacuum": 1.0, "air": 1.00028, "glass": 1.55, "diamond": 2.419, "bromine": 1.661, "helium": 1.00004, "water ice": 1.31,
       "hydrogen": 1.00013, "fused quartz": 1.458, "pyrex": 1.470, "carbon dioxide": 1.00045, "acrylic glass": 1.49, "water": 1.3330,
       "polypropylene": 1.49, "acetone": 1.36, "bk7": 1.5046, "ethanol": 1.361, "sodium chloride": 1.544, "carbon tetrachloride": 1.461,
       "amber": 1.55, "glycerol": 1.4729, "pet": 1.575, "benzene": 1.501, "silicone oil": 1.52045, "none": 0.0}
CONDUCTOR_ETA = {"ag": 0.129, "au": 0.402, "cu": 1.040, "none": 0.0}
CONDUCTOR_K = {"ag": 3.250, "au": 2.540, "cu": 2.583, "none": 1.0}


class MtsError(ValueError):
    pass


def _snake(name):
    """Mitsuba 0.6 writes camelCase (toWorld, intIOR, maxDepth); the exported names are the snake_case ones of Mitsuba 2."""
    s = re.sub(r"(?<=[a-z0-9])([A-Z])", r"_\1", name)
    s = re.sub(r"([A-Z]+)([A-Z][a-z])", r"\1_\2", s)
    return s.lower()


class Obj:
    def __init__(self, tag, plugin, ident):
        self.tag, self.plugin, self.id = tag, plugin, ident
        self.props = {}      # name -> python value (float, int, bool, str, [r, g, b], 16 floats for a transform)
        self.children = []   # (name or None, Obj), in document order


def _numbers(text):
    return [float(t) for t in re.split(r"[\s,]+", text.strip()) if t]


def _mat_mul(a, b):
    return [sum(a[r * 4 + k] * b[k * 4 + c] for k in range(4)) for r in range(4) for c in range(4)]


def _identity():
    return [1.0 if r == c else 0.0 for r in range(4) for c in range(4)]


def _vec3(el, subst, default=None):
    if "value" in el.attrib:
        v = _numbers(subst(el.attrib["value"]))
        return v * 3 if len(v) == 1 else v[:3]
    d = default if default is not None else 0.0
    return [float(subst(el.attrib.get(k, str(d)))) for k in ("x", "y", "z")]


def _normalize(v):
    n = math.sqrt(sum(x * x for x in v))
    return [x / n for x in v] if n > 0 else v


def _cross(a, b):
    return [a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]]


def _transform(el, subst):
    """<transform>: the listed operations applied in order (each one multiplies from the left); row-major 4 x 4."""
    m = _identity()
    for op in el:
        if op.tag == "matrix":
            v = _numbers(subst(op.attrib["value"]))
            if len(v) == 9:
                v = [v[0], v[1], v[2], 0, v[3], v[4], v[5], 0, v[6], v[7], v[8], 0, 0, 0, 0, 1]
            if len(v) != 16:
                raise MtsError("<matrix> needs 9 or 16 values")
            t = v
        elif op.tag == "translate":
            x, y, z = _vec3(op, subst)
            t = [1, 0, 0, x, 0, 1, 0, y, 0, 0, 1, z, 0, 0, 0, 1]
        elif op.tag == "scale":
            x, y, z = _vec3(op, subst, 1.0)
            t = [x, 0, 0, 0, 0, y, 0, 0, 0, 0, z, 0, 0, 0, 0, 1]
        elif op.tag == "rotate":
            ax = _normalize(_vec3(op, subst))
            ang = math.radians(float(subst(op.attrib.get("angle", "0"))))
            c, s = math.cos(ang), math.sin(ang)
            x, y, z = ax
            t = [c + x * x * (1 - c), x * y * (1 - c) - z * s, x * z * (1 - c) + y * s, 0,
                 y * x * (1 - c) + z * s, c + y * y * (1 - c), y * z * (1 - c) - x * s, 0,
                 z * x * (1 - c) - y * s, z * y * (1 - c) + x * s, c + z * z * (1 - c), 0, 0, 0, 0, 1]
        elif op.tag in ("lookat", "lookAt"):
            o = _numbers(subst(op.attrib["origin"]))
            tg = _numbers(subst(op.attrib["target"]))
            up = _numbers(subst(op.attrib.get("up", "0, 1, 0")))
            d = _normalize([tg[i] - o[i] for i in range(3)])
            left = _normalize(_cross(up, d))
            nup = _cross(d, left)
            t = [left[0], nup[0], d[0], o[0], left[1], nup[1], d[1], o[1], left[2], nup[2], d[2], o[2], 0, 0, 0, 1]
        else:
            raise MtsError(f"transform operation <{op.tag}> is not supported")
        m = _mat_mul([float(x) for x in t], m)
    return m


def parse(path, defines=None):
    """-> (scene Obj, version tuple). `<include>` files are read relative to the including file."""
    defaults = dict(defines or {})
    by_id = {}

    def subst(text):
        def rep(mo):
            k = mo.group(1)
            if k not in defaults:
                raise MtsError(f"undefined parameter ${k}")
            return str(defaults[k])
        return re.sub(r"\$(\w+)", rep, text)

    def read(el, base_dir, old):
        plugin = subst(el.attrib.get("type", ""))
        o = Obj(el.tag, plugin, el.attrib.get("id"))
        for ch in el:
            tag = ch.tag
            name = ch.attrib.get("name")
            key = (_snake(name) if old else name) if name is not None else None
            if tag == "default":
                defaults.setdefault(ch.attrib["name"], ch.attrib["value"])
            elif tag == "alias":
                by_id[ch.attrib["as"]] = by_id[ch.attrib["id"]]
            elif tag == "include":
                inc = os.path.join(base_dir, subst(ch.attrib["filename"]))
                sub = ET.parse(inc).getroot()
                inner = read(sub, os.path.dirname(inc), old)
                o.children += inner.children
            elif tag == "ref":
                rid = subst(ch.attrib["id"])
                if rid not in by_id:
                    raise MtsError(f"<ref id='{rid}'> names nothing defined before it")
                o.children.append((key, by_id[rid]))
            elif tag in OBJECT_TAGS:
                c = read(ch, base_dir, old)
                if c.id:
                    by_id[c.id] = c
                o.children.append((key, c))
            elif tag in ("float", "integer", "boolean", "string"):
                v = subst(ch.attrib["value"])
                o.props[key] = float(v) if tag == "float" else int(float(v)) if tag == "integer" else (v.strip().lower() == "true") if tag == "boolean" else v
            elif tag in ("rgb", "srgb", "color"):
                v = subst(ch.attrib["value"]).strip()
                if v.startswith("#"):
                    c = [int(v[i:i + 2], 16) / 255.0 for i in (1, 3, 5)]
                else:
                    c = _numbers(v)
                    c = c * 3 if len(c) == 1 else c[:3]
                o.props[key] = c
            elif tag == "spectrum":
                v = subst(ch.attrib.get("value", "")).strip()
                if not v or ":" in v or "filename" in ch.attrib:
                    raise MtsError(f"spectrum '{name}': only uniform spectra are supported")
                c = _numbers(v)
                o.props[key] = [c[0]] * 3 if len(c) == 1 else c[:3]
            elif tag == "blackbody":
                t = float(subst(ch.attrib["temperature"]))
                sc = float(subst(ch.attrib.get("scale", "1")))
                o.props[key] = f"blackbody({t:g})" + (f"*{sc:g}" if sc != 1 else "")  # MtsConverter.cpp:137-143
            elif tag in ("point", "vector"):
                o.props[key] = _vec3(ch, subst)
            elif tag == "transform":
                o.props[key] = _transform(ch, subst)
            elif tag == "animation":
                raise MtsError("animations are not supported")
            else:
                raise MtsError(f"element <{tag}> is not supported")
        return o

    root = ET.parse(path).getroot()
    if root.tag != "scene":
        raise MtsError("not a Mitsuba scene (no <scene> root)")
    version = tuple(int(x) for x in re.findall(r"\d+", root.attrib.get("version", "2.0.0"))[:3])
    return read(root, os.path.dirname(os.path.abspath(path)), version[0] == 0), version


def _rename(name):
    return "transform" if name == "to_world" else name  # translate(), MtsConverter.cpp:23-28


def _unique(objs):
    seen, out = set(), []
    for o in objs:
        if id(o) not in seen:
            seen.add(id(o))
            out.append(o)
    return out


def convert(scene):
    """Obj tree of a <scene> -> the dict igcli reads as JSON (export_scene, MtsConverter.cpp:796-955)."""
    out = {}
    kids = lambda o, tag: [(n, c) for n, c in o.children if c.tag == tag]  # noqa: E731

    # ---- sensor (+ film), integrator (:237-273,799-813)
    for _, c in scene.children:
        if c.tag == "sensor":
            out["camera"] = dict({"type": c.plugin}, **{_rename(k): v for k, v in c.props.items()})
            for _, f in kids(c, "film"):
                out["film"] = {"size": [int(f.props.get("width", 0)), int(f.props.get("height", 0))]}
        elif c.tag == "integrator":
            out["technique"] = dict({"type": c.plugin}, **{_rename(k): v for k, v in c.props.items()})

    # ---- the object lists, in the order of the extract_* functions (:615-794); one entry per object
    def textures_of(o, acc):
        acc += [c for _, c in kids(o, "texture")]
        for n, c in o.children:
            if n is None and c.tag in ("bsdf", "emitter", "shape"):
                textures_of(c, acc)
        for n, c in o.children:
            if n is not None and c.tag in ("bsdf", "emitter"):  # (named nested bsdfs carry textures too)
                textures_of(c, acc)
        return acc

    def bsdfs_of(o, acc):
        for _, c in kids(o, "bsdf"):
            acc.append(c)
            bsdfs_of(c, acc)
        for n, c in o.children:
            if c.tag == "shape":
                bsdfs_of(c, acc)
        return acc

    def media_of(o, acc):
        for _, c in kids(o, "medium"):
            acc.append(c)
        for _, c in kids(o, "shape"):
            media_of(c, acc)
        return acc

    def shapes_of(o, acc):
        for _, c in kids(o, "shape"):
            if c.plugin == "shapegroup":
                shapes_of(c, acc)
            elif c.plugin != "instance":
                acc.append(c)
        return acc

    textures = _unique(textures_of(scene, []))
    bsdfs = _unique(bsdfs_of(scene, []))
    media = _unique(media_of(scene, []))
    shapes = _unique(shapes_of(scene, []))
    tex_id = {id(o): f"__texture_{i}" for i, o in enumerate(textures)}
    bsdf_id = {id(o): f"__bsdf_{i}" for i, o in enumerate(bsdfs)}
    medium_id = {id(o): f"__medium_{i}" for i, o in enumerate(media)}
    shape_id = {id(o): f"__shape_{i}" for i, o in enumerate(shapes)}

    def with_textures(entry, o):
        for n, c in kids(o, "texture"):
            if id(c) not in tex_id:  # a texture nested where the exporter does not collect them (media, emitters, shapes)
                raise MtsError(f"<{o.tag} type=\"{o.plugin}\">: nested texture '{n}' is not supported there")
            entry[n if n is not None else "texture"] = tex_id[id(c)]
        return entry

    if textures:  # export_texture (:275-292)
        out["textures"] = [dict({"name": tex_id[id(t)], "type": t.plugin}, **{_rename(k): v for k, v in t.props.items()}) for t in textures]

    # ---- bsdfs (export_bsdf, :294-373), after the two the entities fall back to
    out["bsdfs"] = [{"name": "__black", "type": "diffuse"}, {"name": "__pass", "type": "passthrough"}]
    for b in bsdfs:
        e = {"name": bsdf_id[id(b)], "type": b.plugin}
        for k, v in b.props.items():
            if k == "material":
                continue
            if k in ("int_ior", "ext_ior") and isinstance(v, str):
                v = IOR.get(v.lower(), 0.0)
            e[_rename(k)] = v
        if isinstance(b.props.get("material"), str):  # "we lose color information" (:324)
            m = b.props["material"].lower()
            e["eta"], e["k"] = CONDUCTOR_ETA.get(m, 0.0), CONDUCTOR_K.get(m, 1.0)
        with_textures(e, b)
        for n, c in kids(b, "bsdf"):
            e[n if n is not None else "bsdf"] = bsdf_id[id(c)]
        out["bsdfs"].append(e)

    if media:  # export_medium (:375-412)
        out["media"] = [with_textures(dict({"name": medium_id[id(m)], "type": m.plugin}, **{_rename(k): v for k, v in m.props.items()}), m) for m in media]

    # ---- shapes and entities (:414-438,548-613,725-768)
    if shapes:
        out["shapes"] = []
        for s in shapes:
            e = {"name": shape_id[id(s)], "type": "mitsuba" if s.plugin == "serialized" else s.plugin}
            e.update({_rename(k): v for k, v in s.props.items()})
            out["shapes"].append(e)
        entities = []  # (transform, [shape objects])
        for _, c in kids(scene, "shape"):
            if c.plugin == "instance":
                group = [g for n, g in c.children if g.tag == "shape" and g.plugin == "shapegroup"]
                if group:
                    entities.append((c.props.get("to_world", _identity()), shapes_of(group[0], [])))
            elif c.plugin != "shapegroup":
                entities.append((_identity(), [c]))
        out["entities"] = []
        entity_of_shape = {}
        for i, (transform, members) in enumerate(entities):
            for k, s in enumerate(members):
                name = f"__entity_{i}" if len(members) == 1 else f"__entity_{i}_{k}"
                entity_of_shape.setdefault(id(s), name)
                e = {"name": name, "shape": shape_id[id(s)], "transform": transform}
                has_bsdf = has_media = False
                for n, c in kids(s, "bsdf"):
                    e[n if n is not None else "bsdf"] = bsdf_id[id(c)]
                    has_bsdf = True
                for n, c in kids(s, "medium"):
                    if n in ("exterior", "interior"):
                        e["outer_medium" if n == "exterior" else "inner_medium"] = medium_id[id(c)]
                        has_media = True
                if not has_bsdf:
                    e["bsdf"] = "__pass" if has_media else "__black"
                out["entities"].append(e)

    # ---- lights: area emitters of shapes first, then the scene's own emitters (:440-546,770-794,927-954)
    lights = []
    for _, s in kids(scene, "shape"):
        for _, em in kids(s, "emitter"):
            # (the reference numbers the entity by the shape's index; the name of the shape's own entity is what is meant)
            e = {"name": f"__light_{len(lights)}", "type": "area", "entity": entity_of_shape.get(id(s), "__entity_0")}
            e.update({_rename(k): v for k, v in em.props.items()})
            lights.append(with_textures(e, em))
    for _, em in kids(scene, "emitter"):
        parts = [("_sun", "sun"), ("_sky", "sky")] if em.plugin == "sunsky" else [("", em.plugin)]
        base = f"__light_{len(lights)}"
        for suffix, plugin in parts:
            e = {"name": base + suffix, "type": plugin}
            e.update({_rename(k): v for k, v in em.props.items()})
            lights.append(with_textures(e, em))
    if lights:
        out["lights"] = lights
    return out


def convert_file(path, defines=None):
    scene, _ = parse(path, defines)
    return convert(scene)


def main(argv=None):
    import argparse
    ap = argparse.ArgumentParser(prog="python -m ignis_amd.mts", description="Mitsuba XML -> Ignis JSON (the .xml leg of `igutil convert`)")
    ap.add_argument("input")
    ap.add_argument("-o", "--output", help="output .json (default: the input with .json)")
    ap.add_argument("-D", "--define", action="append", default=[], metavar="NAME=VALUE", help="value of a $NAME parameter")
    args = ap.parse_args(argv)
    defines = dict(d.split("=", 1) for d in args.define)
    try:
        data = convert_file(args.input, defines)
    except (MtsError, ET.ParseError, OSError, KeyError, ValueError) as e:  # (KeyError: a ref / alias naming an id nobody defines)
        print(f"error: {e}", file=sys.stderr)
        return 1
    dst = args.output or os.path.splitext(args.input)[0] + ".json"
    with open(dst, "w") as f:
        json.dump(data, f, indent=1)
    return 0


if __name__ == "__main__":
    sys.exit(main())
