"""ctypes mirrors of include/ig_tables.h (POD scene tables) and the host loader ABI.

The loader itself is native (libig_host.so, include/igh_host.h); this module only
declares the structures so Python callers (tests, bench, the Runtime mirror) can pass
`igd_scene` pointers between the native libraries and inspect tables with numpy.
"""
import ctypes as C
import os

import numpy as np

_LIB_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib")


class Node8(C.Structure):
    _fields_ = [("bounds", (C.c_float * 8) * 6), ("child", C.c_int32 * 8), ("pad", C.c_int32 * 8)]


class Tri4(C.Structure):
    _fields_ = [("v0", (C.c_float * 4) * 3), ("e1", (C.c_float * 4) * 3), ("e2", (C.c_float * 4) * 3),
                ("n", (C.c_float * 4) * 3), ("prim_id", C.c_int32 * 4)]


class EntityLeaf1(C.Structure):
    _fields_ = [("min", C.c_float * 3), ("entity_id", C.c_int32), ("max", C.c_float * 3), ("shape_id", C.c_int32),
                ("local", C.c_float * 12), ("flags", C.c_uint32), ("mat_id", C.c_int32), ("user", C.c_int32 * 2)]


class LookupEntry(C.Structure):
    _fields_ = [("type_id", C.c_uint32), ("flags", C.c_uint32), ("offset", C.c_uint64)]


class Material(C.Structure):
    _fields_ = [("bsdf_type", C.c_int32), ("light_id", C.c_int32), ("flags", C.c_uint32), ("tex_id", C.c_int32),
                ("p", C.c_float * 12), ("q", C.c_float * 8), ("tex_refl", C.c_int32), ("pad", C.c_int32 * 3),
                ("r", C.c_float * 8)]


class Light(C.Structure):
    _fields_ = [("type", C.c_int32), ("entity_id", C.c_int32), ("pad", C.c_int32 * 2), ("d", C.c_float * 32)]


class Camera(C.Structure):
    _fields_ = [("eye", C.c_float * 3), ("dir", C.c_float * 3), ("up", C.c_float * 3), ("fov", C.c_float),
                ("fov_is_vertical", C.c_int32), ("near_clip", C.c_float), ("far_clip", C.c_float),
                ("aspect_ratio", C.c_float), ("type", C.c_int32), ("fisheye_mode", C.c_int32),
                ("fisheye_mask", C.c_int32), ("scale", C.c_float), ("aperture_radius", C.c_float),
                ("focal_length", C.c_float), ("pixel_sampler", C.c_int32)]


class Technique(C.Structure):
    _fields_ = [("max_depth", C.c_int32), ("min_depth", C.c_int32), ("clamp", C.c_float), ("nee", C.c_int32),
                ("light_selector", C.c_int32), ("type", C.c_int32), ("aov_mis", C.c_int32), ("debug_mode", C.c_int32),
                ("photon_count", C.c_int32), ("max_light_depth", C.c_int32), ("merge_radius", C.c_float), ("reserved", C.c_int32)]


class Texture(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("channels", C.c_uint32), ("filter", C.c_uint32),
                ("wrap_u", C.c_uint32), ("wrap_v", C.c_uint32), ("offset", C.c_uint64)]


class Medium(C.Structure):
    _fields_ = [("sigma_a", C.c_float * 3), ("sigma_s", C.c_float * 3), ("g", C.c_float), ("type", C.c_int32)]


class Scene(C.Structure):
    _fields_ = [
        ("entities", C.POINTER(C.c_float)), ("entity_count", C.c_uint32),
        ("shape_lookups", C.POINTER(LookupEntry)), ("shape_count", C.c_uint32),
        ("shape_data", C.POINTER(C.c_uint8)), ("shape_data_size", C.c_uint64),
        ("primbvh", C.POINTER(C.c_uint8)), ("primbvh_size", C.c_uint64),
        ("scene_nodes", C.POINTER(Node8)), ("scene_node_count", C.c_uint32),
        ("scene_leaves", C.POINTER(EntityLeaf1)), ("scene_leaf_count", C.c_uint32),
        ("materials", C.POINTER(Material)), ("material_count", C.c_uint32),
        ("entity_per_material", C.POINTER(C.c_int32)),
        ("lights", C.POINTER(Light)), ("light_count", C.c_uint32), ("infinite_light_count", C.c_uint32),
        ("light_hierarchy", C.POINTER(C.c_float)), ("light_hierarchy_nodes", C.c_uint32),
        ("light_codes", C.POINTER(C.c_uint32)),
        ("camera", Camera), ("technique", Technique),
        ("bbox_min", C.c_float * 3), ("bbox_max", C.c_float * 3),
        ("film_width", C.c_int32), ("film_height", C.c_int32), ("scene_radius", C.c_float),
        ("textures", C.POINTER(Texture)), ("texture_count", C.c_uint32),
        ("texture_data", C.POINTER(C.c_uint8)), ("texture_data_size", C.c_uint64),
        ("cdf_data", C.POINTER(C.c_float)), ("cdf_data_count", C.c_uint64),
        ("sphere_nodes", C.POINTER(Node8)), ("sphere_node_count", C.c_uint32),
        ("sphere_leaves", C.POINTER(EntityLeaf1)), ("sphere_leaf_count", C.c_uint32),
        ("light_cdf", C.POINTER(C.c_float)), ("light_cdf_count", C.c_uint32),
        ("media", C.POINTER(Medium)), ("media_count", C.c_uint32),
        ("expr_code", C.POINTER(C.c_uint32)), ("expr_code_count", C.c_uint32),
    ]


assert C.sizeof(Node8) == 256 and C.sizeof(Tri4) == 208 and C.sizeof(EntityLeaf1) == 96
assert C.sizeof(Material) == 144 and C.sizeof(Light) == 144


class HostOptions(C.Structure):
    _fields_ = [("film_width", C.c_int32), ("film_height", C.c_int32)]


_host = None


def host_lib():
    """Loads libig_host.so (built by __graft_entry__.build()); fails loudly if missing."""
    global _host
    if _host is None:
        path = os.path.join(_LIB_DIR, "libig_host.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` first")
        lib = C.CDLL(path)
        lib.igh_load_file.restype = C.c_void_p
        lib.igh_load_file.argtypes = [C.c_char_p, C.POINTER(HostOptions)]
        lib.igh_load_string.restype = C.c_void_p
        lib.igh_load_string.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(HostOptions)]
        lib.igh_tables.restype = C.POINTER(Scene)
        lib.igh_tables.argtypes = [C.c_void_p]
        lib.igh_entity_name.restype = C.c_char_p
        lib.igh_entity_name.argtypes = [C.c_void_p, C.c_uint32]
        lib.igh_material_name.restype = C.c_char_p
        lib.igh_material_name.argtypes = [C.c_void_p, C.c_uint32]
        lib.igh_free.restype = None
        lib.igh_free.argtypes = [C.c_void_p]
        lib.igh_save_exr.restype = C.c_int32
        lib.igh_save_exr.argtypes = [C.c_char_p, C.POINTER(C.c_float), C.c_int32, C.c_int32, C.c_float, C.POINTER(C.c_char_p)]
        lib.igh_read_float_image.restype = C.c_int32
        lib.igh_read_float_image.argtypes = [C.c_char_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_float), C.c_uint64]
        lib.igh_read_image8.restype = C.c_int32
        lib.igh_read_image8.argtypes = [C.c_char_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint8), C.c_uint64]
        lib.igh_eval_expression.restype = C.c_int32
        lib.igh_eval_expression.argtypes = [C.c_char_p, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_int32), C.POINTER(C.c_uint32)]
        lib.igh_last_error.restype = C.c_char_p
        _host = lib
    return _host


def save_exr(path, rgb, scale=1.0, meta=None):
    """Image::save for the framebuffer (igh_save_exr): float32 [H, W, 3] -> OpenEXR with channels B, G, R."""
    import numpy as np
    a = np.ascontiguousarray(rgb, dtype=np.float32)
    if a.ndim != 3 or a.shape[2] != 3:
        raise ValueError("expected an [H, W, 3] image")
    items = []
    for k, v in (meta or {}).items():
        items += [str(k).encode(), str(v).encode()]
    arr = (C.c_char_p * (len(items) + 1))(*items, None)
    rc = host_lib().igh_save_exr(str(path).encode(), a.ctypes.data_as(C.POINTER(C.c_float)), a.shape[1], a.shape[0], float(scale), arr)
    if rc != 0:
        raise RuntimeError(host_lib().igh_last_error().decode())


EXPR_VARS = ("uvw", "P", "V", "N", "Ng", "Nx", "Ny", "frontside")  # enum ige_var (include/ig_expr.h)
EXPR_TYPES = ("bool", "int", "num", "vec2", "vec3", "vec4", "str")


def eval_expression(source, **variables):
    """igh_eval_expression: compile a PExpr string as the loader does and run it once; returns (type name, value) with the value
    a bool / int / float or a tuple of the vector's components. Variables (uvw, P, V, N, Ng, Nx, Ny, frontside) default to zero."""
    import numpy as np
    vals = np.zeros((len(EXPR_VARS), 4), np.float32)
    for k, v in variables.items():
        row = EXPR_VARS.index(k)
        a = np.atleast_1d(np.asarray(v, np.float32))
        vals[row, :] = a[0] if a.size == 1 else 0
        if a.size > 1:
            vals[row, :a.size] = a
    out = (C.c_float * 4)()
    ty, words = C.c_int32(), C.c_uint32()
    if host_lib().igh_eval_expression(source.encode(), vals.ctypes.data_as(C.POINTER(C.c_float)), out, ty, words) != 0:
        raise RuntimeError(host_lib().igh_last_error().decode())
    name = EXPR_TYPES[ty.value]
    if name == "bool":
        return name, out[0] != 0
    if name == "int":
        return name, int(out[0])
    if name == "num":
        return name, float(out[0])
    return name, tuple(float(out[i]) for i in range(int(name[-1])))


def read_image8(path):
    """The loader's PNG / JPEG readers (igh_read_image8): uint8 [H, W, channels] as stored, rows top to bottom."""
    import numpy as np
    w, h, c = C.c_uint32(), C.c_uint32(), C.c_uint32()
    if host_lib().igh_read_image8(str(path).encode(), w, h, c, None, 0) != 0:
        raise RuntimeError(host_lib().igh_last_error().decode())
    a = np.empty((h.value, w.value, c.value), np.uint8)
    if host_lib().igh_read_image8(str(path).encode(), w, h, c, a.ctypes.data_as(C.POINTER(C.c_uint8)), a.size) != 0:
        raise RuntimeError(host_lib().igh_last_error().decode())
    return a


def read_float_image(path):
    """Image::load for OpenEXR / Radiance HDR files (igh_read_float_image): float32 [H, W, 1 or 4], rows top to bottom."""
    import numpy as np
    w, h, c = C.c_uint32(), C.c_uint32(), C.c_uint32()
    if host_lib().igh_read_float_image(str(path).encode(), w, h, c, None, 0) != 0:
        raise RuntimeError(host_lib().igh_last_error().decode())
    a = np.empty((h.value, w.value, c.value), np.float32)
    if host_lib().igh_read_float_image(str(path).encode(), w, h, c, a.ctypes.data_as(C.POINTER(C.c_float)), a.size) != 0:
        raise RuntimeError(host_lib().igh_last_error().decode())
    return a


class LoadedScene:
    """Owns an igh_scene; `.tables` is the borrowed igd_scene pointer (valid while this object lives)."""

    def __init__(self, handle):
        self._h = handle
        self.tables = host_lib().igh_tables(handle)

    @staticmethod
    def from_file(path, width=0, height=0):
        opts = HostOptions(int(width), int(height))
        h = host_lib().igh_load_file(os.fsencode(path), C.byref(opts))
        if not h:
            raise RuntimeError(host_lib().igh_last_error().decode())
        return LoadedScene(h)

    @staticmethod
    def from_string(text, base_dir="", width=0, height=0):
        opts = HostOptions(int(width), int(height))
        h = host_lib().igh_load_string(text.encode(), os.fsencode(base_dir), C.byref(opts))
        if not h:
            raise RuntimeError(host_lib().igh_last_error().decode())
        return LoadedScene(h)

    @property
    def scene(self):
        return self.tables.contents

    def entity_name(self, i):
        s = host_lib().igh_entity_name(self._h, i)
        return s.decode() if s else None

    def material_name(self, i):
        s = host_lib().igh_material_name(self._h, i)
        return s.decode() if s else None

    def scene_nodes(self):
        sc = self.scene
        return np.ctypeslib.as_array(C.cast(sc.scene_nodes, C.POINTER(C.c_float)), shape=(sc.scene_node_count, 64)).copy()

    def shape_mesh(self, shape_id):
        """(vertices [n, 3], normals [n, 3], indices [f, 3], texcoords [n, 2]) of a triangle-mesh shape, decoded from its
        record in the "shapes" dyn-table (src/runtime/shape/TriMeshProvider.cpp:575-596)."""
        import numpy as np
        sc = self.scene
        off = sc.shape_lookups[shape_id].offset
        blob = np.ctypeslib.as_array(sc.shape_data, shape=(sc.shape_data_size,))
        faces, nv, nn, nt = (int(x) for x in blob[off:off + 16].view(np.uint32))
        f = blob[off:].view(np.float32)
        v0 = 12
        n0 = v0 + nv * 4
        i0 = n0 + nn * 4
        t0 = i0 + faces * 4
        return (f[v0:n0].reshape(nv, 4)[:, :3].copy(), f[n0:i0].reshape(nn, 4)[:, :3].copy(),
                f[i0:t0].view(np.int32).reshape(faces, 4)[:, :3].copy(), f[t0:t0 + nt * 2].reshape(nt, 2).copy())

    def primbvh_bytes(self):
        sc = self.scene
        return bytes(C.string_at(sc.primbvh, sc.primbvh_size))

    def close(self):
        if self._h:
            host_lib().igh_free(self._h)
            self._h = None
            self.tables = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
