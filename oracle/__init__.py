"""ctypes wrapper of the CPU oracle (oracle/liboracle.so).

TEST INFRASTRUCTURE ONLY: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg import this package. The product package `ignis_amd` never does.
"""
import ctypes as C
import os

import numpy as np

_DIR = os.path.dirname(os.path.abspath(__file__))


class Settings(C.Structure):
    _fields_ = [("spi", C.c_int32), ("width", C.c_int32), ("height", C.c_int32), ("iteration", C.c_int32),
                ("frame", C.c_int32), ("seed", C.c_int32), ("threads", C.c_int32),
                ("xmin", C.c_int32), ("ymin", C.c_int32), ("xmax", C.c_int32), ("ymax", C.c_int32),
                ("row_offset", C.c_int32), ("row_stride", C.c_int32)]


class Stats(C.Structure):
    _fields_ = [("camera_rays", C.c_uint64), ("bounce_rays", C.c_uint64), ("shadow_rays", C.c_uint64),
                ("nodes", C.c_uint64), ("tris", C.c_uint64), ("leaves", C.c_uint64), ("unoccluded", C.c_uint64),
                ("max_stack", C.c_int32), ("threads_used", C.c_int32)]

    def as_dict(self):
        return {k: int(getattr(self, k)) for k, _ in self._fields_}


_lib = None


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(_DIR, "liboracle.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} is missing: run `make -C oracle` (or __graft_entry__.build())")
        l = C.CDLL(path)
        fp = C.POINTER(C.c_float)
        ip = C.POINTER(C.c_int32)
        up = C.POINTER(C.c_uint32)
        l.oracle_render.restype = C.c_int
        l.oracle_render.argtypes = [C.c_void_p, C.POINTER(Settings), fp, C.POINTER(Stats)]
        l.oracle_render_ex.restype = C.c_int
        l.oracle_render_ex.argtypes = [C.c_void_p, C.POINTER(Settings), fp, C.POINTER(Stats), fp, fp]
        l.oracle_render_aovs.restype = C.c_int
        l.oracle_render_aovs.argtypes = [C.c_void_p, C.POINTER(Settings), fp, C.POINTER(Stats), fp, fp, fp, fp]
        l.oracle_generate_rays.restype = C.c_int
        l.oracle_generate_rays.argtypes = [C.c_void_p, C.POINTER(Settings), C.c_int64, C.c_int64, fp, up]
        l.oracle_trace.restype = C.c_int
        l.oracle_trace.argtypes = [C.c_void_p, C.c_int64, fp, C.c_uint32, C.c_int, ip, ip, fp, fp, fp, C.POINTER(Stats)]
        l.oracle_trace_bruteforce.restype = C.c_int
        l.oracle_trace_bruteforce.argtypes = [C.c_void_p, C.c_int64, fp, fp, ip, ip]
        l.oracle_intersect_tri.restype = C.c_int
        l.oracle_intersect_tri.argtypes = [fp, fp, C.c_float, C.c_float, fp, fp, fp, fp, fp]
        l.oracle_intersect_box.restype = C.c_int
        l.oracle_intersect_box.argtypes = [fp, fp, C.c_float, C.c_float, fp, fp, fp]
        l.oracle_random_seed.restype = C.c_uint32
        l.oracle_random_seed.argtypes = [C.c_int32] * 6
        l.oracle_random_f32.restype = None
        l.oracle_random_f32.argtypes = [C.c_uint32, C.c_uint32, C.c_int32, fp, up]
        l.oracle_detmath.restype = None
        l.oracle_detmath.argtypes = [C.c_int, C.c_int64, fp, fp]
        l.oracle_hardware_threads.restype = C.c_int
        _lib = l
    return _lib


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


def _up(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint32))


def _scene_ptr(scene):
    """Accepts an ignis_amd.tables.LoadedScene or a raw pointer."""
    t = getattr(scene, "tables", scene)
    return C.cast(t, C.c_void_p)


def make_settings(spi, width, height, iteration=0, frame=0, seed=0, threads=0, window=None, rows=None):
    s = Settings(spi, width, height, iteration, frame, seed, threads, 0, 0, 0, 0, 0, 1)
    if rows is not None:
        s.row_offset, s.row_stride = rows
    if window is not None:
        s.xmin, s.ymin, s.xmax, s.ymax = window
    return s


def render(scene, spi, width, height, iteration=0, frame=0, seed=0, threads=0, fb=None, window=None, rows=None, aovs=None, mis_aovs=None):
    """One iteration of the reference CPU pipeline; returns (fb[h,w,3] float32 accumulated, stats dict).
    aovs: optional (normals, albedo) float32 [h, w, 3] arrays, accumulated by iteration 0 (the info-buffer AOVs).
    mis_aovs: optional ("Direct Weights", "NEE Weights") arrays of a path tracer with aov_mis, accumulated like fb."""
    if fb is None:
        fb = np.zeros((height, width, 3), dtype=np.float32)
    assert fb.dtype == np.float32 and fb.flags.c_contiguous and fb.shape == (height, width, 3)
    cfg = make_settings(spi, width, height, iteration, frame, seed, threads, window, rows)
    st = Stats()
    if mis_aovs is not None:
        for a in mis_aovs:
            assert a.dtype == np.float32 and a.flags.c_contiguous and a.shape == (height, width, 3)
        rc = lib().oracle_render_aovs(_scene_ptr(scene), C.byref(cfg), _fp(fb), C.byref(st), None, None, _fp(mis_aovs[0]), _fp(mis_aovs[1]))
        if rc != 0:
            raise RuntimeError("oracle_render failed")
        return fb, st.as_dict()
    if aovs is not None:
        for a in aovs:
            assert a.dtype == np.float32 and a.flags.c_contiguous and a.shape == (height, width, 3)
        rc = lib().oracle_render_ex(_scene_ptr(scene), C.byref(cfg), _fp(fb), C.byref(st), _fp(aovs[0]), _fp(aovs[1]))
        if rc != 0:
            raise RuntimeError("oracle_render failed")
        return fb, st.as_dict()
    rc = lib().oracle_render(_scene_ptr(scene), C.byref(cfg), _fp(fb), C.byref(st))
    if rc != 0:
        raise RuntimeError("oracle_render failed")
    return fb, st.as_dict()


def generate_rays(scene, spi, width, height, first_id, count, iteration=0, frame=0, seed=0):
    rays = np.empty((count, 8), dtype=np.float32)
    ctr = np.empty(count, dtype=np.uint32)
    cfg = make_settings(spi, width, height, iteration, frame, seed)
    rc = lib().oracle_generate_rays(_scene_ptr(scene), C.byref(cfg), first_id, count, _fp(rays), _up(ctr))
    if rc != 0:
        raise RuntimeError("oracle_generate_rays failed")
    return rays, ctr


def trace(scene, rays, flags=0, any_hit=False):
    """rays: (n, 8) float32 [org, dir, tmin, tmax]. Returns dict of ent_id, prim_id, t, u, v + stats."""
    rays = np.ascontiguousarray(rays, dtype=np.float32)
    n = rays.shape[0]
    ent = np.empty(n, dtype=np.int32)
    prim = np.empty(n, dtype=np.int32)
    t = np.empty(n, dtype=np.float32)
    u = np.empty(n, dtype=np.float32)
    v = np.empty(n, dtype=np.float32)
    st = Stats()
    rc = lib().oracle_trace(_scene_ptr(scene), n, _fp(rays), flags, 1 if any_hit else 0, _ip(ent), _ip(prim), _fp(t), _fp(u), _fp(v), C.byref(st))
    if rc != 0:
        raise RuntimeError("oracle_trace failed")
    return {"ent_id": ent, "prim_id": prim, "t": t, "u": u, "v": v, "stats": st.as_dict()}


def trace_bruteforce(scene, rays):
    rays = np.ascontiguousarray(rays, dtype=np.float32)
    n = rays.shape[0]
    t = np.empty(n, dtype=np.float32)
    ent = np.empty(n, dtype=np.int32)
    prim = np.empty(n, dtype=np.int32)
    rc = lib().oracle_trace_bruteforce(_scene_ptr(scene), n, _fp(rays), _fp(t), _ip(ent), _ip(prim))
    if rc != 0:
        raise RuntimeError("oracle_trace_bruteforce failed")
    return {"t": t, "ent_id": ent, "prim_id": prim}


def intersect_tri(org, dir, tmin, tmax, v0, e1, e2, n):
    a = [np.asarray(x, dtype=np.float32) for x in (org, dir, v0, e1, e2, n)]
    out = np.zeros(3, dtype=np.float32)
    hit = lib().oracle_intersect_tri(_fp(a[0]), _fp(a[1]), tmin, tmax, _fp(a[2]), _fp(a[3]), _fp(a[4]), _fp(a[5]), _fp(out))
    return bool(hit), out


def intersect_box(org, dir, tmin, tmax, bmin, bmax):
    a = [np.asarray(x, dtype=np.float32) for x in (org, dir, bmin, bmax)]
    out = np.zeros(2, dtype=np.float32)
    hit = lib().oracle_intersect_box(_fp(a[0]), _fp(a[1]), tmin, tmax, _fp(a[2]), _fp(a[3]), _fp(out))
    return bool(hit), out


def random_seed(sample, iteration, frame, x, y, user):
    return int(lib().oracle_random_seed(sample, iteration, frame, x, y, user))


def random_sequence(seed, first_counter, count):
    f = np.empty(count, dtype=np.float32)
    r = np.empty(count, dtype=np.uint32)
    lib().oracle_random_f32(seed, first_counter, count, _fp(f), _up(r))
    return f, r


def detmath(which, x):
    x = np.ascontiguousarray(x, dtype=np.float32)
    if which == "atan2":  # x: [n, 2] pairs (y, x)
        y = np.empty(x.shape[0], np.float32)
        lib().oracle_detmath(4, C.c_int64(x.shape[0]), _fp(x), _fp(y))
        return y
    y = np.empty_like(x)
    lib().oracle_detmath({"sin": 0, "cos": 1, "acos": 2, "asin": 3, "exp": 5}[which], x.size, _fp(x), _fp(y))
    return y


def align_vectors(a, b, v):
    """(M v, M) for M = mat3x3_align_vectors(a, b); M as a [3, 3] array of columns."""
    x = [np.asarray(t, dtype=np.float32) for t in (a, b, v)]
    out, m = np.zeros(3, np.float32), np.zeros(9, np.float32)
    lib().oracle_align_vectors(_fp(x[0]), _fp(x[1]), _fp(x[2]), _fp(out), _fp(m))
    return out, m.reshape(3, 3)


def ensure_valid_reflection(ng, i, n):
    x = [np.asarray(t, dtype=np.float32) for t in (ng, i, n)]
    out = np.zeros(3, np.float32)
    lib().oracle_ensure_valid_reflection(_fp(x[0]), _fp(x[1]), _fp(x[2]), _fp(out))
    return out


def vndf_ggx(alpha_u, alpha_v, seed, wo):
    wo = np.asarray(wo, dtype=np.float32)
    n = np.zeros(3, np.float32)
    pdf, dcos = C.c_float(0), C.c_float(0)
    lib().oracle_vndf_ggx(C.c_float(alpha_u), C.c_float(alpha_v), C.c_uint32(seed), _fp(wo), _fp(n), C.byref(pdf), C.byref(dcos))
    return n, pdf.value, dcos.value


def image_lookup(scene, tex_id, uv):
    """Bitmap texture lookup at uv [n, 2] -> rgb [n, 3] (texture/image.art filters and borders)."""
    uv = np.ascontiguousarray(uv, dtype=np.float32)
    out = np.zeros((uv.shape[0], 3), np.float32)
    lib().oracle_image_lookup(scene.tables, C.c_int32(tex_id), C.c_int64(uv.shape[0]), _fp(uv), _fp(out))
    return out


def bsdf_eval(scene, mat_id, wo, wi, entering=True):
    """eval (cosine included, as in the reference) and pdf of material `mat_id` on a flat surface with normal +z."""
    wi = np.ascontiguousarray(wi, dtype=np.float32).reshape(-1, 3)
    wo = np.ascontiguousarray(wo, dtype=np.float32)
    pdf = np.zeros(wi.shape[0], np.float32)
    col = np.zeros((wi.shape[0], 3), np.float32)
    rc = lib().oracle_bsdf_probe(scene.tables, C.c_int32(mat_id), C.c_int32(int(entering)), C.c_int32(0), _fp(wo),
                                 C.c_int64(wi.shape[0]), C.c_uint32(0), _fp(wi), _fp(pdf), _fp(col), None)
    if rc != 0:
        raise RuntimeError("oracle_bsdf_probe failed")
    return col, pdf


def bsdf_sample(scene, mat_id, wo, n, seed=1, entering=True):
    """n BSDF samples: directions, pdfs, weights (eval / pdf), eta; rejected samples have pdf 0."""
    wo = np.ascontiguousarray(wo, dtype=np.float32)
    wi = np.zeros((n, 3), np.float32)
    pdf = np.zeros(n, np.float32)
    col = np.zeros((n, 3), np.float32)
    eta = np.ones(n, np.float32)
    rc = lib().oracle_bsdf_probe(scene.tables, C.c_int32(mat_id), C.c_int32(int(entering)), C.c_int32(1), _fp(wo),
                                 C.c_int64(n), C.c_uint32(seed), _fp(wi), _fp(pdf), _fp(col), _fp(eta))
    if rc != 0:
        raise RuntimeError("oracle_bsdf_probe failed")
    return wi, pdf, col, eta


def cdf1d(data, mode, u):
    """core/cdf.art 1D CDF over `data` = [x1, ..., 1] (leading 0 implied). mode "discrete" / "continuous" / "pdf" ->
    (off, pos, pdf)."""
    data = np.ascontiguousarray(data, dtype=np.float32)
    off, pos, pdf = C.c_int32(0), C.c_float(0), C.c_float(0)
    lib().oracle_cdf1d(_fp(data), C.c_int32(data.size), C.c_int32({"discrete": 0, "continuous": 1, "pdf": 2}[mode]), C.c_float(u),
                       C.byref(off), C.byref(pos), C.byref(pdf))
    return off.value, pos.value, pdf.value


def cdf2d(table, size_x, size_y, mode, u):
    """2D marginal / conditional CDF: "continuous" -> (pos [2], pdf) for u [2]; "pdf" -> pdf at position u."""
    table = np.ascontiguousarray(table, dtype=np.float32)
    u = np.ascontiguousarray(u, dtype=np.float32)
    pos, pdf = np.zeros(2, np.float32), C.c_float(0)
    lib().oracle_cdf2d(_fp(table), C.c_int32(size_x), C.c_int32(size_y), C.c_int32(1 if mode == "continuous" else 2), _fp(u), _fp(pos), C.byref(pdf))
    return pos, pdf.value


def warp(which, a, b):
    """core/warp.art forward maps as the oracle's shading code calls them: "disk" square_to_concentric_disk -> (x, y);
    "sphere" equal_area_square_to_sphere -> (x, y, z); "dir" dir_from_spherical(theta, phi) -> (x, y, z);
    "spherical" spherical_from_dir(dir_from_spherical(theta, phi)) -> (theta, phi)."""
    out = np.zeros(3, np.float32)
    k = {"disk": 0, "sphere": 1, "dir": 2, "spherical": 3}[which]
    lib().oracle_warp(C.c_int32(k), C.c_float(a), C.c_float(b), _fp(out))
    return out[:2] if k in (0, 3) else out


def interval_search(arr, value, strict=False):
    """interval::binary_search (core/interval.art:7-23) with the predicate arr[i] <= value (or < value)."""
    arr = np.ascontiguousarray(arr, dtype=np.int32)
    lib().oracle_interval_search.restype = C.c_int32
    return int(lib().oracle_interval_search(_ip(arr), C.c_int32(arr.size), C.c_int32(value), C.c_int32(1 if strict else 0)))


def hardware_threads():
    return int(lib().oracle_hardware_threads())


def set_ppm_sound_directions(on):
    """ORACLE-ONLY experiment switch (oracle.cpp): photon directions through a sound snorm16 pair instead of encode_normal_32 as written."""
    lib().oracle_set_ppm_sound_directions(1 if on else 0)
