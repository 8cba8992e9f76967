"""ctypes binding of the MI355X render device (libig_device_hip.so, include/igd_device.h).

There is no CPU fallback: if the library is missing, or no gfx950 device is visible, creating a
device raises. numpy is only used to hand host buffers across the C ABI.
"""
import ctypes as C
import os

import numpy as np

from .tables import Scene, _LIB_DIR


class Setup(C.Structure):
    _fields_ = [("gpu_index", C.c_int32), ("acquire_stats", C.c_int32), ("debug_trace", C.c_int32),
                ("is_interactive", C.c_int32), ("stream_capacity", C.c_uint64), ("info_aovs", C.c_int32),
                ("blocking_render", C.c_int32)]


class RenderSettings(C.Structure):
    _fields_ = [("rays", C.POINTER(C.c_float)), ("spi", C.c_int32), ("width", C.c_int32), ("height", C.c_int32),
                ("iteration", C.c_int32), ("frame", C.c_int32), ("user_seed", C.c_int32),
                ("row_offset", C.c_int32), ("row_stride", C.c_int32), ("iterations", C.c_int32)]


class Stats(C.Structure):
    _fields_ = [("camera_rays", C.c_uint64), ("bounce_rays", C.c_uint64), ("shadow_rays", C.c_uint64),
                ("unoccluded", C.c_uint64),
                ("nodes_primary", C.c_uint64), ("tris_primary", C.c_uint64), ("leaves_primary", C.c_uint64),
                ("nodes_secondary", C.c_uint64), ("tris_secondary", C.c_uint64), ("leaves_secondary", C.c_uint64),
                ("traverse_primary_launches", C.c_uint64), ("traverse_secondary_launches", C.c_uint64),
                ("ms_generate", C.c_double), ("ms_traverse_primary", C.c_double), ("ms_shade", C.c_double),
                ("ms_traverse_secondary", C.c_double), ("ms_resolve", C.c_double), ("ms_total", C.c_double),
                ("rounds", C.c_uint32), ("pad", C.c_uint32), ("tail_rays", C.c_uint64), ("ms_tail", C.c_double),
                ("section_passes", C.c_uint64 * 6), ("section_lanes", C.c_uint64 * 6), ("ms_ray_sort", C.c_double), ("stream_bytes", C.c_uint32 * 8)]

    def as_dict(self):
        d = {k: getattr(self, k) for k, _ in self._fields_ if k not in ("pad", "section_passes", "section_lanes", "stream_bytes")}
        d["section_passes"], d["section_lanes"] = list(self.section_passes), list(self.section_lanes)
        d["stream_bytes"] = list(self.stream_bytes)
        for k in ("nodes", "tris", "leaves"):
            d[k] = d[k + "_primary"] + d[k + "_secondary"]
        return d


# Every symbol include/igd_device.h declares (checked by tests/test_abi.py).
EXPORTS = [
    "igd_get_abi_version", "igd_device_count", "igd_create", "igd_destroy", "igd_assign_scene", "igd_render",
    "igd_resize", "igd_release_all", "igd_framebuffer_width", "igd_framebuffer_height", "igd_framebuffer_host",
    "igd_framebuffer_device", "igd_clear_framebuffer", "igd_sync_framebuffer_to_device", "igd_get_stats",
    "igd_reset_stats", "igd_traverse", "igd_set_parameter_i32", "igd_set_parameter_f32", "igd_set_parameter_vec3",
    "igd_synchronize", "igd_last_error", "igd_buffer_size", "igd_buffer_copy", "igd_buffer_ptr",
    "igd_node_bytes",
    "igd_comm_available", "igd_comm_unique_id", "igd_comm_init", "igd_comm_world_size", "igd_comm_gather_rows", "igd_comm_allreduce_f64", "igd_comm_destroy",
]

_lib = None


def lib():
    """Loads libig_device_hip.so; raises (never falls back) when it has not been built."""
    global _lib
    if _lib is None:
        # IGD_LIBRARY: kernel-variant experiments only (tools/build_variant.sh); the product library is the in-tree one
        path = os.environ.get("IGD_LIBRARY") or os.path.join(_LIB_DIR, "libig_device_hip.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} is missing: the HIP extension must be built (__graft_entry__.build()); "
                               "ignis_amd has no CPU fallback")
        l = C.CDLL(path)
        fp, ip = C.POINTER(C.c_float), C.POINTER(C.c_int32)
        l.igd_get_abi_version.restype = C.c_uint32
        l.igd_device_count.restype = C.c_int32
        l.igd_create.restype = C.c_void_p
        l.igd_create.argtypes = [C.POINTER(Setup)]
        l.igd_destroy.restype = None
        l.igd_destroy.argtypes = [C.c_void_p]
        l.igd_assign_scene.restype = C.c_int32
        l.igd_assign_scene.argtypes = [C.c_void_p, C.POINTER(Scene)]
        l.igd_render.restype = C.c_int32
        l.igd_render.argtypes = [C.c_void_p, C.POINTER(RenderSettings)]
        l.igd_resize.restype = C.c_int32
        l.igd_resize.argtypes = [C.c_void_p, C.c_int32, C.c_int32]
        l.igd_release_all.restype = C.c_int32
        l.igd_release_all.argtypes = [C.c_void_p]
        l.igd_framebuffer_width.restype = C.c_int32
        l.igd_framebuffer_width.argtypes = [C.c_void_p]
        l.igd_framebuffer_height.restype = C.c_int32
        l.igd_framebuffer_height.argtypes = [C.c_void_p]
        l.igd_framebuffer_host.restype = fp
        l.igd_framebuffer_host.argtypes = [C.c_void_p, C.c_char_p, C.c_int32]
        l.igd_framebuffer_device.restype = C.c_void_p
        l.igd_framebuffer_device.argtypes = [C.c_void_p, C.c_char_p]
        l.igd_clear_framebuffer.restype = C.c_int32
        l.igd_clear_framebuffer.argtypes = [C.c_void_p, C.c_char_p]
        l.igd_sync_framebuffer_to_device.restype = C.c_int32
        l.igd_sync_framebuffer_to_device.argtypes = [C.c_void_p, C.c_char_p, fp]
        l.igd_get_stats.restype = C.c_int32
        l.igd_get_stats.argtypes = [C.c_void_p, C.POINTER(Stats)]
        l.igd_reset_stats.restype = C.c_int32
        l.igd_reset_stats.argtypes = [C.c_void_p]
        l.igd_traverse.restype = C.c_int32
        l.igd_traverse.argtypes = [C.c_void_p, C.c_int64, fp, C.c_uint32, C.c_int32, ip, ip, fp, fp, fp, C.c_int32,
                                   C.POINTER(C.c_double)]
        l.igd_set_parameter_i32.restype = C.c_int32
        l.igd_set_parameter_i32.argtypes = [C.c_void_p, C.c_char_p, C.c_int32]
        l.igd_set_parameter_f32.restype = C.c_int32
        l.igd_set_parameter_f32.argtypes = [C.c_void_p, C.c_char_p, C.c_float]
        l.igd_set_parameter_vec3.restype = C.c_int32
        l.igd_set_parameter_vec3.argtypes = [C.c_void_p, C.c_char_p, fp]
        l.igd_synchronize.restype = C.c_int32
        l.igd_synchronize.argtypes = [C.c_void_p]
        l.igd_node_bytes.restype = C.c_int32
        l.igd_node_bytes.argtypes = [C.c_void_p]
        l.igd_comm_available.restype = C.c_int32
        l.igd_comm_available.argtypes = []
        l.igd_comm_unique_id.restype = C.c_int32
        l.igd_comm_unique_id.argtypes = [C.POINTER(C.c_uint8)]
        l.igd_comm_init.restype = C.c_int32
        l.igd_comm_init.argtypes = [C.c_void_p, C.POINTER(C.c_uint8), C.c_int32, C.c_int32]
        l.igd_comm_world_size.restype = C.c_int32
        l.igd_comm_world_size.argtypes = [C.c_void_p]
        l.igd_comm_gather_rows.restype = C.c_int32
        l.igd_comm_gather_rows.argtypes = [C.c_void_p, C.c_int32]
        l.igd_comm_allreduce_f64.restype = C.c_int32
        l.igd_comm_allreduce_f64.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.c_int32, C.c_int32]
        l.igd_comm_destroy.restype = C.c_int32
        l.igd_comm_destroy.argtypes = [C.c_void_p]
        l.igd_last_error.restype = C.c_char_p
        l.igd_buffer_size.restype = C.c_uint64
        l.igd_buffer_size.argtypes = [C.c_void_p, C.c_char_p]
        l.igd_buffer_copy.restype = C.c_int32
        l.igd_buffer_copy.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_uint64]
        l.igd_buffer_ptr.restype = C.c_void_p
        l.igd_buffer_ptr.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_uint64)]
        _lib = l
    return _lib


class DeviceError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"[{code}] {message}")
        self.code = code


def _check(rc):
    if rc != 0:
        raise DeviceError(rc, lib().igd_last_error().decode())


def device_count():
    return int(lib().igd_device_count())


class Device:
    """One MI355X render device (IRenderDevice counterpart)."""

    def __init__(self, gpu_index=0, acquire_stats=False, stream_capacity=0, info_aovs=False, blocking_render=False, interactive=False):
        # acquire_stats: False/0 off, 1 HIP-event stage timers, True/2 timers + traversal work counters
        level = 2 if acquire_stats is True else int(acquire_stats)
        # info_aovs: the "Normals" / "Albedo" AOVs of the denoiser's info buffer, read with framebuffer("Normals") / ("Albedo")
        # blocking_render: every render() call completes before it returns (the reference's contract) instead of being deferred
        setup = Setup(int(gpu_index), level, 0, int(bool(interactive)), int(stream_capacity), int(bool(info_aovs)), int(bool(blocking_render)))
        self._h = lib().igd_create(C.byref(setup))
        if not self._h:
            raise DeviceError(-2, lib().igd_last_error().decode())

    def close(self):
        if getattr(self, "_h", None):
            lib().igd_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def assign_scene(self, scene):
        """scene: ignis_amd.tables.LoadedScene (or anything with a `.tables` igd_scene pointer)."""
        _check(lib().igd_assign_scene(self._h, scene.tables))

    def render(self, spi, width, height, iteration=0, frame=0, seed=0, rays=None, row_offset=0, row_stride=1, iterations=1):
        rs = RenderSettings(None, int(spi), int(width), int(height), int(iteration), int(frame), int(seed),
                            int(row_offset), int(row_stride), int(iterations))
        keep = None
        if rays is not None:
            keep = np.ascontiguousarray(rays, dtype=np.float32)
            assert keep.ndim == 2 and keep.shape[1] == 8
            rs.rays = keep.ctypes.data_as(C.POINTER(C.c_float))
            rs.width, rs.height = keep.shape[0], 1
        _check(lib().igd_render(self._h, C.byref(rs)))

    def set_parameter(self, name, value):
        """Registry parameter of the next render: int, float or a 3-vector (see include/igd_device.h)."""
        n = name.encode()
        if isinstance(value, (bool, int, np.integer)):
            _check(lib().igd_set_parameter_i32(self._h, n, int(value)))
        elif isinstance(value, (float, np.floating)):
            _check(lib().igd_set_parameter_f32(self._h, n, float(value)))
        else:
            v = np.ascontiguousarray(value, dtype=np.float32).reshape(-1)
            if v.size != 3:
                raise ValueError("vector parameters have 3 components here")
            _check(lib().igd_set_parameter_vec3(self._h, n, v.ctypes.data_as(C.POINTER(C.c_float))))

    def buffer(self, name):
        """Device-resident table by name (IRenderDevice::copyBufferToHost): uint8 array, or None for an unknown name."""
        n = name.encode()
        size = int(lib().igd_buffer_size(self._h, n))
        if size == 0:
            return None
        out = np.empty(size, np.uint8)
        _check(lib().igd_buffer_copy(self._h, n, out.ctypes.data_as(C.c_void_p), size))
        return out

    def buffer_device_ptr(self, name):
        """(device pointer, size in bytes) of a named buffer (IRenderDevice::getBufferForDevice); (None, 0) if unknown."""
        size = C.c_uint64(0)
        p = lib().igd_buffer_ptr(self._h, name.encode(), C.byref(size))
        return (int(p) if p else None), int(size.value)

    def node_bytes(self):
        """Bytes per inner BVH node the traversal kernels fetch for the assigned scene (256: Node8, 128: quantised)."""
        return int(lib().igd_node_bytes(self._h))

    def synchronize(self):
        """Waits for the part of the last render that overlaps the next one (tail paths + resolve) and reports
        its errors; the framebuffer / stats accessors do this implicitly."""
        _check(lib().igd_synchronize(self._h))

    def resize(self, width, height):
        _check(lib().igd_resize(self._h, width, height))

    def release_all(self):
        _check(lib().igd_release_all(self._h))

    def framebuffer(self, name=None, copy=True):
        """float32 [height, width, 3]: sum over iterations (divide by the iteration count to display)."""
        p = lib().igd_framebuffer_host(self._h, name.encode() if name else None, 1)
        if not p:
            raise DeviceError(-1, lib().igd_last_error().decode())
        w, h = lib().igd_framebuffer_width(self._h), lib().igd_framebuffer_height(self._h)
        a = np.ctypeslib.as_array(p, shape=(h, w, 3))
        return a.copy() if copy else a

    def framebuffer_device_ptr(self, name=None):
        return lib().igd_framebuffer_device(self._h, name.encode() if name else None)

    def framebuffer_size(self):
        return lib().igd_framebuffer_width(self._h), lib().igd_framebuffer_height(self._h)

    def clear_framebuffer(self, name=None):
        _check(lib().igd_clear_framebuffer(self._h, name.encode() if name else None))

    def upload_framebuffer(self, data, name=None):
        data = np.ascontiguousarray(data, dtype=np.float32)
        _check(lib().igd_sync_framebuffer_to_device(self._h, name.encode() if name else None,
                                                    data.ctypes.data_as(C.POINTER(C.c_float))))

    def stats(self):
        s = Stats()
        _check(lib().igd_get_stats(self._h, C.byref(s)))
        return s.as_dict()

    def reset_stats(self):
        _check(lib().igd_reset_stats(self._h))

    def traverse(self, rays, flags=0, any_hit=False, repeat=1):
        """Closest-hit / any-hit traversal of a host ray list (n, 8). Returns dict incl. kernel_ms."""
        rays = np.ascontiguousarray(rays, dtype=np.float32)
        n = rays.shape[0]
        ent = np.empty(n, dtype=np.int32)
        prim = np.empty(n, dtype=np.int32)
        t = np.empty(n, dtype=np.float32)
        u = np.empty(n, dtype=np.float32)
        v = np.empty(n, dtype=np.float32)
        ms = C.c_double(0)
        fp, ip = C.POINTER(C.c_float), C.POINTER(C.c_int32)
        _check(lib().igd_traverse(self._h, n, rays.ctypes.data_as(fp), int(flags), int(bool(any_hit)),
                                  ent.ctypes.data_as(ip), prim.ctypes.data_as(ip), t.ctypes.data_as(fp),
                                  u.ctypes.data_as(fp), v.ctypes.data_as(fp), int(repeat), C.byref(ms)))
        return {"ent_id": ent, "prim_id": prim, "t": t, "u": u, "v": v, "kernel_ms": ms.value}
