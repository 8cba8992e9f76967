"""Host-side mirror of the reference's Python API for the hot path.

Mirrors the names and semantics of the nanobind module `ignis`
(src/frontend/python/runtime.cpp:223-312 over src/runtime/Runtime.cpp:334-446): `loadFromFile`,
`loadFromString`, `RuntimeOptions`, `Ray`, and a `Runtime` with `step`, `trace`, `reset`,
`getFramebufferForHost`, `clearFramebuffer`, `IterationCount`, `SampleCount`, ... so the
reference's integrator tests (src/tests/integrator/*.py) read the same against this backend.
The render work is done by the MI355X device behind include/igd_device.h; scene loading by the
native host library behind include/igh_host.h. There is no CPU render path.
"""
import math

import numpy as np

from .device import Device
from .tables import LoadedScene, save_exr


class DenoiserSettings:
    """RuntimeSettings.h:6-10 (nanobind: src/frontend/python/runtime.cpp:152-156)."""

    def __init__(self):
        self.Enabled = False
        self.HighQuality = True
        self.Prefilter = False


# Runtime::hasDenoiser (Runtime.cpp:756): the reference answers with its build flag IG_HAS_DENOISER; here a denoiser is a callable
# registered once per process, `fn(color, normals, albedo, settings) -> denoised`, all float32 [height, width, 3] with the colour
# already divided by the iteration count left to the callable (OIDN is handed the accumulated buffers too, extra/OIDN.cpp:103-126).
_denoiser = None


def registerDenoiser(fn):
    """Installs (or with None removes) the process-wide denoiser the runtimes call after every step, where the reference's
    build links OpenImageDenoise (extra/OIDN.cpp). This image has no OIDN: the hook is the boundary."""
    global _denoiser
    if fn is not None and not callable(fn):
        raise TypeError("registerDenoiser expects a callable or None")
    _denoiser = fn


def hasDenoiser():
    return _denoiser is not None


class RuntimeOptions:
    """Subset of RuntimeOptions (src/runtime/RuntimeSettings.h:12-59) the hot path consumes."""

    def __init__(self):
        self.Device = 0               # Target.Device: HIP ordinal
        self.AcquireStats = False
        self.SPI = 0                  # 0 = recommendSPI (Runtime.cpp:71-79)
        self.Seed = 0
        self.OverrideFilmSize = (0, 0)
        self.StreamCapacity = 0       # rays in flight, 0 = device default
        self.IsTracer = False
        # Denoiser.Enabled (RuntimeSettings.h): the runtime then wraps the technique with the info buffer and the device keeps the
        # "Normals" / "Albedo" AOVs (InfoBufferTechnique.cpp, Runtime.cpp:246-264); the denoiser itself (OIDN) is not part of this path
        self.EnableInfoAOVs = False
        self.Denoiser = DenoiserSettings()
        # tile sharding across devices: this runtime renders rows offset, offset+stride, ...
        self.RowOffset = 0
        self.RowStride = 1

    @staticmethod
    def makeDefault(trace=False):
        o = RuntimeOptions()
        o.IsTracer = bool(trace)
        return o


class Ray:
    """RuntimeStructs.h:39-43."""

    def __init__(self, org, dir, tmin=0.0, tmax=3.4028234664e+38):
        self.Origin = tuple(float(x) for x in org)
        self.Direction = tuple(float(x) for x in dir)
        self.Range = (float(tmin), float(tmax))


class CameraOrientation:
    """src/runtime/camera/CameraOrientation.h: Eye, Dir, Up."""

    def __init__(self, eye=(0, 0, 0), dir=(0, 0, 1), up=(0, 1, 0)):
        self.Eye = tuple(float(x) for x in eye)
        self.Dir = tuple(float(x) for x in dir)
        self.Up = tuple(float(x) for x in up)


def recommend_spi(width, height, interactive=False):
    """recommendSPI for a GPU target (src/runtime/Runtime.cpp:71-79)."""
    spi_f = 8 // 2 if interactive else 8
    spi = int(math.ceil(spi_f / ((width / 1000.0) * (height / 1000.0))))
    return max(1, min(64, spi))


class Runtime:
    def __init__(self, scene: LoadedScene, opts: RuntimeOptions):
        self._scene = scene
        self._opts = opts
        sc = scene.scene
        self._width, self._height = int(sc.film_width), int(sc.film_height)
        self._spi = opts.SPI if opts.SPI > 0 else recommend_spi(self._width, self._height)
        # Runtime.cpp:247: lopts.Denoiser.Enabled = !IsTracer && Denoiser.Enabled && hasDenoiser(), with the warning of :260-262
        self._denoise = bool(opts.Denoiser.Enabled) and not opts.IsTracer and hasDenoiser()
        if opts.Denoiser.Enabled and not opts.IsTracer and not hasDenoiser():
            import sys
            print("[ignis_amd] warning: Trying to use denoiser but no denoiser is available", file=sys.stderr)
        self._device = Device(opts.Device, opts.AcquireStats, opts.StreamCapacity,
                              info_aovs=(opts.EnableInfoAOVs or self._denoise) and not opts.IsTracer)
        self._device.assign_scene(scene)
        self._iteration = 0
        self._samples = 0
        self._frame = 0
        cam = sc.camera
        self._initial_orientation = CameraOrientation(tuple(cam.eye), tuple(cam.dir), tuple(cam.up))
        # the global registry of Runtime (Runtime.cpp:696-719): what the device does not read is only stored
        self.IntParameters, self.FloatParameters, self.VectorParameters = {}, {}, {}
        self.ColorParameters, self.StringParameters = {}, {}
        if not opts.IsTracer:
            self._device.resize(self._width, self._height)

    @property
    def device(self):
        """The render device (ignis_amd.Device) behind this runtime: what ignis_amd.comm.Comm attaches its communicator to."""
        return self._device

    # -- context manager like RuntimeWrap (runtime.cpp:313-320)
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.shutdown()
        return False

    def shutdown(self):
        if self._device is not None:
            self._device.close()
            self._device = None
        if self._scene is not None:
            self._scene.close()
            self._scene = None

    # -- Runtime::step (Runtime.cpp:334-387)
    def step(self, ignoreDenoiser=False):
        if self._opts.IsTracer:
            raise RuntimeError("Trying to use step() in a trace driver!")
        self._device.render(self._spi, self._width, self._height, iteration=self._iteration, frame=self._frame,
                            seed=self._opts.Seed, row_offset=self._opts.RowOffset, row_stride=self._opts.RowStride)
        self._samples += self._spi
        self._iteration += 1
        if self._denoise and not ignoreDenoiser:
            self._runDenoiser()

    # -- OIDN::run on the host route (extra/OIDN.cpp:100-127): colour, "Normals", "Albedo" in, "Denoised" out and uploaded
    def _runDenoiser(self):
        color = self._device.framebuffer(None)
        normals = self._device.framebuffer("Normals")
        albedo = self._device.framebuffer("Albedo")
        out = np.ascontiguousarray(_denoiser(color, normals, albedo, self._opts.Denoiser), dtype=np.float32)
        if out.shape != color.shape:
            raise ValueError(f"the denoiser returned an image of shape {out.shape}, expected {color.shape}")
        self._device.upload_framebuffer(out, "Denoised")

    # -- several iterations as one wavefront (igd_render_settings.iterations): bit-identical to `count` step() calls, but
    # the launches are `count` times larger — what a small film needs to fill the GPU
    def stepMany(self, count):
        if self._opts.IsTracer:
            raise RuntimeError("Trying to use step() in a trace driver!")
        count = max(1, int(count))
        self._device.render(self._spi, self._width, self._height, iteration=self._iteration, frame=self._frame, seed=self._opts.Seed,
                            row_offset=self._opts.RowOffset, row_stride=self._opts.RowStride, iterations=count)
        self._samples += self._spi * count
        self._iteration += count
        if self._denoise:
            self._runDenoiser()

    def recommendedBatch(self):
        """Iterations per call that make a call about as large as a full 1080p iteration at spi 8 (16.6 M camera rays)."""
        rays = self._width * self._height * self._spi / max(1, self._opts.RowStride)
        return int(max(1, min(64, (1 << 24) // max(1, int(rays)))))

    # -- Runtime::trace (Runtime.cpp:389-446): returns (n, 3) radiance, accumulated over calls
    def trace(self, rays):
        if not self._opts.IsTracer:
            raise RuntimeError("Trying to use trace() in a camera driver!")
        arr = np.empty((len(rays), 8), dtype=np.float32)
        for i, r in enumerate(rays):
            d = np.asarray(r.Direction, dtype=np.float32)
            n = np.linalg.norm(d)
            arr[i, 0:3] = r.Origin
            arr[i, 3:6] = d / n if n > 0 else d  # rays are normalised on upload (Device.cpp:602-643)
            arr[i, 6:8] = r.Range
        self._device.render(self._spi, len(rays), 1, iteration=self._iteration, frame=self._frame,
                            seed=self._opts.Seed, rays=arr)
        self._samples += self._spi
        self._iteration += 1
        return self._device.framebuffer().reshape(-1, 3)[: len(rays)]

    def reset(self):
        self._device.clear_framebuffer()
        self._iteration = 0
        self._samples = 0

    def getFramebufferForHost(self, aov=""):
        return self._device.framebuffer(aov or None)

    def clearFramebuffer(self, aov=""):
        self._device.clear_framebuffer(aov or None)

    # -- Runtime::setParameter overloads (Runtime.cpp:696-719); the device reads the ones igd_device.h lists
    def setParameter(self, name, value):
        if isinstance(value, str):
            self.StringParameters[name] = value
            return
        if isinstance(value, (bool, int, np.integer)):
            self.IntParameters[name] = int(value)
        elif isinstance(value, (float, np.floating)):
            self.FloatParameters[name] = float(value)
        else:
            v = tuple(float(x) for x in value)
            if len(v) == 4:
                self.ColorParameters[name] = v
                return
            self.VectorParameters[name] = v
            value = v
        self._device.set_parameter(name, value)

    # -- Runtime::setCameraOrientation (Runtime.cpp:736-741)
    def setCameraOrientation(self, orientation):
        self.setParameter("__camera_eye", orientation.Eye)
        self.setParameter("__camera_dir", orientation.Dir)
        self.setParameter("__camera_up", orientation.Up)

    def getCameraOrientation(self):
        g = self.VectorParameters
        return CameraOrientation(g.get("__camera_eye", (0, 0, 0)), g.get("__camera_dir", (0, 0, 1)), g.get("__camera_up", (0, 1, 0)))

    InitialCameraOrientation = property(lambda self: self._initial_orientation)

    # -- Runtime::saveFramebuffer (Runtime.cpp:794-876): mean over the iterations so far, channels B, G, R
    def saveFramebuffer(self, path, fb=None):
        """`fb`: the accumulated film to write instead of this device's own (the multi-GPU CLI hands rank 0 the gathered one)."""
        fb = self._device.framebuffer() if fb is None else np.asarray(fb, dtype=np.float32)
        if self._opts.IsTracer:
            fb = fb.reshape(1, -1, 3)
        o = self.getCameraOrientation() if "__camera_eye" in self.VectorParameters else self._initial_orientation
        meta = {"igTechniqueType": self.Technique, "igCameraType": self.Camera, "igSeed": self.Seed, "igSPP": self._samples,
                "igSPI": self._spi, "igIteration": self._iteration, "igFrame": self._frame, "igTargetString": "MI355X (gfx950, HIP)",
                "igCameraEye": o.Eye, "igCameraUp": o.Up, "igCameraDir": o.Dir}
        try:
            save_exr(path, fb, 1.0 / self._iteration if self._iteration > 0 else 1.0, meta)
            return True
        except RuntimeError:
            return False

    def synchronize(self):
        """Everything submitted so far (deferred iterations, overlapped tails, resolves) has reached the framebuffer."""
        self._device.synchronize()

    def framebufferTensor(self, torch, on_device=True):
        """The accumulated film as a torch tensor [H, W, 3] for the one collective of the tile-sharded path (ignis_amd/sharding.py):
        a zero-copy view of the device framebuffer for the RCCL backend, a host copy for gloo."""
        self._device.synchronize()
        if not on_device:
            return torch.from_numpy(np.ascontiguousarray(self._device.framebuffer()))

        class _View:  # __cuda_array_interface__ over the device pointer (no copy)
            pass
        v = _View()
        v.__cuda_array_interface__ = {"shape": (self._height, self._width, 3), "typestr": "<f4", "data": (self._device.framebuffer_device_ptr(), False), "version": 2}
        return torch.as_tensor(v, device=torch.device("cuda", self._opts.Device))

    def incFrameCount(self):
        self._frame += 1

    def getStatistics(self):
        return self._device.stats()

    IterationCount = property(lambda self: self._iteration)
    SampleCount = property(lambda self: self._samples)
    FrameCount = property(lambda self: self._frame)
    FramebufferWidth = property(lambda self: self._width)
    FramebufferHeight = property(lambda self: self._height)
    Seed = property(lambda self: self._opts.Seed)
    SPI = property(lambda self: self._spi)
    Technique = property(lambda self: "path")
    Camera = property(lambda self: "perspective")

    @property
    def SceneBoundingBox(self):
        sc = self._scene.scene
        return tuple(sc.bbox_min), tuple(sc.bbox_max)


def _film_override(opts):
    w, h = opts.OverrideFilmSize
    return int(w), int(h)


def loadFromFile(path, opts=None):
    """ignis.loadFromFile (runtime.cpp:322-340)."""
    opts = opts or RuntimeOptions.makeDefault()
    w, h = _film_override(opts)
    return Runtime(LoadedScene.from_file(str(path), w, h), opts)


def loadFromString(text, opts=None, dir=""):
    """ignis.loadFromString (runtime.cpp:342-360)."""
    opts = opts or RuntimeOptions.makeDefault()
    w, h = _film_override(opts)
    return Runtime(LoadedScene.from_string(text, str(dir), w, h), opts)


def loadFromScene(scene, *args):
    """ignis.loadFromScene(scene[, dir][, opts]) (runtime.cpp:340-350, Runtime::loadFromScene, Runtime.cpp:219): a scene assembled
    or edited through ignis_amd.scene.Scene. The one loader of this backend reads JSON, so the scene is lowered to that text."""
    import json
    import os
    from .scene import Scene
    if not isinstance(scene, Scene):
        raise TypeError("loadFromScene expects an ignis_amd.Scene")
    opts, dir = None, ""
    for a in args:
        if isinstance(a, RuntimeOptions):
            opts = a
        elif isinstance(a, (str, os.PathLike)):
            dir = os.fspath(a)
        else:
            raise TypeError("loadFromScene(scene[, dir][, opts])")
    return loadFromString(json.dumps(scene.toJSON()), opts, dir)
