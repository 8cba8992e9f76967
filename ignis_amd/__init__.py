"""ignis_amd — MI355X-native ray-tracing core for the Ignis renderer's hot path.

Layout (only what the path needs):
  csrc/host/    scene loader -> SceneDatabase-compatible tables   (libig_host.so, include/igh_host.h)
  csrc/device/  HIP kernels + render device                       (libig_device_hip.so, include/igd_device.h)
  tables.py / device.py   ctypes views of the two C ABIs
  runtime.py    mirror of the reference's Python `ignis` module API for this path
"""
from .runtime import (CameraOrientation, DenoiserSettings, Ray, Runtime, RuntimeOptions, hasDenoiser, loadFromFile,  # noqa: F401
                      loadFromScene, loadFromString, registerDenoiser)  # noqa: F401
from .device import Device, DeviceError, device_count  # noqa: F401
from .tables import LoadedScene  # noqa: F401
from .scene import Scene, SceneObject, SceneParser, SceneProperty  # noqa: F401

__version__ = "0.1.0"
