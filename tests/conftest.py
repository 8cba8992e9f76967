import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

SCENES = os.path.join(ROOT, "scenes")
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _native_libs():
    """Build the native libraries once per session if they are missing (seconds; no GPU needed)."""
    need = [os.path.join(ROOT, "ignis_amd", "lib", "libig_host.so"),
            os.path.join(ROOT, "ignis_amd", "lib", "libig_device_hip.so"),
            os.path.join(ROOT, "oracle", "liboracle.so")]
    if not all(os.path.exists(p) for p in need):
        import __graft_entry__
        __graft_entry__.build()


@pytest.fixture(scope="session")
def diamond_scene():
    from ignis_amd.tables import LoadedScene
    return LoadedScene.from_file(os.path.join(SCENES, "diamond_scene.json"), 128, 128)


@pytest.fixture(scope="session", params=["tail", "rounds", "tail-wide", "rounds-sorted", "tail-wide8"])
def gpu_device(request):
    """The device every feature-parity test renders on, in three schedules. "tail": the product's default — a
    stream of <= 1 Mi paths is handed to k_tail before round 0, which is where every small-film test ends up. "rounds":
    IGD_TAIL_THRESHOLD=0, the same test through the wavefront kernels the benchmark runs (k_shade + k_traverse rounds to the
    last path). "tail-wide": IGD_TAIL_WIDE=64, k_tail with every closest-hit ray traversed by a whole wave (wide_core.h, what
    the product does for waves that follow <= 4 paths). "rounds-sorted": the rounds with every bounce and shadow stream traversed in
    (direction octant, origin cell) order (raysort.hip; IGD_RAY_SORT=1, what igd_assign_scene switches on for BVHs beyond 64 MB).
    "tail-wide8": IGD_TAIL_WIDE=0 IGD_TAIL_WIDE8=64, k_tail with every closest-hit ray traversed by a group of eight lanes, eight rays of a
    wave at a time (group_core.h, what waves that follow 5 - IGD_TAIL_WIDE8 paths do). The schedule is read when the device is created."""
    from ignis_amd import Device
    env = {"rounds": {"IGD_TAIL_THRESHOLD": "0"}, "tail-wide": {"IGD_TAIL_WIDE": "64"},
           "rounds-sorted": {"IGD_TAIL_THRESHOLD": "0", "IGD_RAY_SORT": "1", "IGD_RAY_SORT_MIN": "0"},
           "tail-wide8": {"IGD_TAIL_WIDE": "0", "IGD_TAIL_WIDE8": "64"}}.get(request.param, {})
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        dev = Device(0, acquire_stats=True)
    finally:
        for k, v in old.items():
            if v is None:
                del os.environ[k]
            else:
                os.environ[k] = v
    yield dev
    dev.close()


def flat_scene(lights=(), max_depth=2, size=(64, 64)):
    """The reference's integrator test scene (src/tests/integrator/common/__init__.py:37-65)."""
    return {
        "technique": {"type": "path", "max_depth": max_depth},
        "camera": {"type": "perspective", "fov": 90, "near_clip": 0.01, "far_clip": 100,
                   "transform": [1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, -1]},
        "film": {"size": list(size)},
        "bsdfs": [{"type": "diffuse", "name": "ground", "reflectance": [1, 1, 1]}],
        "shapes": [{"type": "rectangle", "name": "Bottom", "width": 2, "height": 2, "flip_normals": True}],
        "entities": [{"name": "Bottom", "shape": "Bottom", "bsdf": "ground"}],
        "lights": list(lights),
    }
