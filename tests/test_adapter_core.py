"""The part of the IRenderDevice adapter that compiles without reference headers (ignis_amd/csrc/adapter/hip_adapter_core.*):
scene hand-over from SceneDatabase-shaped byte tables, registry forwarding, named buffers, statistics mapping. The shell that
derives from IG::IRenderDevice (HipRenderDevice.cpp) needs Eigen through IG_Config.h and is NOT compiled here."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import ROOT, SCENES


def _lib():
    from ignis_amd import device, tables
    device.lib()
    tables.host_lib()
    l = C.CDLL(os.path.join(ROOT, "ignis_amd", "lib", "libig_adapter_core.so"))
    l.iga_create.restype = C.c_void_p
    l.iga_create.argtypes = [C.c_int, C.c_int, C.c_int]
    l.iga_destroy.argtypes = [C.c_void_p]
    l.iga_ok.argtypes = [C.c_void_p]
    l.iga_error.restype = C.c_char_p
    l.iga_error.argtypes = [C.c_void_p]
    l.iga_set_scene_file.argtypes = [C.c_void_p, C.c_char_p]
    l.iga_assign_scene.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64), C.POINTER(C.c_int32), C.c_uint64]
    l.iga_used_runtime_tables.argtypes = [C.c_void_p]
    l.iga_check_database.restype = C.c_char_p
    l.iga_check_database.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64), C.POINTER(C.c_int32), C.c_uint64]
    l.iga_forward_int.argtypes = [C.c_void_p, C.c_char_p, C.c_int32]
    l.iga_forward_float.argtypes = [C.c_void_p, C.c_char_p, C.c_float]
    l.iga_forward_vector.argtypes = [C.c_void_p, C.c_char_p, C.c_float, C.c_float, C.c_float]
    l.iga_render.argtypes = [C.c_void_p, C.POINTER(C.c_float)] + [C.c_uint64] * 6
    l.iga_framebuffer_host.restype = C.POINTER(C.c_float)
    l.iga_framebuffer_host.argtypes = [C.c_void_p, C.c_char_p]
    l.iga_buffer_size.restype = C.c_uint64
    l.iga_buffer_size.argtypes = [C.c_void_p, C.c_char_p]
    l.iga_copy_buffer.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_uint64]
    l.iga_drain_statistics.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
    return l


def _database(scene):
    """The loader's own tables shaped like the runtime's SceneDatabase (byte arrays, as FixTable / DynTable / SceneBVH hold them)."""
    s = scene.scene
    arrs = [
        np.ctypeslib.as_array(C.cast(s.entities, C.POINTER(C.c_uint8)), shape=(s.entity_count * 36 * 4,)).copy(),
        np.ctypeslib.as_array(C.cast(s.shape_lookups, C.POINTER(C.c_uint8)), shape=(s.shape_count * 16,)).copy(),
        np.ctypeslib.as_array(s.shape_data, shape=(s.shape_data_size,)).copy(),
        np.ctypeslib.as_array(s.primbvh, shape=(s.primbvh_size,)).copy(),
        np.ctypeslib.as_array(C.cast(s.scene_nodes, C.POINTER(C.c_uint8)), shape=(s.scene_node_count * 256,)).copy(),
        np.ctypeslib.as_array(C.cast(s.scene_leaves, C.POINTER(C.c_uint8)), shape=(s.scene_leaf_count * 96,)).copy(),
    ]
    mats = np.frombuffer(arrs[0], np.int32).reshape(-1, 36)[:, 34]
    per_material = np.bincount(mats, minlength=int(mats.max()) + 1).astype(np.int32)
    return arrs, per_material


def _pack(arrs):
    ptrs = (C.c_void_p * 6)(*[a.ctypes.data for a in arrs])
    sizes = (C.c_uint64 * 6)(*[a.size for a in arrs])
    return ptrs, sizes


def test_adapter_core_exports_and_database_check(diamond_scene):
    l = _lib()
    arrs, per_mat = _database(diamond_scene)
    pm = per_mat.ctypes.data_as(C.POINTER(C.c_int32))
    ptrs, sizes = _pack(arrs)
    assert l.iga_check_database(diamond_scene._h, ptrs, sizes, pm, per_mat.size) == b""
    # a GPU target of the reference builds BVH2 tables (64-byte nodes): refused, the host library's tables are used instead
    bvh2 = list(arrs)
    bvh2[4] = arrs[4][:64 * 3].copy()
    ptrs2, sizes2 = _pack(bvh2)
    assert b"<8, 4> layout" in l.iga_check_database(diamond_scene._h, ptrs2, sizes2, pm, per_mat.size)
    # the runtime numbers entities in unordered_map order: a different order is detected
    swapped = list(arrs)
    e = arrs[0].copy().reshape(-1, 144)
    e[[0, 1]] = e[[1, 0]]
    swapped[0] = e.reshape(-1)
    ptrs3, sizes3 = _pack(swapped)
    assert b"numbered entities differently" in l.iga_check_database(diamond_scene._h, ptrs3, sizes3, pm, per_mat.size)
    wrong = per_mat.copy()
    wrong[0] += 1
    assert b"entity_per_material" in l.iga_check_database(diamond_scene._h, ptrs, sizes, wrong.ctypes.data_as(C.POINTER(C.c_int32)), wrong.size)


@pytest.mark.gpu
def test_adapter_core_renders_from_runtime_tables(diamond_scene):
    """assignScene with SceneDatabase-shaped tables, the registry re-sent before every iteration (as IRenderDevice::render
    receives it), render x 3, framebuffer, named buffers, statistics: identical to driving the C ABI directly."""
    from ignis_amd import Device
    l = _lib()
    path = os.path.join(SCENES, "diamond_scene.json").encode()
    arrs, per_mat = _database(diamond_scene)
    ptrs, sizes = _pack(arrs)
    core = l.iga_create(0, 1, 0)
    assert l.iga_ok(core), l.iga_error(core)
    assert not l.iga_assign_scene(core, ptrs, sizes, per_mat.ctypes.data_as(C.POINTER(C.c_int32)), per_mat.size)
    assert b"no scene description" in l.iga_error(core)
    assert l.iga_set_scene_file(core, path), l.iga_error(core)
    assert l.iga_assign_scene(core, ptrs, sizes, per_mat.ctypes.data_as(C.POINTER(C.c_int32)), per_mat.size), l.iga_error(core)
    assert l.iga_used_runtime_tables(core)
    cam = diamond_scene.scene.camera
    for it in range(3):
        l.iga_forward_int(core, b"__tech_max_depth", int(diamond_scene.scene.technique.max_depth))
        l.iga_forward_float(core, b"__tech_clamp", float(diamond_scene.scene.technique.clamp))
        l.iga_forward_vector(core, b"__camera_eye", cam.eye[0], cam.eye[1], cam.eye[2])
        assert l.iga_render(core, None, 4, 128, 128, it, 0, 5), l.iga_error(core)
    fb = np.ctypeslib.as_array(l.iga_framebuffer_host(core, b""), shape=(128, 128, 3)).copy()
    stats = (C.c_uint64 * 11)()
    l.iga_drain_statistics(core, stats)

    dev = Device(0, acquire_stats=1)
    dev.assign_scene(diamond_scene)
    for it in range(3):
        dev.render(4, 128, 128, iteration=it, seed=5)
    np.testing.assert_array_equal(fb, dev.framebuffer())
    # ... and to the oracle's image of the same three iterations (the adapter is one more way into the HIP path, checked like the others)
    import oracle
    ref = np.zeros((128, 128, 3), np.float32)
    for it in range(3):
        ref += oracle.render(diamond_scene, 4, 128, 128, iteration=it, seed=5)[0]
    assert float(np.linalg.norm(fb - ref) / np.linalg.norm(ref)) <= 1e-4
    st = dev.stats()
    assert (stats[0], stats[1], stats[2]) == (st["camera_rays"], st["shadow_rays"], st["bounce_rays"])
    assert stats[4] == st["camera_rays"] + st["bounce_rays"] and stats[6] == st["shadow_rays"]
    again = (C.c_uint64 * 11)()
    l.iga_drain_statistics(core, again)
    assert list(again)[:3] == [0, 0, 0]  # drained: nothing is counted twice

    # named buffers (IRenderDevice::getBufferSizeInBytes / copyBufferToHost): the tables under the reference's names
    for name, arr in (("entities", arrs[0]), ("trimesh_primbvh", arrs[3]), ("scene_bvh_nodes", arrs[4]), ("scene_bvh_leaves", arrs[5]), ("shapes", arrs[2])):
        assert l.iga_buffer_size(core, name.encode()) == arr.size, name
        out = np.empty(arr.size, np.uint8)
        assert l.iga_copy_buffer(core, name.encode(), out.ctypes.data, out.size)
        np.testing.assert_array_equal(out, arr, err_msg=name)
    assert l.iga_buffer_size(core, b"no such buffer") == 0
    np.testing.assert_array_equal(dev.buffer("Color").view(np.float32).reshape(128, 128, 3), fb)
    assert dev.buffer("nope") is None
    dev.close()

    # tables the runtime built for a GPU target (BVH2) are refused with a reason and the host library's are used: same image
    bvh2 = list(arrs)
    bvh2[4] = arrs[4][:64 * 3].copy()
    ptrs2, sizes2 = _pack(bvh2)
    assert l.iga_assign_scene(core, ptrs2, sizes2, per_mat.ctypes.data_as(C.POINTER(C.c_int32)), per_mat.size)
    assert not l.iga_used_runtime_tables(core) and b"<8, 4> layout" in l.iga_error(core)
    l.iga_destroy(core)


@pytest.mark.gpu
def test_blocking_render_contract(diamond_scene):
    """igd_setup.blocking_render: every igd_render completes before it returns (the reference's contract) — same image as the
    deferred default."""
    from ignis_amd import Device
    a = Device(0, acquire_stats=1, blocking_render=True)
    b = Device(0, acquire_stats=1)
    for d in (a, b):
        d.assign_scene(diamond_scene)
        for it in range(3):
            d.render(4, 128, 128, iteration=it, seed=8)
    np.testing.assert_array_equal(a.framebuffer(), b.framebuffer())
    a.close()
    b.close()


@pytest.mark.gpu
def test_adapter_core_renders_the_light_tracer():
    """The adapter keeps the "Normals" / "Albedo" film buffers (igd_setup.info_aovs) for the runtime's denoiser; a light tracer scene
    renders through it all the same (no camera-flagged ray ever reaches the info-buffer wrapper, technique/internal/infobuffer.art:13,
    so both stay zero) and gives the image of the direct C ABI."""
    from ignis_amd import Device, LoadedScene
    l = _lib()
    path = os.path.join(SCENES, "evaluation", "cycles-lights-lt.json")
    core = l.iga_create(0, 1, 0)
    assert l.iga_ok(core), l.iga_error(core)
    assert l.iga_set_scene_file(core, str(path).encode()), l.iga_error(core)
    assert l.iga_assign_scene(core, None, None, None, 0), l.iga_error(core)  # (no database: the host library's tables)
    for it in range(2):
        assert l.iga_render(core, None, 4, 96, 96, it, 0, 7), l.iga_error(core)
    fb = np.ctypeslib.as_array(l.iga_framebuffer_host(core, b""), shape=(96, 96, 3)).copy()
    nrm = np.ctypeslib.as_array(l.iga_framebuffer_host(core, b"Normals"), shape=(96, 96, 3)).copy()
    assert fb.sum() > 0 and not nrm.any()
    sc = LoadedScene.from_file(str(path), 96, 96)
    dev = Device(0, acquire_stats=1)
    dev.assign_scene(sc)
    for it in range(2):
        dev.render(4, 96, 96, iteration=it, seed=7)
    ref = dev.framebuffer()
    dev.close()
    assert float(np.linalg.norm(fb - ref) / np.linalg.norm(ref)) <= 1e-5  # (float atomics in the connection splats: order dependent)
    l.iga_destroy(core)
