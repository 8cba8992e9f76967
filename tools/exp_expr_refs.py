import sys, os, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0,'tests'); sys.path.insert(0,'tests/golden')
import numpy as np
import exr_decode
from ignis_amd.tables import LoadedScene
from ignis_amd.device import Device
from test_reference_images import reference_for, robust_mean_ratio, error_image, box
dev = Device()
for stem in ["cycles-roughness-base","cycles-roughness-rxry","cycles-roughness-raniso","cycles-normalmap","cycles-bumpmap"]:
    try:
        sc = LoadedScene.from_file(f"scenes/evaluation/{stem}.json",256,256)
    except Exception as e:
        print(stem, "LOAD FAIL", e); continue
    ref = exr_decode.read_rgb(reference_for(stem))
    dev.assign_scene(sc); dev.resize(256,256); dev.clear_framebuffer()
    for spp_it in (64, 256, 1024):
        dev.clear_framebuffer()
        t=time.time()
        for it in range(spp_it):
            dev.render(16,256,256,iteration=it,seed=1)
        fb = dev.framebuffer()/spp_it
        print(stem, spp_it*16, "spp ratio %.4f err1 %.4e err8 %.4e  t %.1fs" % (robust_mean_ratio(fb,ref), error_image(fb,ref), error_image(box(fb,8),box(ref,8)), time.time()-t), flush=True)
    np.save(f"gpurun_out/{stem}.npy", fb)
