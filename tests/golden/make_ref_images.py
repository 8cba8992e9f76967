#!/usr/bin/env python3
"""Brings the reference's own end-to-end check into the repo as data fixtures (run in the build container, where
/root/reference exists; the GPU box only sees the committed copies).

The reference validates its renderers against images of other renderers (Mitsuba, Cycles, Radiance):
`scripts/RunEvaluations.py` renders every `scenes/evaluation/*.json` at 1024 spp, compares with
`scenes/evaluation/references/ref-<scene>*.exr` through `error_image` (relative squared error, clipped at its 99th
percentile) and passes a scene whose error is below 1e-3 or a per-scene bound (`predef_eps`); `scripts/evaluation/MakeHtml.py`
reports the same pairs. This script copies
  * the scene descriptions and their meshes         -> scenes/evaluation/            (data: JSON, PLY, OBJ)
  * the reference images, byte for byte             -> tests/golden/references/*.exr (data: OpenEXR, ZIP or PIZ)
and checks that every image decodes with tests/golden/exr_decode.py. Nothing else is read from the reference.
"""
import glob
import os
import shutil
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference/scenes/evaluation"
sys.path.insert(0, HERE)


def main():
    import numpy as np
    import exr_decode
    dst_scenes = os.path.join(ROOT, "scenes", "evaluation")
    dst_refs = os.path.join(HERE, "references")
    os.makedirs(os.path.join(dst_scenes, "meshes"), exist_ok=True)
    os.makedirs(dst_refs, exist_ok=True)
    for p in sorted(glob.glob(os.path.join(REF, "*.json"))):
        shutil.copyfile(p, os.path.join(dst_scenes, os.path.basename(p)))
    for p in sorted(glob.glob(os.path.join(REF, "meshes", "*"))):
        shutil.copyfile(p, os.path.join(dst_scenes, "meshes", os.path.basename(p)))
    # files the scenes name outside their own directory (../meshes/Room.obj, ../textures/...), where the reference has them
    import re
    for p in sorted(glob.glob(os.path.join(REF, "*.json"))):
        for rel in re.findall(r'"filename"\s*:\s*"(\.\./[^"]+)"', open(p).read()):
            src = os.path.normpath(os.path.join(REF, rel))
            dst = os.path.normpath(os.path.join(dst_scenes, rel))
            if os.path.isfile(src) and os.path.getsize(src) < (1 << 20) and not rel.endswith(".xml"):
                os.makedirs(os.path.dirname(dst), exist_ok=True)
                shutil.copyfile(src, dst)
                os.chmod(dst, 0o644)
    n = 0
    for p in sorted(glob.glob(os.path.join(REF, "references", "*.exr"))):
        out = os.path.join(dst_refs, os.path.basename(p))
        shutil.copyfile(p, out)
        img = exr_decode.read_rgb(out)
        assert img.shape == (256, 256, 3) and np.isfinite(img).all(), p
        n += 1
    for d in (dst_scenes, os.path.join(dst_scenes, "meshes"), dst_refs):
        for f in os.listdir(d):
            os.chmod(os.path.join(d, f), 0o644)
    print(f"{n} reference images copied and decoded")


if __name__ == "__main__":
    main()
