"""Work per ray on the tables the loader builds, counted by the CPU oracle (no GPU): nodes visited, triangles tested and entity
leaves scanned per ray of one small-film iteration, next to the shape of the tree (tools/bvh_stats.py). The builder's
environment knobs apply (IGH_BVH_REINSERT=0: without the reinsertion pass; IGH_BVH_REINSERT_ITERS, IGH_BVH_REINSERT_RATIO).
usage: python tools/bvh_visits.py scene.json [width height spi]"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import oracle  # noqa: E402
from ignis_amd.tables import LoadedScene  # noqa: E402

scene = sys.argv[1]
w, h, spi = (int(x) for x in sys.argv[2:5]) if len(sys.argv) > 4 else (192, 108, 2)
t0 = time.time()
sc = LoadedScene.from_file(scene, w, h)
t1 = time.time()
_, st = oracle.render(sc, spi, w, h, iteration=0, seed=1)
rays = st["camera_rays"] + st["bounce_rays"] + st["shadow_rays"]
knobs = " ".join(f"{k}={v}" for k, v in sorted(os.environ.items()) if k.startswith("IGH_"))
print(f"{os.path.basename(scene)} [{knobs or 'defaults'}]: load + build {t1 - t0:.1f} s; {rays} rays: "
      f"{st['nodes'] / rays:.3f} nodes, {st['tris'] / rays:.3f} triangles, {st['leaves'] / rays:.3f} entity leaves per ray")
