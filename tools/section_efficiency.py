"""Useful share of the section instructions k_traverse issues: lanes with work / (64 x wave-level executions), per section and
kernel, from the counters of the STATS kernel variants (igd_stats.section_passes / section_lanes).
usage: python tools/section_efficiency.py [scene.json] [width height spi iterations]"""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from ignis_amd import Device, LoadedScene  # noqa: E402

scene = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(__file__), "..", "scenes", "diamond_scene.json")
w, h, spi, its = (int(x) for x in sys.argv[2:6]) if len(sys.argv) > 5 else (1920, 1080, 8, 4)
sc = LoadedScene.from_file(scene, w, h)
dev = Device(0, acquire_stats=True)
dev.assign_scene(sc)
for it in range(its):
    dev.render(spi, w, h, iteration=it, seed=1)
st = dev.stats()
dev.close()
names = ("entity leaf", "inner node", "triangle packet")
out = {"scene": os.path.basename(scene), "rays": st["camera_rays"] + st["bounce_rays"] + st["shadow_rays"]}
for k, kernel in ((0, "closest hit"), (3, "any hit")):
    rows = {}
    tot_p = tot_l = 0
    for i, n in enumerate(names):
        p, l = st["section_passes"][k + i], st["section_lanes"][k + i]
        rows[n] = {"passes": p, "lanes": l, "useful_share": round(l / (64.0 * p), 4) if p else None}
        tot_p += p
        tot_l += l
    rows["all sections"] = {"passes": tot_p, "lanes": tot_l, "useful_share": round(tot_l / (64.0 * tot_p), 4) if tot_p else None}
    out[kernel] = rows
print(json.dumps(out, indent=1))
