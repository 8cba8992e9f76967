"""How often each part of k_traverse runs: a render with a device library built with -DIG_TRAV_PROFILE=1 (closest hit) or =2 (any hit)
(tools/build_variant.sh tprof1 -DIG_TRAV_PROFILE=1). The variant reports through the section counters of igd_stats.
With the static instruction counts of the blocks (tools/isa_histogram.py --blocks) this gives the dynamic instruction mix.
usage: IGD_LIBRARY=ignis_amd/lib/var/libig_device_hip_tprof1.so python tools/trav_events.py [scene.json] [width height spi iterations]"""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from ignis_amd import Device, LoadedScene  # noqa: E402

scene = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(__file__), "..", "scenes", "diamond_scene.json")
w, h, spi, its = (int(x) for x in sys.argv[2:6]) if len(sys.argv) > 5 else (1920, 1080, 8, 4)
sc = LoadedScene.from_file(scene, w, h)
dev = Device(0)
dev.assign_scene(sc)
dev.render(spi, w, h, iteration=0, seed=1, iterations=its)
st = dev.stats()
dev.close()
names = ["main-loop passes", "refill blocks", "rays begun (lanes)", "leaf sections", "leaf scan iterations", "leaf enter blocks", "node sections", "node second halves",
         "tri sections", "tri packet iterations", "tri second halves", "settle iterations"]
ev = list(st["section_passes"]) + list(st["section_lanes"])
rays = ev[2] or 1
out = {"scene": os.path.basename(scene), "library": os.path.basename(os.environ.get("IGD_LIBRARY", "")), "rays": rays,
       "events": {n: int(c) for n, c in zip(names, ev)},
       "per_64_rays": {n: round(c * 64.0 / rays, 3) for n, c in zip(names, ev)}}
print(json.dumps(out, indent=1))
