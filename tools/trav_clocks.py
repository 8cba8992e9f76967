"""Where a wave of k_traverse spends its cycles: runs a render with a device library built with -DIG_TRAV_CLOCKS
(tools/build_variant.sh tclocks -DIG_TRAV_CLOCKS; the marks drain the memory counters, so a phase carries the latency of its own loads)
and prints the shares for the closest-hit and the any-hit launches. The variant reports through the section counters of igd_stats.
usage: IGD_LIBRARY=ignis_amd/lib/var/libig_device_hip_tclocks.so python tools/trav_clocks.py [scene.json] [width height spi iterations]"""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from ignis_amd import Device, LoadedScene  # noqa: E402

scene = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(__file__), "..", "scenes", "diamond_scene.json")
w, h, spi, its = (int(x) for x in sys.argv[2:6]) if len(sys.argv) > 5 else (1920, 1080, 8, 8)
sc = LoadedScene.from_file(scene, w, h)
dev = Device(0)
dev.assign_scene(sc)
dev.render(spi, w, h, iteration=0, seed=1, iterations=its)
st = dev.stats()
dev.close()
names = ["epilogue (hit stores / splat), loop bookkeeping", "entity-leaf section", "inner-node section", "triangle section", "settle + quorum at the top of a pass", "refill (batch reservation, ray loads, begin())"]
out = {"scene": os.path.basename(scene)}
for kernel, acc in (("closest hit", st["section_passes"]), ("any hit", st["section_lanes"])):
    total = float(sum(acc)) or 1.0
    out[kernel] = {"wave_cycles": int(total), "phases": {n: round(c / total, 4) for n, c in zip(names, acc) if n != "-"}}
print(json.dumps(out, indent=1))
