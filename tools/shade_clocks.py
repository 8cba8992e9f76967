"""Where a wave of k_shade spends its cycles: runs a render with a device library built with -DIG_SHADE_CLOCKS
(tools/build_variant.sh clocks -DIG_SHADE_CLOCKS; the marks drain the memory counters, so a phase carries the latency of its own loads)
and prints the shares. The variant reports through the section counters of igd_stats (stats kernels off).
usage: IGD_LIBRARY=ignis_amd/lib/var/libig_device_hip_clocks.so python tools/shade_clocks.py [scene.json] [width height spi iterations]"""
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
acc = list(st["section_passes"]) + list(st["section_lanes"])
names = ["sort by material", "load the ray's columns", "surface element (entity, indices, vertices)", "material record + BSDF set-up", "emission (on_hit)",
         "next event estimation (on_shadow)", "bounce (on_bounce); a miss: all of it", "accumulator read-modify-write", "ballots / bins / first barrier of the append",
         "the append's stores (drained) and last barrier", "loop overhead + prologue", "reservation: bin scan + atomic with return by one thread, barrier"]
total = float(sum(acc)) or 1.0
out = {"scene": os.path.basename(scene), "wave_cycles": int(total), "phases": {n: {"cycles": int(c), "share": round(c / total, 4)} for n, c in zip(names, acc) if n != "-"}}
print(json.dumps(out, indent=1))
