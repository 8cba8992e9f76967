"""Where a k_tail wave's cycles go in one pass of the tail: a render with a device library built with -DIG_TAIL_CLOCKS=<pass>
(tools/build_variant.sh tailclk6 -DIG_TAIL_CLOCKS=6). Shader-clock cycles (memory drained at every mark) summed over the waves that took
part in that pass, and the longest wave. The variant reports through the section counters of igd_stats.
usage: IGD_LIBRARY=ignis_amd/lib/var/libig_device_hip_tailclk6.so python tools/tail_clocks.py [scene.json] [width height spi iterations]"""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from ignis_amd import Device, LoadedScene  # noqa: E402

scene = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(__file__), "..", "scenes", "diamond_scene.json")
w, h, spi, its = (int(x) for x in sys.argv[2:6]) if len(sys.argv) > 5 else (1920, 1080, 8, 3)
sc = LoadedScene.from_file(scene, w, h)
dev = Device(0)
dev.assign_scene(sc)
dev.render(spi, w, h, iteration=0, seed=1, iterations=its)
st = dev.stats()
dev.close()
names = ["refill", "closest-hit traversal", "shading", "any-hit traversal + splat", "loop / spill of long paths"]
acc, aux = st["section_passes"], st["section_lanes"]
waves = aux[0] or 1
total = float(sum(acc[:5])) or 1.0
print(json.dumps({"scene": os.path.basename(scene), "library": os.path.basename(os.environ.get("IGD_LIBRARY", "")), "waves_in_pass": int(waves),
                  "cycles_per_wave": round(total / waves), "longest_wave_cycles": int(aux[1]), "traversal_passes_per_wave": round(acc[5] / waves, 1),
                  "shares": {n: round(c / total, 4) for n, c in zip(names, acc[:5])},
                  "cycles_per_traversal_pass": round(acc[1] / max(acc[5], 1))}, indent=1))
