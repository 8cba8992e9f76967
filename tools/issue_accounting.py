"""One number for the VALU issue share of k_traverse<closest>, derived from files:

  dynamic opcode histogram = static opcode counts of the kernel's parts (tools/isa_regions.py --json, built with -DIG_ISA_MARKS)
                             x how often each part runs per ray (tools/trav_events.py, the -DIG_TRAV_PROFILE build)
  scaled to the measured SQ_INSTS_VALU per ray (parts contain blocks that are skipped when no lane takes them)
  x the price of each opcode class (profiles/<tag>_valu_calibration.txt, 4 waves per SIMD)
  = SIMD cycles of VALU issue per ray  ->  mean cycles per VALU instruction (what bench.py and tools/prof_summary.py price with)
  /  SIMD cycles available per ray (1024 SIMDs x clock x launch time / rays)  = issue share; the same for the scalar unit.

usage: python tools/issue_accounting.py <tag>      reads profiles/<tag>_{valu_calibration.txt,trav_events_closest.json,traffic.json,rocprofv3_pmc.txt}
       writes profiles/<tag>_issue_accounting.json, prints it"""
import collections
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
P = lambda n: os.path.join(ROOT, "profiles", f"{tag}_{n}")
SHADER_GHZ = 2.1  # effective clock of the traversal launches under load (bench.py)

# ---- prices: cycles per wave64 instruction on one SIMD at 4 waves per SIMD
price = {}
for line in open(P("valu_calibration.txt")):
    m = re.match(r"^(\S.*?)\s{2,}4\s+\d+\s+([\d.]+)", line)
    if m:
        price[m.group(1).strip()] = float(m.group(2))
two_op = price["v_add_f32"]
wide = price["v_max_f32"]
classes = {
    "two-operand fp32 / integer / logic / move": (two_op, re.compile(r"^v_(add|sub|subrev|mul|fmac|and|or|xor|not|mov|lshlrev|lshrrev|ashrrev|cvt|mbcnt_lo|mbcnt_hi|bcnt)_")),
    "v_fma_f32": (price["v_fma_f32"], re.compile(r"^v_fma_f32")),
    "min / max / compare / select / three-operand integer": (wide, re.compile(r"^v_(max|min|max3|min3|med3|cmp|cmpx|cndmask|lshl_add|add3|mad|bfe|bfi|bitop3|and_or|or3|xad|lshl_or|readfirstlane|div_scale|div_fmas|div_fixup|pk_)")),
    "v_rcp_f32 / transcendental": (price["v_rcp_f32"], re.compile(r"^v_(rcp|rsq|sqrt|exp|log|sin|cos)_")),
}


def cost_of(op):
    for name, (c, rx) in classes.items():
        if rx.match(op):
            return name, c * (2.0 if op.endswith(("_b64", "_u64")) and not op.startswith("v_cmp") else 1.0)
    return "min / max / compare / select / three-operand integer", wide  # unknown VOP3: the wide price


# ---- static opcode counts per part
regions = json.loads(subprocess.run([sys.executable, os.path.join(ROOT, "tools", "isa_regions.py"), "closest", "--json"], check=True, capture_output=True, text=True).stdout)
ev = json.load(open(P("trav_events_closest.json")))["per_64_rays"]
freq = {  # executions per 64 rays of each part
    "pass.head": ev["main-loop passes"], "pass.quorum": ev["main-loop passes"], "epilogue": ev["main-loop passes"],
    "refill": ev["refill blocks"], "refill.end": ev["refill blocks"],
    "leaf.scan": ev["leaf scan iterations"], "leaf.enter": ev["leaf enter blocks"], "leaf.settle": ev["leaf sections"], "leaf.end": ev["leaf sections"],
    "node.half0": ev["node sections"], "node.half1": ev["node second halves"], "node.settle": ev["node sections"], "node.end": ev["node sections"],
    "tri.half0": ev["tri packet iterations"], "tri.half1": ev["tri second halves"], "tri.settle": ev["tri sections"], "tri.end": ev["tri sections"],
    "settle.cull": ev["settle iterations"], "settle.classify": ev["settle iterations"], "settle.ret": ev["settle iterations"], "settle.end": ev["settle iterations"],
    "prologue": 0.0,
}
dyn = collections.Counter()
salu_dyn = 0.0
for part, ops in regions["parts"].items():
    f = freq.get(part, 0.0)
    for op, n in ops.items():
        if op.startswith("v_"):
            dyn[op] += n * f
        elif op.startswith("s_") and not op.startswith(("s_waitcnt", "s_nop", "s_cbranch", "s_branch", "s_load", "s_endpgm")):
            salu_dyn += n * f

# ---- measured side
tj = json.load(open(P("traffic.json")))
pr = tj["closest_hit_per_ray"]
tk = next(v for k, v in tj["kernels"].items() if k.startswith("k_traverse<false, false, false"))
valu_per_64 = pr["valu_insts"] * 64.0  # wave-instructions per 64 rays
secs = tk["avg_ns_under_pmc"] * 1e-9
rays_per_launch = pr["rays_per_launch_profiled"]
salu_per_launch = None
pmc_txt = P("rocprofv3_pmc.txt")
if os.path.exists(pmc_txt):
    for line in open(pmc_txt):
        if "k_traverse<false, false, false" in line and "SQ_INSTS_SALU" in line:
            salu_per_launch = float(line.split()[-2])
predicted = sum(dyn.values())
scale = valu_per_64 / predicted
by_class = collections.OrderedDict()
cycles = 0.0
for op, n in dyn.items():
    name, c = cost_of(op)
    e = by_class.setdefault(name, {"cycles_per_inst": round(c, 3), "insts_per_64_rays": 0.0})
    e["insts_per_64_rays"] += n * scale
    cycles += n * scale * c
for e in by_class.values():
    e["insts_per_64_rays"] = round(e["insts_per_64_rays"], 1)
cpi = cycles / valu_per_64
avail = 1024 * SHADER_GHZ * 1e9 * secs / (rays_per_launch / 64.0)  # SIMD cycles per 64 rays
out = {
    "kernel": "k_traverse<closest>", "tag": tag,
    "valu_insts_per_64_rays_measured": round(valu_per_64, 1), "valu_insts_per_64_rays_static_x_events": round(predicted, 1),
    "static_x_events_over_measured": round(predicted / valu_per_64, 3),
    "note": "static counts of a part include blocks the wave skips when no lane takes them (hits, pushes, the division of a hit), hence the ratio above 1; the histogram is scaled to the measured total",
    "classes": by_class,
    "top_opcodes_per_64_rays": {op: round(n * scale, 1) for op, n in dyn.most_common(14)},
    "valu_cycles_per_inst": round(cpi, 3),
    "valu_issue_frac": round(cycles / avail, 4),
    "salu_insts_per_64_rays_measured": round(salu_per_launch / (rays_per_launch / 64.0), 1) if salu_per_launch else None,
    "salu_cycles_per_inst": price.get("s_and_b64 (scalar unit)"),
    "salu_issue_frac": round(salu_per_launch / (rays_per_launch / 64.0) * price["s_and_b64 (scalar unit)"] / avail, 4) if salu_per_launch and "s_and_b64 (scalar unit)" in price else None,
    "launch_seconds_under_pmc": secs, "shader_ghz": SHADER_GHZ,
    "sources": [f"profiles/{tag}_valu_calibration.txt", f"profiles/{tag}_trav_events_closest.json", f"profiles/{tag}_traffic.json", "tools/isa_regions.py (static counts of the committed kernel)"],
}
json.dump(out, open(P("issue_accounting.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
