"""Static instructions per part of k_traverse, from an assembly built with -DIG_ISA_MARKS (comment lines `; @@ name` at the
borders of the parts, traverse_core.h IG_MARK). Instructions are attributed to the last mark seen in layout order; the hot
loop is laid out in source order, cold blocks the compiler moved are attributed to whatever precedes them (small).
usage: python tools/isa_regions.py [any|closest] [--json] [-D...]      (--json: opcode counts per part, for tools/issue_accounting.py)"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLAGS = "--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -fPIC -Wno-everything".split()
which = "closest"
extra = []
as_json = False
for a in sys.argv[1:]:
    if a in ("any", "closest"):
        which = a
    elif a == "--json":
        as_json = True
    else:
        extra.append(a)
with tempfile.TemporaryDirectory() as tmp:
    out = os.path.join(tmp, "t.s")
    subprocess.run(["/opt/rocm/bin/hipcc", *FLAGS, "-I" + os.path.join(ROOT, "include"), "--cuda-device-only", "-S", "-DIG_ISA_MARKS", *extra, "-mllvm",
                    "-amdgpu-sched-strategy=max-memory-clause", os.path.join(ROOT, "ignis_amd", "csrc", "device", "traverse.hip"), "-o", out], check=True, stderr=subprocess.DEVNULL)
    text = open(out).read()
sym = "_ZN5igdev10k_traverseILb%dELb0ELb0ELb0ELb0ELb0EEEvNS_12TraverseArgsE" % (1 if which == "any" else 0)  # <ANY_HIT, no stats, not DEEP, no spheres, Node8 records, stream order>
m = re.search(r"^%s:.*?^\.Lfunc_end" % re.escape(sym), text, re.S | re.M)
body = m.group(0)
cur = "prologue"
order = []
cnt = collections.OrderedDict()
ops_by_part = collections.OrderedDict()
for line in body.split("\n"):
    mm = re.match(r"\s*; @@ (\S+)", line)
    if mm:
        cur = mm.group(1)
        continue
    mm = re.match(r"^\s+([a-z_0-9]+)(\s|$)", line)
    if mm and not line.strip().startswith((".", ";")):
        op = mm.group(1)
        ops_by_part.setdefault(cur, collections.Counter())[op] += 1
        c = cnt.setdefault(cur, collections.Counter())
        kind = "valu" if op.startswith("v_") else "salu" if op.startswith("s_") and not op.startswith(("s_cbranch", "s_branch", "s_waitcnt", "s_nop", "s_load", "s_endpgm")) else \
            "vmem" if op.startswith(("global_", "buffer_", "flat_")) else "lds" if op.startswith("ds_") else "other"
        c[kind] += 1
        if op.startswith("v_cndmask"):
            c["cndmask"] += 1
        if op.startswith("v_mov"):
            c["mov"] += 1
if as_json:
    import json
    print(json.dumps({"kernel": f"k_traverse<{which}>", "parts": {k: dict(v) for k, v in ops_by_part.items()}}))
    sys.exit(0)
print(f"# k_traverse<{which}>: static instructions by part (layout order)")
print(f"{'part':18s} {'valu':>5s} {'salu':>5s} {'vmem':>5s} {'lds':>4s} {'other':>5s} {'cndmask':>7s} {'mov':>4s}")
tot = collections.Counter()
for k, c in cnt.items():
    print(f"{k:18s} {c['valu']:5d} {c['salu']:5d} {c['vmem']:5d} {c['lds']:4d} {c['other']:5d} {c['cndmask']:7d} {c['mov']:4d}")
    tot.update(c)
print(f"{'total':18s} {tot['valu']:5d} {tot['salu']:5d} {tot['vmem']:5d} {tot['lds']:4d} {tot['other']:5d} {tot['cndmask']:7d} {tot['mov']:4d}")
