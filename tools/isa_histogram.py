"""Static opcode histogram of the kernels of one device translation unit, from the gfx950 assembly hipcc emits.
usage: python tools/isa_histogram.py [traverse|shade|tail|photon] [-D...]  [--kernel SUBSTR] [--top N] [--blocks]
Prints per kernel: instruction totals by class (VALU / SALU / VMEM / LDS / branch), the opcodes the verdicts track
(v_cndmask, v_mov, v_div_*), and the N most frequent opcodes. `--blocks` adds the count per basic block label."""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLAGS = "--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -fPIC -Wall -Wno-unused-parameter".split()
SCHED = {"traverse": "max-memory-clause", "shade": "max-ilp", "photon": "max-ilp", "tail": "max-ilp"}


def assemble(unit, extra):
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, unit + ".s")
        cmd = ["/opt/rocm/bin/hipcc", *FLAGS, "-I" + os.path.join(ROOT, "include"), "--cuda-device-only", "-S", *extra]
        if unit in SCHED:
            cmd += ["-mllvm", "-amdgpu-sched-strategy=" + SCHED[unit]]
        cmd += [os.path.join(ROOT, "ignis_amd", "csrc", "device", unit + ".hip"), "-o", out]
        subprocess.run(cmd, check=True)
        return open(out).read()


def classify(op):
    if op.startswith(("v_", )):
        return "VALU"
    if op.startswith(("s_cbranch", "s_branch", "s_setpc", "s_swappc", "s_endpgm")):
        return "branch"
    if op.startswith(("s_waitcnt", "s_nop", "s_sleep", "s_barrier")):
        return "wait"
    if op.startswith(("s_load", "s_buffer_load", "s_store")):
        return "SMEM"
    if op.startswith("s_"):
        return "SALU"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "VMEM"
    if op.startswith("ds_"):
        return "LDS"
    return "other"


def main():
    args = sys.argv[1:]
    unit = "traverse"
    extra, kernel_filter, top, blocks = [], None, 16, False
    i = 0
    while i < len(args):
        a = args[i]
        if a == "--kernel":
            kernel_filter = args[i + 1]
            i += 1
        elif a == "--top":
            top = int(args[i + 1])
            i += 1
        elif a == "--blocks":
            blocks = True
        elif a.startswith("-"):
            extra.append(a)
        else:
            unit = a
        i += 1
    text = assemble(unit, extra)
    demangle = lambda n: subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip()
    cur = None
    kernels = collections.OrderedDict()
    block = None
    for line in text.split("\n"):
        m = re.match(r"^(_Z\w+|k_\w+):\s*(;.*)?$", line)
        if m and "$" not in m.group(1):
            cur = m.group(1)
            kernels[cur] = {"ops": collections.Counter(), "blocks": collections.OrderedDict(), "meta": {}}
            block = "entry"
            continue
        if cur is None:
            continue
        if line.startswith(".Lfunc_end"):
            cur = None
            continue
        m = re.match(r"^(\.LBB\w+):", line)
        if m:
            block = m.group(1)
            continue
        m = re.match(r"^\s+([a-z_0-9]+)(\s|$)", line)
        if m and not line.strip().startswith((".", ";")):
            op = m.group(1)
            kernels[cur]["ops"][op] += 1
            kernels[cur]["blocks"].setdefault(block, collections.Counter())[op] += 1
    # resource lines follow each kernel as comments: "; NumVgprs: 116" etc.
    for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", text, re.S):
        name, body = m.group(1), m.group(2)
        if name in kernels:
            for key in ("next_free_vgpr", "private_segment_fixed_size", "group_segment_fixed_size"):
                mm = re.search(r"\.amdhsa_" + key + r" (\d+)", body)
                if mm:
                    kernels[name]["meta"][key] = int(mm.group(1))
    for name, k in kernels.items():
        dn = re.sub(r"\(.*", "", demangle(name).replace("void igdev::", "").replace("igdev::", ""))
        if kernel_filter and kernel_filter not in dn:
            continue
        ops = k["ops"]
        cls = collections.Counter()
        for op, n in ops.items():
            cls[classify(op)] += n
        total = sum(ops.values())
        print(f"== {dn}   {k['meta']}")
        print("   total %d : " % total + "  ".join(f"{c} {n}" for c, n in cls.most_common()))
        track = {"v_cndmask_b32": 0, "v_mov_b32": 0, "v_div_scale_f32": 0, "v_div_fmas_f32": 0, "v_div_fixup_f32": 0, "v_rcp_f32": 0, "v_readfirstlane_b32": 0,
                 "v_accvgpr_write_b32": 0, "v_accvgpr_read_b32": 0}
        for op in list(track):
            track[op] = sum(n for o, n in ops.items() if o.startswith(op.rsplit("_", 1)[0]) and o.split("_e")[0] == op) or ops.get(op, 0)
        print("   tracked : " + "  ".join(f"{o} {n}" for o, n in track.items()))
        print("   top     : " + "  ".join(f"{o} {n}" for o, n in ops.most_common(top)))
        if blocks:
            for b, c in k["blocks"].items():
                n = sum(c.values())
                if n >= 8:
                    print(f"     {b:14s} {n:5d}  valu {sum(v for o, v in c.items() if o.startswith('v_')):4d}  cndmask {sum(v for o, v in c.items() if o.startswith('v_cndmask')):3d}  mov {sum(v for o, v in c.items() if o.startswith('v_mov_b32')):3d}  vmem {sum(v for o, v in c.items() if classify(o) == 'VMEM'):3d}  lds {sum(v for o, v in c.items() if classify(o) == 'LDS'):3d}")


if __name__ == "__main__":
    main()
