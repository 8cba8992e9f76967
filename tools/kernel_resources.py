"""Register / scratch / LDS budget of every kernel in the device library, from the code-object metadata
(llvm-readelf --notes of the gfx950 code object inside libig_device_hip.so).
usage: python tools/kernel_resources.py [library]  > profiles/rNN_kernel_resources.txt"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "ignis_amd", "lib", "libig_device_hip.so")
llvm = "/opt/rocm/lib/llvm/bin"
notes = ""
with tempfile.TemporaryDirectory() as tmp:
    # the .hip_fatbin section holds one offload bundle per translation unit
    subprocess.run([os.path.join(llvm, "llvm-objcopy"), "--dump-section", f".hip_fatbin={tmp}/fat.bin", lib], check=True)
    blob = open(f"{tmp}/fat.bin", "rb").read()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    starts = [m.start() for m in re.finditer(re.escape(magic), blob)]
    for n, st in enumerate(starts):
        part = blob[st:starts[n + 1] if n + 1 < len(starts) else len(blob)]
        open(f"{tmp}/b{n}.bin", "wb").write(part)
        co = f"{tmp}/dev{n}.co"
        subprocess.run([os.path.join(llvm, "clang-offload-bundler"), "--type=o", "--unbundle", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                        f"--input={tmp}/b{n}.bin", f"--output={co}"], check=True, capture_output=True)
        notes += subprocess.run([os.path.join(llvm, "llvm-readelf"), "--notes", co], check=True, capture_output=True, text=True).stdout
demangle = lambda n: subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip()
rows = []
cur = {}
for line in notes.split("\n"):
    m = re.match(r"\s*-?\s*\.(\w+):\s*(.*)", line)
    if not m:
        continue
    k, v = m.group(1), m.group(2).strip()
    if k == "wavefront_size":  # last key of a kernel's record
        rows.append(cur)
        cur = {}
    elif k in ("name", "vgpr_count", "vgpr_spill_count", "sgpr_count", "sgpr_spill_count", "private_segment_fixed_size", "group_segment_fixed_size", "agpr_count"):
        if k == "name" and not v.startswith("_Z") and not v.startswith("k_"):
            continue
        cur[k] = v
print(f"# {os.path.relpath(lib, ROOT)}: gfx950 code-object metadata (llvm-readelf --notes)")
print(f"{'kernel':78s} {'vgpr':>5s} {'spill':>6s} {'scratch_B':>9s} {'sgpr':>5s} {'lds_B':>7s} {'waves/SIMD':>10s}")
for r in sorted(rows, key=lambda r: r.get("name", "")):
    if "name" not in r or "rocprim" in r["name"] or "hipcub" in r["name"]:
        continue  # (the library kernels of the photon grid's sort and scan: not ours)
    n = demangle(r["name"]).replace("void igdev::", "").replace("igdev::", "")
    n = re.sub(r"\(.*", "", n)
    vg = int(r.get("vgpr_count", 0)) + int(r.get("agpr_count", 0))
    alloc = (vg + 7) // 8 * 8
    occ = min(8, 512 // alloc) if alloc else 8
    print(f"{n[:78]:78s} {vg:5d} {int(r.get('vgpr_spill_count', 0)):6d} {int(r.get('private_segment_fixed_size', 0)):9d} {int(r.get('sgpr_count', 0)):5d} {int(r.get('group_segment_fixed_size', 0)):7d} {occ:10d}")
