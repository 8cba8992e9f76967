"""Writes ignis_amd/data/hosek_rgb.f64: the RGB coefficient tables of the Hosek-Wilkie sky model ("An Analytic Model for Full
Spectral Sky-Dome Radiance", SIGGRAPH 2012, sample implementation 1.4a), read from the copy the reference ships
(src/runtime/skysun/model/ArHosekSkyModelData_RGB.h). Constant published data, stored as raw little-endian doubles:
per channel (R, G, B) 1080 configuration values (2 albedos x 10 turbidities x 6 Bezier control points x 9 coefficients) followed by
120 radiance values (2 x 10 x 6). The model itself is restated in ignis_amd/csrc/host/hosek.h.
usage (in the build container, where /root/reference exists): python tools/make_hosek_tables.py"""
import os
import re
import struct
import sys

src = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/src/runtime/skysun/model/ArHosekSkyModelData_RGB.h"
text = open(src).read()
out = b""
for ch in (1, 2, 3):
    for name, n in ((f"datasetRGB{ch}", 1080), (f"datasetRGBRad{ch}", 120)):
        m = re.search(r"double\s+" + name + r"\[\]\s*=\s*\{(.*?)\};", text, re.S)
        body = re.sub(r"//[^\n]*", "", m.group(1))
        vals = [float(v) for v in re.findall(r"[-+]?\d*\.?\d+(?:[eE][-+]?\d+)?", body)]
        assert len(vals) == n, (name, len(vals))
        out += struct.pack(f"<{n}d", *vals)
dst = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ignis_amd", "data", "hosek_rgb.f64")
open(dst, "wb").write(out)
print(dst, len(out), "bytes")
