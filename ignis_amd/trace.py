"""igtrace counterpart (src/frontend/trace/main.cpp): radiance along explicit rays.

    python -m ignis_amd.trace scene.json --input rays.txt [--spp 64] [--output out.txt] [--seed S]

Input: one ray per line, `ox oy oz dx dy dz [tmin [tmax]]` (main.cpp:17-60; a range with tmax <= tmin means unbounded).
Output: one `r\\tg\\tb` line per ray in scientific notation, the mean over spp iterations of one sample each
(the tracer fixes SPI = 1, main.cpp:76-78,63-67).
"""
import argparse
import sys

import numpy as np

from .runtime import Ray, RuntimeOptions, loadFromFile

FLT_MAX = 3.4028234664e+38


def read_rays(stream):
    rays = []
    for line in stream:
        line = line.strip()
        if not line:
            break
        try:
            data = [float(x) for x in line.split()]
        except ValueError:
            continue
        if len(data) < 6:
            continue
        tmin = data[6] if len(data) > 6 else 0.0
        tmax = data[7] if len(data) > 7 else 0.0
        if tmax <= tmin:
            tmax = FLT_MAX
        rays.append(Ray(data[0:3], data[3:6], tmin, tmax))
    return rays


def main(argv=None):
    ap = argparse.ArgumentParser(prog="ignis_amd.trace", description=__doc__.splitlines()[0])
    ap.add_argument("scene")
    ap.add_argument("-i", "--input", default=None, help="ray file (default: stdin)")
    ap.add_argument("-o", "--output", default=None, help="output file (default: stdout)")
    ap.add_argument("--spp", type=int, default=64)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--gpu", type=int, default=0)
    args = ap.parse_args(argv)

    rays = read_rays(open(args.input) if args.input else sys.stdin)
    if not rays:
        print("No rays given", file=sys.stderr)
        return 1
    opts = RuntimeOptions.makeDefault(trace=True)
    opts.SPI = 1
    opts.Seed = args.seed
    opts.Device = args.gpu
    with loadFromFile(args.scene, opts) as rt:
        data = None
        for _ in range(max(1, args.spp)):
            data = rt.trace(rays)
        spp = rt.SampleCount
    out = open(args.output, "w") if args.output else sys.stdout
    for r, g, b in np.asarray(data, dtype=np.float64) / spp:
        out.write(f"{r:e}\t{g:e}\t{b:e}\n")
    if args.output:
        out.close()
    return 0


if __name__ == "__main__":
    sys.exit(main())
