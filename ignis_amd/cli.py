"""igcli counterpart for the MI355X backend (src/frontend/cli/main.cpp:60-185): render a scene file for a number of
samples per pixel and write the EXR, printing the reference's statistics lines (ray counts, min/med/max Msamples/s).

    python -m ignis_amd.cli scenes/diamond_scene.json --spp 64 -o out.exr [--width W --height H --spi N --seed S --stats]
"""
import argparse
import math
import sys
import time

from .runtime import RuntimeOptions, loadFromFile


def beautiful_time(ms):
    s = ms / 1000.0
    return f"{s:.3f}s" if s < 60 else f"{int(s // 60)}m {s % 60:.1f}s"


def main(argv=None):
    ap = argparse.ArgumentParser(prog="ignis_amd.cli", description=__doc__.splitlines()[0])
    ap.add_argument("scene")
    ap.add_argument("-o", "--output", default="output.exr")
    ap.add_argument("--spp", type=int, default=None, help="samples per pixel (rounded up to a multiple of the spi)")
    ap.add_argument("--spi", type=int, default=0, help="samples per iteration, 0 = recommended")
    ap.add_argument("--time", type=float, default=None, help="render for this many seconds instead of a fixed spp")
    ap.add_argument("--width", type=int, default=0)
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--gpu", type=int, default=0)
    ap.add_argument("--batch", type=int, default=0, help="iterations rendered as one wavefront per call (0 = as many as fill the GPU, 1 = like the reference)")
    ap.add_argument("--stats", action="store_true", help="acquire ray statistics and stage timers")
    ap.add_argument("--full-stats", action="store_true", help="also count traversal work (slower kernels)")
    args = ap.parse_args(argv)
    if args.spp is None and args.time is None:
        args.spp = 64

    t_all = time.perf_counter()
    opts = RuntimeOptions.makeDefault()
    opts.Device = args.gpu
    opts.SPI = args.spi
    opts.Seed = args.seed
    opts.OverrideFilmSize = (args.width, args.height)
    opts.AcquireStats = True if args.full_stats else (1 if args.stats else False)
    t0 = time.perf_counter()
    rt = loadFromFile(args.scene, opts)
    t_loading = time.perf_counter() - t0

    spi = rt.SPI
    desired_iter = int(math.ceil((args.spp or 0) / spi))
    if args.spp and args.spp % spi:
        print(f"Given spp {args.spp} is not a multiple of the spi {spi}. Using spp {desired_iter * spi} instead", file=sys.stderr)
    print("Started rendering...", file=sys.stderr)
    samples_sec, t_render = [], 0.0
    batch = args.batch if args.batch > 0 else rt.recommendedBatch()
    while True:
        count = batch if desired_iter <= 0 else min(batch, desired_iter - rt.IterationCount)
        t0 = time.perf_counter()
        rt.stepMany(count)
        rt._device.synchronize()  # per-call timing like the reference's blocking step()
        dt = time.perf_counter() - t0
        t_render += dt
        samples_sec += [spi * rt.FramebufferWidth * rt.FramebufferHeight * count / dt] * count
        if desired_iter > 0 and rt.IterationCount >= desired_iter:
            break
        if args.time is not None and t_render > args.time:
            break

    t0 = time.perf_counter()
    ok = rt.saveFramebuffer(args.output)
    t_saving = time.perf_counter() - t0
    print(f"Result saved to {args.output}" if ok else f"Failed to save EXR file {args.output}", file=sys.stderr)
    ms_all = (time.perf_counter() - t_all) * 1e3
    if args.stats or args.full_stats:
        st = rt.getStatistics()
        total = st["camera_rays"] + st["bounce_rays"] + st["shadow_rays"]
        print("Statistics:\n"
              f"  Ray Count: {total}\n    Camera: {st['camera_rays']}\n    Bounce: {st['bounce_rays']}\n    Shadow: {st['shadow_rays']}\n"
              f"  Mrays/s (render time): {total / t_render / 1e6:.1f}")
        if args.full_stats:
            print(f"  Traversal: nodes {st['nodes']}, triangles {st['tris']}, entity leaves {st['leaves']}")
    print(f"  Iterations: {rt.IterationCount}\n  SPP: {rt.SampleCount}\n  SPI: {spi}\n  Time: {beautiful_time(ms_all)}\n"
          f"    Loading> {beautiful_time(t_loading * 1e3)}\n    Render>  {beautiful_time(t_render * 1e3)}\n    Saving>  {beautiful_time(t_saving * 1e3)}")
    rt.shutdown()
    s = sorted(samples_sec)
    print(f"# {s[0] * 1e-6:.3f}/{s[len(s) // 2] * 1e-6:.3f}/{s[-1] * 1e-6:.3f} (min/med/max Msamples/s)")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
