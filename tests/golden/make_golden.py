"""Generates the committed fixtures under tests/golden/ from the CPU oracle (run in the authoring
container: `python tests/golden/make_golden.py`). The reference itself cannot be built or imported
here, so these vectors pin the ORACLE's behaviour over time and give the GPU tests
inputs/outputs that do not depend on liboracle.so being rebuilt identically."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import oracle  # noqa: E402
from ignis_amd.tables import LoadedScene  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def main():
    seeds = np.array([0, 1, 0xDEADBEEF, 0x811C9DC5], dtype=np.uint32)
    raw = np.stack([oracle.random_sequence(int(s), 1, 16)[1] for s in seeds])
    np.savez(os.path.join(OUT, "rng_tea.npz"), seeds=seeds, raw=raw)

    sc = LoadedScene.from_file(os.path.join(ROOT, "scenes", "diamond_scene.json"), 128, 128)
    rays, _ = oracle.generate_rays(sc, 1, 128, 128, 0, 4096, seed=1)
    hit = oracle.trace(sc, rays, flags=1)
    np.savez_compressed(os.path.join(OUT, "diamond_hits_4096.npz"), rays=rays, ent_id=hit["ent_id"], prim_id=hit["prim_id"],
                        t=hit["t"], u=hit["u"], v=hit["v"])

    fb, st = oracle.render(sc, 4, 64, 64, iteration=0, seed=1)
    np.savez_compressed(os.path.join(OUT, "diamond_radiance_64x64_spi4.npz"), fb=fb,
                        stats=np.array([st[k] for k in ("camera_rays", "bounce_rays", "shadow_rays", "nodes", "tris", "leaves", "unoccluded")], dtype=np.uint64))
    print("fixtures written to", OUT)


if __name__ == "__main__":
    main()
