#!/usr/bin/env python3
"""Seeded procedural stand-in for the reference's Bedroom / Living-room configs (SURVEY.md 8d, configs 3 and 5;
the assets are not in the reference tree). States what it is: a synthetic scene, NOT those assets.

    python tools/make_standin_scene.py OUT_DIR [--triangles 1000000] [--seed 7] [--instances 96] [--materials divergent|lean]

Writes OUT_DIR/standin.json + OUT_DIR/meshes/*.ply (binary little-endian PLY, the loader's own format)
+ OUT_DIR/textures/*.png:
  * one heightfield terrain (about 60 % of the unique triangles) inside a closed room,
  * 8 unique noise-displaced icosphere "rocks" (the rest), instanced `--instances` times with random
    rotation / scale / translation,
  * 32 materials. `--materials divergent` (the default, what BASELINE config 3 is for: "divergent BSDF/shading stress")
    cycles {principled, rough plastic, rough dielectric, blend of two named BSDFs, bump-mapped diffuse with bitmap
    reflectance + bitmap height map, rough conductor, diffuse (every other one a checkerboard), smooth dielectric}:
    every material class of the shading kernels (basic / principled / coated / blend) and bitmap lookups take part.
    `--materials lean` is the round 2 - 4 mix {diffuse, rough conductor, smooth dielectric, checkerboard diffuse}, which
    the lean shading kernel covers: the HBM-regime traversal workload of tools/run_standin.sh. The geometry (terrain,
    rocks, instance transforms, instance -> material slot) is the same bytes in both modes,
  * 4 rectangular area lights under the ceiling.
Everything derives from numpy's PCG64 seeded with --seed, so the same arguments give the same bytes.
"""
import argparse
import json
import os
import struct
import zlib

import numpy as np


def write_png(path, img):
    """8-bit PNG (gray for HxW, RGB for HxWx3), filter type 0 rows, one IDAT."""
    img = np.ascontiguousarray(img, dtype=np.uint8)
    h, w = img.shape[:2]
    ctype = 0 if img.ndim == 2 else 2
    raw = np.concatenate([np.zeros((h, 1), np.uint8), img.reshape(h, -1)], axis=1).tobytes()

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, ctype, 0, 0, 0))
                + chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))


def write_ply(path, verts, faces):
    verts = np.ascontiguousarray(verts, dtype="<f4")
    faces = np.ascontiguousarray(faces, dtype="<i4")
    with open(path, "wb") as f:
        f.write((f"ply\nformat binary_little_endian 1.0\nelement vertex {len(verts)}\n"
                 "property float x\nproperty float y\nproperty float z\n"
                 f"element face {len(faces)}\nproperty list uchar int vertex_indices\nend_header\n").encode())
        f.write(verts.tobytes())
        rec = np.zeros(len(faces), dtype=[("n", "u1"), ("i", "<i4", 3)])
        rec["n"] = 3
        rec["i"] = faces
        f.write(rec.tobytes())


def value_noise(rng, n, octaves=5):
    """Smooth 2-D noise on an (n+1) x (n+1) grid: sum of bilinearly upsampled random lattices."""
    out = np.zeros((n + 1, n + 1))
    amp = 1.0
    for o in range(octaves):
        cells = 4 << o
        lat = rng.random((cells + 1, cells + 1))
        x = np.linspace(0, cells, n + 1)
        i = np.minimum(x.astype(int), cells - 1)
        t = x - i
        t = t * t * (3 - 2 * t)
        rows = lat[i, :] * (1 - t)[:, None] + lat[i + 1, :] * t[:, None]
        out += amp * (rows[:, i] * (1 - t)[None, :] + rows[:, i + 1] * t[None, :])
        amp *= 0.5
    return out / 2.0


def terrain(rng, n, size, height):
    h = value_noise(rng, n) * height
    xs = np.linspace(-size, size, n + 1)
    gx, gz = np.meshgrid(xs, xs, indexing="ij")
    verts = np.stack([gx, h, gz], axis=-1).reshape(-1, 3)
    idx = np.arange((n + 1) * (n + 1)).reshape(n + 1, n + 1)
    a, b, c, d = idx[:-1, :-1].ravel(), idx[1:, :-1].ravel(), idx[:-1, 1:].ravel(), idx[1:, 1:].ravel()
    faces = np.concatenate([np.stack([a, c, b], 1), np.stack([b, c, d], 1)])
    return verts, faces


def icosphere(level):
    t = (1 + 5 ** 0.5) / 2
    v = np.array([[-1, t, 0], [1, t, 0], [-1, -t, 0], [1, -t, 0], [0, -1, t], [0, 1, t], [0, -1, -t], [0, 1, -t],
                  [t, 0, -1], [t, 0, 1], [-t, 0, -1], [-t, 0, 1]], dtype=np.float64)
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    f = np.array([[0, 11, 5], [0, 5, 1], [0, 1, 7], [0, 7, 10], [0, 10, 11], [1, 5, 9], [5, 11, 4], [11, 10, 2],
                  [10, 7, 6], [7, 1, 8], [3, 9, 4], [3, 4, 2], [3, 2, 6], [3, 6, 8], [3, 8, 9], [4, 9, 5],
                  [2, 4, 11], [6, 2, 10], [8, 6, 7], [9, 8, 1]])
    for _ in range(level):
        edges = np.sort(np.concatenate([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]]), axis=1)
        uniq, inv = np.unique(edges, axis=0, return_inverse=True)
        mid = v[uniq[:, 0]] + v[uniq[:, 1]]
        mid /= np.linalg.norm(mid, axis=1, keepdims=True)
        m = len(v) + inv.reshape(3, -1)  # midpoint ids of edges (01), (12), (20) per face
        v = np.concatenate([v, mid])
        a, b, c = f[:, 0], f[:, 1], f[:, 2]
        f = np.concatenate([np.stack([a, m[0], m[2]], 1), np.stack([b, m[1], m[0]], 1),
                            np.stack([c, m[2], m[1]], 1), np.stack([m[0], m[1], m[2]], 1)])
    return v, f


def rock(rng, level):
    v, f = icosphere(level)
    # low-frequency radial displacement from a few random plane waves
    disp = np.zeros(len(v))
    for _ in range(6):
        d = rng.normal(size=3)
        disp += rng.uniform(0.03, 0.12) * np.sin(v @ d * rng.uniform(1.5, 5.0) + rng.uniform(0, 6.28))
    return v * (1 + disp)[:, None], f


def rot(rng):
    q = rng.normal(size=4)
    q /= np.linalg.norm(q)
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("out")
    ap.add_argument("--triangles", type=int, default=1_000_000, help="unique triangles (approximate lower bound)")
    ap.add_argument("--seed", type=int, default=7)
    ap.add_argument("--instances", type=int, default=96)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--max-depth", type=int, default=16)
    ap.add_argument("--materials", choices=("divergent", "lean"), default="divergent")
    args = ap.parse_args()
    rng = np.random.default_rng(args.seed)
    os.makedirs(os.path.join(args.out, "meshes"), exist_ok=True)

    # split the triangle budget: rocks = 8 x 20 * 4^level, terrain = the rest
    level = 2
    while 8 * 20 * 4 ** (level + 1) <= 0.4 * args.triangles and level < 7:
        level += 1
    rock_tris = 8 * 20 * 4 ** level
    n = int(np.ceil(np.sqrt(max(args.triangles - rock_tris, 2) / 2)))
    size, room_h = 10.0, 8.0

    shapes, entities, bsdfs, lights = [], [], [], []
    tv, tf = terrain(rng, n, size, 1.2)
    write_ply(os.path.join(args.out, "meshes", "terrain.ply"), tv, tf)
    shapes.append({"type": "external", "name": "terrain", "filename": "meshes/terrain.ply"})
    for r in range(8):
        rv, rf = rock(rng, level)
        write_ply(os.path.join(args.out, "meshes", f"rock{r}.ply"), rv, rf)
        shapes.append({"type": "external", "name": f"rock{r}", "filename": f"meshes/rock{r}.ply"})

    # (both modes take the same draws from `rng`, so what follows — the instance transforms and material slots — does not depend on the mode)
    lean_bsdfs = []
    for m in range(32):
        kind = m % 4
        col = [round(float(x), 4) for x in rng.uniform(0.2, 0.9, 3)]
        if kind == 0:
            lean_bsdfs.append({"type": "diffuse", "name": f"mat{m}", "reflectance": col})
        elif kind == 1:
            lean_bsdfs.append({"type": "conductor", "name": f"mat{m}", "roughness": round(float(rng.uniform(0.05, 0.5)), 4),
                               "eta": [0.2, 0.92, 1.1], "k": [3.9, 2.45, 2.14], "specular_reflectance": col})
        elif kind == 2:
            lean_bsdfs.append({"type": "dielectric", "name": f"mat{m}", "int_ior": round(float(rng.uniform(1.3, 2.0)), 4)})
        else:
            lean_bsdfs.append({"type": "diffuse", "name": f"mat{m}", "reflectance": f"check{m}"})
    textures = [{"type": "checkerboard", "name": f"check{m}", "scale_x": 8, "scale_y": 8,
                 "color0": [0.8, 0.8, 0.8], "color1": [round(float(x), 4) for x in rng.uniform(0.05, 0.4, 3)]}
                for m in range(3, 32, 4)]
    if args.materials == "lean":
        bsdfs += lean_bsdfs
    else:
        mr = np.random.default_rng([args.seed, 0x6d617473])  # the materials' own stream
        os.makedirs(os.path.join(args.out, "textures"), exist_ok=True)
        for t in range(4):
            # a colour map and a height map per bitmap material: smooth noise, 256 x 256
            hm = value_noise(mr, 255, octaves=4 + t % 2)
            hm = (hm - hm.min()) / max(hm.max() - hm.min(), 1e-9)
            write_png(os.path.join(args.out, "textures", f"height{t}.png"), np.round(hm * 255))
            base = mr.uniform(0.25, 0.95, 3)
            alt = mr.uniform(0.05, 0.6, 3)
            cm = hm[..., None] * base + (1 - hm[..., None]) * alt
            write_png(os.path.join(args.out, "textures", f"color{t}.png"), np.round(cm * 255))
            textures.append({"type": "image", "name": f"height{t}", "filename": f"textures/height{t}.png", "filter_type": "bilinear", "linear": True})
            textures.append({"type": "image", "name": f"color{t}", "filename": f"textures/color{t}.png", "filter_type": "bilinear"})

        def u(lo, hi):
            return round(float(mr.uniform(lo, hi)), 4)

        def colour(lo=0.2, hi=0.9):
            return [round(float(x), 4) for x in mr.uniform(lo, hi, 3)]
        for m in range(32):
            kind, rep = m % 8, m // 8
            name = f"mat{m}"
            if kind == 0:
                b = {"type": "principled", "name": name, "base_color": colour(), "metallic": u(0, 1) if rep % 2 else 0.0, "roughness": u(0.05, 0.8),
                     "specular_tint": u(0, 1), "sheen": u(0, 1) if rep == 1 else 0.0, "sheen_tint": u(0, 1),
                     "clearcoat": u(0.2, 1) if rep >= 2 else 0.0, "clearcoat_gloss": u(0, 1), "ior": u(1.3, 1.8),
                     "specular_transmission": u(0.3, 0.9) if rep == 3 else 0.0}
                bsdfs.append(b)
            elif kind == 1:
                bsdfs.append({"type": "roughplastic" if rep % 2 == 0 else "plastic", "name": name, "diffuse_reflectance": colour(), "int_ior": u(1.3, 1.7),
                              **({"roughness": u(0.03, 0.4)} if rep % 2 == 0 else {})})
            elif kind == 2:
                bsdfs.append({"type": "roughdielectric", "name": name, "int_ior": u(1.3, 2.0), "roughness": u(0.02, 0.3)})
            elif kind == 3:
                # a blend needs two named inner BSDFs (BlendBSDF.cpp:14-56); neither is a blend or a bump map
                bsdfs.append({"type": "diffuse", "name": f"{name}-a", "reflectance": colour()})
                second = ({"type": "conductor", "name": f"{name}-b", "roughness": u(0.05, 0.4), "eta": [0.2, 0.92, 1.1], "k": [3.9, 2.45, 2.14]} if rep % 2 == 0 else
                          {"type": "principled", "name": f"{name}-b", "base_color": colour(), "metallic": u(0.3, 1), "roughness": u(0.1, 0.6)})
                bsdfs.append(second)
                bsdfs.append({"type": "blend", "name": name, "first": f"{name}-a", "second": f"{name}-b", "weight": u(0.2, 0.8)})
            elif kind == 4:
                bsdfs.append({"type": "diffuse", "name": f"{name}-inner", "reflectance": f"color{rep}"})
                bsdfs.append({"type": "bumpmap", "name": name, "bsdf": f"{name}-inner", "map": f"height{rep}", "strength": u(0.5, 2.0)})
            elif kind == 5:
                bsdfs.append({"type": "conductor", "name": name, "roughness": u(0.05, 0.5), "eta": [0.2, 0.92, 1.1], "k": [3.9, 2.45, 2.14],
                              "specular_reflectance": colour()})
            elif kind == 6:
                bsdfs.append({"type": "diffuse", "name": name, "reflectance": f"check{4 * rep + 3}" if rep % 2 else colour()})
            else:
                bsdfs.append({"type": "dielectric", "name": name, "int_ior": u(1.3, 2.0)})
    bsdfs.append({"type": "diffuse", "name": "mat-room", "reflectance": [0.7, 0.7, 0.7]})
    bsdfs.append({"type": "diffuse", "name": "mat-light", "reflectance": [0, 0, 0]})

    # the terrain takes half of the camera hits: the lean mix keeps it plain diffuse, the divergent one gives it the bump-mapped bitmap material
    entities.append({"name": "terrain", "shape": "terrain", "bsdf": "mat0" if args.materials == "lean" else "mat4"})
    for i in range(args.instances):
        s = float(rng.uniform(0.25, 0.9))
        R = rot(rng) * s
        p = [float(rng.uniform(-size * 0.9, size * 0.9)), float(rng.uniform(0.8, room_h * 0.6)), float(rng.uniform(-size * 0.9, size * 0.9))]
        M = [float(R[0, 0]), float(R[0, 1]), float(R[0, 2]), p[0], float(R[1, 0]), float(R[1, 1]), float(R[1, 2]), p[1],
             float(R[2, 0]), float(R[2, 1]), float(R[2, 2]), p[2], 0, 0, 0, 1]
        entities.append({"name": f"rock{i}", "shape": f"rock{i % 8}", "bsdf": f"mat{int(rng.integers(0, 32))}", "transform": M})

    # room: floor is the terrain; 4 walls + ceiling as rectangles (2 x 2 in the xy plane, normal +z)
    def wall(name, M):
        shapes.append({"type": "rectangle", "name": name, "transform": M})
        entities.append({"name": name, "shape": name, "bsdf": "mat-room"})
    S, Hh = size, room_h / 2
    wall("wall-back", [S, 0, 0, 0, 0, Hh, 0, Hh, 0, 0, 1, -S, 0, 0, 0, 1])
    wall("wall-front", [-S, 0, 0, 0, 0, Hh, 0, Hh, 0, 0, -1, S, 0, 0, 0, 1])
    wall("wall-left", [0, 0, 1, -S, 0, Hh, 0, Hh, -S, 0, 0, 0, 0, 0, 0, 1])
    wall("wall-right", [0, 0, -1, S, 0, Hh, 0, Hh, S, 0, 0, 0, 0, 0, 0, 1])
    wall("ceiling", [S, 0, 0, 0, 0, 0, -1, room_h, 0, S, 0, 0, 0, 0, 0, 1])
    for l in range(4):
        cx, cz = (-1) ** l * S * 0.45, (-1) ** (l // 2) * S * 0.45
        name = f"light{l}"
        # 1.5 x 1.5 rectangle just under the ceiling, facing down
        shapes.append({"type": "rectangle", "name": name, "transform": [0.75, 0, 0, cx, 0, 0, -1, room_h - 0.05, 0, 0.75, 0, cz, 0, 0, 0, 1]})
        entities.append({"name": name, "shape": name, "bsdf": "mat-light"})
        lights.append({"type": "area", "name": name, "entity": name, "radiance": [40, 38, 34]})

    scene = {
        "technique": {"type": "path", "max_depth": args.max_depth},
        "camera": {"type": "perspective", "fov": 60, "near_clip": 0.05, "far_clip": 200,
                   "transform": [{"lookat": {"origin": [S * 0.85, room_h * 0.55, S * 0.85], "target": [0, 1.5, 0], "up": [0, 1, 0]}}]},
        "film": {"size": [args.width, args.height]},
        "textures": textures, "bsdfs": bsdfs, "shapes": shapes, "entities": entities, "lights": lights,
    }
    with open(os.path.join(args.out, "standin.json"), "w") as f:
        json.dump(scene, f, indent=1)
    unique = len(tf) + rock_tris
    inst = len(tf) + (rock_tris // 8) * args.instances
    print(json.dumps({"scene": os.path.join(args.out, "standin.json"), "unique_triangles": int(unique),
                      "instanced_triangles": int(inst), "entities": len(entities), "materials": len(bsdfs), "material_mix": args.materials, "seed": args.seed}))


if __name__ == "__main__":
    main()
