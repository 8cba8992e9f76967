"""Writes scenes/textures/sky_gradient.png: a small synthetic latitude-longitude environment map (horizon-to-zenith gradient,
one bright sun disc, a dark ground half) for scenes/diamond_scene_principled.json. Stand-in data, not a captured panorama."""
import os
import struct
import sys
import zlib

import numpy as np


def write_png_rgb(path, img):
    h, w, _ = img.shape
    raw = b"".join(b"\x00" + img[y].astype(np.uint8).tobytes() for y in range(h))

    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xFFFFFFFF)

    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0))
                + chunk(b"IDAT", zlib.compress(raw, 9)) + chunk(b"IEND", b""))


def main(out):
    w, h = 128, 64
    v = (np.arange(h) + 0.5) / h                  # 0 = top row = zenith
    u = (np.arange(w) + 0.5) / w
    up = np.clip(1 - 2 * v, 0, 1)[:, None]        # 1 at the zenith, 0 at the horizon and below
    sky = np.stack([40 + 60 * (1 - up), 70 + 70 * (1 - up), 140 + 60 * (1 - up)], -1) * np.ones((h, w, 1))
    ground = np.array([35, 30, 25]) * np.ones((h, w, 3))
    img = np.where((v < 0.5)[:, None, None], sky, ground)
    d2 = ((u[None, :] - 0.3) * 2) ** 2 + (v[:, None] - 0.22) ** 2
    img = np.where((d2 < 0.0012)[..., None], np.array([255, 250, 235]), img)
    write_png_rgb(out, np.clip(img, 0, 255))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(__file__), "..", "scenes", "textures", "sky_gradient.png"))
