"""Tiny PNG/PPM writers (no external deps) for looking at renders."""
import struct
import zlib

import numpy as np


def write_png(path, rgba8_top_down):
    a = np.ascontiguousarray(rgba8_top_down, dtype=np.uint8)
    h, w, ch = a.shape
    assert ch in (3, 4)
    raw = b"".join(b"\x00" + a[y].tobytes() for y in range(h))

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)

    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 6 if ch == 4 else 2, 0, 0, 0)) +
                chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))


def linear_to_display(rgb_bottom_up, gamma=True):
    c = np.clip(rgb_bottom_up, 0, 1)
    if gamma:
        c = c ** (1 / 2.2)
    return (np.floor(c * 255 + 0.5)).astype(np.uint8)[::-1]
