"""Scene container + render parameters (host-side records of the drop-in boundary, include/ptx.h).

Reference call sites these mirror: the eleven uploadTexture() calls of main.js:272-282 (scene arrays)
and the gl.uniform* block of main.js:290-308 (params).
"""
import json
import os
import struct
from dataclasses import dataclass, field

import numpy as np

_MAGIC = 0x53585450  # 'PTXS'
_ORDER = [("coords", "f", 9, "T"), ("normals", "f", 3, "T"), ("colors", "f", 3, "T"), ("emissions", "f", 3, "T"),
          ("light_tri", "i", 1, "L"), ("node_min", "f", 3, "N"), ("node_max", "f", 3, "N"), ("node_miss", "i", 1, "N"),
          ("node_first", "i", 1, "N"), ("node_count", "i", 1, "N"), ("inorder_tri", "i", 1, "T")]

GOLDEN_SCENES = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "scenes")


@dataclass
class Scene:
    T: int
    N: int
    L: int
    coords: np.ndarray
    normals: np.ndarray
    colors: np.ndarray
    emissions: np.ndarray
    light_tri: np.ndarray
    node_min: np.ndarray
    node_max: np.ndarray
    node_miss: np.ndarray
    node_first: np.ndarray
    node_count: np.ndarray
    inorder_tri: np.ndarray
    name: str = ""

    def validate(self):
        for name, kind, per, dim in _ORDER:
            a = getattr(self, name)
            want = per * getattr(self, dim)
            if a.size != want:
                raise ValueError(f"scene.{name}: {a.size} elements, expected {want}")
            if a.dtype != (np.float32 if kind == "f" else np.int32) or not a.flags.c_contiguous:
                raise ValueError(f"scene.{name}: must be contiguous {'float32' if kind == 'f' else 'int32'}")


def load_scene(path_or_name):
    """Reads a .ptxs file (host/scenefile.js layout). A bare name resolves inside tests/golden/scenes."""
    path = path_or_name
    if not os.path.exists(path):
        path = os.path.join(GOLDEN_SCENES, path_or_name + ".ptxs")
    with open(path, "rb") as f:
        raw = f.read()
    magic, version, T, N, L = struct.unpack_from("<5I", raw, 0)
    if magic != _MAGIC or version != 1:
        raise ValueError(f"{path}: not a PTXS v1 file")
    off = 32
    arrays = {}
    dims = {"T": T, "N": N, "L": L}
    for name, kind, per, dim in _ORDER:
        n = per * dims[dim]
        arrays[name] = np.frombuffer(raw, dtype=np.float32 if kind == "f" else np.int32, count=n, offset=off).copy()
        off += 4 * n
    if off != len(raw):
        raise ValueError(f"{path}: trailing bytes")
    s = Scene(T=T, N=N, L=L, name=os.path.splitext(os.path.basename(path))[0], **arrays)
    s.validate()
    return s


@dataclass
class Params:
    width: int
    height: int
    cam_src: np.ndarray
    cam_up: np.ndarray
    cam_right: np.ndarray
    cam_left_bottom: np.ndarray
    max_path_length: int = 8
    sample_count: int = 16

    def __post_init__(self):
        # host values are f64 (classes/camera.js); they are rounded to f32 once, at the boundary (main.js:292-296)
        for k in ("cam_src", "cam_up", "cam_right", "cam_left_bottom"):
            setattr(self, k, np.asarray(getattr(self, k), dtype=np.float64).astype(np.float32))


def load_camera(name):
    with open(os.path.join(GOLDEN_SCENES, name + ".json")) as f:
        return json.load(f)["camera"]


# BASELINE.json configs: scene fixture, W, H, frames, sample_count, bounces
CONFIGS = {
    "C1": dict(scene="cornell", width=256, height=256, frames=1, sample_count=16, max_path_length=4),
    "C2": dict(scene="my_cornell_6", width=1024, height=1024, frames=16, sample_count=16, max_path_length=8),
    "C3": dict(scene="teapot", width=1920, height=1080, frames=32, sample_count=16, max_path_length=8),
    "C4": dict(scene="bedroom2_v5", width=1920, height=1080, frames=64, sample_count=16, max_path_length=8),
    "C5": dict(scene="pixar_room", width=3840, height=2160, frames=256, sample_count=16, max_path_length=12),
    "headline": dict(scene="pixar_room", width=1920, height=1080, frames=64, sample_count=16, max_path_length=8),
}


def config_params(scene_name, width, height, sample_count=16, max_path_length=8):
    cam = load_camera(scene_name)
    return Params(width=width, height=height, cam_src=cam["cam_src"], cam_up=cam["cam_up"], cam_right=cam["cam_right"],
                  cam_left_bottom=cam["cam_left_bottom"], max_path_length=max_path_length, sample_count=sample_count)


def timestamps(n_frames, first=0):
    """Fixed timestamp list of SURVEY.md 8d: frame f uses 1000 + f."""
    return np.arange(first, first + n_frames, dtype=np.int32) + 1000
