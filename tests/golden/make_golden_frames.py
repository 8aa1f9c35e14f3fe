"""Writes tests/golden/frames.npz: small golden outputs of the CPU oracle (frame sums + primary-hit ids) that the
GPU tests compare against bit for bit.  Inputs are the committed scene fixtures; run from the repo root:
    python tests/golden/make_golden_frames.py
The oracle itself is pinned by the reference-derived vectors listed in DESIGN.md ("Oracle pin")."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import ptx_import  # noqa: E402
from oracle import orc  # noqa: E402

m = ptx_import.load()
out = {}
for name, w, h, spp, bounces, nfr in [("cornell", 32, 32, 4, 4, 2), ("my_cornell_6", 48, 32, 4, 8, 2), ("bedroom2_v5", 48, 27, 2, 8, 2),
                                      ("teapot", 48, 27, 2, 8, 1), ("pixar_room", 64, 36, 4, 8, 2)]:
    s = m.load_scene(name)
    p = m.config_params(name, w, h, spp, bounces)
    acc, _ = orc.render_frames(s, p, m.scene.timestamps(nfr))
    _, tri, _ = orc.primary_hits(s, p)
    key = f"{name}-{w}-{h}-{spp}-{bounces}-{nfr}"
    out[key + "_sum"] = acc
    out[key + "_tri"] = tri
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "frames.npz"), **out)
print("wrote", len(out), "arrays")
