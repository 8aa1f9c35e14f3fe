"""north_star's first bar at full size: the primary-ray closest-hit triangle ids (and t bits) of every pixel of a configuration, HIP
against the CPU oracle's literal walk of the reference tree, under all four traversal options.
    python tools/full_size_primary_hits.py headline C5 ..."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401,E402
import numpy as np  # noqa: E402
import ptx_import  # noqa: E402
from oracle import orc  # noqa: E402


def main():
    m = ptx_import.load()
    rc = 0
    report = []
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    for cfg in sys.argv[1:] or ["headline"]:
        c = m.CONFIGS[cfg]
        s = m.load_scene(c["scene"])
        p = m.config_params(c["scene"], c["width"], c["height"], c["sample_count"], c["max_path_length"])
        ctx = m.Context(0)
        ctx.upload_scene(s)
        ctx.set_params(p)
        t0 = time.time()
        t_o, tri_o, _ = orc.primary_hits(s, p)
        dt = time.time() - t0
        row = {"config": cfg, "scene": c["scene"], "pixels": c["width"] * c["height"], "pixels_with_a_hit": int((tri_o >= 0).sum()), "oracle_seconds": round(dt, 1)}
        for traversal, prune in ((3, 2), (3, 1), (2, 2), (1, 2), (0, 2)):
            ctx.set_option("traversal", traversal)
            ctx.set_option("prune", prune)
            t, tri = ctx.primary_hits()
            bad = int(((tri != tri_o) | (t.view(np.uint32) != t_o.view(np.uint32))).sum())
            row[f"traversal={traversal},prune={prune}"] = bad
            rc = rc or (1 if bad else 0)
        print(json.dumps(row), flush=True)
        report.append(row)
        ctx.close()
    json.dump(report, open(os.path.join(ROOT, "gpurun_out", "full_size_primary_hits.json"), "w"), indent=1)
    return rc


if __name__ == "__main__":
    sys.exit(main())
