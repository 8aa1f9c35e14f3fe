"""The 8-bit side of the path at full size: the reference's progressive RGBA8 blend (option compat_rgba8, both unorm8 store rules)
and the display pass, HIP against the oracle's byte arithmetic, over a few frames of a configuration.  The float frames that go in
are the HIP path's own (they equal the oracle's bit for bit: tools/full_frame_vs_oracle.py), so no CPU rendering is needed.
    python tools/full_size_display_check.py headline C5 [--frames N]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401,E402
import numpy as np  # noqa: E402
import ptx_import  # noqa: E402
from oracle import orc  # noqa: E402


def main():
    m = ptx_import.load()
    args = sys.argv[1:]
    n = 4
    if "--frames" in args:
        k = args.index("--frames")
        n = int(args[k + 1])
        del args[k:k + 2]
    rc = 0
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    report = []
    for cfg in args or ["headline"]:
        c = m.CONFIGS[cfg]
        s = m.load_scene(c["scene"])
        p = m.config_params(c["scene"], c["width"], c["height"], c["sample_count"], c["max_path_length"])
        ctx = m.Context(0)
        ctx.upload_scene(s)
        ctx.set_params(p)
        frames = []
        for f in range(n):                                   # the float frames, one by one
            ctx.reset_accum()
            ctx.render_frames([1000 + f], first_frame_number=f)
            frames.append(np.ascontiguousarray(ctx.read_sum()).copy())
        for rules in (1, 2):                                 # compat_rgba8: 1 = round(c * 255), 2 = SwiftShader's store rule
            ctx.set_option("compat_rgba8", rules)
            ctx.reset_accum()
            ctx.render_frames([1000 + f for f in range(n)], first_frame_number=0)
            prev = None
            for f in range(n):
                prev = orc.blend_rgba8(p, frames[f], prev, f, store=rules - 1)
            row = {"config": cfg, "frames": n, "compat_rgba8": rules}
            for gamma in (False, True):
                for flip in (False, True):
                    got = ctx.read_compat_rgba8(gamma, flip)
                    want = orc.display_rgba8(prev, store=rules - 1) if gamma else prev
                    want = want[::-1] if flip else want
                    bad = int((got != want).any(axis=2).sum())
                    row[f"gamma={int(gamma)},flip={int(flip)}"] = bad
                    rc = rc or (1 if bad else 0)
            print(json.dumps(row), flush=True)
            report.append(row)
        ctx.set_option("compat_rgba8", 0)
        ctx.close()
    json.dump(report, open(os.path.join(ROOT, "gpurun_out", "full_size_display_check.json"), "w"), indent=1)
    return rc


if __name__ == "__main__":
    sys.exit(main())
