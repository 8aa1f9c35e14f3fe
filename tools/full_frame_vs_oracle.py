"""One WHOLE frame of a BASELINE configuration, every pixel: the HIP path (library defaults) against the CPU oracle, bit for bit, and
the reference ray counters.  The suite compares row samples at full size (the oracle needs about a second of 16 cores per million
samples); this tool does the whole frame once and writes gpurun_out/full_frame_vs_oracle_<config>_frame<F>.json.
    python tools/full_frame_vs_oracle.py headline C4 C3 C2 [C5] [--frame F] [--size W H]
The oracle is the checker here, as in tests/ -- nothing under path-tracing_amd/ uses it."""
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
    args = sys.argv[1:]
    frame = 0
    if "--frame" in args:
        k = args.index("--frame")
        frame = int(args[k + 1])
        del args[k:k + 2]
    size = None
    if "--size" in args:                                       # another image size than the configuration's (aspect < 1, edge tiles)
        k = args.index("--size")
        size = (int(args[k + 1]), int(args[k + 2]))
        del args[k:k + 3]
    modes = [""]
    if "--modes" in args:                                      # further option sets for the HIP side, e.g. "memo=0;pipeline=1;prune=1,anyhit=0"
        k = args.index("--modes")
        modes += [x for x in args[k + 1].split(";") if x]
        del args[k:k + 2]
    rc = 0
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    for cfg in args or ["headline"]:
        c = dict(m.CONFIGS[cfg])
        if size:
            c["width"], c["height"] = size
        s = m.load_scene(c["scene"])
        p = m.config_params(c["scene"], c["width"], c["height"], c["sample_count"], c["max_path_length"])
        ctx = m.Context(0)
        ctx.upload_scene(s)
        ctx.set_params(p)
        renders = {}
        defaults = dict(pipeline=2, traversal=3, prune=2, anyhit=1, memo=1, overlap=1, margin_lat=3000)
        for mode in modes:
            for k, v in defaults.items():
                ctx.set_option(k, v)
            for kv in filter(None, mode.split(",")):
                ctx.set_option(kv.split("=")[0], int(kv.split("=")[1]))
            ctx.reset_accum(); ctx.reset_counters()
            ctx.render_frames([1000 + frame], first_frame_number=frame)
            renders[mode or "defaults"] = (np.ascontiguousarray(ctx.read_sum()).copy(), ctx.counters())
            print(f"{cfg} frame {frame}: rendered with [{mode or 'defaults'}]", flush=True)
        got, gc = renders["defaults"]
        t0 = time.time()
        threads = min(16, orc.max_threads())
        want = np.zeros((p.height, p.width, 3), dtype=np.float32)
        oc = {}
        bands = 16                                             # a line of progress per band: long silent runs are taken for hung ones
        for b in range(bands):
            y0, y1 = p.height * b // bands, p.height * (b + 1) // bands
            _, cb = orc.render_frame(s, p, 1000 + frame, y0=y0, y1=y1, nthreads=threads, out=want)
            for k, v in cb.items():
                oc[k] = oc.get(k, 0) + v
            print(f"{cfg} frame {frame}: oracle rows {y0}..{y1 - 1} done, {time.time() - t0:.0f} s", flush=True)
        dt = time.time() - t0
        bad = int((got.view(np.uint32) != want.view(np.uint32)).any(axis=2).sum())
        same_counts = all(gc[k] == oc[k] for k in ("samples", "rays", "ray_hits", "light_attempts"))
        other = {}
        for mode, (img, cn) in renders.items():
            other[mode] = {"differing_pixels": int((img.view(np.uint32) != want.view(np.uint32)).any(axis=2).sum()),
                           "reference_counters_equal": all(cn[k] == oc[k] for k in ("samples", "rays", "ray_hits", "light_attempts"))}
            rc = rc or (1 if other[mode]["differing_pixels"] or not other[mode]["reference_counters_equal"] else 0)
        out = {"config": cfg, "option_sets": other, "scene": c["scene"], "width": c["width"], "height": c["height"], "bounces": c["max_path_length"], "frame": frame,
               "pixels": c["width"] * c["height"], "differing_pixels": bad, "reference_counters_equal": same_counts,
               "counters": {k: int(oc[k]) for k in ("samples", "rays", "ray_hits", "light_attempts")},
               "oracle_seconds": round(dt, 1), "oracle_threads": threads}
        print(json.dumps(out), flush=True)
        json.dump(out, open(os.path.join(ROOT, "gpurun_out", f"full_frame_vs_oracle_{cfg}_{c['width']}x{c['height']}_frame{frame}.json"), "w"), indent=1)
        rc = rc or (1 if bad or not same_counts else 0)
        ctx.close()
    return rc


if __name__ == "__main__":
    sys.exit(main())
