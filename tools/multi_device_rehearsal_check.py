"""The one-process multi-device context at full size, rehearsed on ONE GPU (the same device listed G times: every code path of
ptx_create(ids, G) except the RCCL call, which k_sum_accums stands in for): a whole BASELINE job split by frame over G sub-contexts
must equal its definition -- device g sums frames g, g + G, ... in order, the devices' sums are added in device order -- bit for bit,
with the per-frame images taken from a single-device context.
    python tools/multi_device_rehearsal_check.py headline [--devices 2] [--frames N]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401,E402
import numpy as np  # noqa: E402
import ptx_import  # noqa: E402


def main():
    m = ptx_import.load()
    args = sys.argv[1:]
    G, limit = 2, None
    for flag in ("--devices", "--frames"):
        if flag in args:
            k = args.index(flag)
            if flag == "--devices":
                G = int(args[k + 1])
            else:
                limit = int(args[k + 1])
            del args[k:k + 2]
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    rc, report = 0, []
    for cfg in args or ["headline"]:
        c = m.CONFIGS[cfg]
        n = c["frames"] if limit is None else limit
        s = m.load_scene(c["scene"])
        p = m.config_params(c["scene"], c["width"], c["height"], c["sample_count"], c["max_path_length"])
        ts = [1000 + f for f in range(n)]
        single = m.Context(0)
        single.upload_scene(s); single.set_params(p)
        parts = [None] * G
        for f in range(n):
            single.reset_accum()
            single.render_frames([ts[f]], first_frame_number=f)
            img = np.ascontiguousarray(single.read_sum())
            g = f % G
            parts[g] = img.copy() if parts[g] is None else parts[g] + img
            if f % 16 == 15:
                print(f"{cfg}: {f + 1}/{n} single frames", flush=True)
        want = parts[0]
        for g in range(1, G):
            if parts[g] is not None:
                want = want + parts[g]
        single.close()
        multi = m.Context([0] * G)
        multi.upload_scene(s); multi.set_params(p)
        t0 = time.time()
        half = n // 2
        multi.render_frames(ts[:half], first_frame_number=0)          # two calls and a read in between, as a progressive host does
        _ = multi.read_sum()
        multi.render_frames(ts[half:], first_frame_number=half)
        got = np.ascontiguousarray(multi.read_sum())
        dt = time.time() - t0
        bad = int((got.view(np.uint32) != want.view(np.uint32)).any(axis=2).sum())
        row = {"config": cfg, "devices": G, "frames": n, "frames_done": multi.frames_done, "differing_pixels": bad, "seconds": round(dt, 2)}
        print(json.dumps(row), flush=True)
        report.append(row)
        rc = rc or (1 if bad or multi.frames_done != n else 0)
        multi.close()
    json.dump(report, open(os.path.join(ROOT, "gpurun_out", f"multi_device_rehearsal_{G}.json"), "w"), indent=1)
    return rc


if __name__ == "__main__":
    sys.exit(main())
