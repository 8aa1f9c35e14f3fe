"""Whole BASELINE jobs, frame by frame: the default walk (leaf-SAH tree, margin rule, any-hit light rays) against the unpruned
walk of the same tree (option traversal = 2), every pixel of every frame bit for bit.  The headline job is a test
(tests/test_prune_certificate.py); this tool does the same for any configuration -- C5 alone is 256 frames of 3840 x 2160 at 12
bounces, too long for the suite -- and writes what it found to gpurun_out/whole_job_check_<config>.json (copied to profiles/).

    python tools/whole_job_check.py C2 C3 C4 C5 [--frames N] [--first-frame F] [--mode default|provable|margin0|unpruned] [--pipeline 0|1|2] [--ref-literal]
(--mode provable: option prune = 1 against the unpruned walk; margin0: round 2's margin rule, option margin_lat = 0)
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401,E402  (one HIP runtime per process)
import numpy as np  # noqa: E402
import ptx_import  # noqa: E402


def main():
    m = ptx_import.load()
    args = sys.argv[1:]
    limit = None
    if "--frames" in args:
        k = args.index("--frames")
        limit = int(args[k + 1])
        del args[k:k + 2]
    first = 0
    if "--first-frame" in args:                      # other timestamps than the job's own: frames first .. first + n - 1
        k = args.index("--first-frame")
        first = int(args[k + 1])
        del args[k:k + 2]
    ref_opts, ref_name = dict(traversal=2, prune=2), "unpruned walk of the leaf-SAH tree (traversal=2)"
    if "--ref-literal" in args:                      # reference = the shader's literal walk of the reference tree (traversal = 0, pipeline 0)
        args.remove("--ref-literal")
        ref_opts, ref_name = dict(traversal=0, prune=2, pipeline=0), "literal threaded walk of the reference tree (traversal=0, pipeline 0)"
    pipeline = 2
    if "--pipeline" in args:                         # the kernel of the mode under test (the reference stays as chosen above)
        k = args.index("--pipeline")
        pipeline = int(args[k + 1])
        del args[k:k + 2]
    mode_name = "default"
    if "--mode" in args:
        k = args.index("--mode")
        mode_name = args[k + 1]
        del args[k:k + 2]
    mode_opts = {"default": dict(traversal=3, prune=2, margin_lat=3000), "provable": dict(traversal=3, prune=1, margin_lat=3000),
                 "margin0": dict(traversal=3, prune=2, margin_lat=0),
                 "unpruned": dict(traversal=2, prune=2, margin_lat=3000)}[mode_name]
    mode_opts = dict(mode_opts, pipeline=pipeline)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    rc = 0
    for cfg in args or ["headline"]:
        c = m.CONFIGS[cfg]
        n = c["frames"] if limit is None else limit          # more than the job has: further timestamps (1000 + frame)
        ctx = m.Context(0)
        ctx.upload_scene(m.load_scene(c["scene"]))
        ctx.set_params(m.config_params(c["scene"], c["width"], c["height"], c["sample_count"], c["max_path_length"]))
        ctx.set_option("pipeline", 2)
        rays = samples = 0
        differing = []
        running = None                                   # float32 sum of the per-frame images of the mode under test, in frame order
        ms = {"default": 0.0, "unpruned": 0.0}
        t_start = time.time()
        for f in range(first, first + n):
            img = {}
            for mode, opts in (("default", mode_opts), ("unpruned", ref_opts)):
                ctx.set_option("pipeline", 2)
                for k, v in opts.items():
                    ctx.set_option(k, v)
                ctx.reset_accum()
                ctx.reset_counters()
                t0 = time.time()
                ctx.render_frames([1000 + f], first_frame_number=f)
                ctx.synchronize()
                ms[mode] += (time.time() - t0) * 1e3
                img[mode] = np.ascontiguousarray(ctx.read_sum()).view(np.uint32)
                if mode == "default":
                    cn = ctx.counters()
                    rays += cn["rays"]
                    samples += cn["samples"]
            running = img["default"].view(np.float32).copy() if running is None else running + img["default"].view(np.float32)
            bad = int((img["default"] != img["unpruned"]).sum())
            if bad:
                differing.append({"frame": f, "words": bad})
                rc = 1
            if f % 8 == 7 or f == first + n - 1:
                print(f"{cfg}: {f + 1 - first}/{n} frames, {rays / 1e9:.1f} G rays, {len(differing)} differing, {time.time() - t_start:.0f} s", flush=True)
        # the same frames in launches of up to 64 (tile-major task list, frames summed by k_accumulate in frame order): the
        # accumulation buffer must equal the float32 sum of the single-frame images above, bit for bit
        ctx.set_option("pipeline", 2)
        for k, v in mode_opts.items():
            ctx.set_option(k, v)
        ctx.reset_accum()
        t0 = time.time()
        ctx.render_frames([1000 + f for f in range(first, first + n)], first_frame_number=first)
        ctx.synchronize()
        batched_ms = (time.time() - t0) * 1e3 / n
        batched = np.ascontiguousarray(ctx.read_sum()).view(np.uint32)
        batched_bad = int((batched != running.view(np.uint32)).sum())
        rc = rc or (1 if batched_bad else 0)
        print(f"{cfg}: {n} frames in one call: {batched_ms:.2f} ms per frame, {batched_bad} words differ from the sum of the single frames", flush=True)
        out = {"config": cfg, "one_call_of_all_frames": {"ms_per_frame": round(batched_ms, 2), "words_differing_from_sum_of_single_frames": batched_bad}, "scene": c["scene"], "width": c["width"], "height": c["height"], "bounces": c["max_path_length"],
               "frames_compared": n, "first_frame": first, "frames_of_job": c["frames"], "samples": samples, "reference_rays": rays,
               "differing_frames": differing, "ms_per_frame_one_frame_per_call": {k: round(v / n, 2) for k, v in ms.items()},
               "mode": mode_name, "options": mode_opts,
               "compared": f"{mode_name} {mode_opts} (anyhit=1) vs {ref_name}: accumulation buffer bits of every frame"}
        json.dump(out, open(os.path.join(ROOT, "gpurun_out", f"whole_job_check_{cfg}{'' if mode_name == 'default' else '_' + mode_name}{'' if first == 0 else '_from' + str(first)}{'_vs_literal' if 'literal' in ref_name else ''}{'' if pipeline == 2 else '_pipeline' + str(pipeline)}.json"), "w"), indent=1)
        print(json.dumps(out), flush=True)
        ctx.close()
    return rc


if __name__ == "__main__":
    sys.exit(main())
