#!/usr/bin/env python3
"""Headline benchmark: progressive path tracing of pixar_room at 1920x1080, 16 spp per frame, 8 bounces
(BASELINE.json metric "Msamples/s + Mrays/s"; SURVEY.md 8d "Headline").

  python bench.py --gpus N --steps K --warmup W
A step = one frame (W*H*16 camera paths) per GPU.  Rank r renders frames {r, r+N, ...} of the fixed timestamp
list (frame f -> timestamp 1000+f) over the full image; the float accumulation buffers are summed with ONE
all-reduce (RCCL) at the end of the timed region.  Per-GPU work is fixed => weak scaling.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # first: one HIP runtime per process
import torch.distributed as dist
import numpy as np

import ptx_import

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s


def algorithmic_bytes(fr):
    """36*Nv + 40*Nt + 36*Nh + 76*Nl + 32*Npf, counts of the REFERENCE algorithm."""
    return 36 * fr["node_visits"] + 40 * fr["tri_tests"] + 36 * fr["ray_hits"] + 76 * fr["light_attempts"] + 32 * fr["pixel_frames"]


def measured_traffic(frames_per_launch):
    """HBM bytes per path-tracing launch from the committed PMC passes (profiles/r01_traffic.json, written by
    tools/traffic_from_pmc.py: FETCH_SIZE x2 gfx950 correction + WRITE_SIZE, KB -> B, per frame) x the frames one launch
    of this run renders; None when no such profile is committed."""
    try:
        t = json.load(open(os.path.join(ROOT, "profiles", "r01_traffic.json")))
        return t["hbm_bytes_per_frame"] * frames_per_launch
    except Exception:
        return None


def cpu_baseline(m, scene, params, seconds_target=15.0):
    """Times the CPU oracle (kind 'port') on an evenly spread row sample of frame 0 of the same workload."""
    from oracle import orc
    threads = min(orc.max_threads(), 16)
    # calibrate on ~2 rows per thread, then size the sample for ~seconds_target
    t0 = time.time()
    _, c = orc.render_rows(scene, params, 1000, 0, params.height, max(1, params.height // (2 * threads)), nthreads=threads)
    dt = time.time() - t0
    rate = c["samples"] / max(dt, 1e-6)
    rows = int(max(2, min(params.height, rate * seconds_target / (params.width * params.sample_count))))
    step = max(1, params.height // rows)
    t0 = time.time()
    _, c = orc.render_rows(scene, params, 1000, 0, params.height, step, nthreads=threads)
    dt = time.time() - t0
    nrows = len(range(0, params.height, step))
    return {"value": c["samples"] / dt / 1e6, "unit": "Msamples/s", "cores": threads, "kind": "port",
            "mrays_per_s": c["rays"] / dt / 1e6,
            "sample": f"CPU oracle (oracle/ptx_oracle.c, OpenMP), frame 0 of the same workload, every {step}th row "
                      f"({nrows} rows x {params.width} px x {params.sample_count} spp = {c['samples']} samples) in {dt:.1f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=32)      # one launch of K frames per GPU: the drain at the end of a launch (about 40 ms
                                                            # while the last paths finish) is amortised over the frames, as in a long render
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="headline")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--traversal", type=int, default=None, help="0 literal, 1 pruned, 2 leaf-SAH, 3 leaf-SAH + pruned (library default)")
    ap.add_argument("--pipeline", type=int, default=None, help="0 one ray at a time per lane, 1 wave-scheduled with per-lane leaf queues, 2 wave-scheduled with a shared leaf queue (library default)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend; 'gloo' + PTX_BENCH_ONE_DEVICE=1 rehearses N>1 on one GPU")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    one_device = os.environ.get("PTX_BENCH_ONE_DEVICE", "0") == "1"      # rehearsal of the N>1 code path on a 1-GPU box
    dev_index = 0 if (world == 1 or one_device) else local_rank
    torch.cuda.set_device(dev_index)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(args.backend)
    n_gpus = world
    dev = torch.device("cuda", dev_index)
    host_reduce = world > 1 and args.backend != "nccl"

    m = ptx_import.load()
    cfg = m.CONFIGS[args.config]
    scene = m.load_scene(cfg["scene"])
    params = m.config_params(cfg["scene"], cfg["width"], cfg["height"], cfg["sample_count"], cfg["max_path_length"])
    counts = json.load(open(os.path.join(ROOT, "tests", "golden", "workload_counts.json")))[args.config]["frames"]

    ctx = m.Context(dev.index)
    ctx.upload_scene(scene)
    ctx.set_params(params)
    if args.traversal is not None:
        ctx.set_option("traversal", args.traversal)
    if args.pipeline is not None:
        ctx.set_option("pipeline", args.pipeline)
    K, W = args.steps, args.warmup
    accum = torch.zeros(params.height, params.width, 4, dtype=torch.float32, device=dev)

    def frames_of(step0, nsteps):
        return [((step0 + j) * n_gpus + rank) for j in range(nsteps)]

    def run(frame_ids):
        ctx.render_frames([1000 + f for f in frame_ids])
        if world > 1:  # one all-reduce of the float accumulation buffer per frame batch
            ctx.accum_export_device(accum.data_ptr())
            if host_reduce:                      # gloo rehearsal: stage through the host
                h = accum.cpu()
                dist.all_reduce(h)
                accum.copy_(h)
            else:
                dist.all_reduce(accum)           # RCCL over xGMI
            torch.cuda.current_stream().synchronize()
            ctx.accum_import_device(accum.data_ptr(), ctx.frames_done * world)
        ctx.synchronize()

    if W > 0:
        run(frames_of(0, W))
    ctx.reset_accum()
    ctx.reset_counters()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ids = frames_of(W, K)
    run(ids)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    rdev = torch.device("cpu") if host_reduce else dev
    tmax = torch.tensor([dt], dtype=torch.float64, device=rdev)
    c = ctx.counters()
    tot = torch.tensor([c["samples"], c["rays"], c["rays_traced"]], dtype=torch.float64, device=rdev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(tot)
    dt = float(tmax.item())
    samples, rays, rays_traced = [float(x) for x in tot.tolist()]

    if rank == 0:
        # roofline of the dominant kernel (k_render): algorithmic bytes of the frames this rank's launches processed
        # / HIP-event time of those launches on the library's stream
        launches = max(1, c["kernel_launches"])
        bytes_total = sum(algorithmic_bytes(counts[f % len(counts)]) for f in ids)
        kernel_s = c["kernel_ms"] / 1e3
        achieved = bytes_total / kernel_s / 1e9
        line = {
            "metric": "Msamples/s (pixar_room 1920x1080, 16 spp/frame, 8 bounces; Mrays/s alongside)",
            "value": samples / dt / 1e6, "unit": "Msamples/s", "mrays_per_s": rays / dt / 1e6,
            "mrays_traced_per_s": rays_traced / dt / 1e6,
            "n_gpus": n_gpus, "steps": K, "warmup": W, "ms_per_step": dt * 1e3 / K, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "reference scene (tests/golden/scenes/pixar_room.ptxs, flattened from resources/pixar_room/v1 by the JS host); fixed timestamps 1000+f",
            "config": {"workload": f"{args.config}: {cfg['scene']} {cfg['width']}x{cfg['height']}, {cfg['sample_count']} spp/frame, {cfg['max_path_length']} bounces, one frame per step per GPU",
                       "frames_per_gpu": K, "traversal": {0: "literal (reference tree)", 1: "reference tree + best-t pruning", 2: "leaf-SAH tree", 3: "leaf-SAH tree + best-t pruning"}[3 if args.traversal is None else args.traversal],
                       "pipeline": {0: "one ray per lane", 1: "wave-scheduled, per-lane leaf queues", 2: "wave-scheduled, shared leaf queue"}[2 if args.pipeline is None else args.pipeline],
                       "parallelism": f"frame-split x{n_gpus} + 1 all-reduce"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": measured_traffic(K / launches), "kernel": {0: "k_render", 1: "k_render_wave", 2: "k_render_pool"}[2 if args.pipeline is None else args.pipeline], "kernel_ms_avg": c["kernel_ms"] / launches, "launches": launches,
                         "algorithmic_bytes_per_launch": bytes_total / launches,
                         "note": "algorithmic bytes = 36Nv+40Nt+36Nh+76Nl+32Npf with the REFERENCE algorithm's counts (tests/golden/workload_counts.json); the scene is L2-resident, so this is an effective rate"},
        }
        if not args.no_cpu_baseline and n_gpus == 1:
            line["cpu_baseline"] = cpu_baseline(m, scene, params)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
