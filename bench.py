#!/usr/bin/env python3
"""Headline benchmark: progressive path tracing of pixar_room at 1920x1080, 16 spp per frame, 8 bounces
(BASELINE.json metric "Msamples/s + Mrays/s"; SURVEY.md 8d "Headline").

  python bench.py --gpus N --steps K --warmup W
A step = one frame (W*H*16 camera paths) per GPU.  Rank r renders frames {r, r+N, ...} of the fixed timestamp
list (frame f -> timestamp 1000+f) over the full image; the float accumulation buffers are summed with ONE
all-reduce at the end of the timed region: ptx_allreduce, the library's own RCCL communicator
(include/ptx.h), whose unique id travels over torch.distributed.  Per-GPU work is fixed => weak scaling.
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
SIMDS, CLOCK_GHZ = 1024, 2.4                      # 256 CUs x 4 SIMD-32; peak engine clock (hipDeviceProp clockRate)
# One wave64 VALU instruction occupies a SIMD for 2.1 cycles at best (v_fma/mul/add/mov/and/xor with >= 2 waves per SIMD;
# v_min/max/cmp/cndmask/cvt/shift/v_pk_* 4, v_rcp/v_sqrt 8): measured on gfx950 by tools/dbg/valu_issue.hip,
# profiles/r02_valu_issue_batch{1,2}.json.  The roofline prices EVERY instruction at the cheapest class, so it is a true
# upper bound of the issue rate.
VALU_CYCLES_PER_WAVE_INSTR = 2.1


def algorithmic_bytes(fr):
    """36*Nv + 40*Nt + 36*Nh + 76*Nl + 32*Npf, counts of the REFERENCE algorithm."""
    return 36 * fr["node_visits"] + 40 * fr["tri_tests"] + 36 * fr["ray_hits"] + 76 * fr["light_attempts"] + 32 * fr["pixel_frames"]


def executed_bytes(c):
    """Bytes the kernel itself requests for the work it really executes (its own counters from a count=1 pass): 32-B node
    records, 80-B pair records per two triangle tests, 3 x 16 B of attributes per hit that is shaded, 6 x 16 B per light
    sample, the per-pixel primary-hit cache (8 B per sample) and 16 B written + 32 B re-read and accumulated per pixel-frame."""
    return (32 * c["node_visits"] + 40 * c["tri_tests"] + 48 * c["ray_hits"] + 96 * c["light_attempts"] + 8 * c["samples"] + 48 * c["pixel_frames"])


def committed_profile():
    """Newest profiles/r*_pmc.json (tools/prof_report.py over tools/profile.sh): PMC counters of this kernel on this workload,
    per frame.  They are NOT measured by this run; the line says so."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc.json")))
    if not files:
        return None, None
    try:
        return json.load(open(files[-1])), os.path.relpath(files[-1], ROOT)
    except Exception:
        return None, None


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
    # N > 1: the all-reduce runs inside the library on its own RCCL communicator (ptx_comm_init / ptx_allreduce).  If that
    # cannot be created on every rank, all ranks fall back together to torch.distributed on the exported buffer, and the
    # output line says so.
    allreduce_via, accum = "none (1 GPU)", None
    if world > 1:
        ok, why = 1, ""
        if host_reduce:
            ok, why = 0, "gloo rehearsal"
        else:
            try:
                m.distributed.init_library_comm(ctx)
            except Exception as e:                      # noqa: BLE001 - reported in the output line
                ok, why = 0, str(e)
        flag = torch.tensor([ok], dtype=torch.int32, device=torch.device("cpu") if host_reduce else dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 1:
            allreduce_via = f"ptx_allreduce (librccl inside libptx_hip.so, {ctx.comm_world} ranks)"
        else:
            allreduce_via = "torch.distributed all_reduce on the exported buffer" + (f" ({why})" if why else "")
            accum = torch.empty(params.height, params.width, 4, dtype=torch.float32, device=dev)
            torch.cuda.synchronize()

    def frames_of(step0, nsteps):
        return [((step0 + j) * n_gpus + rank) for j in range(nsteps)]

    def run(frame_ids):
        ctx.render_frames([1000 + f for f in frame_ids])
        if world > 1 and accum is None:  # one all-reduce of the float accumulation buffer per frame batch, in place
            ctx.allreduce()
        elif world > 1:
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
        # ---- roofline of the dominant kernel (the path tracer).  The scene is ~5 MB and L2 / Infinity-Cache resident, so HBM
        # is not what bounds it (traffic below); the vector ALU's issue rate is.  Primary figure: VALU wave-instructions issued
        # per second against the SIMDs' peak issue rate.  Beside it: the SURVEY 8d "effective" rate on the REFERENCE
        # algorithm's bytes (a speed-up measure, may exceed 1) and the rate on the bytes the kernel really requests.
        launches = max(1, c["kernel_launches"])
        kernel_s = c["kernel_ms"] / 1e3
        frames_timed = len(ids)
        kernel_name = {0: "k_render", 1: "k_render_wave", 2: "k_render_pool"}[2 if args.pipeline is None else args.pipeline]
        bytes_total = sum(algorithmic_bytes(counts[f % len(counts)]) for f in ids)
        prof, prof_file = committed_profile()
        default_kernel = args.pipeline is None and args.traversal is None and args.config == "headline"
        valu = None
        if prof and "valu" in prof and default_kernel:
            n_inst = prof["valu"]["wave_instr_per_frame"]
            peak_rate = SIMDS * CLOCK_GHZ * 1e9 / VALU_CYCLES_PER_WAVE_INSTR               # wave-instructions per second, whole chip
            achieved_rate = n_inst * frames_timed / kernel_s
            valu = {"wave_instr_per_frame": n_inst, "cycles_per_wave_instr": VALU_CYCLES_PER_WAVE_INSTR, "simds": SIMDS, "clock_ghz": CLOCK_GHZ,
                    "issue_peak_ms_per_frame": n_inst / peak_rate * 1e3, "kernel_ms_per_frame": c["kernel_ms"] / frames_timed,
                    "lane_utilisation": prof["valu"]["lane_utilisation"], "frac": achieved_rate / peak_rate,
                    "counters_from": f"{prof_file} (SQ_INSTS_VALU, SQ_THREAD_CYCLES_VALU of the same kernel and workload; committed profile, not measured in this run)",
                    "issue_cost_from": "profiles/r02_valu_issue_batch1.json (tools/dbg/valu_issue.hip)"}
        # the vector-memory (texture-address) path: busy cycles per CU from the committed profile against the kernel's cycles.
        # What a gather costs that unit is measured in profiles/r02_gather_rate.json (max(16, 0.6 x lines + 3 x L1 misses) cycles
        # per load instruction); with the profile's loads, cache accesses and misses the model gives the busy time it measures.
        vmem = None
        if prof and "ta" in prof and default_kernel:
            pf = prof["per_frame"]
            busy_ms = pf["TA_TA_BUSY_sum"] / 256.0 / (CLOCK_GHZ * 1e9) * 1e3
            loads = pf.get("SQ_INSTS_VMEM_RD")
            vmem = {"unit": "texture-address path (TA/TCP), one per CU", "busy_ms_per_frame": busy_ms, "kernel_ms_per_frame": c["kernel_ms"] / frames_timed,
                    "frac": busy_ms / (c["kernel_ms"] / frames_timed), "wave_loads_per_frame": loads,
                    "busy_cycles_per_wave_load": pf["TA_TA_BUSY_sum"] / loads if loads else None,
                    "model_cycles_per_wave_load": (0.6 * pf["TCP_TOTAL_CACHE_ACCESSES_sum"] + 3.0 * pf["TCP_TCC_READ_REQ_sum"]) / loads if loads and "TCP_TCC_READ_REQ_sum" in pf else None,
                    "counters_from": f"{prof_file} (TA_TA_BUSY_sum, SQ_INSTS_VMEM_RD, TCP_TOTAL_CACHE_ACCESSES_sum, TCP_TCC_READ_REQ_sum; committed profile, not measured in this run)",
                    "cost_model_from": "profiles/r02_gather_rate.json (tools/dbg/gather_rate.hip)"}
        # executed work: one frame with the kernel's own node / triangle counters switched on, outside the timed region
        ctx.set_option("count", 1)
        ctx.reset_counters()
        ctx.render_frames([1000 + ids[0]])
        ctx.synchronize()
        ce = ctx.counters()
        ctx.set_option("count", 0)
        actual_b = executed_bytes(ce)
        actual = {"bytes_per_frame": actual_b, "node_visits_per_ray": ce["node_visits"] / max(1, ce["rays_traced"]), "tri_tests_per_ray": ce["tri_tests"] / max(1, ce["rays_traced"]),
                  "achieved_gbs": actual_b * frames_timed / kernel_s / 1e9,
                  "note": "bytes the kernel requests for the work it executes (count=1 pass of frame 0 outside the timed region); served by L1/L2, not HBM"}
        traffic = prof["hbm"]["hbm_bytes_per_frame"] * frames_timed / launches if (prof and "hbm" in prof and default_kernel) else None
        eff = bytes_total / kernel_s / 1e9
        roof = {"kernel": kernel_name, "kernel_ms_avg": c["kernel_ms"] / launches, "launches": launches, "frames_per_launch": frames_timed / launches,
                "traffic": traffic, "traffic_source": f"{prof_file}: FETCH_SIZE x2 + WRITE_SIZE per frame x frames per launch (committed profile, not measured in this run)" if traffic else None,
                "hbm_effective": {"bound": "hbm", "achieved": eff, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": eff / HBM_PEAK_GBS, "algorithmic_bytes_per_launch": bytes_total / launches,
                                  "note": "SURVEY 8d: 36Nv+40Nt+36Nh+76Nl+32Npf with the REFERENCE algorithm's counts (tests/golden/workload_counts.json) / kernel time; "
                                          "a speed-up measure over the unpruned reference traversal, not a utilisation: the kernel executes a fraction of that work and none of it comes from HBM"},
                "actual": actual, "valu": valu, "vmem": vmem}
        if valu:
            roof.update({"bound": "valu", "achieved": valu["wave_instr_per_frame"] * frames_timed / kernel_s / 1e9, "peak": SIMDS * CLOCK_GHZ / VALU_CYCLES_PER_WAVE_INSTR,
                         "unit": "G wave-instr/s", "frac": valu["frac"]})
        else:   # no committed counters for this kernel variant: only the byte rates are known
            roof.update({"bound": "hbm", "achieved": actual["achieved_gbs"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": actual["achieved_gbs"] / HBM_PEAK_GBS})
        line = {
            "metric": "Msamples/s (pixar_room 1920x1080, 16 spp/frame, 8 bounces; Mrays/s alongside)",
            "value": samples / dt / 1e6, "unit": "Msamples/s", "mrays_per_s": rays / dt / 1e6,
            "mrays_traced_per_s": rays_traced / dt / 1e6,
            "n_gpus": n_gpus, "steps": K, "warmup": W, "ms_per_step": dt * 1e3 / K, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "reference scene (tests/golden/scenes/pixar_room.ptxs, flattened from resources/pixar_room/v1 by the JS host); fixed timestamps 1000+f",
            "config": {"workload": f"{args.config}: {cfg['scene']} {cfg['width']}x{cfg['height']}, {cfg['sample_count']} spp/frame, {cfg['max_path_length']} bounces, one frame per step per GPU",
                       "frames_per_gpu": K, "traversal": {0: "literal (reference tree)", 1: "reference tree + best-t pruning", 2: "leaf-SAH tree", 3: "leaf-SAH tree + best-t pruning"}[3 if args.traversal is None else args.traversal],
                       "pipeline": {0: "one ray per lane", 1: "wave-scheduled, per-lane leaf queues", 2: "wave-scheduled, shared leaf queue"}[2 if args.pipeline is None else args.pipeline],
                       "light_rays": "exact any-hit termination + blocker cache (library default; same sums and reference ray counters as closest-hit queries)",
                       "parallelism": f"frame-split x{n_gpus} + 1 all-reduce", "allreduce": allreduce_via, "rccl_ranks": ctx.comm_world},
            "roofline": roof,
        }
        if not args.no_cpu_baseline and n_gpus == 1:
            line["cpu_baseline"] = cpu_baseline(m, scene, params)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
