#!/usr/bin/env python3
"""Headline benchmark: progressive path tracing of pixar_room at 1920x1080, 16 spp per frame, 8 bounces
(BASELINE.json metric "Msamples/s + Mrays/s"; SURVEY.md 8d "Headline").

  python bench.py --gpus N --steps K --warmup W [--scaling weak|strong] [--config headline|C2|C3|C4|C5]

A step = one frame (W*H*16 camera paths).  Rank r of N renders frames {r, r+N, ...} of the fixed timestamp list
(frame f -> timestamp 1000+f) over the full image; the float accumulation buffers are summed with ONE all-reduce at
the end of the timed region: ptx_allreduce, the library's own RCCL communicator (include/ptx.h), whose
unique id travels over torch.distributed.
  --scaling weak (default): K frames PER GPU (fixed per-GPU work).  The line also carries "strong_scaling": the
    north_star job (the config's whole frame list, 64 frames = 1024 spp for the headline) split over the N ranks and timed
    the same way, outside the headline region.
  --scaling strong: the K steps are the job's frames in total (default 64), split over the ranks; `value` is that run.
How the ranks come to be: under a launcher (WORLD_SIZE set: the driver's torch.distributed.run command) this process IS
one rank.  Without one, `--gpus N` with N > 1 starts N ranks itself, as a `python -m torch.distributed.run` CHILD process
before this process has made any GPU call (never a re-exec), and passes the child's line and exit code on.
"""
import argparse
import glob
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s
SIMDS, CLOCK_GHZ = 1024, 2.4                      # 256 CUs x 4 SIMD-32; peak engine clock (hipDeviceProp clockRate)
# One wave64 VALU instruction occupies a SIMD for 2.1 cycles at best (v_fma/mul/add/mov/and/xor with >= 2 waves per SIMD;
# v_min/max/cmp/cndmask/cvt/shift/v_pk_* 4, v_rcp/v_sqrt 8): measured on gfx950 by tools/dbg/valu_issue.hip,
# profiles/r02_valu_issue_batch{1,2}.json.  The roofline prices EVERY instruction at the cheapest class, so it is a true
# upper bound of the issue rate.
VALU_CYCLES_PER_WAVE_INSTR = 2.1
STALE_TOLERANCE = 0.05                            # committed profile vs this run, kernel ms per frame


def algorithmic_bytes(fr):
    """36*Nv + 40*Nt + 36*Nh + 76*Nl + 32*Npf, counts of the REFERENCE algorithm."""
    return 36 * fr["node_visits"] + 40 * fr["tri_tests"] + 36 * fr["ray_hits"] + 76 * fr["light_attempts"] + 32 * fr["pixel_frames"]


def executed_bytes(c):
    """Bytes the kernel itself requests for the work it really executes (its own counters from a count=1 pass): 32-B node
    records, 80-B pair records per two triangle tests, 3 x 16 B of attributes per hit that is shaded, 6 x 16 B per light
    sample, the per-pixel primary-hit cache (8 B per sample) and 16 B written + 32 B re-read and accumulated per pixel-frame."""
    return (32 * c["node_visits"] + 40 * c["tri_tests"] + 48 * c["ray_hits"] + 96 * c["light_attempts"] + 8 * c["samples"] + 48 * c["pixel_frames"])


def committed_profile(config):
    """Newest committed PMC summary of this workload (tools/prof_report.py over tools/profile.sh): profiles/rNN_pmc.json for
    the headline, profiles/rNN_<config>_pmc.json for the others.  The counters are NOT measured by this run; the line says so
    and marks them stale when this run's kernel time per frame has moved away from the profile's."""
    pat = "r[0-9][0-9]_pmc.json" if config == "headline" else f"r[0-9][0-9]_{config.lower()}_pmc.json"
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", pat)))
    if not files:
        return None, None
    try:
        return json.load(open(files[-1])), os.path.relpath(files[-1], ROOT)
    except Exception:
        return None, None


def reference_swiftshader_baseline():
    """The reference shader itself (verbatim text on SwiftShader WebGL2, headless Chromium) as timed in the BUILD container
    when the committed frames were captured (oracle/gen_ref_frames.py stores each draw's wall time): data from the fixtures,
    not re-measured here -- the reference's sources cannot travel to the GPU box."""
    import numpy as np
    out = {}
    for key, rel in (("pixar_room_96x54_16spp_8bounces", "pixar_room_b8.npz"), ("cornell_256x256_16spp_4bounces_C1", "cornell_c1.npz")):
        path = os.path.join(ROOT, "tests", "golden", "ref_frames", rel)
        try:
            d = np.load(path)
            samples = int(d["width"]) * int(d["height"]) * int(d["sample_count"]) * len(d["timestamps"])
            ms = float(np.sum(d["draw_ms"]))
            out[key] = {"ksamples_per_s": samples / ms, "cores": int(d["host_cores"]), "draw_ms": ms, "samples": samples,
                        "fixture": os.path.relpath(path, ROOT)}
        except Exception as e:                      # noqa: BLE001
            out[key] = {"error": str(e)}
    out["note"] = "reference fragment shader on Google SwiftShader (8 host cores of the build container), wall time of one gl draw; baseline only"
    return out


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
                      f"({nrows} rows x {params.width} px x {params.sample_count} spp = {c['samples']} samples) in {dt:.1f} s",
            "reference_swiftshader": reference_swiftshader_baseline()}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None,    # weak: one launch of K frames per GPU (default 32): the drain at the end of a launch
                                                            # (about 40 ms while the last paths finish) is amortised as in a long render;
                                                            # strong: K frames in total (default: the config's job, 64 for the headline)
                    help="timed frames: per GPU (--scaling weak, default 32) or in total (--scaling strong, default = the config's job)")
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak")
    ap.add_argument("--config", default="headline")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the strong-scaling and exact-mode legs outside the headline region")
    ap.add_argument("--traversal", type=int, default=None, help="0 literal, 1 pruned, 2 leaf-SAH, 3 leaf-SAH + pruned (library default)")
    ap.add_argument("--prune", type=int, default=None, help="1 provable rule, 2 margin rule (library default)")
    ap.add_argument("--pipeline", type=int, default=None, help="0 one ray at a time per lane, 1 wave-scheduled with per-lane leaf queues, 2 wave-scheduled with a shared leaf queue (library default)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend; 'gloo' + PTX_BENCH_ONE_DEVICE=1 rehearses N>1 on one GPU")
    ap.add_argument("--dry-run", action="store_true", help="launcher check without a GPU: the ranks meet over gloo and rank 0 prints how many there are")
    return ap.parse_args(argv)


def spawn_ranks(args):
    """--gpus N without a launcher: N fresh rank processes under `python -m torch.distributed.run`, started as a child before
    this process has touched the GPU (importing torch does not); its stdout (rank 0's JSON line) and exit code are passed on."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    child = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    for ln in child.stdout.splitlines():            # stdout carries the one JSON line; whatever else the ranks printed goes to stderr
        print(ln, file=sys.stdout if ln.startswith("{") else sys.stderr, flush=True)
    return child.returncode


def dry_run(args, world, rank):
    import torch
    import torch.distributed as dist
    if world > 1:
        dist.init_process_group("gloo")
    seen = torch.tensor([1], dtype=torch.int64)
    if world > 1:
        dist.all_reduce(seen)
        dist.barrier()
    if rank == 0:
        print(json.dumps({"dry_run": True, "n_gpus": world, "ranks_seen": int(seen.item()), "gpus_arg": args.gpus, "scaling": args.scaling}), flush=True)
    if world > 1:
        dist.destroy_process_group()
    return 0


def main():
    args = parse_args()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return spawn_ranks(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.dry_run:
        return dry_run(args, world, rank)

    import torch  # before the library: one HIP runtime per process
    import torch.distributed as dist
    import ptx_import

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
    rdev = torch.device("cpu") if host_reduce else dev

    m = ptx_import.load()
    cfg = m.CONFIGS[args.config]
    scene = m.load_scene(cfg["scene"])
    params = m.config_params(cfg["scene"], cfg["width"], cfg["height"], cfg["sample_count"], cfg["max_path_length"])
    counts = json.load(open(os.path.join(ROOT, "tests", "golden", "workload_counts.json")))[args.config]["frames"]

    ctx = m.Context(dev.index)
    t0 = time.perf_counter()
    ctx.upload_scene(scene)
    upload_ms = (time.perf_counter() - t0) * 1e3
    ctx.set_params(params)
    if args.traversal is not None:
        ctx.set_option("traversal", args.traversal)
    if args.prune is not None:
        ctx.set_option("prune", args.prune)
    if args.pipeline is not None:
        ctx.set_option("pipeline", args.pipeline)
    W = args.warmup
    strong = args.scaling == "strong"
    K = args.steps if args.steps is not None else (cfg["frames"] if strong else 32)
    # N > 1: the all-reduce runs inside the library on its own RCCL communicator (ptx_comm_init / ptx_allreduce).  If that
    # cannot be created, every rank learns it in the same collective (init_library_comm) and all fall back together to
    # torch.distributed on the exported buffer; the output line says so.
    allreduce_via, accum = "none (1 GPU)", None
    if world > 1:
        why = ""
        if host_reduce:
            why = "gloo rehearsal"
        else:
            try:
                m.distributed.init_library_comm(ctx)
            except Exception as e:                      # noqa: BLE001 - the same on every rank; reported in the output line
                why = str(e)
        if not why:
            allreduce_via = f"ptx_allreduce (librccl inside libptx_hip.so, {ctx.comm_world} ranks)"
        else:
            allreduce_via = f"torch.distributed all_reduce on the exported buffer ({why})"
            accum = torch.empty(params.height, params.width, 4, dtype=torch.float32, device=dev)
            torch.cuda.synchronize()

    def run(frame_ids):
        """This rank's frames + the one all-reduce of the float accumulation buffer."""
        if len(frame_ids):
            ctx.render_frames([1000 + f for f in frame_ids], first_frame_number=int(frame_ids[0]))
        if world > 1 and accum is None:
            ctx.allreduce()                          # in place, RCCL over xGMI inside the library
        elif world > 1:
            ctx.accum_export_device(accum.data_ptr())
            done = torch.tensor([ctx.frames_done], dtype=torch.int64, device=rdev)
            if host_reduce:                          # gloo rehearsal: stage through the host
                h = accum.cpu()
                dist.all_reduce(h)
                accum.copy_(h)
            else:
                dist.all_reduce(accum)               # RCCL over xGMI
            dist.all_reduce(done)
            torch.cuda.current_stream().synchronize()
            ctx.accum_import_device(accum.data_ptr(), int(done.item()))
        ctx.synchronize()

    def timed(frame_ids):
        """Barrier + synchronize on both sides, MAX over ranks; returns (seconds, this job's counters summed over ranks, own counters)."""
        ctx.reset_accum()
        ctx.reset_counters()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(frame_ids)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        c = ctx.counters()
        tmax = torch.tensor([dt], dtype=torch.float64, device=rdev)
        tot = torch.tensor([c["samples"], c["rays"], c["rays_traced"], len(frame_ids)], dtype=torch.float64, device=rdev)
        if world > 1:
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dist.all_reduce(tot)
        return float(tmax.item()), [float(x) for x in tot.tolist()], c

    def my_frames(first, n_total):
        """Frames first .. first + n_total - 1 of the job's list, dealt round-robin (distributed.frames_for_rank)."""
        return [first + f for f in m.frames_for_rank(n_total, rank, world)]

    # one-off costs that the timed region does not contain: the first call fills the per-pixel primary-hit cache (k_primary, once
    # per camera), allocates the frame buffers and loads the code object
    first_call_ms = second_call_ms = None
    if not args.no_extras:                          # (profiling runs skip it: one launch per frame count there)
        t0 = time.perf_counter()
        run(my_frames(0, world))
        first_call_ms = (time.perf_counter() - t0) * 1e3
        t0 = time.perf_counter()
        run(my_frames(0, world))
        second_call_ms = (time.perf_counter() - t0) * 1e3
    if W > 0:
        run(my_frames(0, W * world))
    # ---- the timed region
    ids = my_frames(W * world, K) if strong else my_frames(W * world, K * world)
    dt, (samples, rays, rays_traced, frames_all), c = timed(ids)

    # ---- outside the timed region: the other scaling mode, the exact modes
    extras = not args.no_extras
    other = None
    if extras:
        job = cfg["frames"]
        if strong:       # the weak figure beside a strong headline: 32 frames per GPU
            o_ids, o_label = my_frames(0, 32 * world), {"scaling": "weak", "frames_per_gpu": 32}
        else:
            o_ids, o_label = my_frames(0, job), {"scaling": "strong", "frames_total": job, "frames_per_gpu": job / world,
                                                 "job": f"the config's whole frame list: {job} frames x {cfg['sample_count']} spp = {job * cfg['sample_count']} spp"}
        o_dt, (o_samples, o_rays, _, _), _ = timed(o_ids)
        other = dict(o_label, value=o_samples / o_dt / 1e6, unit="Msamples/s", mrays_per_s=o_rays / o_dt / 1e6, seconds=o_dt,
                     note="same barrier + synchronize bracket and MAX over ranks as the headline region, all-reduce included")
    exact = None
    if extras and n_gpus == 1 and args.traversal is None and args.prune is None and args.pipeline is None:
        # the modes whose equality with the reference's unpruned traversal is PROVEN for every ray (include/ptx.h "prune"):
        # same frames, timed the same way, so the price of the proof stands next to the default in the line
        exact = {"frames": min(8, len(ids)), "default_ms_per_frame": dt * 1e3 / max(1, len(ids)),
                 "note": "default = margin rule with lateral allowance (prune = 2, margin_lat = 3000 ppm): every frame of every BASELINE job equals the "
                         "unpruned walk bit for bit (profiles/r03_whole_job_check.json; round 2's rule without the allowance got 6 pixel-frames "
                         "of C5's 2.1e9 wrong) and ptx_selfcheck_pruning finds 0 differing rays in 1e8 adversarial rays per scene; not proven for "
                         "every conceivable ray; primary rays are never pruned in any mode"}
        sub = ids[:exact["frames"]]
        for label, key, val, back in (("prune=1 (provable rule)", "prune", 1, 2), ("traversal=2 (leaf-SAH tree, no pruning, no any-hit)", "traversal", 2, 3)):
            ctx.set_option(key, val)
            run(sub[:1])
            e_dt, (e_samples, _, _, _), _ = timed(sub)
            ctx.set_option(key, back)
            exact[label] = {"ms_per_frame": e_dt * 1e3 / len(sub), "msamples_per_s": e_samples / e_dt / 1e6, "vs_default": (e_dt / len(sub)) / (dt / max(1, len(ids)))}

    if rank == 0:
        # ---- roofline of the dominant kernel (the path tracer).  The scene is ~5 MB and L2 / Infinity-Cache resident, so HBM
        # is not what bounds it (traffic below); the vector ALU's issue rate is.  Primary figure: VALU wave-instructions issued
        # per second against the SIMDs' peak issue rate.  Beside it: the SURVEY 8d "effective" rate on the REFERENCE
        # algorithm's bytes (a speed-up measure, may exceed 1) and the rate on the bytes the kernel really requests.
        launches = max(1, c["kernel_launches"])
        kernel_s = c["kernel_ms"] / 1e3
        frames_timed = max(1, len(ids))
        kernel_ms_per_frame = c["kernel_ms"] / frames_timed
        pipeline = 2 if args.pipeline is None else args.pipeline
        kernel_name = {0: "k_render", 1: "k_render_wave", 2: "k_render_pool"}[pipeline]
        bytes_total = sum(algorithmic_bytes(counts[f % len(counts)]) for f in ids)
        prof, prof_file = committed_profile(args.config)
        default_kernel = args.pipeline is None and args.traversal is None and args.prune is None
        if not default_kernel:
            prof = None
        stale = None
        prof_ms = (prof or {}).get("kernel_ms_per_frame_timed_launch") or (prof or {}).get("kernel_ms_per_frame")
        if prof_ms:
            drift = kernel_ms_per_frame / prof_ms - 1.0
            stale = {"stale": abs(drift) > STALE_TOLERANCE, "profile_kernel_ms_per_frame": prof_ms, "run_kernel_ms_per_frame": kernel_ms_per_frame,
                     "drift": drift, "tolerance": STALE_TOLERANCE}
            if stale["stale"]:
                print(f"bench.py: WARNING: {prof_file} was taken at {prof_ms:.2f} ms per frame, this run measures {kernel_ms_per_frame:.2f} "
                      f"({drift:+.1%}): the counter-derived roofline figures are STALE -- re-run tools/profile.sh", file=sys.stderr, flush=True)
        valu = None
        if prof and "valu" in prof:
            n_inst = prof["valu"]["wave_instr_per_frame"]
            peak_rate = SIMDS * CLOCK_GHZ * 1e9 / VALU_CYCLES_PER_WAVE_INSTR               # wave-instructions per second, whole chip
            achieved_rate = n_inst * frames_timed / kernel_s
            valu = {"wave_instr_per_frame": n_inst, "cycles_per_wave_instr": VALU_CYCLES_PER_WAVE_INSTR, "simds": SIMDS, "clock_ghz": CLOCK_GHZ,
                    "issue_peak_ms_per_frame": n_inst / peak_rate * 1e3, "kernel_ms_per_frame": kernel_ms_per_frame,
                    "lane_utilisation": prof["valu"]["lane_utilisation"], "frac": achieved_rate / peak_rate,
                    "salu_per_valu": (prof["per_frame"]["SQ_INSTS_SALU"] / n_inst) if "SQ_INSTS_SALU" in prof.get("per_frame", {}) else None,
                    "counters_from": f"{prof_file} (SQ_INSTS_VALU, SQ_THREAD_CYCLES_VALU, SQ_INSTS_SALU of the same kernel and workload; committed profile, not measured in this run)",
                    "issue_cost_from": "profiles/r02_valu_issue_batch1.json (tools/dbg/valu_issue.hip)"}
        # the vector-memory (texture-address) path: busy cycles per CU from the committed profile against the kernel's cycles.
        # What a gather costs that unit is measured in profiles/r02_gather_rate.json (max(16, 0.6 x lines + 3 x L1 misses) cycles
        # per load instruction); with the profile's loads, cache accesses and misses the model gives the busy time it measures.
        vmem = None
        if prof and "ta" in prof:
            pf = prof["per_frame"]
            busy_ms = pf["TA_TA_BUSY_sum"] / 256.0 / (CLOCK_GHZ * 1e9) * 1e3
            loads = pf.get("SQ_INSTS_VMEM_RD")
            vmem = {"unit": "texture-address path (TA/TCP), one per CU", "busy_ms_per_frame": busy_ms, "kernel_ms_per_frame": kernel_ms_per_frame,
                    "frac": busy_ms / kernel_ms_per_frame, "wave_loads_per_frame": loads,
                    "busy_cycles_per_wave_load": pf["TA_TA_BUSY_sum"] / loads if loads else None,
                    "model_cycles_per_wave_load": (0.6 * pf["TCP_TOTAL_CACHE_ACCESSES_sum"] + 3.0 * pf["TCP_TCC_READ_REQ_sum"]) / loads if loads and "TCP_TCC_READ_REQ_sum" in pf else None,
                    "counters_from": f"{prof_file} (TA_TA_BUSY_sum, SQ_INSTS_VMEM_RD, TCP_TOTAL_CACHE_ACCESSES_sum, TCP_TCC_READ_REQ_sum; committed profile, not measured in this run)",
                    "cost_model_from": "profiles/r02_gather_rate.json (tools/dbg/gather_rate.hip)"}
        # executed work: one frame with the kernel's own node / triangle counters switched on, outside the timed region
        ctx.set_option("count", 1)
        ctx.reset_counters()
        ctx.render_frames([1000 + (ids[0] if ids else 0)])
        ctx.synchronize()
        ce = ctx.counters()
        ctx.set_option("count", 0)
        actual_b = executed_bytes(ce)
        actual = {"bytes_per_frame": actual_b, "node_visits_per_ray": ce["node_visits"] / max(1, ce["rays_traced"]), "tri_tests_per_ray": ce["tri_tests"] / max(1, ce["rays_traced"]),
                  "achieved_gbs": actual_b * frames_timed / kernel_s / 1e9,
                  "note": "bytes the kernel requests for the work it executes (count=1 pass of one frame outside the timed region); served by L1/L2, not HBM"}
        traffic = prof["hbm"]["hbm_bytes_per_frame"] * frames_timed / launches if (prof and "hbm" in prof) else None
        eff = bytes_total / kernel_s / 1e9
        roof = {"kernel": kernel_name, "kernel_ms_avg": c["kernel_ms"] / launches, "launches": launches, "frames_per_launch": frames_timed / launches,
                "traffic": traffic, "traffic_source": f"{prof_file}: FETCH_SIZE x2 + WRITE_SIZE per frame x frames per launch (committed profile, not measured in this run)" if traffic else None,
                "hbm_effective": {"bound": "hbm", "achieved": eff, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": eff / HBM_PEAK_GBS, "algorithmic_bytes_per_launch": bytes_total / launches,
                                  "note": "SURVEY 8d: 36Nv+40Nt+36Nh+76Nl+32Npf with the REFERENCE algorithm's counts (tests/golden/workload_counts.json) / kernel time; "
                                          "a speed-up measure over the unpruned reference traversal, not a utilisation: the kernel executes a fraction of that work and none of it comes from HBM"},
                "actual": actual, "valu": valu, "vmem": vmem, "profile_check": stale, "stale": bool(stale and stale["stale"])}
        if valu:
            roof.update({"bound": "valu", "achieved": valu["wave_instr_per_frame"] * frames_timed / kernel_s / 1e9, "peak": SIMDS * CLOCK_GHZ / VALU_CYCLES_PER_WAVE_INSTR,
                         "unit": "G wave-instr/s", "frac": valu["frac"]})
        else:   # no committed counters for this kernel variant: only the byte rates are known
            roof.update({"bound": "hbm", "achieved": actual["achieved_gbs"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": actual["achieved_gbs"] / HBM_PEAK_GBS})
        prune = 2 if args.prune is None else args.prune
        line = {
            "metric": "Msamples/s (pixar_room 1920x1080, 16 spp/frame, 8 bounces; Mrays/s alongside)",
            "value": samples / dt / 1e6, "unit": "Msamples/s", "mrays_per_s": rays / dt / 1e6,
            "mrays_traced_per_s": rays_traced / dt / 1e6,
            "n_gpus": n_gpus, "steps": K, "warmup": W, "ms_per_step": dt * 1e3 / K, "higher_is_better": True,
            "scaling": args.scaling, "vs_baseline": None, "dtype": "f32", "data": f"reference scene (tests/golden/scenes/{cfg['scene']}.ptxs, flattened from the reference's resources/ by the JS host); fixed timestamps 1000+f",
            "config": {"workload": f"{args.config}: {cfg['scene']} {cfg['width']}x{cfg['height']}, {cfg['sample_count']} spp/frame, {cfg['max_path_length']} bounces, "
                                   + (f"{K} frames in total split over the GPUs" if strong else "one frame per step per GPU"),
                       "frames_per_gpu": K / n_gpus if strong else K, "frames_all_gpus": int(frames_all),
                       "traversal": {0: "literal (reference tree)", 1: "reference tree + best-t pruning", 2: "leaf-SAH tree", 3: "leaf-SAH tree + best-t pruning"}[3 if args.traversal is None else args.traversal],
                       "prune_rule": {1: "provable rule (prune = 1)", 2: "margin rule with lateral allowance (prune = 2, margin_lat = 3000 ppm, library default): whole BASELINE jobs bit-equal to the unpruned walk, 0 of 1e8 adversarial rays per scene differ; not proven for every ray"}[prune],
                       "pipeline": {0: "one ray per lane", 1: "wave-scheduled, per-lane leaf queues", 2: "wave-scheduled, shared leaf queue"}[pipeline],
                       "light_rays": "any-hit termination under the selected prune rule + blocker cache (library default; same sums and reference ray counters as closest-hit queries in every frame compared)",
                       "parallelism": f"frame-split x{n_gpus} + 1 all-reduce", "allreduce": allreduce_via, "rccl_ranks": ctx.comm_world,
                       "excluded_one_off_ms": {"upload_scene_host_and_copy": upload_ms, "first_call_wall": first_call_ms, "same_call_again_wall": second_call_ms,
                                               "k_primary_kernel": (prof or {}).get("one_off", {}).get("k_primary_ms"),
                                               "note": "once per scene / per camera, not in the timed region: scene upload (tree build + copies), and in the first render call the "
                                                       "primary-hit cache (k_primary), frame-buffer allocation and code-object load"}},
            "roofline": roof,
        }
        if other:
            line["strong_scaling" if not strong else "weak_scaling"] = other
        if exact:
            line["exact_mode"] = exact
        if not args.no_cpu_baseline and n_gpus == 1:
            line["cpu_baseline"] = cpu_baseline(m, scene, params)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
