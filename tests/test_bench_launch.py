"""bench.py --gpus N without a launcher must start N ranks itself (round-2 review: the flag was parsed and ignored, so the
driver's N > 1 runs would have printed n_gpus 1).  CPU: the launch path with --dry-run (the ranks meet over gloo, no GPU
call anywhere).  GPU: the real N = 2 bench on ONE device (PTX_BENCH_ONE_DEVICE=1, gloo: RCCL cannot put two ranks on one
GPU), a few small steps."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*args, env=None, timeout=600):
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout                   # ONE JSON line on stdout, nothing else
    return json.loads(lines[0])


@pytest.mark.parametrize("n", [1, 2, 3])
def test_gpus_flag_starts_that_many_ranks(n):
    line = run_bench("--gpus", str(n), "--dry-run")
    assert line["n_gpus"] == n and line["ranks_seen"] == n


def test_under_a_launcher_the_process_is_one_rank():
    # the driver's form: torch.distributed.run sets WORLD_SIZE; bench.py must not spawn again
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(29500 + (os.getpid() + 31) % 2000), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [json.loads(ln) for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and lines[0]["n_gpus"] == 2 and lines[0]["ranks_seen"] == 2


@pytest.mark.gpu
def test_bench_two_ranks_on_one_device(ptx):
    line = run_bench("--gpus", "2", "--backend", "gloo", "--steps", "2", "--warmup", "1", "--config", "C2", "--no-cpu-baseline",
                     env={"PTX_BENCH_ONE_DEVICE": "1"})
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["steps"] == 2
    assert line["config"]["frames_all_gpus"] == 4                      # 2 frames per GPU x 2 ranks
    px = 1024 * 1024 * 16
    assert abs(line["value"] * 1e6 * line["ms_per_step"] * 2 / 1e3 - 4 * px) < 1e-3 * 4 * px      # value = all ranks' samples / max time
    st = line["strong_scaling"]
    assert st["scaling"] == "strong" and st["frames_total"] == 16 and st["frames_per_gpu"] == 8 and st["value"] > 0


@pytest.mark.gpu
def test_bench_strong_scaling_mode_one_gpu(ptx):
    line = run_bench("--scaling", "strong", "--steps", "4", "--config", "C2", "--no-cpu-baseline", "--no-extras")
    assert line["n_gpus"] == 1 and line["scaling"] == "strong" and line["config"]["frames_all_gpus"] == 4
