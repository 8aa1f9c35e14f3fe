"""The JS host through the raw N-API addon (node main.js) must produce the same bytes as the ctypes path."""
import os
import shutil
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.skipif(shutil.which("node") is None, reason="node not installed")
def test_node_cli_matches_ctypes_and_oracle(ptx, orc, gpu_ctx, tmp_path):
    name, w, h, spp, b, nfr = "my_cornell_6", 40, 24, 3, 4, 3
    out = str(tmp_path / "sum.f32")
    png = str(tmp_path / "img.png")
    scenes = os.path.join(ROOT, "tests", "golden", "scenes")
    subprocess.check_call(["node", os.path.join(ROOT, "path-tracing_amd", "host", "main.js"), "--scene", os.path.join(scenes, name + ".ptxs"),
                           "--camera-json", os.path.join(scenes, name + ".json"), "--width", str(w), "--height", str(h), "--frames", str(nfr),
                           "--spp", str(spp), "--bounces", str(b), "--dump-sum", out, "--out", png])
    got = np.fromfile(out, dtype=np.float32).reshape(h, w, 3)
    s = ptx.load_scene(name)
    p = ptx.config_params(name, w, h, spp, b)
    want, _ = orc.render_frames(s, p, ptx.scene.timestamps(nfr))
    assert (got.view(np.uint32) == want.view(np.uint32)).all()
    assert open(png, "rb").read(8) == b"\x89PNG\r\n\x1a\n"


@pytest.mark.gpu
@pytest.mark.skipif(shutil.which("node") is None, reason="node not installed")
def test_node_cli_checkpoint_resume_and_compat_output(ptx, orc, tmp_path):
    name, w, h, spp, b = "my_cornell_6", 32, 20, 2, 3
    scenes = os.path.join(ROOT, "tests", "golden", "scenes")
    base = ["node", os.path.join(ROOT, "path-tracing_amd", "host", "main.js"), "--scene", os.path.join(scenes, name + ".ptxs"),
            "--camera-json", os.path.join(scenes, name + ".json"), "--width", str(w), "--height", str(h), "--spp", str(spp), "--bounces", str(b)]
    ck, out = str(tmp_path / "run.ptxc"), str(tmp_path / "sum.f32")
    subprocess.check_call(base + ["--frames", "3", "--checkpoint", ck, "--checkpoint-every", "2"])          # "interrupted" after 3 frames
    assert os.path.getsize(ck) == 20 + w * h * 12
    log = subprocess.check_output(base + ["--frames", "7", "--checkpoint", ck, "--checkpoint-every", "2", "--dump-sum", out]).decode()
    assert "resumed" in log and "at frame 3" in log
    got = np.fromfile(out, dtype=np.float32).reshape(h, w, 3)
    s = ptx.load_scene(name)
    p = ptx.config_params(name, w, h, spp, b)
    want, _ = orc.render_frames(s, p, ptx.scene.timestamps(7))
    assert (got.view(np.uint32) == want.view(np.uint32)).all()
    png = str(tmp_path / "compat.png")
    subprocess.check_call(base + ["--frames", "3", "--compat-out", png])
    assert open(png, "rb").read(8) == b"\x89PNG\r\n\x1a\n"


@pytest.mark.skipif(shutil.which("node") is None, reason="node not installed")
def test_node_addon_loads_and_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    js = "const {PathTracer}=require('%s/path-tracing_amd/host/ptx'); try{new PathTracer(0); console.log('NOERR')}catch(e){console.log('ERR '+e.message)}" % ROOT
    out = subprocess.check_output(["node", "-e", js]).decode()
    assert out.startswith("ERR ptx status -2")
