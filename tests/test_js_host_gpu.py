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
    # the traversal switches of the CLI: provable pruning rule / the reference's literal traversal give the same sum
    for extra in (["--prune", "1"], ["--traversal", "0"]):
        subprocess.check_call(["node", os.path.join(ROOT, "path-tracing_amd", "host", "main.js"), "--scene", os.path.join(scenes, name + ".ptxs"),
                               "--camera-json", os.path.join(scenes, name + ".json"), "--width", str(w), "--height", str(h), "--frames", str(nfr),
                               "--spp", str(spp), "--bounces", str(b), "--dump-sum", out] + extra)
        assert (np.fromfile(out, dtype=np.float32).reshape(h, w, 3).view(np.uint32) == want.view(np.uint32)).all(), extra


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
    # the image the reference saves (progressive 8-bit blend, display pass, row flip) under SwiftShader's GL rules: equal to
    # the oracle's blend + display chain, which tests/test_reference_frames.py pins byte for byte to the reference shaders
    ppm = str(tmp_path / "compat.ppm")
    subprocess.check_call(base + ["--frames", "3", "--compat-out", ppm, "--compat-rules", "2", "--compat-gamma", "1"])
    raw = open(ppm, "rb").read()
    head = ("P6\n%d %d\n255\n" % (w, h)).encode()
    assert raw.startswith(head) and len(raw) == len(head) + w * h * 3
    got8 = np.frombuffer(raw[len(head):], dtype=np.uint8).reshape(h, w, 3)
    prev = None
    for f, ts in enumerate(ptx.scene.timestamps(3)):
        prev = orc.blend_rgba8(p, orc.render_frame(s, p, int(ts))[0], prev, f, store=1)
    assert (got8 == orc.display_rgba8(prev, store=1)[::-1, :, :3]).all()


@pytest.mark.gpu
@pytest.mark.skipif(shutil.which("node") is None, reason="node not installed")
def test_node_async_render_matches_sync(tmp_path):
    """renderFramesAsync (libuv worker, Promise) leaves the event loop free and gives the same bytes as renderFrames."""
    js = """
const path = require('path');
const root = %r;
const { PathTracer } = require(path.join(root, 'path-tracing_amd/host/ptx'));
const { readScene } = require(path.join(root, 'path-tracing_amd/host/scenefile'));
const fs = require('fs');
const scene = readScene(path.join(root, 'tests/golden/scenes/my_cornell_6.ptxs'));
const c = JSON.parse(fs.readFileSync(path.join(root, 'tests/golden/scenes/my_cornell_6.json'), 'utf8')).camera;
const cam = { cameraSource: c.cam_src, cameraUp: c.cam_up, cameraRight: c.cam_right, cameraLeftBottom: c.cam_left_bottom };
(async () => {
  const pt = new PathTracer(0);
  pt.uploadScene(scene); pt.setParams(64, 48, cam, 4, 8);
  pt.renderFrames([1000, 1001, 1002], 0);
  const a = pt.readSum();
  pt.resetAccum();
  let ticks = 0;
  const timer = setInterval(() => { ticks++; }, 1);
  const pending = pt.renderFramesAsync([1000, 1001, 1002], 0);
  let guarded = false, second = false;
  try { pt.readSum(); } catch (e) { guarded = /while renderFramesAsync is running/.test(e.message); }   // one context, one thread
  try { await pt.renderFramesAsync([1], 0); } catch (e) { second = /still running/.test(e.message); }
  // past the JS wrapper, straight at the addon: the handle itself is marked busy and held alive by the job
  let addonBusy = 0;
  try { pt.addon.readSum(pt.handle); } catch (e) { addonBusy += /busy/.test(e.message) ? 1 : 0; }
  try { pt.addon.destroy(pt.handle); } catch (e) { addonBusy += /destroy.. while/.test(e.message) ? 1 : 0; }
  await pending;
  clearInterval(timer);
  const b = pt.readSum();
  let same = a.length === b.length;
  for (let i = 0; same && i < a.length; i++) same = Object.is(a[i], b[i]);
  let rejected = false;
  try { await pt.addon.renderFramesAsync(pt.handle, new Float32Array(2), 0); } catch (e) { rejected = true; }
  console.log(JSON.stringify({ same, frames: pt.framesDone, rejected, guarded, second, addonBusy, len: b.length }));
  pt.destroy();
})().catch((e) => { console.error(e); process.exit(1); });
""" % ROOT
    f = tmp_path / "async.js"
    f.write_text(js)
    import json
    out = json.loads(subprocess.check_output(["node", str(f)]).decode().strip().splitlines()[-1])
    assert out == {"same": True, "frames": 3, "rejected": True, "guarded": True, "second": True, "addonBusy": 2, "len": 64 * 48 * 3}


@pytest.mark.gpu
@pytest.mark.skipif(shutil.which("node") is None, reason="node not installed")
def test_node_cli_multi_device_equals_python_multi_device(ptx, tmp_path):
    """`--devices 0,0,0`: the JS host drives one PathTracer over several devices (here one GPU listed three times: the
    rehearsal form, summed by a kernel) -- same bytes as the ctypes path over the same device list, frames_done summed."""
    name, w, h, spp, b, nfr = "my_cornell_6", 40, 24, 3, 4, 7
    out = str(tmp_path / "sum.f32")
    scenes = os.path.join(ROOT, "tests", "golden", "scenes")
    log = subprocess.check_output(["node", os.path.join(ROOT, "path-tracing_amd", "host", "main.js"), "--scene", os.path.join(scenes, name + ".ptxs"),
                                   "--camera-json", os.path.join(scenes, name + ".json"), "--width", str(w), "--height", str(h), "--frames", str(nfr),
                                   "--spp", str(spp), "--bounces", str(b), "--dump-sum", out, "--devices", "0,0,0"]).decode()
    assert "all-reduce over 3 ranks" in log
    got = np.fromfile(out, dtype=np.float32).reshape(h, w, 3)
    with ptx.Context([0, 0, 0]) as ctx:
        ctx.upload_scene(ptx.load_scene(name))
        ctx.set_params(ptx.config_params(name, w, h, spp, b))
        ctx.render_frames(ptx.scene.timestamps(nfr))
        want = ctx.read_sum()
    assert (got.view(np.uint32) == want.view(np.uint32)).all()


@pytest.mark.skipif(shutil.which("node") is None, reason="node not installed")
def test_node_addon_loads_and_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    js = "const {PathTracer}=require('%s/path-tracing_amd/host/ptx'); try{new PathTracer(0); console.log('NOERR')}catch(e){console.log('ERR '+e.message)}" % ROOT
    out = subprocess.check_output(["node", "-e", js]).decode()
    assert out.startswith("ERR ptx status -2")
