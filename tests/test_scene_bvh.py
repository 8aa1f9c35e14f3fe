"""Host-side scene fixtures + BVH builder -- no GPU."""
import json
import os
import shutil
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "path-tracing_amd", "host")
SCENES = ["cornell", "cornell_faithful", "my_cornell_6", "bedroom2_v5", "teapot", "pixar_room"]

# SURVEY.md 8a "Verified BVH shapes": recorded from the REFERENCE builder (classes/bvh.js under Node)
SURVEY_SHAPES = {"cornell": (32, 15, {4: 8}), "my_cornell_6": (122, 63, {4: 26, 3: 6}), "bedroom2_v5": (2308, 1023, None),
                 "pixar_room": (18154, 8191, {4: 2326, 5: 1770})}
SURVEY_CORNELL = {0: (-1, -2, 0), 1: (8, -2, 0), 2: (5, -2, 0), 3: (4, 0, 4), 4: (5, 4, 4), 14: (-1, 28, 4)}  # node: (miss, first, count)


@pytest.mark.parametrize("name", SCENES)
def test_fixture_invariants(ptx, name):
    s = ptx.load_scene(name)
    N, T = s.N, s.T
    assert sorted(s.inorder_tri.tolist()) == list(range(T))           # every triangle in exactly one leaf slot
    leaves = s.node_first != -2
    assert (s.node_count[~leaves] == 0).all() and (s.node_count[leaves] >= 1).all() and (s.node_count <= 5).all()
    # leaf ranges tile the inorder array in pre-order
    pos = 0
    for i in np.nonzero(leaves)[0]:
        assert s.node_first[i] == pos
        pos += s.node_count[i]
    assert pos == T
    # miss link = pre-order index after the subtree: recompute subtree ends from the structure
    end = np.zeros(N, dtype=np.int64)

    def walk(i):
        if s.node_first[i] != -2:
            end[i] = i + 1
            return i + 1
        r = walk(i + 1)
        e = walk(r)
        end[i] = e
        return e
    import sys
    sys.setrecursionlimit(10000)
    assert walk(0) == N
    want = np.where(end < N, end, -1)
    assert (s.node_miss == want).all()
    # boxes: a leaf's box is the f32 rounding of the union of its triangles' corners; parents contain children
    c = s.coords.reshape(T, 3, 3)
    for i in np.nonzero(leaves)[0][:200]:
        tris = s.inorder_tri[s.node_first[i]:s.node_first[i] + s.node_count[i]]
        assert (s.node_min[3 * i:3 * i + 3] <= c[tris].min(axis=(0, 1)) + 1e-6).all()
        assert (s.node_max[3 * i:3 * i + 3] >= c[tris].max(axis=(0, 1)) - 1e-6).all()
    # light list = triangles with any emission > 0, ascending (utils/triangleMapper.js:62-64)
    em = s.emissions.reshape(T, 3)
    assert s.light_tri.tolist() == np.nonzero((em > 0).any(axis=1))[0].tolist()


@pytest.mark.parametrize("name", sorted(SURVEY_SHAPES))
def test_shapes_recorded_from_reference_builder(ptx, name):
    s = ptx.load_scene(name)
    T, N, hist = SURVEY_SHAPES[name]
    assert (s.T, s.N) == (T, N)
    if hist:
        got = {int(k): int(v) for k, v in zip(*np.unique(s.node_count[s.node_count > 0], return_counts=True))}
        assert got == hist
    if name == "cornell":
        for i, (miss, first, count) in SURVEY_CORNELL.items():
            assert (s.node_miss[i], s.node_first[i], s.node_count[i]) == (miss, first, count)


def test_pixar_room_lights_are_the_lamp_bulb(ptx):
    s = ptx.load_scene("pixar_room")
    assert s.light_tri.tolist() == list(range(17869, 17888))          # LampBulb, 19 tris
    assert (s.emissions.reshape(-1, 3)[17869] == 50).all()            # emissive 1 x 50 (strength ignored)


@pytest.mark.parametrize("name", ["cornell", "my_cornell_6"])
def test_js_builder_equals_python_restatement_f64(ptx, name):
    """The product's JS builder (arrays in the fixture) vs oracle/bvh_oracle.py on the same f64 triangles."""
    from oracle.bvh_oracle import build_bvh
    s = ptx.load_scene(name)
    tris = json.load(open(os.path.join(ROOT, "tests", "golden", "scenes", name + ".json")))["triangles_f64"]
    b = build_bvh(tris)
    assert b["node_count"] == s.N
    assert b["node_miss"] == s.node_miss.tolist() and b["node_first"] == s.node_first.tolist()
    assert b["node_cnt"] == s.node_count.tolist() and b["inorder_tri"] == s.inorder_tri.tolist()
    assert (np.asarray(b["node_min"], dtype=np.float32).ravel().view(np.uint32) == s.node_min.view(np.uint32)).all()
    assert (np.asarray(b["node_max"], dtype=np.float32).ravel().view(np.uint32) == s.node_max.view(np.uint32)).all()


@pytest.mark.skipif(shutil.which("node") is None, reason="node not installed")
@pytest.mark.parametrize("name", ["bedroom2_v5", "pixar_room"])
def test_js_builder_equals_python_restatement_large(ptx, name, tmp_path):
    """Rebuild from the fixture's f32 coordinates with host/bvh.js (node) and with the Python restatement."""
    from oracle.bvh_oracle import build_bvh
    src = os.path.join(ROOT, "tests", "golden", "scenes", name + ".ptxs")
    out = str(tmp_path / "rebuilt.ptxs")
    subprocess.check_call(["node", os.path.join(HOST, "tools", "rebuild_bvh.js"), src, out])
    js = ptx.load_scene(out)
    s = ptx.load_scene(name)
    b = build_bvh(s.coords.astype(np.float64).reshape(-1, 9).tolist())
    assert b["node_miss"] == js.node_miss.tolist() and b["node_first"] == js.node_first.tolist()
    assert b["node_cnt"] == js.node_count.tolist() and b["inorder_tri"] == js.inorder_tri.tolist()
    assert (np.asarray(b["node_min"], dtype=np.float32).ravel().view(np.uint32) == js.node_min.view(np.uint32)).all()
    assert (np.asarray(b["node_max"], dtype=np.float32).ravel().view(np.uint32) == js.node_max.view(np.uint32)).all()


@pytest.mark.skipif(shutil.which("node") is None, reason="node not installed")
def test_js_camera_presets(ptx):
    js = """
    const {cameraFromPreset} = require('%s/camera');
    const out = {};
    for (const n of ['room-v3','cornell','pixar-room','teapot']) { const c = cameraFromPreset(n, 16, 9).getCamera();
      out[n] = ['cameraSource','cameraDirection','cameraUp','cameraRight','cameraLeftBottom'].map(k => [c[k].x, c[k].y, c[k].z]); }
    console.log(JSON.stringify(out));
    """ % HOST
    out = json.loads(subprocess.check_output(["node", "-e", js]))
    for name, (src, d, up, right, lb) in out.items():
        src, d, up, right, lb = map(np.asarray, (src, d, up, right, lb))
        assert abs(np.linalg.norm(d) - 1) < 1e-12 and abs(np.linalg.norm(right) - 1) < 1e-12
        assert np.allclose(up, [0, 1, 0])                             # world up, never re-orthogonalised (camera.js:69)
        assert np.allclose(right, np.cross(d, up) / np.linalg.norm(np.cross(d, up)))
        assert np.allclose(lb, src + d - 0.5 * right - 0.5 * up)      # camera.js:74-81
    # pixar room preset (main.js:157-161): eye (2,2,4) looking at (0,-2,-4)
    src, d = np.asarray(out["pixar-room"][0]), np.asarray(out["pixar-room"][1])
    assert np.allclose(src, [2, 2, 4]) and np.allclose(d, np.array([-2, -4, -8]) / np.linalg.norm([-2, -4, -8]))
    # cornell preset (main.js:152-154): x = 12.4, rotated +pi/2 about y -> looks down -x
    assert np.allclose(out["cornell"][1], [-1, 0, 0], atol=1e-9)
    # fixture side-car agrees with the live JS
    cam = ptx.load_camera("pixar_room")
    assert np.allclose(cam["cam_left_bottom"], out["pixar-room"][4])
