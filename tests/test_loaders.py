"""Scene ingestion of the JS host (host/gltf.js, obj.js, scene.js), which replaces loadModel() of /root/reference/main.js:536-712
(three's GLTFLoader + the flattening loop) and utils/triangleMapper.js:43-79 -- CPU only, needs node.

Two kinds of test:
  * tiny hand-made glTF / GLB / OBJ files whose triangles are computed here independently (numpy, from the rules
    SURVEY.md 8c-3 records: TRS and `matrix` nodes, multi-primitive meshes as groups, u16 / u32 indices, byteStride, default
    material, data-URI buffer, the recentring rule of main.js:553-576, THREE.Triangle.getNormal, emission x 50);
  * in the build container only (the reference tree does not travel): make_fixtures.js over /root/reference/resources must
    reproduce every committed tests/golden/scenes/*.ptxs byte for byte.
three.js itself is not importable here, so these pin the loaders against their written contract, not against three
("parity unpinned" at that boundary)."""
import base64
import json
import os
import shutil
import struct
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "path-tracing_amd", "host")
pytestmark = pytest.mark.skipif(shutil.which("node") is None, reason="node not installed")

DUMP_JS = """
const path = require('path');
const { loadModel, buildSceneArrays, mapTrianglesArrayToTexturizedArray } = require(path.join(%r, 'scene'));
const opts = JSON.parse(process.argv[3] || '{}');
const { trianglesArray, centre } = loadModel(process.argv[2], opts);
const v = (p) => [p.x, p.y, p.z], c = (p) => [p.r, p.g, p.b];
const tris = trianglesArray.map((t) => ({ a: v(t.triangle.a), b: v(t.triangle.b), c: v(t.triangle.c), n: v(t.normal), color: c(t.color), emission: c(t.emission) }));
const s = buildSceneArrays(trianglesArray);
const m = mapTrianglesArrayToTexturizedArray(trianglesArray);
console.log(JSON.stringify({ tris, centre, T: s.T, N: s.N, L: s.L, lightTri: Array.from(s.lightTri), emissions: Array.from(s.emissions),
                             coords: Array.from(s.coords), refLightIndices: m.lightIndices, refEmissions: m.emissions }));
""" % HOST


def load(tmp_path, scene_file, opts=None):
    js = tmp_path / "dump.js"
    js.write_text(DUMP_JS)
    out = subprocess.check_output(["node", str(js), str(scene_file), json.dumps(opts or {})])
    return json.loads(out.decode())


# ---------------------------------------------------------------------------------------------- the hand-made glTF
QUAD = np.array([[0, 0, 0], [1, 0, 0], [1, 1, 0], [0, 1, 0]], dtype=np.float32)       # interleaved, byteStride 16
QUAD_IDX = np.array([0, 1, 2, 0, 2, 3], dtype=np.uint16)
TRI = np.array([[0, 0, 1], [2, 0, 1], [0, 2, 1]], dtype=np.float32)                    # tightly packed
TRI_IDX = np.array([2, 1, 0], dtype=np.uint32)
LONE = np.array([[5, 5, 5], [6, 5, 5], [5, 6, 5]], dtype=np.float32)                   # non-indexed primitive on its own node
S2 = float(np.sqrt(0.5))


def gltf_files():
    """-> (json dict with one buffer, the buffer's bytes).  Layout: strided quad | u16 indices (+pad) | tri | u32 indices | lone."""
    quad_bytes = b"".join(struct.pack("<3f4x", *p) for p in QUAD)            # 16-byte stride
    idx16 = QUAD_IDX.tobytes() + b"\0\0\0\0"                                 # 12 bytes + pad to 16
    blob = quad_bytes + idx16 + TRI.tobytes() + TRI_IDX.tobytes() + LONE.tobytes()
    o_idx16, o_tri, o_idx32, o_lone = 64, 80, 116, 128
    assert len(blob) == o_lone + 36
    doc = {
        "asset": {"version": "2.0"},
        "scene": 0,
        "scenes": [{"nodes": [0, 2]}],
        "nodes": [
            # TRS node: translate (10, 0, 0), rotate 90 degrees about z, scale 2; its child carries the mesh
            {"name": "parent", "translation": [10, 0, 0], "rotation": [0, 0, S2, S2], "scale": [2, 2, 2], "children": [1]},
            # `matrix` node (column-major): translation (0, 5, 0)
            {"name": "child", "matrix": [1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 5, 0, 1], "mesh": 0},
            {"name": "lone", "mesh": 1, "translation": [0, 0, -3]},
        ],
        "meshes": [
            {"name": "two_prims", "primitives": [
                {"attributes": {"POSITION": 0}, "indices": 1, "material": 0},
                {"attributes": {"POSITION": 2}, "indices": 3}]},                       # no material: white / black
            {"name": "lone_mesh", "primitives": [{"attributes": {"POSITION": 4}, "material": 1}]},
        ],
        "materials": [
            {"name": "glow", "pbrMetallicRoughness": {"baseColorFactor": [0.8, 0.1, 0.2, 1]}, "emissiveFactor": [0.5, 0.25, 0],
             "extensions": {"KHR_materials_emissive_strength": {"emissiveStrength": 7}}},      # ignored by main.js:667
            {"name": "plain", "pbrMetallicRoughness": {"baseColorFactor": [0.3, 0.4, 0.5, 1]}},
        ],
        "accessors": [
            {"bufferView": 0, "componentType": 5126, "count": 4, "type": "VEC3", "min": [0, 0, 0], "max": [1, 1, 0]},
            {"bufferView": 1, "componentType": 5123, "count": 6, "type": "SCALAR"},
            {"bufferView": 2, "componentType": 5126, "count": 3, "type": "VEC3"},                  # no min / max: computed
            {"bufferView": 3, "componentType": 5125, "count": 3, "type": "SCALAR"},
            {"bufferView": 4, "componentType": 5126, "count": 3, "type": "VEC3", "min": [5, 5, 5], "max": [6, 6, 5]},
        ],
        "bufferViews": [
            {"buffer": 0, "byteOffset": 0, "byteLength": 64, "byteStride": 16},
            {"buffer": 0, "byteOffset": o_idx16, "byteLength": 12},
            {"buffer": 0, "byteOffset": o_tri, "byteLength": 36},
            {"buffer": 0, "byteOffset": o_idx32, "byteLength": 12},
            {"buffer": 0, "byteOffset": o_lone, "byteLength": 36},
        ],
        "buffers": [{"byteLength": len(blob)}],
    }
    return doc, blob


def expected_gltf_triangles():
    """Independent numpy restatement of main.js:551-679 for the file above."""
    def trs(t, q, s):
        x, y, z, w = q
        r = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                      [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                      [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
        m = np.eye(4)
        m[:3, :3] = r * np.asarray(s)[None, :]
        m[:3, 3] = t
        return m
    parent = trs([10, 0, 0], [0, 0, S2, S2], [2, 2, 2])
    child = parent @ trs([0, 5, 0], [0, 0, 0, 1], [1, 1, 1])          # the group of the two primitives
    lone = trs([0, 0, -3], [0, 0, 0, 1], [1, 1, 1])
    meshes = [  # (parent world matrix, own local position, vertices in index order, local bbox, colour, emissive)
        (child, np.zeros(3), QUAD[QUAD_IDX].astype(np.float64), (QUAD.min(0), QUAD.max(0)), [0.8, 0.1, 0.2], [0.5, 0.25, 0]),
        (child, np.zeros(3), TRI[TRI_IDX].astype(np.float64), (TRI.min(0), TRI.max(0)), [1, 1, 1], [0, 0, 0]),
        (np.eye(4), np.array([0, 0, -3.0]), LONE.astype(np.float64), (LONE.min(0), LONE.max(0)), [0.3, 0.4, 0.5], [0, 0, 0]),
    ]

    def world(pw, pos):
        m = np.eye(4)
        m[:3, 3] = pos
        return pw @ m
    lo, hi = np.full(3, np.inf), np.full(3, -np.inf)
    for pw, pos, _, (bmin, bmax), _, _ in meshes:                     # Box3.setFromObject, non-precise: 8 corners per mesh
        mw = world(pw, pos)
        for c in range(8):
            p = np.array([bmax[0] if c & 1 else bmin[0], bmax[1] if c & 2 else bmin[1], bmax[2] if c & 4 else bmin[2], 1.0])
            q = (mw @ p)[:3]
            lo, hi = np.minimum(lo, q), np.maximum(hi, q)
    centre = (lo + hi) / 2
    tris = []
    for pw, pos, verts, _, col, em in meshes:
        mw = world(pw, pos - centre)                                   # main.js:563-568: added to the mesh's LOCAL position
        w = (mw @ np.c_[verts, np.ones(len(verts))].T).T[:, :3]
        for k in range(0, len(w), 3):
            a, b, c = w[k], w[k + 1], w[k + 2]
            n = np.cross(c - b, a - b)
            n = n / np.linalg.norm(n)
            tris.append(dict(a=a, b=b, c=c, n=n, color=col, emission=em))
    return tris, centre


def check_gltf(got):
    want, centre = expected_gltf_triangles()
    assert np.allclose(got["centre"], centre, atol=1e-12)
    assert got["T"] == len(want) == 4                                   # quad = 2, tri = 1, lone = 1; depth-first pre-order
    for g, w in zip(got["tris"], want):
        for k in ("a", "b", "c", "n", "color", "emission"):
            assert np.allclose(g[k], w[k], atol=1e-9), (k, g[k], w[k])
    # the parent's rotation is applied to the recentring shift (a world translation would give another answer)
    assert not np.allclose(got["tris"][0]["a"], np.array([10 - 2 * 5, 0, 0]) - centre, atol=1e-3)
    # utils/triangleMapper.js:43-79: emission x 50, lights = triangles with any emissive channel > 0 (strength extension ignored)
    assert got["L"] == 2 and got["lightTri"] == [0, 1]
    assert got["refLightIndices"] == [0, 0, 0, 1, 1, 1]                 # the reference triplicates every index
    assert np.allclose(got["emissions"][:6], [25, 12.5, 0, 25, 12.5, 0]) and not any(got["emissions"][6:])
    assert np.allclose(got["coords"][:3], np.float32(want[0]["a"]))     # f64 -> f32 once, at the boundary


def test_gltf_trs_matrix_stride_indices_multiprimitive_default_material(tmp_path):
    doc, blob = gltf_files()
    doc["buffers"][0]["uri"] = "scene.bin"
    (tmp_path / "scene.bin").write_bytes(blob)
    (tmp_path / "scene.gltf").write_text(json.dumps(doc))
    check_gltf(load(tmp_path, tmp_path / "scene.gltf"))


def test_gltf_data_uri_buffer(tmp_path):
    doc, blob = gltf_files()
    doc["buffers"][0]["uri"] = "data:application/octet-stream;base64," + base64.b64encode(blob).decode()
    (tmp_path / "inline.gltf").write_text(json.dumps(doc))
    check_gltf(load(tmp_path, tmp_path / "inline.gltf"))


def test_glb_container(tmp_path):
    doc, blob = gltf_files()
    js = json.dumps(doc).encode()
    js += b" " * (-len(js) % 4)
    body = blob + b"\0" * (-len(blob) % 4)
    glb = struct.pack("<III", 0x46546C67, 2, 12 + 8 + len(js) + 8 + len(body)) + struct.pack("<II", len(js), 0x4E4F534A) + js + \
        struct.pack("<II", len(body), 0x004E4942) + body
    (tmp_path / "scene.glb").write_bytes(glb)
    check_gltf(load(tmp_path, tmp_path / "scene.glb"))


def test_gltf_material_override_and_invisible_strip_modes(tmp_path):
    doc, blob = gltf_files()
    doc["buffers"][0]["uri"] = "scene.bin"
    # the quad as a TRIANGLE_STRIP over vertices 0, 1, 3, 2 (mode 5) -> (0,1,3), (2,3,1) with the odd triangle flipped
    doc["meshes"][0]["primitives"][0]["mode"] = 5
    strip = np.array([0, 1, 3, 2, 0, 0], dtype=np.uint16)
    blob = blob[:64] + strip.tobytes() + blob[76:]
    doc["accessors"][1]["count"] = 4
    (tmp_path / "scene.bin").write_bytes(blob)
    (tmp_path / "scene.gltf").write_text(json.dumps(doc))
    got = load(tmp_path, tmp_path / "scene.gltf", {"materialOverrides": {"plain": {"emissive": [1, 1, 1]}}})
    assert got["T"] == 4 and got["lightTri"] == [0, 1, 3]               # "plain" became an emitter (the C1 override mechanism)
    want, _ = expected_gltf_triangles()
    quad_world = {tuple(np.round(w[k], 9)) for w in want[:2] for k in ("a", "b", "c")}
    assert {tuple(np.round(g[k], 9)) for g in got["tris"][:2] for k in ("a", "b", "c")} == quad_world
    # both strip triangles keep the quad's orientation (normal of the list form)
    assert np.allclose(got["tris"][0]["n"], want[0]["n"], atol=1e-9) and np.allclose(got["tris"][1]["n"], want[0]["n"], atol=1e-9)


# ---------------------------------------------------------------------------------------------- OBJ / MTL
OBJ = """# quad + triangle + pentagon
mtllib mats.mtl
o first
v 0 0 0
v 2 0 0
v 2 2 0
v 0 2 0
v 0 0 4
v 1 0 4
v 0 1 4
usemtl red
f 1 2 3 4
usemtl lamp
f 5/1/1 6/2/1 7/3/1
o second
v 10 0 0
v 11 0 0
v 12 1 0
v 11 2 0
v 10 2 0
f -5 -4 -3 -2 -1
"""
MTL = """newmtl red
Kd 0.9 0.1 0.1
Ke 0 0 0
newmtl lamp
Kd 1 1 1
Ke 0.2 0.4 0.6
"""


def test_obj_fan_triangulation_materials_negative_indices(tmp_path):
    (tmp_path / "model.obj").write_text(OBJ)
    (tmp_path / "mats.mtl").write_text(MTL)
    got = load(tmp_path, tmp_path / "model.obj")
    v = np.array([[0, 0, 0], [2, 0, 0], [2, 2, 0], [0, 2, 0], [0, 0, 4], [1, 0, 4], [0, 1, 4],
                  [10, 0, 0], [11, 0, 0], [12, 1, 0], [11, 2, 0], [10, 2, 0]], dtype=np.float64)
    centre = (v.min(0) + v.max(0)) / 2                                  # no node transforms: plain world recentring
    assert np.allclose(got["centre"], centre)
    faces = [([0, 1, 2], "red"), ([0, 2, 3], "red"), ([4, 5, 6], "lamp"),
             ([7, 8, 9], "lamp"), ([7, 9, 10], "lamp"), ([7, 10, 11], "lamp")]          # fans (v0, vi, vi+1); usemtl persists across `o`
    mats = {"red": ([0.9, 0.1, 0.1], [0, 0, 0]), "lamp": ([1, 1, 1], [0.2, 0.4, 0.6])}
    assert got["T"] == len(faces)
    for g, (idx, mat) in zip(got["tris"], faces):
        a, b, c = v[idx] - centre
        assert np.allclose(g["a"], a) and np.allclose(g["b"], b) and np.allclose(g["c"], c)
        n = np.cross(c - b, a - b)
        assert np.allclose(g["n"], n / np.linalg.norm(n))
        assert np.allclose(g["color"], mats[mat][0]) and np.allclose(g["emission"], mats[mat][1])
    assert got["lightTri"] == [2, 3, 4, 5]
    assert np.allclose(got["emissions"][6:9], [10, 20, 30])


def test_degenerate_triangle_gets_zero_normal(tmp_path):
    (tmp_path / "d.obj").write_text("v 0 0 0\nv 1 1 1\nv 2 2 2\nv 0 1 0\nf 1 2 3\nf 1 2 4\n")
    got = load(tmp_path, tmp_path / "d.obj")
    assert got["tris"][0]["n"] == [0, 0, 0] and abs(np.linalg.norm(got["tris"][1]["n"]) - 1) < 1e-12


# ---------------------------------------------------------------------------------------------- the reference's own scene files
@pytest.mark.skipif(not os.path.isdir("/root/reference/resources"), reason="build container only: the reference tree does not travel")
def test_make_fixtures_reproduces_the_committed_scene_fixtures(tmp_path):
    out = tmp_path / "scenes"
    subprocess.check_call(["node", os.path.join(HOST, "tools", "make_fixtures.js"), "/root/reference/resources", str(out)], stdout=subprocess.DEVNULL)
    golden = os.path.join(ROOT, "tests", "golden", "scenes")
    names = sorted(f for f in os.listdir(golden) if f.endswith(".ptxs"))
    assert len(names) == 6
    for f in names:
        assert (out / f).read_bytes() == open(os.path.join(golden, f), "rb").read(), f
        side_new, side_old = json.loads((out / f.replace(".ptxs", ".json")).read_text()), json.load(open(os.path.join(golden, f.replace(".ptxs", ".json"))))
        assert side_new["camera"] == side_old["camera"] and (side_new["T"], side_new["N"], side_new["L"]) == (side_old["T"], side_old["N"], side_old["L"])


@pytest.mark.skipif(not os.path.isdir("/root/reference/resources"), reason="build container only: the reference tree does not travel")
def test_every_reference_scene_file_loads_with_the_surveyed_shape(tmp_path):
    """T / N / L of the scene files the reference ships (main.js:79-89)."""
    want = {"cornell/glTF/scene.gltf": (32, 15, 0), "my_cornell_6/gltf/my_cornell_6.gltf": (122, 63, 2), "bedroom2/gltf/bedroom2.gltf": (2314, 1023, 18),
            "bedroom2/gltf/v5/bedroom2_v5.gltf": (2308, 1023, 2), "pixar_room/v1/pixar-room.gltf": (18154, 8191, 19), "teapot/teapot.obj": (15704, 8191, 0)}
    for rel, shape in want.items():
        got = load(tmp_path, os.path.join("/root/reference/resources", rel))
        assert (got["T"], got["N"], got["L"]) == shape, rel
