"""Generates tests/golden/ref_frames/*.npz: frames rendered by the REFERENCE shader itself, verbatim, on the
SwiftShader WebGL2 of kaleido's headless Chromium (build container only; TEST INFRASTRUCTURE).

    python oracle/gen_ref_frames.py [case ...]

The shader text is read from /root/reference/shaders at run time and sent in the request; it is never written
into this repository.  Each fixture holds the RGBA8 frames (bottom-up rows, as gl.readPixels returns them), the
uniform block, the scene fixture name and the draw times.  frameNumber is 0 for every frame, so each frame is the
un-blended per-frame image clamp(mean of sample_count paths) quantised by the RGBA8 target.
"""
import base64
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ptx_import  # noqa: E402

REF = "/root/reference/shaders"
OUT = os.path.join(ROOT, "tests", "golden", "ref_frames")

# name: (scene fixture, W, H, sample_count, max_path_length, timestamps)
CASES = {
    "my_cornell_6_b1": ("my_cornell_6", 48, 48, 8, 1, [1000, 1001]),
    "my_cornell_6_b3": ("my_cornell_6", 48, 48, 8, 3, [1000, 1001]),
    "cornell_b1": ("cornell", 40, 40, 8, 1, [1000]),
    "cornell_c1": ("cornell", 256, 256, 16, 4, [1000]),          # BASELINE.json config 1 exactly
    "bedroom2_v5_b1": ("bedroom2_v5", 48, 27, 4, 1, [1000]),
    "pixar_room_b1": ("pixar_room", 48, 27, 4, 1, [1000]),
    "pixar_room_b8": ("pixar_room", 96, 54, 16, 8, [1000]),      # reduced headline (BASELINE.md plan 1)
    "my_cornell_6_tall_b1": ("my_cornell_6", 24, 40, 4, 1, [1000]),
    # progressive: frameNumber = f, frames blended through the two RGBA8 targets (glsl:532-552); wide and tall images
    # exercise the aspect-shifted read of the previous target
    "my_cornell_6_prog_b1": ("my_cornell_6", 48, 48, 4, 1, [1000, 1001, 1002, 1003, 1004]),
    "pixar_room_prog_b1": ("pixar_room", 48, 27, 4, 1, [1000, 1001, 1002, 1003]),
    "my_cornell_6_tall_prog_b1": ("my_cornell_6", 24, 40, 4, 1, [1000, 1001, 1002, 1003]),
    # deeper paths on every scene (sin/cos, NaN slabs, Russian roulette), wide and tall, single and progressive
    "teapot_b8": ("teapot", 64, 36, 8, 8, [1000]),
    "bedroom2_v5_b4": ("bedroom2_v5", 64, 36, 8, 4, [1000, 1001]),
    "my_cornell_6_wide_b8": ("my_cornell_6", 64, 36, 8, 8, [1000]),
    "my_cornell_6_tall_b12": ("my_cornell_6", 24, 40, 4, 12, [1001]),
    "pixar_room_prog_b4": ("pixar_room", 48, 27, 4, 4, [1000, 1001, 1002]),
    "cornell_prog_b4": ("cornell", 40, 40, 4, 4, [2000, 2001, 2002]),
    # "_disp_": the reference's display pass (fragmentShaderOutput.glsl: pow(c, 1/2.2)) drawn into the canvas after every
    # frame as main.js:357-373 does; the canvas pixels are stored as `display` next to the 8-bit targets
    "my_cornell_6_disp_prog_b4": ("my_cornell_6", 48, 48, 4, 4, [1000, 1001, 1002]),
    "pixar_room_disp_b8": ("pixar_room", 64, 36, 8, 8, [1000]),
    "bedroom2_v5_disp_prog_b1": ("bedroom2_v5", 24, 40, 4, 1, [1000, 1001]),
}


# name: (scene fixture, W, H): primary-ray words (hit triangle id, t bits, direction bits) from a DERIVED debug shader
PRIM_CASES = {
    "my_cornell_6_prim": ("my_cornell_6", 64, 36),
    "my_cornell_6_tall_prim": ("my_cornell_6", 24, 40),
    "cornell_prim": ("cornell", 40, 40),
    "bedroom2_v5_prim": ("bedroom2_v5", 48, 27),
    "teapot_prim": ("teapot", 48, 27),
    "pixar_room_prim": ("pixar_room", 64, 36),
}
PRIM_OUT = os.path.join(ROOT, "tests", "golden", "ref_prim")
# name: (scene fixture, W, H, sample_count, timestamp): float bits of the frame's clamp(mean radiance) (glsl:516-529) at
# maxPathLength = 1, where no result depends on sin/cos: the whole shading arithmetic, bit for bit
RAD_CASES = {
    "my_cornell_6_rad_b1": ("my_cornell_6", 48, 48, 8, 1000),
    "my_cornell_6_tall_rad_b1": ("my_cornell_6", 24, 40, 4, 1001),
    "bedroom2_v5_rad_b1": ("bedroom2_v5", 48, 27, 4, 1000),
    "pixar_room_rad_b1": ("pixar_room", 48, 27, 4, 1000),
}


def derive_debug_shader(text):
    """The reference fragment shader with three additions: a global that remembers the inorder triangle
    index of the current best hit inside ray_bvh_intersection_hit_miss, a uniform debugWord, and a tail of main() that
    traces the primary ray once more and writes one 32-bit word (triangle id / t bits / direction bits) as four bytes.
    Everything the words depend on -- ray generation, traversal, both intersection routines -- is the reference's text.
    Built in memory from /root/reference at run time; never stored."""
    def sub_once(t, old, new):
        assert t.count(old) == 1, (t.count(old), old)
        return t.replace(old, new)
    text = sub_once(text, "bool ray_bvh_intersection_hit_miss(out float out_t, out Triangle out_triangle, vec3 origin, vec3 direction) {",
                    "int g_hitIndex = -1;\nuniform int debugWord;\nbool ray_bvh_intersection_hit_miss(out float out_t, out Triangle out_triangle, vec3 origin, vec3 direction) {")
    text = sub_once(text, "            // outColor = vec4(0.0f, 1.0f, 0.0f, 1.0f);\n            out_t = t;\n            out_triangle = triangle;",
                    "            out_t = t;\n            out_triangle = triangle;\n            g_hitIndex = triangleIndex;")
    tail = ("  {\n    float dbg_t; Triangle dbg_tri; g_hitIndex = -1;\n"
            "    ray_bvh_intersection_hit_miss(dbg_t, dbg_tri, cameraSource, ray_direction);\n"
            "    uint w = debugWord == 0 ? uint(g_hitIndex) : (debugWord == 1 ? floatBitsToUint(dbg_t) : (debugWord < 5 ? floatBitsToUint(ray_direction[debugWord - 2]) : floatBitsToUint(currentColor[debugWord - 5])));\n"
            "    outColor = vec4(float(w & 255u), float((w >> 8) & 255u), float((w >> 16) & 255u), float(w >> 24)) / 255.0f;\n  }\n")
    head, sep, rest = text.rpartition("  outColor = vec4(blendedColor.rgb, 1.0f);\n")
    assert sep, "main()'s final store not found"
    return head + sep + tail + rest


def run_prim_case(name):
    m = ptx_import.load()
    rad = name in RAD_CASES
    scene_name, w, h = (RAD_CASES[name] if rad else PRIM_CASES[name])[:3]
    spp, ts0 = (RAD_CASES[name][3:] if rad else (1, 1000))
    s = m.load_scene(scene_name)
    cam = m.load_camera(scene_name)
    f32 = lambda v: [float(np.float32(x)) for x in v]  # noqa: E731
    payload = {
        "width": w, "height": h, "sample_count": spp, "max_path_length": 1 if rad else 0, "timestamps": [ts0], "debug_words": [5, 6, 7] if rad else [0, 1, 2, 3, 4],
        "cam_src": f32(cam["cam_src"]), "cam_dir": f32(cam["cam_dir"]), "cam_up": f32(cam["cam_up"]),
        "cam_right": f32(cam["cam_right"]), "cam_left_bottom": f32(cam["cam_left_bottom"]),
        "vertexShader": open(os.path.join(REF, "vertexShader.glsl")).read(),
        "fragmentShader": derive_debug_shader(open(os.path.join(REF, "fragmentShaderPathTracing.glsl")).read()),
        "scene": {"T": s.T, "N": s.N, "L": s.L, **{k: getattr(s, k).tolist() for k in
                  ("coords", "normals", "colors", "emissions", "light_tri", "node_min", "node_max", "node_miss", "node_first", "node_count", "inorder_tri")}},
    }
    res = call_harness(name, payload)
    words = [np.frombuffer(base64.b64decode(b), dtype="<u4").reshape(h, w) for b in res["frames"]]
    os.makedirs(PRIM_OUT, exist_ok=True)
    if rad:
        np.savez_compressed(os.path.join(PRIM_OUT, name + ".npz"), rgb_bits=np.stack(words, axis=-1), scene=scene_name, width=w, height=h,
                            sample_count=spp, max_path_length=1, timestamp=ts0, renderer=json.dumps(res["info"]))
        print(name, res["info"], "mean", float(np.stack(words, axis=-1).view(np.float32).mean()), flush=True)
        return
    np.savez_compressed(os.path.join(PRIM_OUT, name + ".npz"), tri=words[0].view(np.int32), t_bits=words[1], dir_bits=np.stack(words[2:5], axis=-1),
                        scene=scene_name, width=w, height=h, renderer=json.dumps(res["info"]))
    print(name, res["info"], "hits", int((words[0].view(np.int32) >= 0).sum()), "of", w * h, flush=True)


# Probes of what GLSL leaves to the implementation, evaluated by the same derived shader on the same GL implementation
# (the tail of main() computes them from pixel indices and uniforms; nothing of the scene is used):
#   sincos: float bits of theta = 2 pi i / 16384, sin(theta), cos(theta), sqrt(i / 16384) for i = 0 .. 16383
#   minmax: min/max with NaN and infinite operands (NaN = 0 * (1 / 0) built from the uniform maxPathLength = 0)
#   pow:    float bits of x = i / 16384, pow(x, 1 / 2.2) (the display pass, fragmentShaderOutput.glsl:35), log2(x), exp2(-16 x)
PROBE_CASES = {"sincos": 128, "minmax": 4, "pow": 128}
MINMAX_EXPR = ["min(qn, one)", "min(one, qn)", "max(qn, one)", "max(one, qn)", "min(ninf, qn)", "max(ninf, qn)", "min(qn, ninf)", "max(qn, ninf)",
               "nan2", "min(nan2, one)", "min(one, nan2)", "max(nan2, one)", "max(one, nan2)", "pinf", "ninf", "(qn <= one) ? 1.0f : 0.0f"]


def run_probe_case(name):
    m = ptx_import.load()
    w = h = PROBE_CASES[name]
    s = m.load_scene("my_cornell_6")
    cam = m.load_camera("my_cornell_6")
    f32 = lambda v: [float(np.float32(x)) for x in v]  # noqa: E731
    sh = derive_debug_shader(open(os.path.join(REF, "fragmentShaderPathTracing.glsl")).read())
    a = sh.index("    uint w = debugWord == 0")
    b = sh.index("\n", a)
    if name == "sincos":
        nwords = 4
        probe = ("    uint idx = uint(gl_FragCoord.y) * %du + uint(gl_FragCoord.x);\n    float rr = float(idx) * %sf;\n    float th = 2.0f * M_PI * rr;\n"
                 "    uint w = debugWord == 0 ? floatBitsToUint(th) : (debugWord == 1 ? floatBitsToUint(sin(th)) : (debugWord == 2 ? floatBitsToUint(cos(th)) : floatBitsToUint(sqrt(rr))));"
                 % (w, repr(1.0 / (w * h))))
    elif name == "pow":
        nwords = 4
        probe = ("    uint idx = uint(gl_FragCoord.y) * %du + uint(gl_FragCoord.x);\n    float rr = float(idx) * %sf;\n"
                 "    float dv[4] = float[4](rr, pow(rr, 1.0 / 2.2), log2(rr), exp2(-16.0f * rr));\n    uint w = floatBitsToUint(dv[debugWord]);"
                 % (w, repr(1.0 / (w * h))))
    else:
        nwords = len(MINMAX_EXPR)
        probe = ("    float zz = float(maxPathLength); float pinf = 1.0f / zz; float ninf = -1.0f / zz; float nan2 = zz * pinf;\n"
                 "    float qn = uintBitsToFloat(0x7fc00000u + uint(maxPathLength)); float one = float(sampleCount);\n"
                 "    float dv[%d] = float[%d](%s);\n    uint w = floatBitsToUint(dv[debugWord]);" % (nwords, nwords, ", ".join(MINMAX_EXPR)))
    payload = {
        "width": w, "height": h, "sample_count": 1, "max_path_length": 0, "timestamps": [1000], "debug_words": list(range(nwords)),
        "cam_src": f32(cam["cam_src"]), "cam_dir": f32(cam["cam_dir"]), "cam_up": f32(cam["cam_up"]),
        "cam_right": f32(cam["cam_right"]), "cam_left_bottom": f32(cam["cam_left_bottom"]),
        "vertexShader": open(os.path.join(REF, "vertexShader.glsl")).read(), "fragmentShader": sh[:a] + probe + sh[b:],
        "scene": {"T": s.T, "N": s.N, "L": s.L, **{k: getattr(s, k).tolist() for k in
                  ("coords", "normals", "colors", "emissions", "light_tri", "node_min", "node_max", "node_miss", "node_first", "node_count", "inorder_tri")}},
    }
    res = call_harness(name, payload)
    words = np.stack([np.frombuffer(base64.b64decode(b_), dtype="<u4").reshape(-1) for b_ in res["frames"]])
    os.makedirs(PRIM_OUT, exist_ok=True)
    if name == "sincos":
        np.savez_compressed(os.path.join(PRIM_OUT, "probe_sincos.npz"), theta_bits=words[0], sin_bits=words[1], cos_bits=words[2], sqrt_bits=words[3],
                            renderer=json.dumps(res["info"]))
    elif name == "pow":
        np.savez_compressed(os.path.join(PRIM_OUT, "probe_pow.npz"), x_bits=words[0], pow_bits=words[1], log2_bits=words[2], exp2_bits=words[3],
                            renderer=json.dumps(res["info"]))
    else:
        np.savez_compressed(os.path.join(PRIM_OUT, "probe_minmax.npz"), expr=np.asarray(MINMAX_EXPR), bits=words[:, 0], renderer=json.dumps(res["info"]))
    print(name, res["info"], [hex(int(x)) for x in words[:, 1 if name == "sincos" else 0]], flush=True)


def kaleido_exe():
    import kaleido
    return os.path.join(os.path.dirname(kaleido.__file__), "executable", "kaleido")


def call_harness(name, payload):
    req = {"data": {"data": [{"type": "scatter", "customdata": [payload]}], "layout": {}}, "format": "json", "width": 16, "height": 16, "scale": 1}
    cmd = [kaleido_exe(), "plotly", "--plotlyjs=" + os.path.join(ROOT, "oracle", "webgl_harness.js"), "--no-sandbox", "--disable-gpu",
           "--single-process", "--disable-breakpad"]
    proc = subprocess.run(cmd, input=(json.dumps(req) + "\n").encode(), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=7200)
    lines = [ln for ln in proc.stdout.decode(errors="replace").splitlines() if ln.strip()]
    reply = json.loads(lines[-1])
    res = reply.get("result")
    if isinstance(res, str):
        res = json.loads(res)
    if not res or "error" in res:
        raise RuntimeError(f"{name}: harness failed: {reply} / {proc.stderr.decode(errors='replace')[-500:]}")
    return res


def run_case(name):
    m = ptx_import.load()
    scene_name, w, h, spp, bounces, ts = CASES[name]
    s = m.load_scene(scene_name)
    cam = m.load_camera(scene_name)
    f32 = lambda v: [float(np.float32(x)) for x in v]  # noqa: E731
    payload = {
        "width": w, "height": h, "sample_count": spp, "max_path_length": bounces, "timestamps": ts, "progressive": "_prog_" in name,
        "cam_src": f32(cam["cam_src"]), "cam_dir": f32(cam["cam_dir"]), "cam_up": f32(cam["cam_up"]),
        "cam_right": f32(cam["cam_right"]), "cam_left_bottom": f32(cam["cam_left_bottom"]),
        "vertexShader": open(os.path.join(REF, "vertexShader.glsl")).read(),
        "fragmentShader": open(os.path.join(REF, "fragmentShaderPathTracing.glsl")).read(),
        "scene": {"T": s.T, "N": s.N, "L": s.L, **{k: getattr(s, k).tolist() for k in
                  ("coords", "normals", "colors", "emissions", "light_tri", "node_min", "node_max", "node_miss", "node_first", "node_count", "inorder_tri")}},
    }
    if "_disp_" in name:
        out_text = open(os.path.join(REF, "fragmentShaderOutput.glsl")).read()
        # the file opens with a comment line BEFORE '#version 300 es', which GLSL ES rejects ("#version directive must occur
        # on the first line"; the reference's bundler evidently strips it): drop the lines in front of the directive
        payload["outputFragmentShader"] = out_text[out_text.index("#version"):]
    t0 = time.time()
    res = call_harness(name, payload)
    frames = np.stack([np.frombuffer(base64.b64decode(b), dtype=np.uint8).reshape(h, w, 4) for b in res["frames"]])
    os.makedirs(OUT, exist_ok=True)
    extra = {}
    if "_disp_" in name:
        extra["display"] = np.stack([np.frombuffer(base64.b64decode(b), dtype=np.uint8).reshape(h, w, 4) for b in res["display"]])
    np.savez_compressed(os.path.join(OUT, name + ".npz"), frames=frames, scene=scene_name, **extra, width=w, height=h, sample_count=spp,
                        max_path_length=bounces, timestamps=np.asarray(ts, dtype=np.int32), draw_ms=np.asarray(res["ms"]),
                        renderer=json.dumps(res["info"]), host_cores=os.cpu_count())
    print(name, res["info"], "draw ms", [round(x) for x in res["ms"]], "wall", round(time.time() - t0, 1), "s", flush=True)


if __name__ == "__main__":
    for n in (sys.argv[1:] or list(CASES) + list(PRIM_CASES) + list(RAD_CASES) + list(PROBE_CASES)):
        run_probe_case(n) if n in PROBE_CASES else (run_prim_case(n) if (n in PRIM_CASES or n in RAD_CASES) else run_case(n))
