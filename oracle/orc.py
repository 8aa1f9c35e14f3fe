"""ctypes wrapper of the CPU ORACLE (oracle/libptx_oracle.so).  TEST INFRASTRUCTURE ONLY:
imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, never by the product."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

_f = C.POINTER(C.c_float)
_i = C.POINTER(C.c_int32)


class OrcScene(C.Structure):
    _fields_ = [("T", C.c_uint32), ("N", C.c_uint32), ("L", C.c_uint32), ("coords", _f), ("normals", _f), ("colors", _f),
                ("emissions", _f), ("light_tri", _i), ("node_min", _f), ("node_max", _f), ("node_miss", _i),
                ("node_first", _i), ("node_count", _i), ("inorder_tri", _i)]


class OrcParams(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("cam_src", C.c_float * 3), ("cam_up", C.c_float * 3),
                ("cam_right", C.c_float * 3), ("cam_left_bottom", C.c_float * 3), ("max_path_length", C.c_int32),
                ("sample_count", C.c_int32)]


class OrcCounters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("samples", "rays", "node_visits", "tri_tests", "ray_hits", "light_attempts", "rng_calls")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libptx_oracle.so")
        if not os.path.exists(path):
            build()
        _LIB = C.CDLL(path)
        _LIB.orc_sincos_2pi.argtypes = [C.c_float, _f, _f]
        _LIB.orc_ray_tri.argtypes = [_f, _f, _f, _f, _f, _f]
        _LIB.orc_ray_box.argtypes = [_f, _f, _f, _f]
    return _LIB


def _fp(a):
    return a.ctypes.data_as(_f)


def _ip(a):
    return a.ctypes.data_as(_i)


def c_scene(s):
    cs = OrcScene(s.T, s.N, s.L, _fp(s.coords), _fp(s.normals), _fp(s.colors), _fp(s.emissions), _ip(s.light_tri),
                  _fp(s.node_min), _fp(s.node_max), _ip(s.node_miss), _ip(s.node_first), _ip(s.node_count), _ip(s.inorder_tri))
    cs._keep = s
    return cs


def c_params(p):
    cp = OrcParams()
    cp.width, cp.height = p.width, p.height
    for k in ("cam_src", "cam_up", "cam_right", "cam_left_bottom"):
        getattr(cp, k)[:] = [float(x) for x in getattr(p, k)]
    cp.max_path_length, cp.sample_count = p.max_path_length, p.sample_count
    return cp


def pcg2d(seed):
    s = (C.c_uint32 * 2)(*seed)
    o = (C.c_float * 2)()
    lib().orc_pcg2d(s, o)
    return (int(s[0]), int(s[1])), (float(o[0]), float(o[1]))


def seed_init(px, py, timestamp):
    s = (C.c_uint32 * 2)()
    lib().orc_seed_init(C.c_uint32(px), C.c_uint32(py), C.c_int32(timestamp), s)
    return int(s[0]), int(s[1])


def sincos_2pi(theta):
    s, c = C.c_float(), C.c_float()
    lib().orc_sincos_2pi(C.c_float(theta), C.byref(s), C.byref(c))
    return s.value, c.value


def ray_tri(o, d, v0, v1, v2):
    a = [np.ascontiguousarray(x, dtype=np.float32) for x in (o, d, v0, v1, v2)]
    t = C.c_float(0)
    hit = lib().orc_ray_tri(*[_fp(x) for x in a], C.byref(t))
    return bool(hit), t.value


def ray_box(o, d, bmin, bmax):
    a = [np.ascontiguousarray(x, dtype=np.float32) for x in (o, d, bmin, bmax)]
    return bool(lib().orc_ray_box(*[_fp(x) for x in a]))


def sample_hemisphere(rnd, n):
    r = np.ascontiguousarray(rnd, dtype=np.float32)
    nn = np.ascontiguousarray(n, dtype=np.float32)
    out = np.zeros(3, dtype=np.float32)
    lib().orc_sample_hemisphere(_fp(r), _fp(nn), _fp(out))
    return out


def primary_dirs(p):
    cp = c_params(p)
    out = np.zeros((p.height, p.width, 3), dtype=np.float32)
    d = (C.c_float * 3)()
    for y in range(p.height):
        for x in range(p.width):
            lib().orc_primary_dir(C.byref(cp), C.c_uint32(x), C.c_uint32(y), d)
            out[y, x] = d[:]
    return out


def trace_closest(s, o, d):
    o = np.ascontiguousarray(o, dtype=np.float32).reshape(-1, 3)
    d = np.ascontiguousarray(d, dtype=np.float32).reshape(-1, 3)
    n = o.shape[0]
    t = np.zeros(n, dtype=np.float32)
    tri = np.zeros(n, dtype=np.int32)
    cnt = OrcCounters()
    cs = c_scene(s)
    lib().orc_trace_closest(C.byref(cs), _fp(o), _fp(d), C.c_uint32(n), _fp(t), _ip(tri), C.byref(cnt))
    return t, tri, cnt.as_dict()


def trace_pixel(s, p, timestamp, px, py, cap=1 << 16):
    """One pixel of one frame and the rays it traces: (rgb, o (n,3), d (n,3), t (n,), tri (n,)).  Diagnostic."""
    log = np.zeros((cap, 8), dtype=np.float32)
    rgb = np.zeros(3, dtype=np.float32)
    cs, cp = c_scene(s), c_params(p)
    lib().orc_trace_pixel.restype = C.c_uint32
    n = lib().orc_trace_pixel(C.byref(cs), C.byref(cp), C.c_int32(int(timestamp)), C.c_uint32(px), C.c_uint32(py), _fp(rgb), _fp(log), C.c_uint32(cap))
    n = min(int(n), cap)
    return rgb, log[:n, 0:3].copy(), log[:n, 3:6].copy(), log[:n, 6].copy(), log[:n, 7].copy().view(np.int32)


def primary_hits(s, p):
    t = np.zeros((p.height, p.width), dtype=np.float32)
    tri = np.zeros((p.height, p.width), dtype=np.int32)
    cnt = OrcCounters()
    cs, cp = c_scene(s), c_params(p)
    lib().orc_primary_hits(C.byref(cs), C.byref(cp), _fp(t), _ip(tri), C.byref(cnt))
    return t, tri, cnt.as_dict()


def render_frame(s, p, timestamp, y0=0, y1=None, nthreads=0, out=None):
    """Returns (rgb[H,W,3] float32 = per-frame clamped mean, counters dict). Row 0 = bottom row (GL)."""
    rgb = out if out is not None else np.zeros((p.height, p.width, 3), dtype=np.float32)
    cnt = OrcCounters()
    cs, cp = c_scene(s), c_params(p)
    lib().orc_render_frame(C.byref(cs), C.byref(cp), C.c_int32(int(timestamp)), C.c_uint32(y0),
                           C.c_uint32(p.height if y1 is None else y1), _fp(rgb), C.byref(cnt), C.c_int(nthreads))
    return rgb, cnt.as_dict()


def render_rows(s, p, timestamp, y0, y1, y_step, nthreads=0):
    """Rows y0, y0+y_step, ... only (other rows stay zero). Returns (rgb, counters)."""
    rgb = np.zeros((p.height, p.width, 3), dtype=np.float32)
    cnt = OrcCounters()
    cs, cp = c_scene(s), c_params(p)
    lib().orc_render_rows(C.byref(cs), C.byref(cp), C.c_int32(int(timestamp)), C.c_uint32(y0), C.c_uint32(y1),
                          C.c_uint32(y_step), _fp(rgb), C.byref(cnt), C.c_int(nthreads))
    return rgb, cnt.as_dict()


def render_frames(s, p, timestamps, nthreads=0):
    """Sum over frames of the per-frame clamped mean, added in frame order in f32 (the product's accumulation)."""
    acc = np.zeros((p.height, p.width, 3), dtype=np.float32)
    total = {}
    for ts in timestamps:
        rgb, c = render_frame(s, p, int(ts), nthreads=nthreads)
        acc = (acc + rgb).astype(np.float32)
        for k, v in c.items():
            total[k] = total.get(k, 0) + v
    return acc, total


def blend_rgba8(p, cur_rgb, prev_rgba8, frame_number, store=0):
    """store: 0 = round(c * 255); 1 = SwiftShader's render-target store rule (see ptx_oracle.c to_unorm8)."""
    out = np.zeros((p.height, p.width, 4), dtype=np.uint8)
    cp = c_params(p)
    cur = np.ascontiguousarray(cur_rgb, dtype=np.float32)
    prev = None if prev_rgba8 is None else np.ascontiguousarray(prev_rgba8, dtype=np.uint8)
    lib().orc_blend_rgba8_store(C.byref(cp), _fp(cur), None if prev is None else prev.ctypes.data_as(C.POINTER(C.c_uint8)),
                                C.c_int32(frame_number), C.c_int32(store), out.ctypes.data_as(C.POINTER(C.c_uint8)))
    return out


def pow_(x, y):
    lib().orc_pow.restype = C.c_float
    return float(lib().orc_pow(C.c_float(x), C.c_float(y)))


def display_rgba8(rgba8, store=0):
    """The reference's display pass over an 8-bit target (gamma 1/2.2); store as in blend_rgba8."""
    src = np.ascontiguousarray(rgba8, dtype=np.uint8)
    out = np.zeros_like(src)
    lib().orc_display_rgba8(src.ctypes.data_as(C.POINTER(C.c_uint8)), C.c_uint32(src.shape[0] * src.shape[1]), C.c_int32(store),
                            out.ctypes.data_as(C.POINTER(C.c_uint8)))
    return out


def max_threads():
    return int(lib().orc_max_threads())
