"""ctypes mirror of the C ABI in include/ptx.h (libptx_hip.so).

The class follows the boundary of SURVEY.md 8b one-to-one: Context.upload_scene ~ the eleven
uploadTexture() calls (main.js:272-282), set_params ~ the uniform block (main.js:290-308),
render_frames ~ the per-frame draw (main.js:330-355), read_* ~ gl.readPixels (main.js:459).
There is no fallback: if the HIP library is missing or no gfx950 device is usable, this raises.
"""
import ctypes as C
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

PTX_OK, PTX_ERR_INVALID, PTX_ERR_HIP, PTX_ERR_STATE, PTX_ERR_NOMEM = 0, -1, -2, -3, -4

_f = C.POINTER(C.c_float)
_i = C.POINTER(C.c_int32)

# every symbol include/ptx.h declares (tests check the library exports all of them)
SYMBOLS = ["ptx_device_count", "ptx_create", "ptx_destroy", "ptx_last_error", "ptx_upload_scene", "ptx_set_params",
           "ptx_set_option", "ptx_render_frames", "ptx_render_frames_noalloc", "ptx_synchronize", "ptx_reset_accum", "ptx_frames_done",
           "ptx_read_linear_f32", "ptx_read_sum_f32", "ptx_read_rgba8", "ptx_read_compat_rgba8", "ptx_write_sum_f32", "ptx_trace_closest", "ptx_primary_hits",
           "ptx_primary_dirs", "ptx_get_counters", "ptx_reset_counters", "ptx_accum_export_device",
           "ptx_accum_import_device", "ptx_get_size", "ptx_comm_unique_id", "ptx_comm_init", "ptx_comm_world", "ptx_allreduce", "ptx_selfcheck_pruning"]


class PtxError(RuntimeError):
    def __init__(self, status, message):
        super().__init__(f"ptx status {status}: {message}")
        self.status = status


class CScene(C.Structure):
    _fields_ = [("T", C.c_uint32), ("N", C.c_uint32), ("L", C.c_uint32), ("coords", _f), ("normals", _f), ("colors", _f),
                ("emissions", _f), ("light_tri", _i), ("node_min", _f), ("node_max", _f), ("node_miss", _i),
                ("node_first", _i), ("node_count", _i), ("inorder_tri", _i)]


class CParams(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("cam_src", C.c_float * 3), ("cam_up", C.c_float * 3),
                ("cam_right", C.c_float * 3), ("cam_left_bottom", C.c_float * 3), ("max_path_length", C.c_int32),
                ("sample_count", C.c_int32)]


class CCounters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("samples", "rays", "rays_traced", "node_visits", "tri_tests", "ray_hits",
                                          "light_attempts", "pixel_frames", "kernel_launches")] + [("kernel_ms", C.c_double), ("diag", C.c_uint64 * 12)]

    def as_dict(self):
        d = {n: (float(getattr(self, n)) if n == "kernel_ms" else int(getattr(self, n))) for n, _ in self._fields_ if n != "diag"}
        d["diag"] = [int(x) for x in self.diag]
        return d


class CSelfcheck(C.Structure):
    _fields_ = [("rays", C.c_uint64), ("hits", C.c_uint64), ("mismatches", C.c_uint64), ("first", (C.c_float * 8) * 16), ("first_tri", (C.c_int32 * 2) * 16),
                ("emitter_hits", C.c_uint64), ("anyhit_violations", C.c_uint64)]


def lib_path():
    # PTX_LIB_PATH: A/B builds of the same library (tools/perf.py); default = the in-tree build
    return os.environ.get("PTX_LIB_PATH") or os.path.join(_HERE, "csrc", "libptx_hip.so")


def load_library():
    """Loads csrc/libptx_hip.so (built in-tree by csrc/Makefile / __graft_entry__.build()).

    torch bundles its own libamdhip64.so; if torch is going to share the process it must be imported
    BEFORE this library so both resolve to one HIP runtime (DESIGN.md "One HIP runtime per process")."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = lib_path()
    if not os.path.exists(path):
        raise FileNotFoundError(f"{path} is missing: run `make -C path-tracing_amd/csrc` (hipcc, gfx950). There is no CPU fallback.")
    if "torch" not in sys.modules and os.environ.get("PTX_NO_TORCH", "0") != "1":
        try:
            import torch  # noqa: F401  (plumbing only: makes its bundled HIP runtime the process's single one)
        except Exception:
            pass
    L = C.CDLL(path)
    L.ptx_create.restype = C.c_void_p
    L.ptx_create.argtypes = [C.POINTER(C.c_int), C.c_int]
    L.ptx_destroy.argtypes = [C.c_void_p]
    L.ptx_destroy.restype = None
    L.ptx_last_error.restype = C.c_char_p
    L.ptx_last_error.argtypes = [C.c_void_p]
    L.ptx_upload_scene.argtypes = [C.c_void_p, C.POINTER(CScene)]
    L.ptx_set_params.argtypes = [C.c_void_p, C.POINTER(CParams)]
    L.ptx_set_option.argtypes = [C.c_void_p, C.c_char_p, C.c_int64]
    L.ptx_render_frames.argtypes = [C.c_void_p, _i, C.c_uint32, C.c_uint32]
    L.ptx_render_frames_noalloc.argtypes = [C.c_void_p, _i, C.c_uint32, C.c_uint32]
    L.ptx_synchronize.argtypes = [C.c_void_p]
    L.ptx_reset_accum.argtypes = [C.c_void_p]
    L.ptx_frames_done.argtypes = [C.c_void_p]
    L.ptx_read_linear_f32.argtypes = [C.c_void_p, _f]
    L.ptx_read_sum_f32.argtypes = [C.c_void_p, _f]
    L.ptx_read_rgba8.argtypes = [C.c_void_p, C.POINTER(C.c_uint8), C.c_int, C.c_int]
    L.ptx_read_compat_rgba8.argtypes = [C.c_void_p, C.POINTER(C.c_uint8), C.c_int, C.c_int]
    L.ptx_write_sum_f32.argtypes = [C.c_void_p, _f, C.c_uint32]
    L.ptx_trace_closest.argtypes = [C.c_void_p, _f, _f, C.c_uint32, _f, _i]
    L.ptx_primary_hits.argtypes = [C.c_void_p, _i, _f]
    L.ptx_primary_dirs.argtypes = [C.c_void_p, _f]
    L.ptx_get_counters.argtypes = [C.c_void_p, C.POINTER(CCounters)]
    L.ptx_reset_counters.argtypes = [C.c_void_p]
    L.ptx_accum_export_device.argtypes = [C.c_void_p, C.c_void_p]
    L.ptx_accum_import_device.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
    L.ptx_selfcheck_pruning.argtypes = [C.c_void_p, C.c_uint32, C.c_uint64, C.c_uint32, C.POINTER(CSelfcheck)]
    L.ptx_get_size.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    L.ptx_comm_unique_id.argtypes = [C.c_void_p]
    L.ptx_comm_init.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    L.ptx_comm_world.argtypes = [C.c_void_p]
    L.ptx_allreduce.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_uint32)]
    _LIB = L
    return L


def comm_unique_id():
    """128 bytes from ncclGetUniqueId (rank 0 makes it and hands it to the other ranks; Context.comm_init)."""
    L = load_library()
    buf = C.create_string_buffer(128)
    rc = L.ptx_comm_unique_id(buf)
    if rc != PTX_OK:
        raise PtxError(rc, L.ptx_last_error(None).decode())
    return buf.raw


def _fp(a):
    return a.ctypes.data_as(_f)


def _ip(a):
    return a.ctypes.data_as(_i)


class Context:
    """One GPU, one scene, one progressive accumulation (ptx_ctx)."""

    def __init__(self, device=0):
        """device: one HIP ordinal, or a list of them for one context over several GPUs (include/ptx.h ptx_create)."""
        self._lib = load_library()
        devs = [int(d) for d in device] if isinstance(device, (list, tuple)) else [int(device)]
        ids = (C.c_int * len(devs))(*devs)
        self._h = self._lib.ptx_create(ids, len(devs))
        if not self._h:
            raise PtxError(PTX_ERR_HIP, self._lib.ptx_last_error(None).decode())
        self.params = None
        self.devices = devs
        self.device = devs[0]

    def close(self):
        if getattr(self, "_h", None):
            self._lib.ptx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _check(self, rc):
        if rc != PTX_OK:
            raise PtxError(rc, self._lib.ptx_last_error(self._h).decode())

    # ---- scene + parameters
    def upload_scene(self, s):
        if hasattr(s, "validate"):
            s.validate()          # dtype, contiguity and length of every array: the C side can only check index VALUES
        cs = CScene(s.T, s.N, s.L, _fp(s.coords), _fp(s.normals), _fp(s.colors), _fp(s.emissions), _ip(s.light_tri),
                    _fp(s.node_min), _fp(s.node_max), _ip(s.node_miss), _ip(s.node_first), _ip(s.node_count), _ip(s.inorder_tri))
        self._check(self._lib.ptx_upload_scene(self._h, C.byref(cs)))

    def set_params(self, p):
        cp = CParams()
        cp.width, cp.height = p.width, p.height
        for k in ("cam_src", "cam_up", "cam_right", "cam_left_bottom"):
            getattr(cp, k)[:] = [float(x) for x in getattr(p, k)]
        cp.max_path_length, cp.sample_count = p.max_path_length, p.sample_count
        self._check(self._lib.ptx_set_params(self._h, C.byref(cp)))
        self.params = p

    def set_option(self, key, value):
        self._check(self._lib.ptx_set_option(self._h, key.encode(), int(value)))

    # ---- rendering
    def render_frames(self, timestamps, first_frame_number=0):
        ts = np.ascontiguousarray(timestamps, dtype=np.int32)
        self._check(self._lib.ptx_render_frames(self._h, _ip(ts), ts.size, first_frame_number))

    def render_frames_noalloc(self, timestamps, first_frame_number=0):
        """ptx_render_frames through the static scratch: allocates nothing (include/ptx.h)."""
        ts = np.ascontiguousarray(timestamps, dtype=np.int32)
        self._check(self._lib.ptx_render_frames_noalloc(self._h, _ip(ts), ts.size, first_frame_number))

    def synchronize(self):
        self._check(self._lib.ptx_synchronize(self._h))

    def reset_accum(self):
        self._check(self._lib.ptx_reset_accum(self._h))

    @property
    def frames_done(self):
        return self._lib.ptx_frames_done(self._h)

    # ---- read-back (row 0 = bottom row, like gl.readPixels)
    def _img(self, ch, dtype):
        if self.params is None:
            raise PtxError(PTX_ERR_STATE, "set_params first")
        return np.zeros((self.params.height, self.params.width, ch), dtype=dtype)

    def read_linear(self):
        out = self._img(3, np.float32)
        self._check(self._lib.ptx_read_linear_f32(self._h, _fp(out)))
        return out

    def read_sum(self):
        out = self._img(3, np.float32)
        self._check(self._lib.ptx_read_sum_f32(self._h, _fp(out)))
        return out

    def read_rgba8(self, apply_gamma=True, flip_y=True):
        out = self._img(4, np.uint8)
        self._check(self._lib.ptx_read_rgba8(self._h, out.ctypes.data_as(C.POINTER(C.c_uint8)), int(apply_gamma), int(flip_y)))
        return out

    def read_compat_rgba8(self, apply_gamma=False, flip_y=False):
        """The reference's RGBA8 ping-pong target (option compat_rgba8 = 1 or 2, see include/ptx.h)."""
        out = self._img(4, np.uint8)
        self._check(self._lib.ptx_read_compat_rgba8(self._h, out.ctypes.data_as(C.POINTER(C.c_uint8)), int(apply_gamma), int(flip_y)))
        return out

    def write_sum(self, rgb, frames_done):
        """Checkpoint restore: accumulation := rgb (H, W, 3 float32 as read_sum returned it), frames_done frames in it."""
        rgb = np.ascontiguousarray(rgb, dtype=np.float32)
        if rgb.size != self.params.height * self.params.width * 3:
            raise ValueError("rgb must hold height * width * 3 floats")
        self._check(self._lib.ptx_write_sum_f32(self._h, _fp(rgb), int(frames_done)))

    # ---- parity / roofline hooks
    def trace_closest(self, o, d):
        o = np.ascontiguousarray(o, dtype=np.float32).reshape(-1, 3)
        d = np.ascontiguousarray(d, dtype=np.float32).reshape(-1, 3)
        if o.shape != d.shape:
            raise ValueError("o and d must have the same shape")
        n = o.shape[0]
        t = np.zeros(n, dtype=np.float32)
        tri = np.zeros(n, dtype=np.int32)
        self._check(self._lib.ptx_trace_closest(self._h, _fp(o), _fp(d), n, _fp(t), _ip(tri)))
        return t, tri

    def selfcheck_pruning(self, family, n_rays, seed=1):
        """ptx_selfcheck_pruning: n_rays adversarial rays of `family`, traced with and without pruning on the device.
        Returns dict(rays, hits, mismatches, first=[(o, d, t_exact, t_pruned, tri_exact, tri_pruned), ...])."""
        r = CSelfcheck()
        self._check(self._lib.ptx_selfcheck_pruning(self._h, int(family), int(n_rays), int(seed), C.byref(r)))
        first = []
        for q in range(16):
            if r.first_tri[q][0] != -2 or r.first_tri[q][1] != -2:
                v = [float(x) for x in r.first[q]]
                first.append((v[0:3], v[3:6], v[6], v[7], int(r.first_tri[q][0]), int(r.first_tri[q][1])))
        return {"rays": int(r.rays), "hits": int(r.hits), "mismatches": int(r.mismatches), "first": first,
                "emitter_hits": int(r.emitter_hits), "anyhit_violations": int(r.anyhit_violations)}

    def primary_hits(self):
        tri = np.zeros((self.params.height, self.params.width), dtype=np.int32)
        t = np.zeros((self.params.height, self.params.width), dtype=np.float32)
        self._check(self._lib.ptx_primary_hits(self._h, _ip(tri), _fp(t)))
        return t, tri

    def primary_dirs(self):
        out = self._img(3, np.float32)
        self._check(self._lib.ptx_primary_dirs(self._h, _fp(out)))
        return out

    def counters(self):
        c = CCounters()
        self._check(self._lib.ptx_get_counters(self._h, C.byref(c)))
        return c.as_dict()

    def reset_counters(self):
        self._check(self._lib.ptx_reset_counters(self._h))

    # ---- multi-GPU, one process per GPU (RCCL inside the library)
    def comm_init(self, id128, rank, world):
        self._check(self._lib.ptx_comm_init(self._h, C.c_char_p(bytes(id128)), int(rank), int(world)))

    @property
    def comm_world(self):
        return self._lib.ptx_comm_world(self._h)

    def allreduce(self, device_ptr=None):
        """Sums accumulation + frame counts over the ranks (ptx_allreduce); in place when device_ptr is None.
        Returns the total frame count."""
        total = C.c_uint32(0)
        self._check(self._lib.ptx_allreduce(self._h, C.c_void_p(int(device_ptr)) if device_ptr else None, C.byref(total)))
        return int(total.value)

    # ---- reduction hooks for a caller-side collective
    def accum_export_device(self, device_ptr):
        self._check(self._lib.ptx_accum_export_device(self._h, C.c_void_p(int(device_ptr))))

    def accum_import_device(self, device_ptr, frames_done):
        self._check(self._lib.ptx_accum_import_device(self._h, C.c_void_p(int(device_ptr)), int(frames_done)))
