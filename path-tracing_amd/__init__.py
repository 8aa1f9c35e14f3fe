"""path-tracing_amd: MI355X-native progressive path tracer (the hot path of aminequirogabritos/path-tracing).

The directory name has a hyphen (it mirrors the reference repository's name), so it is imported through
``ptx_import.load()`` at the repo root, which registers it as the module ``path_tracing_amd``.

Python here is plumbing around the C-ABI library ``csrc/libptx_hip.so`` (include/ptx.h):
  scene.py    -- .ptxs scene container reader (written by host/scenefile.js) and camera/params records
  binding.py  -- ctypes mirror of the C ABI: Context(upload_scene / set_params / render_frames / read_*)
The product host is JavaScript (host/*.js + the raw N-API addon in csrc/ptx_napi.c).
"""
from .scene import Scene, Params, load_scene, load_camera, config_params, CONFIGS  # noqa: F401
from .binding import Context, PtxError, lib_path, load_library, comm_unique_id  # noqa: F401
from .distributed import frames_for_rank, allreduce_sum, render_frames_distributed, init_library_comm  # noqa: F401
