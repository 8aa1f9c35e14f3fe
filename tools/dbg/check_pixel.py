"""One frame of a config under the default and the unpruned walk with the library named by PTX_LIB_PATH: differing pixels."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa
import numpy as np
import ptx_import
m = ptx_import.load()
cfg = sys.argv[1]; frames = [int(x) for x in sys.argv[2:]]
c = m.CONFIGS[cfg]
s = m.load_scene(c["scene"]); p = m.config_params(c["scene"], c["width"], c["height"], c["sample_count"], c["max_path_length"])
ctx = m.Context(0); ctx.upload_scene(s); ctx.set_params(p)
for f in frames:
    img = {}
    for name, opts in (("default", dict(traversal=3, prune=2)), ("unpruned", dict(traversal=2, prune=2))):
        for k, v in opts.items():
            ctx.set_option(k, v)
        ctx.reset_accum(); ctx.render_frames([1000 + f], first_frame_number=f)
        img[name] = np.ascontiguousarray(ctx.read_sum()).copy()
    bad = np.argwhere((img["default"].view(np.uint32) != img["unpruned"].view(np.uint32)).any(axis=2))
    print(os.path.basename(os.environ.get("PTX_LIB_PATH", "default")), cfg, "frame", f, "differing pixels", bad.tolist(), flush=True)
