"""Registers the hyphen-named package directory ``path-tracing_amd/`` as module ``path_tracing_amd``."""
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))


def load():
    if "path_tracing_amd" in sys.modules:
        return sys.modules["path_tracing_amd"]
    pkg_dir = os.path.join(ROOT, "path-tracing_amd")
    spec = importlib.util.spec_from_file_location("path_tracing_amd", os.path.join(pkg_dir, "__init__.py"),
                                                  submodule_search_locations=[pkg_dir])
    mod = importlib.util.module_from_spec(spec)
    sys.modules["path_tracing_amd"] = mod
    spec.loader.exec_module(mod)
    return mod
