"""Make the reference's own scripts (config.py / eval.py / train.py) import THIS repo's `net` package.

`python eval.py` puts the script's directory first on sys.path, so the reference's `net/` would win over any
PYTHONPATH entry.  install() registers our package object under the name `net` in sys.modules before the reference
code runs; its `from net import core`, `from net.unit import scale, ...` (config.py:187-191) then resolve to the
HIP-backed modules while `import config`, `load.*`, `tools.*` still come from the reference tree."""
import importlib.util
import os
import sys

PKG_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))  # mdf-net_amd/


def install(reference_root=None):
    if reference_root and reference_root not in sys.path:
        sys.path.insert(0, reference_root)        # what `python <reference>/eval.py` would have done
    if PKG_ROOT not in sys.path:
        sys.path.append(PKG_ROOT)                 # for `mdfnet_hip`; appended so the reference's config.py wins
    net_dir = os.path.join(PKG_ROOT, "net")
    spec = importlib.util.spec_from_file_location("net", os.path.join(net_dir, "__init__.py"),
                                                  submodule_search_locations=[net_dir])
    mod = importlib.util.module_from_spec(spec)
    sys.modules["net"] = mod
    spec.loader.exec_module(mod)
    return mod
