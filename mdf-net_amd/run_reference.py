#!/usr/bin/env python3
"""Run an unmodified reference script on the MI355X hot path:

    python mdf-net_amd/run_reference.py /path/to/MDF-Net/eval.py -p pth/dtu_29.pth -d dtu
"""
import os
import runpy
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from mdfnet_hip import dropin  # noqa: E402

if __name__ == "__main__":
    if len(sys.argv) < 2:
        raise SystemExit(__doc__)
    script = os.path.abspath(sys.argv[1])
    sys.path.pop(0)
    dropin.install(os.path.dirname(script))
    sys.argv = sys.argv[1:]
    runpy.run_path(script, run_name="__main__")
