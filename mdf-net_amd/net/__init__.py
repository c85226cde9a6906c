"""Drop-in for the reference's `net` package (net/core.py, net/loss.py, net/unit/*).

Put `mdf-net_amd/` ahead of the reference on sys.path and the reference's own config.py / train.py /
eval.py import THIS package unchanged: same class names, constructor slots, call signatures and
state_dict keys; the hot operators run hand-written gfx950 kernels through the C ABI in
include/mdfnet_hip.h (no PyTorch fallback for them).
"""
