"""Build libmdfnet_hip.so (gfx950) in-tree with hipcc.  `python -m mdfnet_hip.build [--force]`."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG = os.path.dirname(os.path.abspath(__file__))
TOP = os.path.dirname(PKG)                      # mdf-net_amd/
ROOT = os.path.dirname(TOP)                     # repo root
CSRC = os.path.join(TOP, "csrc")
INCLUDE = os.path.join(ROOT, "include")
OBJ = os.path.join(CSRC, "build")
LIB = os.path.join(PKG, "libmdfnet_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"
# -ffp-contract=off: every fma in the kernels is explicit (bit-exact warp arithmetic)
# -munsafe-fp-atomics: fp32/fp64 atomic adds of the training kernels compile to hardware atomics, not CAS loops
FLAGS = ["-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-munsafe-fp-atomics", f"--offload-arch={ARCH}", "-I", INCLUDE, "-I", CSRC,
         "-Wall", "-Wno-unused-function"] + os.environ.get("MDF_EXTRA_HIPCC_FLAGS", "").split()


def sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith((".hip", ".cpp")))


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _compile(src):
    obj = os.path.join(OBJ, src.rsplit(".", 1)[0] + ".o")
    path = os.path.join(CSRC, src)
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + \
           [os.path.join(INCLUDE, f) for f in os.listdir(INCLUDE)]
    if _stale(obj, [path] + hdrs):
        cmd = [HIPCC] + FLAGS + (["-x", "hip"] if src.endswith(".hip") else []) + ["-c", path, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{r.stdout}\n{r.stderr}")
        if r.stderr.strip():
            sys.stderr.write(r.stderr)
        return obj, True
    return obj, False


def build(force=False, verbose=False):
    os.makedirs(OBJ, exist_ok=True)
    if force:
        for f in os.listdir(OBJ):
            os.remove(os.path.join(OBJ, f))
    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        res = list(ex.map(_compile, sources()))
    objs = [o for o, _ in res]
    if any(ch for _, ch in res) or _stale(LIB, objs):
        cmd = [HIPCC, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
        if verbose:
            print("built", LIB)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
