"""Condense a rocprofv3 --kernel-trace --stats run into profiles/<name>.md (+ the raw kernel_stats.csv).
usage: python scripts/summarize_rocprof.py gpurun_out/prof_r1 profiles/r01_bench_cfg2 <steps_incl_warmup> ["cmd"]"""
import csv, glob, os, shutil, sys

src, dst, steps = sys.argv[1], sys.argv[2], int(sys.argv[3])
cmd = sys.argv[4] if len(sys.argv) > 4 else ""
stats = max(glob.glob(os.path.join(src, "**", "*kernel_stats.csv"), recursive=True), key=os.path.getmtime)   # (gpurun MERGES into gpurun_out/: older runs stay)
rows = list(csv.DictReader(open(stats)))
os.makedirs(os.path.dirname(dst), exist_ok=True)
shutil.copy(stats, dst + "_kernel_stats.csv")
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "mdf-net_amd"))
from mdfnet_hip.kernel_families import is_ours, family, MFMA_CONV      # noqa: E402  (exact function names, CPU-tested against csrc/)
unit = sys.argv[5] if len(sys.argv) > 5 else "forward"
tot = sum(float(r["TotalDurationNs"]) for r in rows)
tune = sum(float(r["TotalDurationNs"]) for r in rows if r["Name"].startswith("naive_conv"))
mine = sum(float(r["TotalDurationNs"]) for r in rows if is_ours(r["Name"]))
conv = sum(float(r["TotalDurationNs"]) for r in rows if family(r["Name"]) == MFMA_CONV)
nconv = sum(int(r["Calls"]) for r in rows if family(r["Name"]) == MFMA_CONV)
with open(dst + ".md", "w") as f:
    f.write(f"# rocprofv3 --kernel-trace --stats summary\n\ncommand: `{cmd}`\n\n")
    f.write(f"{unit} passes in the run (warm-up + timed): {steps}\n\n")
    f.write(f"* all kernels: {tot/1e6:.2f} ms; MIOpen find-mode `naive_conv*` (first call only): {tune/1e6:.2f} ms\n")
    f.write(f"* steady state per {unit}: {(tot-tune)/steps/1e6:.3f} ms GPU-busy, of which hand-written HIP kernels "
            f"{mine/steps/1e6:.3f} ms, of which the fp32-MFMA conv family (every launch of mdf_conv*_fwd, the one-launch prob head "
            f"and the refine tail) {conv/steps/1e6:.3f} ms in {nconv/steps:.1f} launches\n\n| kernel | calls | calls/fwd | total ms | avg us | min us | max us |\n|---|---|---|---|---|---|---|\n")
    for r in rows[:45]:
        name = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:110]
        f.write(f"| `{name}` | {r['Calls']} | {int(r['Calls'])/steps:.1f} | {float(r['TotalDurationNs'])/1e6:.3f} | "
                f"{float(r['AverageNs'])/1e3:.1f} | {float(r['MinNs'])/1e3:.1f} | {float(r['MaxNs'])/1e3:.1f} |\n")
print(open(dst + ".md").read()[:1500])
