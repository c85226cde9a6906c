#!/bin/bash
# dev: SQ counters (three passes) of the kernels whose name contains FILTER, averaged per launch.
# usage (GPU box): scripts/pmc_kernels.sh FILTER script.py [args]      e.g.  scripts/pmc_kernels.sh conv_pair scripts/bench_pair.py
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
filter=$1; shift
out=$R/gpurun_out/pmc_kernels; rm -rf $out; mkdir -p $out
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVES" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU" \
           "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/p$i -- python3 $R/"$@" > $out/p$i.log 2>&1 || { tail -5 $out/p$i.log; }
done
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
dur = collections.defaultdict(list)
for f in glob.glob("$out/p*/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "$filter" in r["Kernel_Name"]: dur[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for f in glob.glob("$out/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "$filter" not in k: continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
for k, d in acc.items():
    ds = sorted(dur[k]); print(k[:110], f"| median {ds[len(ds)//2]/1e3:.1f} us under the counters, {len(ds)} launches")
    g = {c: v / n[(k, c)] for c, v in d.items()}
    for c, v in sorted(g.items()): print(f"   {c:28s} {v:16.0f}")
    if "SQ_WAVE_CYCLES" in g:
        wc = g["SQ_WAVE_CYCLES"]
        print(f"   -> of wave time: parked {g.get('SQ_WAIT_ANY',0)/wc:.0%}, issue-stalled {g.get('SQ_WAIT_INST_ANY',0)/wc:.0%}, issuing {g.get('SQ_ACTIVE_INST_ANY',0)/wc:.0%}; "
              f"MFMA pipe cycles per SIMD {g.get('SQ_VALU_MFMA_BUSY_CYCLES',0)/1024:.0f}")
PY
