"""Side-by-side per-kernel totals of two rocprofv3 --kernel-trace --stats runs (dev A/B tool).
usage: python scripts/cmp_rocprof.py dirA dirB stepsA stepsB"""
import collections, csv, glob, os, re, sys


def load(d, steps):
    f = glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)[0]
    acc = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        n = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")
        n = re.sub(r"\(.*$", "", n)[:70]
        a = acc.setdefault(n, [0.0, 0])
        a[0] += float(r["TotalDurationNs"]) / 1e3 / steps
        a[1] += int(r["Calls"]) / steps
    return acc


a, b = load(sys.argv[1], int(sys.argv[3])), load(sys.argv[2], int(sys.argv[4]))
keys = sorted(set(a) | set(b), key=lambda k: -(a.get(k, [0])[0] + b.get(k, [0])[0]))
ta = tb = 0.0
print(f"{'kernel':70s} {'A us/step':>10s} {'calls':>6s} {'B us/step':>10s} {'calls':>6s} {'B-A':>8s}")
for k in keys:
    x, y = a.get(k, [0.0, 0]), b.get(k, [0.0, 0])
    ta += x[0]; tb += y[0]
    if max(x[0], y[0]) >= 3.0:
        print(f"{k:70s} {x[0]:10.1f} {x[1]:6.1f} {y[0]:10.1f} {y[1]:6.1f} {y[0]-x[0]:8.1f}")
print(f"{'TOTAL':70s} {ta:10.1f} {'':6s} {tb:10.1f} {'':6s} {tb-ta:8.1f}")
