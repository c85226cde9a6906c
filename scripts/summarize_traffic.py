"""HBM traffic per hand-written kernel family from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE), corrected as
MI355X_MICROARCH.md (HBM section) prescribes: counters are in KiB; on gfx950 FETCH_SIZE reports 1/2 of the bytes of wide
coalesced streaming reads -> doubled; WRITE_SIZE is exact for 16-B-per-lane streaming stores.
Per kernel FUNCTION as well (round 5): the PMC bytes of every launch of a function next to the ALGORITHMIC bytes of
the ABI calls that function served in one step (bench.py: roofline.instrumentation.per_kernel, from mdf_last_launch) and their ratio --
the over-fetch is attributable per kernel.
usage: python scripts/summarize_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write <forwards> profiles/r01_traffic.json [bench line .json]"""
import collections, csv, glob, json, os, sys

fetch_dir, write_dir, forwards, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]


sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "mdf-net_amd"))
from mdfnet_hip.kernel_families import family, fetch_size_factor, function_name      # noqa: E402  (the one table of kernel -> family; exact names)
bench_json = sys.argv[5] if len(sys.argv) > 5 else None


per_fn = {"FETCH_SIZE": collections.defaultdict(float), "WRITE_SIZE": collections.defaultdict(float)}
per_fn_n = {"FETCH_SIZE": collections.defaultdict(int), "WRITE_SIZE": collections.defaultdict(int)}


def load(d, counter):
    f = max(glob.glob(d + "/**/*counter_collection.csv", recursive=True), key=os.path.getmtime)   # (gpurun MERGES into gpurun_out/: older runs stay)
    acc = collections.defaultdict(float)
    n = collections.defaultdict(int)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        fam = family(r["Kernel_Name"])
        if fam:
            fn = function_name(r["Kernel_Name"])
            per_fn[counter][fn] += float(r["Counter_Value"]) * (fetch_size_factor(r["Kernel_Name"]) if counter == "FETCH_SIZE" else 1.0)
            per_fn_n[counter][fn] += 1
            # FETCH_SIZE: per-kernel factor (2 = 128-B requests tallied as 64 B: every kernel here, calibrated, see
            # kernel_families.fetch_size_factor); WRITE_SIZE is exact
            acc[fam] += float(r["Counter_Value"]) * (fetch_size_factor(r["Kernel_Name"]) if counter == "FETCH_SIZE" else 1.0)
            n[fam] += 1
    return acc, n


fe, nf = load(fetch_dir, "FETCH_SIZE")
wr, nw = load(write_dir, "WRITE_SIZE")
res = {"method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over bench.py; KiB -> bytes; FETCH_SIZE x2 "
                 "(gfx950 counts 128-B requests as 64 B; calibrated per access shape: profiles/r04_fetch_calibration.md); the counter sits "
                 "between L2 and fabric, Infinity-Cache hits included; per forward = total / forwards",
       "forwards": forwards, "families": {}}
for fam in sorted(set(fe) | set(wr)):
    rd = fe.get(fam, 0.0) * 1024 / forwards          # (the gfx950 factor is already applied per kernel in load())
    wt = wr.get(fam, 0.0) * 1024 / forwards
    res["families"][fam] = {"read_bytes_per_forward": rd, "write_bytes_per_forward": wt, "hbm_bytes_per_forward": rd + wt,
                            "launches_per_forward": nf.get(fam, 0) // forwards}
    print(f"{fam:16s} read {rd/1e6:9.1f} MB  write {wt/1e6:9.1f} MB  per forward ({nf.get(fam,0)//forwards} launches)")
alg = {}
if bench_json and os.path.exists(bench_json):
    line = [ln for ln in open(bench_json).read().splitlines() if ln.startswith("{")][-1]
    alg = (json.loads(line).get("roofline", {}).get("instrumentation", {}) or {}).get("per_kernel", {})
res["per_kernel"] = {}
print("per kernel function: PMC bytes per forward | algorithmic bytes of the calls it served | ratio")
for fn in sorted(set(per_fn["FETCH_SIZE"]) | set(per_fn["WRITE_SIZE"]), key=lambda k: -(per_fn["FETCH_SIZE"].get(k, 0) + per_fn["WRITE_SIZE"].get(k, 0))):
    rd = per_fn["FETCH_SIZE"].get(fn, 0.0) * 1024 / forwards
    wt = per_fn["WRITE_SIZE"].get(fn, 0.0) * 1024 / forwards
    a = alg.get(fn, {}).get("algorithmic_mb_per_step")
    ent = {"read_bytes_per_forward": rd, "write_bytes_per_forward": wt, "launches_per_forward": round(per_fn_n["FETCH_SIZE"].get(fn, 0) / forwards, 1)}
    if a:
        ent["algorithmic_bytes_per_forward"] = a * 1e6
        ent["pmc_over_algorithmic"] = round((rd + wt) / (a * 1e6), 3)
    res["per_kernel"][fn] = ent
    print(f"   {fn:28s} read {rd/1e6:8.1f} MB  write {wt/1e6:8.1f} MB" + (f"  algorithmic {a:8.1f} MB  ratio {(rd + wt) / (a * 1e6):5.2f}" if a else ""))
json.dump(res, open(out, "w"), indent=1)
