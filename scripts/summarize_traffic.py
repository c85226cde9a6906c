"""HBM traffic per hand-written kernel family from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE), corrected as
MI355X_MICROARCH.md (HBM section) prescribes: counters are in KiB; on gfx950 FETCH_SIZE reports 1/2 of the bytes of wide
coalesced streaming reads -> doubled; WRITE_SIZE is exact for 16-B-per-lane streaming stores.
usage: python scripts/summarize_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write <forwards> profiles/r01_traffic.json"""
import collections, csv, glob, json, os, sys

fetch_dir, write_dir, forwards, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]


sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "mdf-net_amd"))
from mdfnet_hip.kernel_families import family, fetch_size_factor      # noqa: E402  (the one table of kernel -> family; exact names)


def load(d, counter):
    f = max(glob.glob(d + "/**/*counter_collection.csv", recursive=True), key=os.path.getmtime)   # (gpurun MERGES into gpurun_out/: older runs stay)
    acc = collections.defaultdict(float)
    n = collections.defaultdict(int)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        fam = family(r["Kernel_Name"])
        if fam:
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
json.dump(res, open(out, "w"), indent=1)
