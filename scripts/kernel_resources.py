"""Register / spill / occupancy table of every kernel in one .hip file (dev tool; cross-compiles, no GPU needed).
usage: python scripts/kernel_resources.py mdf-net_amd/csrc/conv_lds.hip [extra hipcc flags]"""
import os, re, subprocess, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1]
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-ffp-contract=off", "--offload-arch=gfx950", "-I", R + "/include", "-I", os.path.dirname(src),
       "-c", src, "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"] + sys.argv[2:]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = []
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        cur = {"name": re.sub(r"\(anonymous namespace\)::|\(.*$", "", name)}
        rows.append(cur)
        continue
    m = re.search(r"remark:\s+([A-Za-z /\[\]]+): (\d+)", line)
    if m and cur is not None:
        cur[m.group(1).strip()] = int(m.group(2))
print("%-52s %5s %5s %6s %6s %4s %7s" % ("kernel", "VGPR", "AGPR", "vspill", "sspill", "occ", "LDS"))
for r in rows:
    print("%-52s %5d %5d %6d %6d %4d %7d" % (r["name"][:52], r.get("VGPRs", -1), r.get("AGPRs", -1), r.get("VGPRs Spill", -1), r.get("SGPRs Spill", -1),
                                          r.get("Occupancy [waves/SIMD]", -1), r.get("LDS Size [bytes/block]", -1)))
