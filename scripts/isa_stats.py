"""Per-kernel ISA statistics of a .hip file (dev tool): MFMA count, global loads by addressing form, 64-bit VALU address adds, scratch.
usage: python scripts/isa_stats.py mdf-net_amd/csrc/conv_lds.hip"""
import os, re, subprocess, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1]
asm = "/tmp/isa_stats.s"
subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-ffp-contract=off", "--offload-arch=gfx950", "-I", R + "/include", "-I", os.path.dirname(src),
                "-S", "--cuda-device-only", src, "-o", asm] + sys.argv[2:], check=True, stderr=subprocess.DEVNULL)
s = open(asm).read()
print("%-46s %6s %5s %6s %6s %6s %7s %7s %6s" % ("kernel", "lines", "mfma", "gload", "saddr", "vaddr", "vadd64", "scratch", "ds_rd"))
for m in re.finditer(r"\n(_Z\w+):\s*; @", s):
    n = m.group(1)
    a = m.end(); b = s.index(".Lfunc_end", a)
    body = s[a:b].splitlines()
    gl = [l for l in body if "global_load" in l]
    sad = [l for l in gl if re.search(r",\s*v\d+,\s*s\[", l)]
    vad = [l for l in gl if re.search(r",\s*v\[\d+:\d+\],\s*off", l)]
    cnt = lambda *k: len([l for l in body if any(x in l for x in k)])
    name = subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip()
    name = re.sub(r"\(anonymous namespace\)::|\(.*$|^void ", "", name)
    print("%-46s %6d %5d %6d %6d %6d %7d %7d %6d" % (name[:46], len(body), cnt("v_mfma"), len(gl), len(sad), len(vad),
                                                  cnt("v_add_co", "v_addc_co", "v_lshl_add_u64"), cnt("scratch_"), cnt("ds_read")))
