"""Static check of wino3d.hip's pinned accumulators: compiles the file to assembly and verifies that outside the inline-assembly
statements (ASMSTART/ASMEND) the compiler itself touches no accumulator register from a64 up -- a[64:255] hold the 48 pinned
accumulator tiles (wino3d_acc.h), which the register allocator does not know to be live.  Also prints spills / scratch.
usage: python scripts/check_pinned_agprs.py [asm file]   (exit code 1 on a violation)"""
import os, re, subprocess, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PINNED = 64     # a[PINNED:255] are pinned; a[0:PINNED-1] are the compiler's


def compile_asm(out="/tmp/wino3d_check.s", name="wino3d.hip"):
    src = R + "/mdf-net_amd/csrc/" + name
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-ffp-contract=off", "-munsafe-fp-atomics", "--offload-arch=gfx950", "-I", R + "/include",
                    "-I", R + "/mdf-net_amd/csrc", "-S", "--cuda-device-only", src, "-o", out], check=True, stderr=subprocess.DEVNULL)
    return out


def check(path):
    s = open(path).read()
    bad = {}
    report = []
    for m in re.finditer(r"\n(_Z\w*wino[23]d_kernel\w*):", s):
        a = m.end(); b = s.index("s_endpgm", a)
        inasm = False; used = set(); ops = {}
        lines = s[a:b].splitlines()
        for l in lines:
            if "#ASMSTART" in l: inasm = True; continue
            if "#ASMEND" in l: inasm = False; continue
            if inasm: continue
            code = l.split(";")[0]
            for r in re.findall(r"\ba\[(\d+):(\d+)\]|\ba(\d+)\b", code):
                regs = range(int(r[0]), int(r[1]) + 1) if r[0] else [int(r[2])]
                used.update(regs)
                if any(PINNED <= x < (192 if 'wino2d' in m.group(1) else 256) for x in regs): ops[code.split()[0]] = ops.get(code.split()[0], 0) + 1
        hi = 192 if "wino2d" in m.group(1) else 256       # wino2d.hip pins tiles 0..31 only
        low = sorted(x for x in used if PINNED <= x < hi)
        name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip().split("(")[0]
        report.append((name, len(used), low[:8], sum("scratch_" in l for l in lines), sum("v_mfma" in l for l in lines),
                       sum(("v_accvgpr" in l) for l in lines)))
        if low: bad[name] = (low, ops)
    return report, bad


if __name__ == "__main__":
    paths = sys.argv[1:] if len(sys.argv) > 1 else [compile_asm(), compile_asm("/tmp/wino2d_check.s", "wino2d.hip")]
    report, bad = [], {}
    for path in paths:
        r_, b_ = check(path)
        report += r_; bad.update(b_)
    for name, n, low, scr, nm, nacc in report:
        print(f"{name}: compiler-used AGPRs {n}, in the pinned range a{PINNED}+: {low if low else 'none'}; scratch instructions {scr}; MFMAs {nm}; accvgpr moves {nacc}")
    if bad:
        print("VIOLATION: the compiler uses pinned accumulator registers:", {k: v[1] for k, v in bad.items()})
        sys.exit(1)
