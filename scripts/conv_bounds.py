"""Per-layer lower bounds of the conv family from a scripts/kernel_breakdown.py log (CPU only): for every launch
max(algorithmic flops / MFMA rate, algorithmic bytes / HBM rate), summed; once at the spec peaks (157.3 TFLOP/s fp32 MFMA, 8 TB/s) and
once at what this box sustains (MFMA at the ~2.0 GHz the chip holds under this load = 131 TFLOP/s; 4.9 TB/s device copy rate,
scripts/micro/hbm_copy.py).  usage: python scripts/conv_bounds.py gpurun_out/breakdown.log"""
import re, sys
rows = []
for l in open(sys.argv[1]):
    m = re.match(r"\s*(\d+)\s+(\S+)\s+(.*?)\s+([\d.]+) us(?:\s+([\d.]+) TF/s)?(?:\s+([\d.]+) GB/s)?", l)
    if m:
        i, name, tag, us, tf, gb = m.groups()
        us = float(us)
        rows.append((name, tag.strip(), us, float(tf) * us * 1e6 if tf else 0.0, float(gb) * us * 1e3 if gb else 0.0))
conv = [r for r in rows if r[0] in ("conv2d", "conv3d")]
T, F = sum(r[2] for r in conv), sum(r[3] for r in conv)
print(f"{len(conv)} conv launches, {T:.0f} us, {F / 1e9:.1f} GFLOP algorithmic = {F / T / 1e6:.1f} TFLOP/s = {F / T / 1e6 / 157.3:.2f} of 157.3")
for pf, pb, label in ((157.3e12, 8e12, "spec peaks (157.3 TFLOP/s, 8 TB/s)"), (157.3e12 * 2.0 / 2.4, 4.9e12, "sustained on this box (131 TFLOP/s at 2.0 GHz, 4.9 TB/s copy rate)")):
    lb = sum(max(r[3] / pf, r[4] / pb) * 1e6 for r in conv)
    hb = [r for r in conv if r[4] / pb > r[3] / pf]
    print(f"{label}: sum of per-layer bounds {lb:.0f} us -> {F / lb / 1e6:.1f} TFLOP/s = {F / lb / 1e6 / 157.3:.2f} of 157.3; "
          f"{len(hb)} launches are HBM-bound by this measure ({sum(r[2] for r in hb):.0f} us today)")
pf, pb = 157.3e12 * 2.0 / 2.4, 4.9e12
out = sorted(((r[2] - max(r[3] / pf, r[4] / pb) * 1e6, r) for r in conv), reverse=True)
print("largest excess over the layer's own sustained bound:")
for d, r in out[:16]:
    b = max(r[3] / pf, r[4] / pb) * 1e6
    print(f"  {r[1]:32s} {r[2]:7.1f} us, bound {b:6.1f} ({'hbm' if r[4] / pb > r[3] / pf else 'mfma'}), excess {d:6.1f}")
print(f"total excess {sum(d for d, _ in out):.0f} us")
