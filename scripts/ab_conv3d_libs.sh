# usage: bash scripts/ab_conv3d_libs.sh <variant .so> [bench_conv3d.py flags]  -- per-layer times of the default library against a variant build
# (scripts/build_variant.sh), interleaved on one box; prints "layer  default_us  variant_us" with the best of three runs each
cd "${GRAFT_REPO_ROOT:?}"
V=$1; shift
for i in 1 2 3; do
  MDF_HIP_LIB=$V timeout -k 10 120 python3 scripts/bench_conv3d.py "$@" 2>/dev/null | grep -v "^sum" | sed -E 's/^(\S+).* ([0-9.]+) us .*/\1 \2/' > /tmp/ab_v$i.txt
  timeout -k 10 120 python3 scripts/bench_conv3d.py "$@" 2>/dev/null | grep -v "^sum" | sed -E 's/^(\S+).* ([0-9.]+) us .*/\1 \2/' > /tmp/ab_d$i.txt
done
python3 - <<'PY'
best = {}
for k in "dv":
    for i in (1, 2, 3):
        for line in open(f"/tmp/ab_{k}{i}.txt"):
            n, t = line.split()
            best[(n, k)] = min(best.get((n, k), 1e9), float(t))
for n in dict.fromkeys(n for n, _ in best):
    print(f"{n:16s} default {best[(n, 'd')]:7.1f} us   variant {best[(n, 'v')]:7.1f} us")
PY
