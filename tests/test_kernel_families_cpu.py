"""CPU: every `__global__` kernel of csrc/*.hip is filed under exactly one measurement family, and the three consumers of that
table (bench.py's rocprof cross-check, scripts/summarize_traffic.py, scripts/summarize_rocprof.py) read it instead of carrying
substring lists of their own.  (Round 3 renamed a kernel and three substring filters silently dropped it: VERDICT r03 weak 1.)"""
import csv
import glob
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "mdf-net_amd", "csrc")


def test_every_global_kernel_is_in_exactly_one_family():
    from mdfnet_hip import kernel_families as KF
    found = KF.globals_in_sources(CSRC)
    assert len(found) >= 50                                  # the scan itself works (51 kernels when written)
    missing = sorted(set(found) - set(KF.KERNEL_FAMILY))
    stale = sorted(set(KF.KERNEL_FAMILY) - set(found))
    assert not missing, f"__global__ kernels without a family in mdfnet_hip/kernel_families.py: {missing}"
    assert not stale, f"kernel_families.py names kernels that csrc/ no longer defines: {stale}"
    # a dict cannot hold a key twice, but the source text can: one family per kernel there too
    text = open(os.path.join(ROOT, "mdf-net_amd", "mdfnet_hip", "kernel_families.py")).read()
    body = text[text.index("KERNEL_FAMILY = {"):text.index("\n}\n")]
    keys = re.findall(r'"([a-z_0-9]+)":', body)
    assert len(keys) == len(set(keys)), sorted(k for k in set(keys) if keys.count(k) > 1)


def test_every_launched_global_is_scanned():
    """The scan's ground truth against a second reading of the sources: every `name<...><<<` / `name<<<` launch site and every
    hipLaunchKernelGGL / &name reference to a *_kernel resolves to a scanned `__global__`."""
    from mdfnet_hip import kernel_families as KF
    found = KF.globals_in_sources(CSRC)
    for f in sorted(os.listdir(CSRC)):
        if not f.endswith((".hip", ".h")):
            continue
        text = open(os.path.join(CSRC, f)).read()
        for m in re.finditer(r"\b([a-z_0-9]+_kernel)\b\s*(?:<[^;{}]*?>)?\s*<<<", text):
            assert m.group(1) in found, (f, m.group(1))


def test_rocprof_names_resolve():
    from mdfnet_hip import kernel_families as KF
    n = "void (anonymous namespace)::conv_lds_kernel<32, 32, 16, 3, 3, 1, 1, 2, 1, 0>((anonymous namespace)::LdsConvParams)"
    assert KF.function_name(n) == "conv_lds_kernel" and KF.family(n) == KF.MFMA_CONV
    assert KF.family("(anonymous namespace)::conv_pair_kernel((anonymous namespace)::PairParams)") == KF.MFMA_CONV
    assert KF.family("void (anonymous namespace)::convtr_all_kernel<16, 8, 2>((anonymous namespace)::ConvParams)") == KF.MFMA_CONV
    assert KF.family("void (anonymous namespace)::warp_vec8_kernel<32>((anonymous namespace)::Params)") == KF.WARP
    assert KF.family("void at::native::vectorized_elementwise_kernel<4, at::native::FillFunctor<float>, std::array<char*, 1ul> >"
                     "(int, at::native::FillFunctor<float>, std::array<char*, 1ul>)") is None
    # a kernel whose name merely CONTAINS one of ours is not ours
    assert KF.family("void other::my_warp_kernel_v2<4>(int)") is None


def test_committed_profiles_have_no_unclassified_kernel_of_ours():
    """Every kernel in an anonymous namespace of the committed kernel-stats files that still exists in csrc/ has a family; the
    conv family of the newest eval profile has the launch count the stats file itself shows for one forward."""
    from mdfnet_hip import kernel_families as KF
    newest = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_bench_cfg2_kernel_stats.csv")))[-1]
    found = KF.globals_in_sources(CSRC)
    with open(newest) as f:
        rows = list(csv.DictReader(f))
    for r in rows:
        fn = KF.function_name(r["Name"])
        if fn in found:
            assert KF.family(r["Name"]) is not None, fn
    fwd = [int(r["Calls"]) for r in rows if KF.function_name(r["Name"]) in ("conv_pair_kernel", "conv_pair_valu_kernel")]
    assert len(fwd) == 1
    launches = sum(int(r["Calls"]) for r in rows if KF.family(r["Name"]) == KF.MFMA_CONV)
    assert launches % fwd[0] == 0, (launches, fwd[0])


def test_consumers_carry_no_name_lists_of_their_own():
    for rel in ("bench.py", "scripts/summarize_traffic.py", "scripts/summarize_rocprof.py"):
        text = open(os.path.join(ROOT, rel)).read()
        assert "kernel_families" in text, rel
        code = "\n".join(l for l in text.splitlines() if not l.strip().startswith("#"))
        code = re.sub(r'""".*?"""', "", code, flags=re.S)
        hits = re.findall(r'"(?:conv_lds_kernel|conv3d_kernel|warp_vec8_kernel|wgrad_lds_kernel)"\s+in\s', code)
        assert not hits, (rel, hits)
