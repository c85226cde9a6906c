"""wino3d.hip keeps its 48 accumulator tiles in a[64:255] by explicit register names (wino3d_acc.h); the register allocator does
not know they are live.  Every inline-assembly statement clobbers the whole range, so no value of the compiler's can sit there
across one -- this test compiles the file to assembly (no GPU needed) and checks that nothing outside the assembly statements
touches the range, that nothing spills to scratch, and that the generated header is in step with its generator."""
import os
import subprocess
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R, "scripts"))


import pytest


@pytest.mark.parametrize("src", ["wino3d.hip", "wino2d.hip"])
def test_compiler_stays_out_of_the_pinned_accumulators(tmp_path, src):
    import check_pinned_agprs as chk
    asm = chk.compile_asm(str(tmp_path / (src + ".s")), src)
    report, bad = chk.check(asm)
    assert report, "no pinned-accumulator kernel found in the assembly"
    assert not bad, f"the compiler uses pinned accumulator registers: {bad}"
    for name, n_agpr, low, scratch, n_mfma, n_moves in report:
        assert scratch == 0, f"{name}: {scratch} scratch instructions (spills)"
        assert n_mfma >= 3 * 16 * 4, name


def test_generated_header_is_current(tmp_path):
    hdr = os.path.join(R, "mdf-net_amd", "csrc", "wino3d_acc.h")
    before = open(hdr).read()
    subprocess.run([sys.executable, os.path.join(R, "scripts", "gen", "gen_wino3d_acc.py")], check=True, capture_output=True)
    assert open(hdr).read() == before, "mdf-net_amd/csrc/wino3d_acc.h differs from what scripts/gen/gen_wino3d_acc.py writes"
