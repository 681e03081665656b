"""No register spills / scratch in the kernels a default plan launches (device ISA metadata of a fresh hipcc -S; no GPU).

A spilled SGPR costs v_readlane / v_writelane pairs and a spilled VGPR scratch traffic INSIDE the MFMA stream, where every
vector instruction takes MFMA issue time: the hot kernels are written to stay within the file."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

# file -> substrings of the (demangled) kernels that must be spill-free
MUST_BE_CLEAN = {
    "wstat.hip": ["k_wstat"],          # forward forms, gated / fused / plain dgrad forms
    "wgrad.hip": ["k_wgrad_stat"],
}


@pytest.fixture(scope="module")
def isa():
    import isa_report
    if not os.path.exists(isa_report.HIPCC):
        pytest.skip("hipcc not available")
    return isa_report.report(sorted(MUST_BE_CLEAN))


@pytest.mark.parametrize("fname", sorted(MUST_BE_CLEAN))
def test_default_path_kernels_do_not_spill(isa, fname):
    checked = 0
    for name, d in isa[fname]:
        if any(s in name for s in MUST_BE_CLEAN[fname]):
            checked += 1
            assert d["vgpr_spill"] == 0 and d["sgpr_spill"] == 0 and d["scratch"] == 0, (name, d)
    assert checked > 0, f"no kernel of {fname} matched {MUST_BE_CLEAN[fname]}"
