"""No register spills / scratch in the kernels a default plan launches (device ISA metadata of a fresh hipcc -S; no GPU).

A spilled SGPR costs v_readlane / v_writelane pairs and a spilled VGPR scratch traffic INSIDE the MFMA stream, where every
vector instruction takes MFMA issue time: the hot kernels are written to stay within the file."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

# file -> substrings of the (mangled) names of the kernels that must be spill-free: everything a default plan of the
# headline configuration (config 2, T = 50) launches, plus the ring's kernels.  (Length-prefixed tokens such as
# "12k_policy_fwdI" keep a name from matching its longer siblings - k_policy_fwd_gumbel belongs to the discrete config 5.)
MUST_BE_CLEAN = {
    "wstat.hip": ["k_wstat"],          # forward forms, gated / fused / plain dgrad forms
    "wgrad.hip": ["k_wgrad_stat"],     # with and without riders
    "chain.hip": ["k_chain"],
    "rowdgrad.hip": ["k_rowdgrad", "k_rowdot"],
    "gemm.hip": ["k_gemm_groupedILi5ELi16ELi1EE", "k_gemm_groupedILi1ELi16ELi1EE", "k_gemm_groupedILi2ELi16ELi1EE"],   # 64x64, 128x32, 32x128
    "kernels.hip": ["6k_prepE", "12k_policy_fwdI", "12k_policy_bwdE", "6k_lossE", "13k_loss_finishE", "13k_head_finishE",
                    "18k_sum_parts_colsumE", "11k_summariesE", "14k_reduce_slabsE", "17k_reduce_partialsE", "13k_adam_polyakE",
                    "k_skinny_wgrad", "14k_stream_wgradE", "17k_boot_lowerboundE", "12k_head_dgradE", "11k_act_layerE"],
    "ring.hip": ["k_gather_windows", "k_scatter_rows", "k_pack_slots", "k_mc_return", "k_her_relabel", "k_episode_expand", "k_her_vmap"],
}


@pytest.fixture(scope="module")
def isa():
    import isa_report
    if not os.path.exists(isa_report.HIPCC):
        pytest.skip("hipcc not available")
    return isa_report.report(sorted(MUST_BE_CLEAN))


@pytest.mark.parametrize("fname", sorted(MUST_BE_CLEAN))
def test_default_path_kernels_do_not_spill(isa, fname):
    for pat in MUST_BE_CLEAN[fname]:
        hits = [(name, d) for name, d in isa[fname] if pat in name]
        assert hits, f"no kernel of {fname} matches {pat!r}"
        for name, d in hits:
            assert d["vgpr_spill"] == 0 and d["sgpr_spill"] == 0 and d["scratch"] == 0, (name, d)
