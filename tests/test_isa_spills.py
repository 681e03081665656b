"""No register spills / scratch in the kernels a default plan launches (device ISA metadata of a fresh hipcc -S; no GPU).

A spilled SGPR costs v_readlane / v_writelane pairs and a spilled VGPR scratch traffic INSIDE the MFMA stream, where every
vector instruction takes MFMA issue time: the hot kernels are written to stay within the file."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

# Every kernel of these files must be spill-free except the ones listed: that covers every kernel a default plan of ANY
# BASELINE config launches (config 2 / 3: k_wstat*, k_wgrad_stat, k_fwd3 / k_chain<2>, k_rowdgrad*, k_rowdot, the tile shapes, the update /
# loss / policy kernels; config 4: + k_chain<1>, the 128x32 / 32x128 tile shapes; config 5: + the Gumbel policy kernels, one-hot,
# im2col / col2im / column sums and the implicit-GEMM convolutions of conv.hip; act(): k_act_layer, k_act_policy<1>, k_act_policy_gauss) and the ring's kernels.
# Allowed to spill: the two-output 64x64 tile (the fallback of critic layer 0 when the weight-stationary kernel does not take a
# launch: few rows), which no default plan of a BASELINE config launches at full size.
MAY_SPILL = {
    "wstat.hip": [], "wgrad.hip": [], "chain.hip": [], "fwdchain.hip": [], "rowdgrad.hip": [], "ring.hip": [], "conv.hip": [],
    "gemm.hip": ["k_gemm_groupedILi4E"],
    "kernels.hip": [],
}
# kernels that must exist (a renamed kernel would silently drop out of the gate above)
MUST_EXIST = {
    "wstat.hip": ["k_wstat"], "wgrad.hip": ["k_wgrad_stat"], "chain.hip": ["k_chainILi1E", "k_chainILi2E"], "fwdchain.hip": ["k_fwd3ILi1E", "k_fwd3ILi2E"],
    "rowdgrad.hip": ["k_rowdgrad", "k_rowdgrad_chainILi1E", "k_rowdgrad_chainILi2E", "k_rowdgrad_chainILi4E", "k_rowdot"],
    "gemm.hip": ["k_gemm_groupedILi5ELi16EE", "k_gemm_groupedILi1ELi16EE", "k_gemm_groupedILi2ELi16EE"],   # 64x64, 128x32, 32x128
    "conv.hip": ["k_conv_fwd_u8", "10k_conv_fwdI", "k_conv_dgrad", "12k_conv_wgradI", "k_conv_wgrad_u8"],
    "kernels.hip": ["6k_prepE", "12k_policy_fwdI", "12k_policy_bwdE", "17k_policy_bwd_dpreE", "19k_head_dgrad_maskedE", "19k_policy_fwd_gumbelE", "19k_policy_bwd_gumbelE", "6k_lossE",
                    "13k_loss_finishE", "13k_head_finishE", "18k_sum_parts_colsumE", "11k_summariesE", "14k_reduce_slabsE",
                    "17k_reduce_partialsE", "13k_adam_polyakE", "k_skinny_wgrad", "14k_stream_wgradI", "17k_boot_lowerboundE",
                    "12k_head_dgradE", "11k_act_layerE", "12k_act_policyI", "18k_act_policy_gaussI", "8k_onehotE", "k_im2col", "13k_col2im_maskE"],
    "ring.hip": ["k_gather_windows", "k_scatter_rows", "k_pack_slots", "k_mc_return", "k_her_relabel", "k_episode_expand", "k_her_vmap"],
}


@pytest.fixture(scope="module")
def isa():
    import isa_report
    if not os.path.exists(isa_report.HIPCC):
        pytest.skip("hipcc not available")
    return isa_report.report(sorted(MAY_SPILL))


@pytest.mark.parametrize("fname", sorted(MAY_SPILL))
def test_default_path_kernels_do_not_spill(isa, fname):
    for pat in MUST_EXIST[fname]:
        assert any(pat in name for name, _ in isa[fname]), f"no kernel of {fname} matches {pat!r}"
    for name, d in isa[fname]:
        if any(pat in name for pat in MAY_SPILL[fname]):
            continue
        assert d["vgpr_spill"] == 0 and d["sgpr_spill"] == 0 and d["scratch"] == 0, (name, d)
