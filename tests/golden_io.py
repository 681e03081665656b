"""Load the committed golden fixtures (tests/golden/*.npz) back into nested dicts."""
import os

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    flat = np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False)
    root = {}
    for key in flat.files:
        parts = key.split("/")
        d = root
        for p in parts[:-1]:
            d = d.setdefault(p, {})
        d[parts[-1]] = flat[key]
    return root


def spec_from_case(case, hyper=None):
    """Build an oracle Spec from the ``case`` dict stored in an update_* fixture."""
    from oracle.update import Spec
    g = lambda k, d=None: (case[k].item() if k in case and case[k].shape == () else
                           (tuple(int(x) for x in case[k]) if k in case else d))
    return Spec(obs=g("obs"), act=g("act"), goal=g("goal", 0) or 0, discrete=bool(g("discrete", False)),
                C=g("C"), Q=g("Q"), latent=g("latent"), enc_features=g("enc_features"),
                enc_hidden=g("enc_hidden"), joint_hidden=g("joint_hidden"), pi_hidden=g("pi_hidden"),
                critic_hidden=g("critic_hidden"), distributional=bool(g("distributional", True)),
                lowerbound=bool(g("lowerbound", True)), max_entropy=bool(g("max_entropy", True)),
                hard_updates=bool(g("hard_updates", False)), T=g("T"), B=g("B"),
                bootstrap=bool(g("bootstrap", False)),
                burn_in=int(g("T") * float(g("burn_in_portion", 0) or 0)),
                gru=str(case["gru"]) if "gru" in case else "")


UPDATE_CASES = ["tqc_small", "tqc_c5q2", "tqc_goal", "sac_min", "tqc_discrete", "tqc_nolb", "sac_boot", "tqc_burn", "gru_zero", "gru_learned", "gru_store"]
ACT_CASES = ["tqc_c5q2", "tqc_goal", "tqc_discrete", "gru_store"]
