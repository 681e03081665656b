"""Latency of fdql_agent_act at config-2 dimensions, beside the same
network chain in torch eager on the same arena.  Usage: python tools/act_bench.py [rows ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F

from fastdeepqlearning_amd.core import NativeAgent, make_config


def eager(ag, obs, eps):
    t = ag.tensors

    def mlp(prefix, x, n):
        feats, h = [x], x
        for i in range(n):
            h = F.leaky_relu(F.linear(h, t[f"{prefix}.feature_extractor.{i}.0.weight"],
                                      t[f"{prefix}.feature_extractor.{i}.0.bias"]), 0.01)
            feats.append(h)
        return F.linear(torch.cat(feats, -1), t[f"{prefix}.head.weight"], t[f"{prefix}.head.bias"])

    s = mlp("encoder.joiner", mlp("encoder.visible_layer_encoders.obs_1d", obs, 1), 1)
    mean, log_std = torch.chunk(mlp("actor_critic.actor", s, 1), 2, -1)
    return torch.tanh(mean + log_std.clamp(-20, 2).exp() * eps)


def main():
    dev = torch.device("cuda:0")
    cfg = make_config(17, 6, 2, 256)
    ag = NativeAgent(cfg, dev)
    ag.init_weights(0)
    for rows in [int(a) for a in sys.argv[1:]] or [1, 8, 32, 256]:
        obs = torch.randn(rows, 17, device=dev)
        eps = torch.randn(rows, 6, device=dev)
        res = {}
        for name, fn in (("native", lambda: ag.act(obs, noise=eps, want_info=False)), ("eager", lambda: eager(ag, obs, eps))):
            for _ in range(20):
                fn()
            torch.cuda.synchronize()
            n = 300
            t0 = time.perf_counter()
            for _ in range(n):
                fn()
            torch.cuda.synchronize()
            res[name] = (time.perf_counter() - t0) / n * 1e6
        # GPU-side time of the native chain alone
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(100):
            ag.act(obs, noise=eps, want_info=False)
        e1.record()
        torch.cuda.synchronize()
        print(f"rows={rows:5d}  native {res['native']:7.1f} us/call (GPU stream {e0.elapsed_time(e1) * 10:.1f} us)   "
              f"torch eager {res['eager']:7.1f} us/call", flush=True)


if __name__ == "__main__":
    main()
