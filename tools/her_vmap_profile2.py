#!/usr/bin/env python3
"""cProfile of bench.config3_her_vmap's ingest (the facade's write stack from Replay.make) - why it is slower than the bare wrapper."""
import cProfile
import os
import pstats
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench

dev = torch.device("cuda:0")
pr = cProfile.Profile()
pr.enable()
out = bench.config3_her_vmap(dev, episodes=1200, steps=10)
pr.disable()
print({k: v for k, v in out.items() if "ingest" in k})
pstats.Stats(pr).sort_stats("cumulative").print_stats(30)
