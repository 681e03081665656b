#!/usr/bin/env python3
"""Condenses the rocprofv3 passes of `tools/collect_profiles.sh s`: per k_gather_windows launch shape (grid size) the bytes the
L2 exchanged with the fabric (FETCH_SIZE x 2: gfx950 reports half of a wide coalesced read, MI355X_MICROARCH.md "HBM";
WRITE_SIZE as is), the kernel-trace duration and the resulting GB/s, beside the algorithmic bytes of tools/sampler_bench.py."""
import collections
import csv
import glob
import os
import sys

out = sys.argv[1]


def newest(sub, pat):
    f = sorted(glob.glob(os.path.join(out, sub, "**", pat), recursive=True), key=os.path.getmtime)
    return list(csv.DictReader(open(f[-1]))) if f else []


def per_grid(rows, counter, want):
    acc = collections.OrderedDict()
    for r in rows:
        if r["Counter_Name"] != counter or want not in r["Kernel_Name"]:
            continue
        g = int(r.get("Grid_Size", r.get("Grid_Size_X", 0)) or 0)
        a = acc.setdefault(g, [0.0, 0])
        a[0] += float(r["Counter_Value"])
        a[1] += 1
    return acc


def durations(rows, want):
    acc = collections.OrderedDict()
    for r in rows:
        if want not in r["Kernel_Name"]:
            continue
        g = int(r.get("Grid_Size", r.get("Grid_Size_X", 0)) or 0)
        a = acc.setdefault(g, [0.0, 0])
        a[0] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
        a[1] += 1
    return acc


shapes = []
for kern in ("k_gather_windows", "k_conv"):
    fetch = per_grid(newest("pmc_fetch", "*_counter_collection.csv"), "FETCH_SIZE", kern)
    write = per_grid(newest("pmc_write", "*_counter_collection.csv"), "WRITE_SIZE", kern)
    dur = durations(newest("trace", "*_kernel_trace.csv"), kern)
    if not fetch:
        continue
    print(f"# {kern}: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) + --kernel-trace durations (third pass), mean per launch")
    print(f"{'grid':>10s} {'launches':>8s} {'read MB (2 x FETCH_SIZE)':>26s} {'write MB':>10s} {'us':>9s} {'GB/s (PMC bytes)':>17s}")
    for g, (v, n) in fetch.items():
        rd = 2.0 * v / n * 1024 / 1e6
        wr = write[g][0] / write[g][1] * 1024 / 1e6 if g in write else float("nan")
        us = dur[g][0] / dur[g][1] if g in dur else float("nan")
        print(f"{g:10d} {n:8d} {rd:26.2f} {wr:10.2f} {us:9.1f} {(rd + wr) / us * 1e3 if us == us else float('nan'):17.1f}")
        if kern == "k_gather_windows" and wr == wr:
            shapes.append((rd, wr, us, g))

# config 4 at B = 1024 (50 x 1024 rows of 399 floats = 81.72 MB each way): the launch shape whose written bytes are closest,
# for bench.py's sampler_roofline.traffic (quoted only on the csrc revision it was measured on)
if shapes:
    import hashlib
    import json
    want = 50 * 1024 * 399 * 4 / 1e6
    rd, wr, us, g = min(shapes, key=lambda t: abs(t[1] - want))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    h = hashlib.sha256()
    d = os.path.join(root, "fastdeepqlearning_amd", "csrc")
    for f in sorted(glob.glob(os.path.join(d, "*.hip")) + glob.glob(os.path.join(d, "*.h"))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    json.dump({"csrc_sha": h.hexdigest()[:16],
               "config4": {"grid": g, "hbm_read_MB_per_launch": rd, "hbm_write_MB_per_launch": wr, "hbm_bytes_per_launch": (rd + wr) * 1e6,
                           "kernel_trace_us": us, "algorithmic_MB_each_way": want}},
              open(os.path.join(out, "sampler_traffic.json"), "w"))
