"""Condense the rocprofv3 outputs of tools/collect_profiles.sh into the small files kept under profiles/."""
import collections
import csv
import glob
import json
import os
import sys

out = sys.argv[1]


def counter_rows(sub):
    f = sorted(glob.glob(os.path.join(out, sub, "**", "*_counter_collection.csv"), recursive=True), key=os.path.getmtime)
    return list(csv.DictReader(open(f[-1]))) if f else []   # newest run (gpurun merges runs into one directory)


def per_kernel(rows, counter):
    """(kernel name, grid) -> mean counter value per dispatch, in dispatch order of first appearance."""
    acc = collections.OrderedDict()
    for r in rows:
        if r["Counter_Name"] != counter or "fdql::" not in r["Kernel_Name"]:
            continue
        key = (r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0], int(r.get("Grid_Size", r.get("Grid_Size_X", 0)) or 0))
        a = acc.setdefault(key, [0.0, 0])
        a[0] += float(r["Counter_Value"])
        a[1] += 1
    return acc


DENSE = ("k_gemm", "k_chain", "k_fwd3", "k_rowdgrad", "k_wstat", "k_wgrad_stat", "k_conv")   # the MFMA kernels
fetch = per_kernel(counter_rows("pmc_fetch"), "FETCH_SIZE")
write = per_kernel(counter_rows("pmc_write"), "WRITE_SIZE")
WORKLOAD = sys.argv[2] if len(sys.argv) > 2 else "tools/profile_stages.py (config 2, T=50, B=256)"
lines = ["# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (two separate passes, --kernel-trace only) over",
         f"# {WORKLOAD}; mean per dispatch.  HBM_read = 2 x FETCH_SIZE KiB (gfx950",
         "# reports half of wide coalesced reads, MI355X_MICROARCH.md; check: k_reduce_slabs must read 32 slabs x 4.07 MB",
         "# = 130 MB).  HBM_write = WRITE_SIZE KiB.  Values are L2<->fabric traffic (Infinity-Cache hits included)."]
tot = collections.defaultdict(lambda: [0.0, 0.0, 0])
for key in fetch:
    rd = 2.0 * fetch[key][0] / fetch[key][1] * 1024 / 1e6
    wr = (write[key][0] / write[key][1] * 1024 / 1e6) if key in write else float("nan")
    lines.append(f"{key[0]:60s} grid={key[1]:9d} dispatches={fetch[key][1]:4d} HBM_read_MB={rd:9.2f} HBM_write_MB={wr:9.2f}")
    t = tot[key[0]]
    t[0] += rd * fetch[key][1]
    t[1] += wr * fetch[key][1]
    t[2] += fetch[key][1]
open(os.path.join(out, "hbm_traffic_pmc.txt"), "w").write("\n".join(lines) + "\n")
# dominant kernel = the kernel instantiation bench.py names in its roofline object
shape_id = {"128x128": 0, "128x32": 1, "32x128": 2, "64x128": 3, "64x64dual": 4, "64x64": 5, "64x64hf": 6}
dom_shape = "64x64"
try:
    for line in open(os.path.join(out, "bench_n1.json")):
        if line.startswith("{"):
            kn = json.loads(line)["roofline"]["kernel"]
            dom_shape = kn.split("<")[1].split(" ")[0] if "<" in kn else ("rows" if "k_rowgemm" in kn else "chain")
except (OSError, KeyError, IndexError, ValueError):
    pass
dom_tag = f"k_gemm_grouped<{shape_id[dom_shape]}," if dom_shape in shape_id else ("k_rowgemm" if dom_shape == "rows" else "k_chain")
try:
    if "k_wstat" in kn:
        dom_tag = kn.split(" (")[0].rstrip(">")   # bench.py names the instantiation: "k_wstat<0, 0, 2> (weight-stationary ..."; rocprofv3's name has one more template argument
    elif "k_wgrad_stat" in kn:
        dom_tag = "k_wgrad_stat"
except NameError:
    pass
dom = [k for k in tot if dom_tag in k]


def csrc_hash():
    import hashlib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    h = hashlib.sha256()
    d = os.path.join(root, "fastdeepqlearning_amd", "csrc")
    for f in sorted(glob.glob(os.path.join(d, "*.hip")) + glob.glob(os.path.join(d, "*.h"))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


traffic = None
if dom:
    t = tot[dom[0]]
    traffic = {"kernel": dom[0], "dispatches_measured": t[2], "hbm_read_MB_per_launch": t[0] / t[2],
               "hbm_write_MB_per_launch": t[1] / t[2], "hbm_bytes_per_launch": (t[0] + t[1]) / t[2] * 1e6,
               "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; FETCH_SIZE doubled (gfx950)",
               "workload": "config 2, T=50, B=256", "csrc_sha": csrc_hash()}
# matrix-pipe utilisation per GEMM kernel: SQ_VALU_MFMA_BUSY_CYCLES sums the busy cycles of the 1024 SIMDs' MFMA pipes,
# SQ_BUSY_CYCLES the cycles the 32 shader engines (8 XCD x 4) had work: utilisation = (MFMA/1024) / (BUSY/32); the
# kernel trace of the same pass gives the duration, hence the shader clock the kernel actually ran at.
sq_rows = counter_rows("pmc_sq")
if sq_rows:
    per = collections.OrderedDict()
    for r in sq_rows:
        if not any(t in r["Kernel_Name"] for t in DENSE):
            continue
        key = (r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0], int(r.get("Grid_Size", r.get("Grid_Size_X", 0)) or 0))
        d = per.setdefault(key, collections.defaultdict(float))
        d[r["Counter_Name"]] += float(r["Counter_Value"])
        d["n_" + r["Counter_Name"]] += 1
    dur = collections.defaultdict(lambda: [0.0, 0])
    kt = sorted(glob.glob(os.path.join(out, "pmc_sq", "**", "*_kernel_trace.csv"), recursive=True), key=os.path.getmtime)
    if kt:
        for r in csv.DictReader(open(kt[-1])):
            if any(t in r["Kernel_Name"] for t in DENSE):
                key = (r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0], int(r.get("Grid_Size", r.get("Grid_Size_X", 0)) or 0))
                dur[key][0] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
                dur[key][1] += 1
    sq_lines = ["# rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY over",
                "# tools/profile_stages.py --reps 1 (config 2, T=50, B=256); per dispatch means.",
                "# mfma_util = (MFMA_BUSY / 1024 SIMDs) / (BUSY_CYCLES / 32 shader engines); clock = (BUSY_CYCLES / 32) / duration;",
                "# waves = SQ_WAVE_CYCLES x 4 / 1024 / (BUSY_CYCLES / 32) = resident waves per SIMD; parked = SQ_WAIT_ANY / SQ_WAVE_CYCLES",
                "# (waves sitting at s_waitcnt / s_barrier)."]
    for key, d in per.items():
        n = max(d["n_SQ_BUSY_CYCLES"], 1.0)
        busy = d["SQ_BUSY_CYCLES"] / n / 32.0
        mf = d["SQ_VALU_MFMA_BUSY_CYCLES"] / max(d["n_SQ_VALU_MFMA_BUSY_CYCLES"], 1.0) / 1024.0
        wc = d["SQ_WAVE_CYCLES"] / max(d["n_SQ_WAVE_CYCLES"], 1.0)
        wa = d["SQ_WAIT_ANY"] / max(d["n_SQ_WAIT_ANY"], 1.0)
        us = dur[key][0] / dur[key][1] if dur[key][1] else float("nan")
        sq_lines.append(f"{key[0]:58s} grid={key[1]:8d} n={int(n):3d} dur_us={us:8.1f} mfma_util={mf / busy if busy else 0:5.3f} "
                        f"clock_GHz={busy / us / 1e3 if us == us and us > 0 else float('nan'):5.2f} waves_per_simd={wc * 4 / 1024 / busy if busy else 0:4.1f} "
                        f"parked={wa / wc if wc else 0:4.2f}")
    open(os.path.join(out, "mfma_utilisation_pmc.txt"), "w").write("\n".join(sq_lines) + "\n")
    if traffic is not None:
        wsum = usum = csum = 0.0
        for key, d in per.items():
            if dom_tag not in key[0] or not dur[key][1]:
                continue
            n = max(d["n_SQ_BUSY_CYCLES"], 1.0)
            busy = d["SQ_BUSY_CYCLES"] / n / 32.0
            mf = d["SQ_VALU_MFMA_BUSY_CYCLES"] / max(d["n_SQ_VALU_MFMA_BUSY_CYCLES"], 1.0) / 1024.0
            us = dur[key][0] / dur[key][1]
            wgt = us * dur[key][1]
            wsum += wgt; usum += wgt * (mf / busy if busy else 0.0); csum += wgt * (busy / us / 1e3)
        if wsum > 0:
            traffic["mfma_pipe_busy_frac"] = round(usum / wsum, 3)
            traffic["shader_clock_ghz_under_load"] = round(csum / wsum, 2)
            traffic["pmc_source"] = ("rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES, time-weighted over the "
                                     "kernel's launches; the 157.3 TFLOP/s peak assumes 2.4 GHz")
        # the same figures for every dense kernel: two launches of a step are within a microsecond of each other
        # (k_wstat<0,0,2> and k_wgrad_stat at config 2), and which of them bench.py finds "dominant" changes from run to run
        allk = {}
        for name, t in tot.items():
            if not any(tag in name for tag in DENSE) or not t[2]:
                continue
            e = {"hbm_read_MB_per_launch": t[0] / t[2], "hbm_write_MB_per_launch": t[1] / t[2],
                 "hbm_bytes_per_launch": (t[0] + t[1]) / t[2] * 1e6, "dispatches_measured": t[2]}
            wsum = usum = csum = 0.0
            for key, d in per.items():
                if key[0] != name or not dur[key][1]:
                    continue
                n = max(d["n_SQ_BUSY_CYCLES"], 1.0)
                busy = d["SQ_BUSY_CYCLES"] / n / 32.0
                mf = d["SQ_VALU_MFMA_BUSY_CYCLES"] / max(d["n_SQ_VALU_MFMA_BUSY_CYCLES"], 1.0) / 1024.0
                us = dur[key][0] / dur[key][1]
                wgt = us * dur[key][1]
                wsum += wgt; usum += wgt * (mf / busy if busy else 0.0); csum += wgt * (busy / us / 1e3)
            if wsum > 0:
                e["mfma_pipe_busy_frac"] = round(usum / wsum, 3)
                e["shader_clock_ghz_under_load"] = round(csum / wsum, 2)
            allk[name] = e
        traffic["kernels"] = allk
# instruction mix per dense kernel: what shares the issue stream with the fp32 MFMAs.  Cost model (tools/proto/mfma_coissue*.hip,
# profiles/r02_rowgemm_notes.txt): an MFMA 32x32x2 holds the SIMD 64 cycles; a VALU instruction beside it ~5 (9 for the first
# after an MFMA), an LDS instruction ~7, a vector-memory instruction 25-57: predicted fraction of the MFMA rate
#   = 64 M / (64 M + 5 V + 7 L + 40 G)   with M = MFMA, V = other VALU, L = LDS, G = vector-memory instructions.
ins_rows = counter_rows("pmc_insts")
if ins_rows:
    per = collections.OrderedDict()
    for r in ins_rows:
        if not any(t in r["Kernel_Name"] for t in DENSE):
            continue
        key = (r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0], int(r.get("Grid_Size", r.get("Grid_Size_X", 0)) or 0))
        d = per.setdefault(key, collections.defaultdict(float))
        d[r["Counter_Name"]] += float(r["Counter_Value"])
        d["n_" + r["Counter_Name"]] += 1
    il = ["# rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU over",
          "# tools/profile_stages.py --reps 1 (config 2, T=50, B=256); wave-instructions per dispatch (means over a kernel's dispatches of one grid).",
          "# other_VALU = SQ_INSTS_VALU - SQ_INSTS_MFMA.  per_MFMA columns: instructions issued per MFMA instruction.",
          "# predicted = 64 M / (64 M + 5 V + 7 L + 40 G): fraction of the bare fp32-MFMA rate left by the instruction mix alone",
          "# (no stalls, no start-up); the measured fraction of a launch is in mfma_utilisation_pmc.txt (mfma_util)."]
    for key, d in per.items():
        g = lambda c: d[c] / max(d["n_" + c], 1.0)   # noqa: E731
        M = g("SQ_INSTS_MFMA")
        if M <= 0:
            continue
        V, L, G, S = g("SQ_INSTS_VALU") - M, g("SQ_INSTS_LDS"), g("SQ_INSTS_VMEM_RD") + g("SQ_INSTS_VMEM_WR"), g("SQ_INSTS_SALU")
        pred = 64 * M / (64 * M + 5 * max(V, 0) + 7 * L + 40 * G)
        il.append(f"{key[0]:58s} grid={key[1]:8d} n={int(d['n_SQ_INSTS_MFMA']):3d} MFMA={M:12.0f} other_VALU/MFMA={V / M:6.3f} LDS/MFMA={L / M:6.3f} "
                  f"VMEM/MFMA={G / M:6.3f} SALU/MFMA={S / M:6.3f} predicted={pred:5.3f}")
    open(os.path.join(out, "valu_per_mfma.txt"), "w").write("\n".join(il) + "\n")
if traffic is not None:
    json.dump(traffic, open(os.path.join(out, "dominant_kernel_traffic.json"), "w"), indent=1)
# kernel stats csv of the bench run
st = sorted(glob.glob(os.path.join(out, "stats", "**", "*kernel_stats.csv"), recursive=True), key=os.path.getmtime)
if st:
    rows = list(csv.DictReader(open(st[-1])))
    with open(os.path.join(out, "rocprofv3_kernel_stats_bench.csv"), "w") as f:
        w = csv.DictWriter(f, fieldnames=rows[0].keys())
        w.writeheader()
        w.writerows(rows)
print("summaries written to", out)
