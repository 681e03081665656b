#!/usr/bin/env python3
"""Register / spill / scratch report of every kernel in fastdeepqlearning_amd/csrc/*.hip (device ISA metadata, no GPU needed).

    python tools/isa_report.py [file.hip ...]        # table on stdout
Used by tests/test_isa_spills.py: the kernels a default plan launches must not spill."""
import concurrent.futures as cf
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "fastdeepqlearning_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def demangle(names):
    try:
        out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt"] + names, capture_output=True, text=True, check=True).stdout
        return out.strip().split("\n")
    except Exception:   # noqa: BLE001
        return names


def kernels_of(path):
    """[(demangled name, {vgpr, agpr, sgpr, vgpr_spill, sgpr_spill, scratch, lds})] of one .hip file."""
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        subprocess.run([HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "--cuda-device-only", "-S", path, "-o", out,
                        "-I", CSRC] + (["-fno-slp-vectorize"] if os.path.basename(path) == "wstat.hip" else []), check=True, capture_output=True)
        text = open(out).read()
    res = []
    for blk in re.split(r"\n  - \.agpr_count", text)[1:]:
        blk = ".agpr_count" + blk
        g = lambda k: int(re.search(r"\." + k + r":\s+(\d+)", blk).group(1))   # noqa: E731
        res.append((re.search(r"\.name:\s+(\S+)", blk).group(1),
                    dict(vgpr=g("vgpr_count"), agpr=g("agpr_count"), sgpr=g("sgpr_count"), vgpr_spill=g("vgpr_spill_count"),
                         sgpr_spill=g("sgpr_spill_count"), scratch=g("private_segment_fixed_size"), lds=g("group_segment_fixed_size"))))
    names = demangle([n for n, _ in res])
    return [(nm, d) for nm, (_, d) in zip(names, res)]


def report(files=None):
    files = files or sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))
    with cf.ThreadPoolExecutor(max_workers=min(4, len(files))) as ex:
        return dict(zip(files, ex.map(lambda f: kernels_of(os.path.join(CSRC, f)), files)))


if __name__ == "__main__":
    for f, ks in report(sys.argv[1:] or None).items():
        print(f"== {f}")
        for name, d in ks:
            flag = "  <-- SPILLS" if d["vgpr_spill"] or d["sgpr_spill"] or d["scratch"] else ""
            print(f"  vgpr {d['vgpr']:3d} agpr {d['agpr']:3d} sgpr {d['sgpr']:3d} spill v{d['vgpr_spill']} s{d['sgpr_spill']} "
                  f"scratch {d['scratch']:4d}  {name[:150]}{flag}")
