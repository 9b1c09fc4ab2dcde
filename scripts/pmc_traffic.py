"""Aggregate rocprofv3 --pmc passes of bench.py into the per-launch HBM traffic of the dominant kernel.

Usage (on the GPU box, after two separate counter passes of the same command):

    rocprofv3 --pmc FETCH_SIZE -d gpurun_out/pmc_fetch --output-format csv -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-public-api --no-alone
    rocprofv3 --pmc WRITE_SIZE -d gpurun_out/pmc_write --output-format csv -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-public-api --no-alone
    python3 scripts/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write k_moran_score gpurun_out/pmc_out

Writes <out>/<kernel>_pmc_traffic.json (read by bench.py for roofline.traffic) and
<out>/pmc_fetch_write_by_kernel.csv. Units and the gfx950 correction follow
/opt/skills/guides/MI355X_MICROARCH.md: FETCH_SIZE / WRITE_SIZE are KiB; FETCH_SIZE counts a 128-B request
made of 16-B-per-lane loads as 64 B, so kernels whose loads are all 16 B per lane are doubled (argument
`--double-fetch`, the default for the Moran permutation kernels, which only issue 16-B loads).
"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def short(name: str) -> str:
    m = re.match(r"(?:void )?([A-Za-z_0-9:]+(?:<[^(]*>)?)", name)
    s = m.group(1) if m else name
    return s if len(s) < 80 else s[:80]


def collect(d: str, counter: str):
    acc = defaultdict(lambda: [0, 0.0])
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                if row["Counter_Name"] != counter:
                    continue
                a = acc[short(row["Kernel_Name"])]
                a[0] += 1
                a[1] += float(row["Counter_Value"])
    return acc


def main() -> None:
    fetch_dir, write_dir, kernel, out = sys.argv[1:5]
    double_fetch = "--no-double-fetch" not in sys.argv
    cells = int(os.environ.get("SC_CELLS", 1_000_000))
    perms = int(os.environ.get("SC_PERMS", 1000))
    genes = int(os.environ.get("SC_GENES", 500))
    os.makedirs(out, exist_ok=True)
    fetch = collect(fetch_dir, "FETCH_SIZE")
    write = collect(write_dir, "WRITE_SIZE")
    with open(os.path.join(out, "pmc_fetch_write_by_kernel.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["counter", "kernel", "dispatches", "sum_KiB", "avg_KiB_per_dispatch"])
        for cname, acc in (("FETCH_SIZE", fetch), ("WRITE_SIZE", write)):
            for k, (nd, tot) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
                w.writerow([cname, k, nd, tot, tot / nd])
    # every form of the kernel (k_moran_score_wg for full chunks, k_moran_score for the pipeline's short first / last
    # chunk): bench.py's roofline averages over all scoring launches of a step, so does this
    fk = [k for k in fetch if k.split("<")[0].startswith(kernel)]
    if not fk:
        raise SystemExit(f"kernel {kernel} not in {fetch_dir}: {sorted(fetch)[:20]}")
    nd = sum(fetch[k][0] for k in fk)
    ftot = sum(fetch[k][1] for k in fk)
    wd = sum(write.get(k, [0, 0.0])[0] for k in fk) or nd
    wtot = sum(write.get(k, [0, 0.0])[1] for k in fk)
    f_kib = ftot / nd
    w_kib = wtot / max(wd, 1)
    # r04: the uint8 form streams its lag operand as 4-byte words (16-bit neighbour sums, 256 B per cell and 128-gene group);
    # 4-byte-per-lane loads are counted in full, so that part of FETCH_SIZE is NOT doubled (SC_LAG4B_BYTES: its bytes per
    # launch; default = cells x 256 B x ceil(genes / 128) when a 16-bit-lag form of the kernel was measured)
    lag4 = 0.0
    if any(", true>" in k for k in fk):
        lag4 = float(os.environ.get("SC_LAG4B_BYTES", cells * 256.0 * -(-genes // 128)))
    hbm = (f_kib * (2.0 if double_fetch else 1.0) + w_kib) * 1024.0 - (lag4 if double_fetch else 0.0)
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from spatialcore_amd import _lib

    res = {
        "kernel": kernel,
        "kernel_forms": sorted(fk),
        "source_hash": _lib.source_hash(),
        "cells": cells,
        "perms": perms,
        "genes_per_gpu": genes,
        "dispatches_measured": nd,
        "FETCH_SIZE_KiB_per_launch": f_kib,
        "WRITE_SIZE_KiB_per_launch": w_kib,
        "fetch_doubled": double_fetch,
        "four_byte_lag_bytes_not_doubled": lag4,
        "hbm_bytes_per_launch": hbm,
        "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in two separate passes of "
                  "`python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-public-api --no-alone`, aggregated by scripts/pmc_traffic.py; "
                  "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half the bytes of 16-B-per-lane "
                  "reads; every load of this kernel is 16 B per lane except the 16-bit lag words of the uint8 form, whose "
                  "bytes are subtracted once from the doubled figure); the counters are L2 memory-side requests, "
                  "Infinity-Cache hits included",
    }
    with open(os.path.join(out, f"{kernel}_pmc_traffic.json"), "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
