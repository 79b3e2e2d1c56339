"""Summarise the rocprofv3 --pmc passes of tools/pmc_run.sh into one small JSON.

usage: python tools/pmc_summarize.py <pmc_out_dir> <out.json>

For every *_counter_collection.csv below <pmc_out_dir>: per library kernel (k_*) and
counter, the number of dispatches and the mean counter value per dispatch.  Adds the
derived HBM figures per kernel with the gfx950 correction of MI355X_MICROARCH.md
(FETCH_SIZE counts 128-B requests of 16 B/lane reads at 64 B: doubled for the LDS-DMA
operand reads; WRITE_SIZE is exact).  rocprofv3 reports both in KiB.
"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def short(name):
    m = re.search(r"\b(k_[a-z0-9_]+)", name)
    return m.group(1) if m else None


def main(src, dst, B=633, N=784, mode="invert"):
    acc = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
    csv.field_size_limit(1 << 30)
    for path in sorted(glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True)):
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                k = short(row["Kernel_Name"])
                if k is None:
                    continue
                a = acc[k][row["Counter_Name"]]
                a[0] += 1
                a[1] += float(row["Counter_Value"])
    what = (f"one batched inversion of {B} SPD matrices of order {N}" if mode == "invert" else
            f"gpsr_fit with one Adam iteration (two evaluations per GP) of {B} GPs with n = {N}, Matern-3/2, "
            f"GPSR_FUSED={os.environ.get('GPSR_FUSED', '0')}" if mode == "fit1" else
            f"one whole fit evaluation (covariance, sweeps, alpha, gradient statistics) of {B} GPs with n = {N}, Matern-3/2")
    out = {"command": "tools/pmc_run.sh: rocprofv3 --pmc <group> --kernel-trace, one counter group per run, on "
                      f"tools/pmc_invert.py {B} {N} 1 {mode} ({what})",
           "units": "FETCH_SIZE / WRITE_SIZE in KiB as rocprofv3 reports them; other counters raw",
           "kernels": {k: {c: {"launches": v[0], "mean_per_launch": v[1] / v[0]} for c, v in sorted(cs.items())}
                       for k, cs in sorted(acc.items())}}
    out["hbm_bytes_per_launch"] = {}
    for kname, cs in out["kernels"].items():
        if "FETCH_SIZE" in cs and "WRITE_SIZE" in cs:
            f, w = cs["FETCH_SIZE"]["mean_per_launch"] * 1024, cs["WRITE_SIZE"]["mean_per_launch"] * 1024
            out["hbm_bytes_per_launch"][kname] = {"fetch_raw": f, "fetch_corrected_x2": 2 * f, "write": w,
                                                  "traffic_corrected": 2 * f + w}
    tot = sum(v["traffic_corrected"] * out["kernels"][k]["FETCH_SIZE"]["launches"] for k, v in out["hbm_bytes_per_launch"].items())
    raw = sum((v["fetch_raw"] + v["write"]) * out["kernels"][k]["FETCH_SIZE"]["launches"] for k, v in out["hbm_bytes_per_launch"].items())
    out["hbm_bytes_per_gp"] = {"corrected_upper_bound": tot / B, "uncorrected": raw / B,
                               "note": "all launches of the run (k_* kernels) / number of GPs"}
    out["hbm_note"] = ("x2 is the guide's gfx950 correction for 16 B/lane reads (the LDS-DMA operand chunks); the C-tile "
                       "reads are 8 B/lane and uncalibrated, so the corrected figure is an upper bound")
    with open(dst, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", dst, "kernels:", ", ".join(out["kernels"]))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], *([int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]] if len(sys.argv) > 5 else []))
