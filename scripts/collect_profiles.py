#!/usr/bin/env python3
"""gpurun_out/prof (written by scripts/profile_bench.sh on the GPU box) -> profiles/rNN_*:
the rocprofv3 kernel summary, the bench lines of the profiled and the unprofiled run, and the HBM
traffic per launch (FETCH_SIZE x 2 [gfx950 correction] + WRITE_SIZE, KiB -> bytes).

    python scripts/collect_profiles.py r01
"""
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "prof")
DST = os.path.join(ROOT, "profiles")
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"



def one(pattern):
    files = glob.glob(os.path.join(SRC, pattern), recursive=True)
    assert files, "nothing matches " + pattern
    return files[0]


shutil.copy(one("stats/**/*kernel_stats.csv"), os.path.join(DST, tag + "_bench_n1_kernel_stats.csv"))
shutil.copy(os.path.join(SRC, "bench_profiled_stdout.json"), os.path.join(DST, tag + "_bench_n1_stdout.json"))
shutil.copy(os.path.join(SRC, "bench_stdout.json"), os.path.join(DST, tag + "_bench_n1_with_cpu_baseline.json"))
line = json.loads(open(os.path.join(SRC, "bench_profiled_stdout.json")).read().strip().splitlines()[-1])
cfg = line["config"]

per = {}
for counter in ("FETCH_SIZE", "WRITE_SIZE"):
    f = one("pmc_%s/**/*counter_collection.csv" % counter)
    acc = {}
    for row in csv.DictReader(open(f)):
        name, val = row["Kernel_Name"], float(row["Counter_Value"])
        if "k_stats<" in name:                         # the per-component-shift kernel: returns at once in this run
            continue
        for key in ("k_logpdf", "k_resp", "k_stats", "k_estep_fused"):     # pmc_get_timings' kernel names
            if key in name:                            # ("k_stats" = k_stats_gemm, the form pmc_estep runs at K = 32)
                break
        else:
            continue
        acc.setdefault(key, {}).setdefault(row["Dispatch_Id"], 0.0)
        acc[key][row["Dispatch_Id"]] += val
    for key, d in acc.items():
        vals = sorted(d.values())
        per.setdefault(key, {})[counter + "_KiB"] = sum(vals) / len(vals)
for key, d in per.items():
    d["hbm_bytes_per_launch"] = (2 * d["FETCH_SIZE_KiB"] + d["WRITE_SIZE_KiB"]) * 1024
out = {"N": cfg["N_per_gpu"], "K": cfg["K"], "D": cfg["D"],
       "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) on `python bench.py --no-cpu-baseline "
               "--steps 3 --warmup 1`; values in KiB as reported; FETCH_SIZE is doubled (gfx950 correction, "
               "MI355X_MICROARCH.md HBM section) when converted to bytes",
       "kernels": per}
json.dump(out, open(os.path.join(DST, tag + "_traffic_n1.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
print(open(os.path.join(DST, tag + "_bench_n1_kernel_stats.csv")).read()[:900])
