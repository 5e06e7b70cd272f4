#!/usr/bin/env python3
"""gpurun_out/sq (scripts/profile_counters.sh) -> profiles/rNN_sq_counters_n1.json: per-launch averages of
the SQ counters for the three bench kernels and the derived quantities quoted in DESIGN.md (effective
clock = SQ_BUSY_CYCLES / 32 shader engines / duration; pipe busy = (4 x non-MFMA VALU instructions +
SQ_VALU_MFMA_BUSY_CYCLES) / 1024 SIMDs / cycles)."""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
out = {}
for f in glob.glob(os.path.join(ROOT, "gpurun_out", "sq", "pass*", "**", "*counter_collection.csv"), recursive=True):
    acc = {}
    for row in csv.DictReader(open(f)):
        name = row["Kernel_Name"]
        key = "k_logpdf" if "k_logpdf" in name else "k_resp" if "k_resp" in name else \
            "k_stats" if "k_stats_gemm" in name else None      # (plain k_stats<...> returns at once in this run)
        if key is None:
            continue
        d = acc.setdefault((key, row["Counter_Name"]), {})
        d[row["Dispatch_Id"]] = d.get(row["Dispatch_Id"], 0.0) + float(row["Counter_Value"])
        dur = acc.setdefault((key, "duration_ns"), {})
        dur[row["Dispatch_Id"]] = float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
    for (key, counter), d in acc.items():
        out.setdefault(key, {})[counter] = sum(d.values()) / len(d)
for key, c in out.items():
    cycles = c["SQ_BUSY_CYCLES"] / 32.0
    c["derived"] = {
        "effective_clock_GHz": cycles / c["duration_ns"],
        "mfma_busy_frac": c["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / cycles,
        "valu_busy_frac_excl_mfma": 4.0 * (c["SQ_INSTS_VALU"] - c["SQ_INSTS_MFMA"]) / 1024.0 / cycles,
        "wave_cycles_waiting_frac": c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"],
        "wave_cycles_issue_stalled_frac": c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"],
    }
    c["derived"]["pipe_busy_frac"] = c["derived"]["mfma_busy_frac"] + c["derived"]["valu_busy_frac_excl_mfma"]
dst = os.path.join(ROOT, "profiles", tag + "_sq_counters_n1.json")
json.dump({"command": "python bench.py --no-cpu-baseline --steps 3 --warmup 1  (N=1e7, K=32, D=20)", "kernels": out},
          open(dst, "w"), indent=1)
for key, c in out.items():
    print(key, {k: round(v, 3) for k, v in c["derived"].items()})
