#!/usr/bin/env python3
"""Per-kernel SQ counters of any command (run on the GPU box):
    python scripts/kprof.py [--filter k_estep] [--out gpurun_out/kprof.json] -- python scripts/kbench.py --D 2 ...
Three rocprofv3 passes (kernel trace + <= 6-8 SQ counters each, no other trace domains) and a table of the
per-launch averages with the derived quantities used in DESIGN.md."""
import argparse
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

SETS = ["SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVES",
        "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS",
        "SQ_INSTS_SALU SQ_INSTS_VMEM SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SMEM"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--filter", default="")
    ap.add_argument("--out", default=None)
    ap.add_argument("--dir", default="/tmp/kprof")
    ap.add_argument("cmd", nargs=argparse.REMAINDER)
    args = ap.parse_args()
    cmd = args.cmd[1:] if args.cmd and args.cmd[0] == "--" else args.cmd
    shutil.rmtree(args.dir, ignore_errors=True)
    os.makedirs(args.dir)
    env = dict(os.environ, TMPDIR="/tmp")
    for i, s in enumerate(SETS):
        r = subprocess.run(["rocprofv3", "--kernel-trace", "--pmc"] + s.split() +
                           ["--output-format", "csv", "-d", os.path.join(args.dir, "pass%d" % i), "-o", "t", "--"] + cmd,
                           cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True, timeout=600)
        if r.returncode != 0:
            print("pass %d rc=%d\n%s" % (i, r.returncode, r.stderr[-2000:]), file=sys.stderr)
    out = {}
    for f in glob.glob(os.path.join(args.dir, "pass*", "**", "*counter_collection.csv"), recursive=True):
        acc = {}
        for row in csv.DictReader(open(f)):
            name = row["Kernel_Name"]
            if args.filter not in name:
                continue
            key = name.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").split("(")[0][:80]
            d = acc.setdefault((key, row["Counter_Name"]), {})
            d[row["Dispatch_Id"]] = d.get(row["Dispatch_Id"], 0.0) + float(row["Counter_Value"])
            dur = acc.setdefault((key, "duration_ns"), {})
            dur[row["Dispatch_Id"]] = float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
            acc.setdefault((key, "vgpr"), {})[row["Dispatch_Id"]] = float(row.get("VGPR_Count", 0) or 0)
            acc.setdefault((key, "lds"), {})[row["Dispatch_Id"]] = float(row.get("LDS_Block_Size", 0) or 0)
        for (key, counter), d in acc.items():
            out.setdefault(key, {})[counter] = sum(d.values()) / len(d)
            out[key]["launches"] = len(d)
    for key, c in sorted(out.items()):
        g = lambda n: c.get(n, float("nan"))
        cycles = g("SQ_BUSY_CYCLES") / 32.0                     # per shader engine -> chip cycles
        der = {
            "ms": g("duration_ns") * 1e-6,
            "clock_GHz": cycles / g("duration_ns"),
            "mfma_busy": g("SQ_VALU_MFMA_BUSY_CYCLES") / 1024.0 / cycles,
            "valu_busy_excl_mfma": 4.0 * (g("SQ_INSTS_VALU") - g("SQ_INSTS_MFMA")) / 1024.0 / cycles,
            "waves_per_simd": g("SQ_WAVE_CYCLES") * 4 / 1024.0 / cycles,   # quad-cycles of resident waves / SIMD cycles
            "waiting_frac": g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES"),
            "issue_stalled_frac": g("SQ_WAIT_INST_ANY") / g("SQ_WAVE_CYCLES"),
            "active_frac": g("SQ_ACTIVE_INST_ANY") / g("SQ_WAVE_CYCLES"),
            "lds_conflict_frac": g("SQ_LDS_BANK_CONFLICT") / max(g("SQ_LDS_IDX_ACTIVE"), 1.0),
            "lds_busy": g("SQ_LDS_IDX_ACTIVE") / 256.0 / cycles,
            "valu_per_wave": g("SQ_INSTS_VALU") / g("SQ_WAVES"),
            "mfma_per_wave": g("SQ_INSTS_MFMA") / g("SQ_WAVES"),
            "smem_per_wave": g("SQ_INSTS_SMEM") / g("SQ_WAVES"),
            "lds_per_wave": g("SQ_INSTS_LDS") / g("SQ_WAVES"),
            "salu_per_wave": g("SQ_INSTS_SALU") / g("SQ_WAVES"),
            "vmem_per_wave": g("SQ_INSTS_VMEM") / g("SQ_WAVES"),
        }
        der["pipe_busy"] = der["mfma_busy"] + der["valu_busy_excl_mfma"]
        c["derived"] = der
        print("%s  (%d launches, vgpr %d, lds %d)" % (key, c["launches"], g("vgpr"), g("lds")))
        print("   " + "  ".join("%s=%.3g" % kv for kv in der.items()))
    if args.out:
        json.dump(out, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
