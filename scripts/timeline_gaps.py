#!/usr/bin/env python3
"""Where a bench step's time outside its three big kernels goes: the kernel timeline of
`rocprofv3 --kernel-trace --output-format csv -- python bench.py --no-cpu-baseline --no-configs --no-traffic`
cut into steps (a step starts with k_logpdf), every launch with its duration and the idle gap in front of it.

    python scripts/timeline_gaps.py gpurun_out/timeline      # directory holding *kernel_trace.csv
"""
import csv
import glob
import os
import sys


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return name.split("(")[0][:60]


def main():
    root = sys.argv[1]
    files = glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True)
    assert files, "no kernel_trace.csv under " + root
    rows = []
    for r in csv.DictReader(open(files[0])):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])))
    rows.sort()
    starts = [i for i, r in enumerate(rows) if r[2].startswith("k_logpdf")]
    steps = [rows[a:b] for a, b in zip(starts, starts[1:])]
    steps = steps[len(steps) // 2:]                      # the timed half (warm-up and timed steps look alike)
    print("%d steps analysed" % len(steps))
    acc = {}
    total = 0.0
    for st in steps:
        prev_end = None
        for s, e, name in st:
            a = acc.setdefault(name, [0, 0.0, 0.0])
            a[0] += 1
            a[1] += (e - s) * 1e-3
            if prev_end is not None:
                a[2] += max(0, s - prev_end) * 1e-3
            prev_end = max(prev_end or e, e)
    for st, nxt in zip(steps, steps[1:]):
        total += (nxt[0][0] - st[0][0]) * 1e-3
    n = len(steps)
    print("%-62s %6s %10s %12s" % ("kernel", "calls", "us/step", "gap before"))
    for name, (c, dur, gap) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
        print("%-62s %6.1f %10.1f %12.1f" % (name, c / n, dur / n, gap / n))
    busy = sum(v[1] for v in acc.values()) / n
    gaps = sum(v[2] for v in acc.values()) / n
    print("kernels %.1f us + gaps inside a step %.1f us per step; step to step %.1f us"
          % (busy, gaps, total / max(1, n - 1)))


if __name__ == "__main__":
    main()
