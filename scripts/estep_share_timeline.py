#!/usr/bin/env python3
"""GaussianInference.E_step at one GPU's share of BASELINE config 4 (N = 1.25e6, K = 64, D = 20) in a loop -- the workload
of `configs.cfg4_share_of_8` -- for a kernel timeline:

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/share_tl -o t -- python scripts/estep_share_timeline.py
    python scripts/estep_share_timeline.py --analyse gpurun_out/share_tl

The analysis cuts the trace into E-steps (an E-step starts with k_pack_build), lists every kernel's mean duration and the
idle gap in front of it, and the step-to-step period: period - kernels - gaps inside = the host time between two E-steps
that nothing on the device overlaps."""
import argparse
import csv
import glob
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

ap = argparse.ArgumentParser()
ap.add_argument("--analyse", default=None)
ap.add_argument("--N", type=int, default=1_250_000)
ap.add_argument("--K", type=int, default=64)
ap.add_argument("--reps", type=int, default=200)
args = ap.parse_args()

if args.analyse:
    files = glob.glob(os.path.join(args.analyse, "**", "*kernel_trace.csv"), recursive=True)
    assert files, "no kernel_trace.csv under " + args.analyse
    rows = []
    for r in csv.DictReader(open(files[0])):
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:56]
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name))
    rows.sort()
    starts = [i for i, r in enumerate(rows) if r[2].startswith("k_pack_build")]
    steps = [rows[a:b] for a, b in zip(starts, starts[1:])]
    steps = steps[len(steps) // 2:]
    acc, n = {}, len(steps)
    for st in steps:
        prev = None
        for s, e, name in st:
            a = acc.setdefault(name, [0, 0.0, 0.0])
            a[0] += 1
            a[1] += (e - s) * 1e-3
            if prev is not None:
                a[2] += max(0, s - prev) * 1e-3
            prev = max(prev or e, e)
    period = sum((b[0][0] - a[0][0]) * 1e-3 for a, b in zip(steps, steps[1:])) / max(1, n - 1)
    print("%d E-steps analysed" % n)
    print("%-58s %6s %10s %12s" % ("kernel (launch order)", "calls", "us/step", "gap before"))
    order = []
    for _, _, name in steps[0]:
        if name not in order:
            order.append(name)
    for name in order:
        c, dur, gap = acc[name]
        print("%-58s %6.1f %10.1f %12.1f" % (name, c / n, dur / n, gap / n))
    busy = sum(v[1] for v in acc.values()) / n
    gaps = sum(v[2] for v in acc.values()) / n
    print("kernels %.1f us + gaps inside an E-step %.1f us; E-step to E-step %.1f us -> host between E-steps %.1f us"
          % (busy, gaps, period, period - busy - gaps))
    sys.exit(0)

import numpy as np  # noqa: E402
from bench import mk  # noqa: E402
from pypmc_amd.density.mixture import create_gaussian_mixture  # noqa: E402
from pypmc_amd.mix_adapt.variational import GaussianInference  # noqa: E402

mix = create_gaussian_mixture(*mk(args.K, 20, 3))
np.random.seed(9)
x = mix.propose(args.N, device=True)
vb = GaussianInference(x, initial_guess=mix)
for _ in range(30):
    vb.E_step()
t0 = time.perf_counter()
for _ in range(args.reps):
    vb.E_step()
print("E_step: %.4f ms" % ((time.perf_counter() - t0) / args.reps * 1e3))
import cProfile  # noqa: E402
import pstats  # noqa: E402
pr = cProfile.Profile()
pr.enable()
for _ in range(args.reps):
    vb.E_step()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
