#!/usr/bin/env python3
"""Is the headline power-bound?  Socket power, power cap and shader clock while each hot kernel runs in a loop (GPU box).

    python scripts/power_probe.py [seconds per workload]

A child process runs one workload back to back (N = 1e7, K = 32, D = 20: the IS pass, the responsibilities + statistics
of the E-step, the statistics kernel alone via pmc_estep_from_u is not separable here, so: `is`, `estep`, `step`);
the parent samples the amdgpu hwmon files (power1_average / power1_input, power1_cap, freq1_input) every 50 ms, or
`rocm-smi --showpower --showclocks --json` where sysfs is not visible.  Prints per workload: launches per second, mean
and max power against the cap, mean shader clock.
"""
import glob
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import sys, time
sys.path.insert(0, %r)
import numpy as np, torch
from bench import mk, gauss_params, vb_params, K, D, K_T
from pypmc_amd.backend import HipBackend, ComponentSet
what, seconds = sys.argv[1], float(sys.argv[2])
be = HipBackend(0)
N = 10_000_000
mu, cov, w = mk(K, D, 1); tmu, tcov, tw = mk(K_T, D, 11)
inv, ln = gauss_params(mu, cov); tinv, tln = gauss_params(tmu, tcov)
W, beta, nu, ln_pi, ln_lambda = vb_params(mu, cov, w, N)
prop = ComponentSet(0, mu, inv, c0=ln, weight=w); tgt = ComponentSet(0, tmu, tinv, c0=tln, weight=tw)
post = ComponentSet(2, mu, W, c0=D / beta, c1=nu, c2=ln_pi, c3=ln_lambda - D * np.log(2. * np.pi))
pp, pt, pv = be.pack(prop), be.pack(tgt), be.pack(post)
g = torch.Generator(device='cuda').manual_seed(1)
x = torch.randn(N, D, dtype=torch.float64, device='cuda', generator=g) * 1.2
x += torch.tensor(mu, device='cuda')[torch.randint(0, K, (N,), device='cuda', generator=g)]
stats = be.zeros(be.stats_len(K, D))
def one():
    if what in ('is', 'step'): be.importance_weights(x, prop, tgt, pack=pp, target_pack=pt)
    if what in ('estep', 'step'): be.estep(x, post, 0, pack=pv, out=stats)
    if what == 'idle': time.sleep(0.01)
one(); torch.cuda.synchronize()
print('READY', flush=True)
t0 = time.time(); n = 0
while time.time() - t0 < seconds:
    for _ in range(20): one()
    torch.cuda.synchronize(); n += 20
print('DONE %%d %%.3f' %% (n, time.time() - t0), flush=True)
""" % ROOT


def hwmons():
    """every amdgpu hwmon directory the node shows (sysfs lists all of its GPUs, not only the one this box may use)"""
    return sorted(d for d in glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")
                  if os.path.exists(os.path.join(d, "power1_cap")))


def busiest(cands, seconds=1.0):
    """the card that draws the most while OUR workload runs = the one the workload runs on"""
    acc = {d: 0.0 for d in cands}
    t0 = time.time()
    while time.time() - t0 < seconds:
        for d in cands:
            acc[d] += sample(d)[0]
        time.sleep(0.05)
    return max(acc, key=acc.get) if acc else None


def read_int(path):
    try:
        return int(open(path).read().strip())
    except (OSError, ValueError):
        return None


def sample(hw):
    if hw:
        p = read_int(os.path.join(hw, "power1_average"))
        if p is None:
            p = read_int(os.path.join(hw, "power1_input"))
        return (p or 0) * 1e-6, (read_int(os.path.join(hw, "power1_cap")) or 0) * 1e-6, \
            (read_int(os.path.join(hw, "freq1_input")) or 0) * 1e-6
    try:
        out = subprocess.run(["rocm-smi", "--showpower", "--showmaxpower", "--showclocks", "--json"],
                             stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, timeout=10).stdout
        card = list(json.loads(out).values())[0]
        power = cap = clk = 0.0
        for k, v in card.items():
            kl = k.lower()
            try:
                if "power" in kl and "max" in kl:
                    cap = float(str(v).split()[0])
                elif "power" in kl:
                    power = float(str(v).split()[0])
                elif "sclk" in kl and "clock" in kl:
                    clk = float(str(v).strip("()").lower().replace("mhz", ""))
            except ValueError:
                pass
        return power, cap, clk
    except Exception:                                            # noqa: BLE001
        return 0.0, 0.0, 0.0


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 6.0
    cands = hwmons()
    hw = None
    print("%-8s %12s %10s %10s %10s %10s %8s" % ("workload", "launches/s", "mean W", "max W", "cap W", "sclk MHz", "samples"))
    for what in ("is", "estep", "step", "idle"):
        child = subprocess.Popen([sys.executable, "-c", CHILD, what, str(seconds)], stdout=subprocess.PIPE, text=True, cwd=ROOT)
        assert child.stdout.readline().startswith("READY")
        time.sleep(0.5)                                            # let the power average settle into the loop
        if what == "is" and cands:
            hw = busiest(cands)
            print("source:", hw, "(of %d cards in sysfs)" % len(cands))
        ps, cs, caps = [], [], []
        t0 = time.time()
        while time.time() - t0 < seconds - 1.0:
            p, cap, clk = sample(hw)
            ps.append(p), cs.append(clk), caps.append(cap)
            time.sleep(0.05)
        done = child.stdout.readline().split()
        child.wait()
        rate = float(done[1]) / float(done[2]) if len(done) == 3 else 0.0
        print("%-8s %12.1f %10.1f %10.1f %10.1f %10.0f %8d" % (what, rate, sum(ps) / len(ps), max(ps), max(caps),
                                                              sum(cs) / len(cs), len(ps)))


if __name__ == "__main__":
    main()
