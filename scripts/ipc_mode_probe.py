#!/usr/bin/env python3
"""Where does HSA_ENABLE_IPC_MODE_LEGACY=0 come from?  (verdict r3 weak 9)

Device memory shared between two PROCESSES -- what RCCL does between the ranks of a node, and what
torch.multiprocessing does with a CUDA tensor -- goes through hipIpcGetMemHandle / hipIpcOpenMemHandle.  This probe runs
that exchange in child interpreters with the variable unset, set to 1 and set to 0, and prints what happened:

    python scripts/ipc_mode_probe.py            (GPU box)  ->  profiles/r04_ipc_mode.txt
"""
import os
import subprocess
import sys
import tempfile

CHILD = r'''
import os, sys
import torch
import torch.multiprocessing as mp

def consumer(q, back):
    t = q.get()                      # hipIpcOpenMemHandle of the producer's allocation
    back.put(float(t.sum().item()))

if __name__ == "__main__":
    mp.set_start_method("spawn")
    q, back = mp.Queue(), mp.Queue()
    p = mp.Process(target=consumer, args=(q, back))
    p.start()
    x = torch.arange(1024, dtype=torch.float64, device="cuda")
    q.put(x)                         # hipIpcGetMemHandle
    print("sum seen by the other process:", back.get(timeout=45), "expected", float(x.sum().item()))
    p.join()
'''

child = os.path.join(tempfile.mkdtemp(prefix="ipc_probe_"), "child.py")      # (spawn re-imports the main module: a real file)
open(child, "w").write(CHILD)
for label, value in (("unset", None), ("1", "1"), ("0", "0")):
    env = dict(os.environ)
    env.pop("HSA_ENABLE_IPC_MODE_LEGACY", None)
    if value is not None:
        env["HSA_ENABLE_IPC_MODE_LEGACY"] = value
    try:
        r = subprocess.run([sys.executable, child], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                           timeout=90)
        lines = [l for l in (r.stdout + r.stderr).splitlines() if l.strip() and "amdgpu.ids" not in l]
        tail = [l for l in lines if "Error" in l or "error" in l or "sum seen" in l][-3:] or lines[-3:]
        print("HSA_ENABLE_IPC_MODE_LEGACY %-5s  rc %d   %s" % (label, r.returncode, " | ".join(tail)[:600]))
    except subprocess.TimeoutExpired:
        print("HSA_ENABLE_IPC_MODE_LEGACY %-5s  timed out after 90 s" % label)
