"""Build libpmc_hip.so (the gfx950 kernels + C ABI) in-tree with hipcc.

    python -m pypmc_amd.build [-j JOBS] [--force]

One translation unit per compiled sample dimension (pypmc_amd/csrc/pmc_dims.h) plus the
dispatcher; objects are cached under pypmc_amd/csrc/build/ and linked into
pypmc_amd/lib/libpmc_hip.so.  hipcc cross-compiles for gfx950 without a GPU present.
"""
import argparse
import concurrent.futures as cf
import os
import re
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "build")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libpmc_hip.so")
# Development aid for A/B measurements: PMC_VARIANT=<name> builds lib/libpmc_hip_<name>.so from a copy of the main
# build's objects in which the units listed in PMC_VARIANT_UNITS (comma separated object stems, e.g.
# "pmc_persample_d20_p0,pmc_stats_d20_p0") are recompiled with PMC_EXTRA_FLAGS; PMC_HIP_LIBRARY=<path> makes
# pypmc_amd._lib load it.  The product is always lib/libpmc_hip.so.
VARIANT = os.environ.get("PMC_VARIANT", "")
if VARIANT:
    OBJ = os.path.join(CSRC, "build_" + VARIANT)
    LIB = os.path.join(LIBDIR, "libpmc_hip_%s.so" % VARIANT)
ARCH = "gfx950"

HIPCC_FLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
               "-fno-fast-math", "-Wall", "-Wno-unused-function", "-Wno-unused-variable",
               "-Wno-unused-but-set-variable"]


def hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found (looked on PATH and in /opt/rocm/bin)")
    return exe


def dim_list():
    """[(D, has_padded_variant)] parsed from pmc_dims.h"""
    text = open(os.path.join(CSRC, "pmc_dims.h")).read()
    body = text[text.index("#define PMC_DIM_LIST"):]
    body = body[:body.index("#define PMC_MAX_DIM")]
    dims = [(int(d), kind == "XP") for kind, d in re.findall(r"\b(XP|X)\((\d+)\)", body)]
    assert dims, "no dimensions parsed from pmc_dims.h"
    return dims


def _newer(target, sources):
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.getmtime(s) <= t for s in sources)


def _compile(job):
    out, src, defs, deps, force = job
    if VARIANT:
        stem = os.path.splitext(os.path.basename(out))[0]
        main = os.path.join(CSRC, "build", os.path.basename(out))
        if stem not in os.environ.get("PMC_VARIANT_UNITS", "").split(","):
            if not os.path.exists(main):
                raise RuntimeError("variant build: %s is missing (build the product first)" % main)
            shutil.copy2(main, out)
            return out, 0.0
        force = True
    if not force and _newer(out, deps):
        return out, 0.0
    import time
    t0 = time.time()
    # PMC_EXTRA_FLAGS: development aid (A/B builds of the whole library with extra -D switches; use with --force)
    cmd = [hipcc()] + HIPCC_FLAGS + os.environ.get("PMC_EXTRA_FLAGS", "").split() + defs + ["-c", src, "-o", out]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed: %s\n%s" % (" ".join(cmd), r.stdout))
    return out, time.time() - t0


def build(jobs=None, force=False, verbose=False):
    os.makedirs(OBJ, exist_ok=True)
    os.makedirs(LIBDIR, exist_ok=True)
    headers = [os.path.join(CSRC, h) for h in ("pmc_dims.h", "pmc_internal.h")]
    headers.append(os.path.join(os.path.dirname(HERE), "include", "pmc_hip.h"))
    headers.append(os.path.join(CSRC, "pmc_device.h"))
    headers.append(os.path.join(CSRC, "pmc_convert.h"))
    asrc = os.path.join(CSRC, "pmc_api.hip")
    work = [(os.path.join(OBJ, "pmc_api.o"), asrc, [], [asrc] + headers, force)]
    csrc = os.path.join(CSRC, "pmc_ctx.hip")           # the handle layer (include/pmc_ctx.h): host code only
    ctx_h = os.path.join(os.path.dirname(HERE), "include", "pmc_ctx.h")
    work.append((os.path.join(OBJ, "pmc_ctx.o"), csrc, [], [csrc, ctx_h, headers[2]], force))
    psrc = os.path.join(CSRC, "pmc_p2p.hip")          # the one-shot exchange among the ranks of a node (HIP IPC)
    work.append((os.path.join(OBJ, "pmc_p2p.o"), psrc, [], [psrc] + headers, force))
    vsrc = os.path.join(CSRC, "pmc_vbstate.hip")      # the K-sized half of a VB iteration as kernels (any D <= 64)
    work.append((os.path.join(OBJ, "pmc_vbstate.o"), vsrc, [], [vsrc] + headers, force))
    tsrc = os.path.join(CSRC, "pmc_tiles.hip")        # one unit for all dimensions (PMC_D is not used by it)
    work.append((os.path.join(OBJ, "pmc_tiles.o"), tsrc, ["-DPMC_D=1"], [tsrc] + headers, force))
    # the run-time-dimension unit (sample dimensions beyond the compiled ones): its own kernels plus the per-sample
    # and propose units compiled for "dimension 0"
    for unit, src in (("big", "pmc_big.hip"), ("persample_d0_p0", "pmc_persample.hip"), ("propose_d0_p0", "pmc_propose.hip")):
        src = os.path.join(CSRC, src)
        work.append((os.path.join(OBJ, "pmc_%s.o" % unit), src, ["-DPMC_D=0", "-DPMC_PADDED=0"], [src] + headers, force))
    for d, padded in dim_list():
        for p in ((0, 1) if padded else (0,)):
            # (mgemm: the Mahalanobis forms as one matrix product; one kernel serves the exact and the padded variant)
            for unit in ("persample", "stats", "propose", "fused") + (("mgemm",) if p == 0 else ()):
                src = os.path.join(CSRC, "pmc_%s.hip" % unit)
                work.append((os.path.join(OBJ, "pmc_%s_d%d_p%d.o" % (unit, d, p)), src,
                             ["-DPMC_D=%d" % d, "-DPMC_PADDED=%d" % p], [src] + headers, force))
    # biggest units first so the pool drains evenly
    work.sort(key=lambda j: -int(re.search(r"_d(\d+)_", j[0]).group(1)) if "_d" in j[0] else 0)
    jobs = jobs or min(8, os.cpu_count() or 1)
    objs = []
    with cf.ThreadPoolExecutor(max_workers=jobs) as ex:
        for out, dt in ex.map(_compile, work):
            objs.append(out)
            if verbose and dt:
                print("  built %-28s %5.1fs" % (os.path.basename(out), dt), flush=True)
    for stale in set(os.path.join(OBJ, f) for f in os.listdir(OBJ) if f.endswith(".o")) - set(objs):
        os.remove(stale)
    if force or not _newer(LIB, objs):
        cmd = [hipcc(), "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB] + sorted(objs) + ["-ldl"]
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed: %s\n%s" % (" ".join(cmd), r.stdout))
    return LIB


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("-j", "--jobs", type=int, default=None)
    ap.add_argument("--force", action="store_true")
    args = ap.parse_args(argv)
    lib = build(args.jobs, args.force, verbose=True)
    print(lib)


if __name__ == "__main__":
    sys.exit(main())
