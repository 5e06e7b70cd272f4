#!/usr/bin/env python3
"""Run the REFERENCE's own unit tests for the hot path against this package (build container only).

    python scripts/run_reference_tests.py [--gpu] > profiles/r02_reference_tests.txt

The reference's test modules (``/root/reference/pypmc/**/*_test.py``) are imported where they lie -- nothing
is copied -- with the name ``pypmc`` bound to ``pypmc_amd``: a meta-path finder answers every import of
``pypmc.X`` with the module object of ``pypmc_amd.X``, and the sub-packages' search paths are extended by
the reference's directories (behind this package's own), so that ``pypmc.density.mixture_test`` is found
there while ``pypmc.density.mixture`` is this package's module.  Two pure-Python helper modules the tests
use and this package has no counterpart of come from the reference the same way
(``pypmc.tools._probability_densities``: test densities).

Without a GPU (this container) the front-end runs on the oracle-backed checker of the CPU test-suite, i.e.
what is tested is the host side of the drop-in: signatures, attributes, error behaviour, update logic,
pruning, convergence control -- by the reference's own assertions.  ``--gpu`` uses the HIP backend
instead (only meaningful where both a GPU and /root/reference exist).

Test classes outside the path (Markov chains, hierarchical clustering, r-value,
plotting, MPI sampler, partition) are not run.
"""
import argparse
import importlib
import importlib.abc
import importlib.machinery
import os
import sys
import unittest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/pypmc"
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

MODULES = ["pypmc.density.base_test", "pypmc.tools.linalg_test", "pypmc.tools.regularize_test", "pypmc.tools.convergence_test",
           "pypmc.density.gauss_test", "pypmc.density.student_t_test", "pypmc.density.mixture_test",
           "pypmc.sampler.importance_sampling_test", "pypmc.mix_adapt.pmc_test", "pypmc.mix_adapt.variational_test"]


class _Alias(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    """pypmc[.X] -> the module object of pypmc_amd[.X]"""

    def find_spec(self, fullname, path=None, target=None):
        if fullname != "pypmc" and not fullname.startswith("pypmc."):
            return None
        return importlib.machinery.ModuleSpec(fullname, self)

    def create_module(self, spec):
        return importlib.import_module("pypmc_amd" + spec.name[len("pypmc"):])

    def exec_module(self, module):
        pass


def install(gpu):
    import pypmc_amd
    from pypmc_amd import backend
    if not gpu:
        from oracle_backend import OracleBackend
        backend.set_default_backend(OracleBackend())
    for sub in ("tools", "density", "sampler", "mix_adapt"):
        pkg = importlib.import_module("pypmc_amd." + sub)
        pkg.__path__.append(os.path.join(REF, sub))        # behind this package's own directory
    sys.meta_path.insert(0, _Alias())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpu", action="store_true")
    ap.add_argument("-v", "--verbose", action="store_true")
    args = ap.parse_args()
    if not os.path.isdir(REF):
        sys.exit("needs /root/reference (build container only)")
    install(args.gpu)
    total = failed = 0
    print("# the reference's unit tests (pypmc 1.2.6, /root/reference/pypmc/**/*_test.py) run against pypmc_amd, backend: %s"
          % ("HIP" if args.gpu else "oracle-backed checker (CPU)"))
    for name in MODULES:
        try:
            mod = importlib.import_module(name)
        except Exception as e:  # noqa: BLE001
            print("%-45s IMPORT FAILED: %s: %s" % (name, type(e).__name__, e))
            failed += 1
            continue
        suite = unittest.defaultTestLoader.loadTestsFromModule(mod)
        res = unittest.TestResult()
        suite.run(res)
        bad = {t.id(): tb for t, tb in res.failures + res.errors}
        nskip = len(res.skipped)
        print("%-45s %3d run, %3d ok, %d failed, %d skipped" % (name, res.testsRun, res.testsRun - len(bad) - nskip,
                                                              len(bad), nskip))
        for tid, tb in sorted(bad.items()):
            last = [ln for ln in tb.strip().splitlines() if ln.strip()][-1]
            print("    FAIL %s: %s" % (tid.split(".", 3)[-1], last[:200]))
            if args.verbose:
                print(tb)
        total += res.testsRun
        failed += len(bad)
    print("# total: %d tests, %d failed" % (total, failed))
    return 1 if failed else 0


if __name__ == "__main__":
    sys.exit(main())
