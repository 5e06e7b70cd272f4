#!/usr/bin/env python3
"""How much of each test tolerance the product path really uses (GPU box):

    python scripts/tolerance_margins.py [oracle]

Runs tests/frontend_cases.py on the HIP backend (or the oracle-backed checker) with numpy.testing.assert_allclose
and frontend_cases.assert_rel wrapped: for every call site the worst  |a - b| / (atol + rtol |b|)  seen (1 = the
tolerance is exhausted) and the rtol that would just have held with the same atol.  Call sites that compare with
golden vectors generated from the reference are the ones DESIGN section 4 quotes.
"""
import os
import sys
import traceback

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import frontend_cases  # noqa: E402

SITES = {}


def site():
    for fr in traceback.extract_stack()[::-1]:
        if fr.filename.endswith("frontend_cases.py") and fr.name not in ("assert_rel",):
            return "%s:%d" % (fr.name, fr.lineno)
    return "?"


def record(a, b, rtol, atol):
    a, b = np.asarray(a, dtype=float), np.asarray(b, dtype=float)
    try:
        a, b = np.broadcast_arrays(a, b)
    except ValueError:
        return
    ok = np.isfinite(a) & np.isfinite(b) & (a != b)
    if not ok.any():
        used, need = 0.0, 0.0
    else:
        diff = np.abs(a - b)[ok]
        used = float(np.max(diff / (atol + rtol * np.abs(b[ok]) + 1e-300)))
        excess = np.maximum(diff - atol, 0.0)
        need = float(np.max(excess / np.maximum(np.abs(b[ok]), 1e-300)))
    s = SITES.setdefault(site(), [0.0, 0.0, rtol, atol])
    s[0], s[1] = max(s[0], used), max(s[1], need)


_allclose = np.testing.assert_allclose
_rel = frontend_cases.assert_rel


def allclose(actual, desired, rtol=1e-7, atol=0, *args, **kw):
    record(actual, desired, rtol, atol)
    return _allclose(actual, desired, rtol, atol, *args, **kw)


def rel(a, b, rtol=frontend_cases.RTOL, atol=0.0, what=""):
    record(a, b, rtol, atol)
    return _rel(a, b, rtol, atol, what)


np.testing.assert_allclose = allclose
frontend_cases.assert_rel = rel


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "oracle":
        from oracle_backend import OracleBackend
        be = OracleBackend()
    else:
        from pypmc_amd.backend import HipBackend
        be = HipBackend()
    for case in frontend_cases.ALL_CASES:
        try:
            case(be)
        except Exception as exc:                                   # noqa: BLE001
            print("CASE FAILED", case.__name__, repr(exc)[:200])
    print("%-44s %10s %10s %12s %12s" % ("call site", "rtol", "atol", "used", "rtol needed"))
    for k, (used, need, rtol, atol) in sorted(SITES.items(), key=lambda kv: -kv[1][1]):
        if rtol > 2e-10 or used > 0.1:
            print("%-44s %10.1e %10.1e %12.3g %12.3g" % (k, rtol, atol, used, need))


if __name__ == "__main__":
    main()
