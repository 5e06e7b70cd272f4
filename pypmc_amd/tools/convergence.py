"""Quality measures of weighted samples (reference: pypmc/tools/convergence.py), computed from
three device reductions  S = sum w,  L = sum w log w (zero weights masked),  Q = sum w^2:

    perplexity = exp(-(L/S - log S)) / N          (convergence.py:31-39)
    ess        = S^2 / (N Q)  = 1 / (1 + C^2)      (convergence.py:67-72)
"""
import numpy as np


def _sums(weights, backend):
    from ..backend import get_backend
    be = get_backend(backend)
    w = weights if not isinstance(weights, (list, tuple, range)) else np.asarray(weights, dtype=np.float64)
    if isinstance(w, np.ndarray):
        w = np.asarray(w, dtype=np.float64).reshape(-1)
    sc = be.tohost(be.weight_sums(w))
    n = int(np.prod(w.shape))
    return float(sc[0]), float(sc[1]), float(sc[2]), n


def perp_from_sums(S, L, N):
    return float(np.exp(-(L / S - np.log(S))) / N)


def ess_from_sums(S, Q, N):
    return float(S * S / (N * Q))


def perp(weights, backend=None):
    """Normalised perplexity of the weights: 0 is terrible, 1 perfect."""
    S, L, _, N = _sums(weights, backend)
    return perp_from_sums(S, L, N)


def ess(weights, backend=None):
    """Normalised effective sample size of the weights: 0 is terrible, 1 perfect."""
    S, _, Q, N = _sums(weights, backend)
    return ess_from_sums(S, Q, N)
