"""Abstract density interface (reference: pypmc/density/base.py:7-66)."""
import numpy as np


class ProbabilityDensity(object):
    """A normalised density q that can be evaluated (log q) and sampled."""
    dim = 0

    def __init__(self):
        raise NotImplementedError('Do not create instances from this class, use derived classes instead.')

    def evaluate(self, x):
        """log q(x) for one point."""
        raise NotImplementedError()

    def multi_evaluate(self, x, out=None):
        """log q(x_n) for every row of ``x``; generic fall-back through ``evaluate``."""
        if out is None:
            out = np.empty(len(x))
        else:
            assert len(out) == len(x)
        for n, point in enumerate(x):
            out[n] = self.evaluate(point)
        return out

    def propose(self, N=1, rng=np.random.mtrand):
        """N samples from q drawn with ``rng``."""
        raise NotImplementedError()
