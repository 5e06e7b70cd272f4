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


class LocalDensity(object):
    """A local density q(x | y) (reference: pypmc/density/base.py:68-105): the proposal interface of the
    Markov-chain sampler, kept because Gauss / StudentT are specified through their local counterparts."""
    dim = 0
    symmetric = False

    def __init__(self):
        raise NotImplementedError('Do not create instances from this class, use derived classes instead.')

    def evaluate(self, x, y):
        """log q(x | y)."""
        raise NotImplementedError()

    def propose(self, y, rng=np.random.mtrand):
        """One sample from q(. | y) drawn with ``rng``."""
        raise NotImplementedError()

