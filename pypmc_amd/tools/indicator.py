"""Support indicators for the target density.  Out of the accelerated path (SURVEY section 2,
row 16): host closures that ImportanceSampler consults before calling the user's target
(reference interface: pypmc/tools/indicator/)."""
import operator

import numpy as np


def merge_function_with_indicator(function, indicator, value_outside):
    """f(x) inside the support, ``value_outside`` elsewhere; f is not called outside."""
    if indicator is None:
        return function
    return lambda x: function(x) if indicator(x) else value_outside


def _checked(dim, test):
    def indicator(x):
        if len(x) != dim:
            raise ValueError('input has wrong dimension (%i instead of %i)' % (len(x), dim))
        return bool(test(np.asarray(x)))
    return indicator


def ball(center, radius=1., bdy=True):
    """Indicator of {x : |x - center| <= radius}; the boundary belongs to it iff ``bdy``."""
    c = np.array(center, dtype=float)
    inside = operator.le if bdy else operator.lt
    return _checked(len(c), lambda x: inside(np.linalg.norm(x - c), radius))


def hyperrectangle(lower, upper, bdy=True):
    """Indicator of the axis-parallel box [lower, upper]; faces belong to it iff ``bdy``."""
    lo, up = np.array(lower, dtype=float), np.array(upper, dtype=float)
    if (up <= lo).any():
        raise ValueError('invalid input; found upper <= lower')
    inside = operator.le if bdy else operator.lt
    return _checked(len(lo), lambda x: inside(lo, x).all() and inside(x, up).all())
