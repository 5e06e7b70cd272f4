from ._history import History, DeviceHistory
from ._linalg import chol_inv_det, bilinear_sym
from ._regularize import regularize, logsumexp, logsumexp2D
from . import convergence, indicator, util
