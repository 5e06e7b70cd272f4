from . import importance_sampling
