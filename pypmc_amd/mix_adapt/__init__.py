from . import variational, pmc
