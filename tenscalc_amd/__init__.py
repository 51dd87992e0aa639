"""tenscalc_amd: batched MI355X-native interior-point back-end for TensCalc's cmex2optimizeCS path."""
from .tcalculus import *  # noqa: F401,F403
from .cmex2optimizeCS import cmex2optimizeCS, SolverSpec  # noqa: F401
