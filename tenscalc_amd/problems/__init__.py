"""Problem statements of the BASELINE configs, written against the
reference's own example scripts."""
from .sls import sls, sls_data  # noqa: F401
from .mpc_dcmotor import mpc_dcmotor, dcmotor_data  # noqa: F401
from .l1l2 import l1l2, l1l2_data  # noqa: F401
from .mpc_quadcopter import mpc_quadcopter, quadcopter_data  # noqa: F401
