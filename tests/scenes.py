"""The synthetic SURVEY 8d scenes live in the product tree (gsrast.workloads: bench.py and the tools use them too); tests import them under the old name."""
from gsrast.workloads import *  # noqa: F401,F403
from gsrast.workloads import _quat_to_rot  # noqa: F401
