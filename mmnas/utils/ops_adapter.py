from mmnas_amd.utils.ops_adapter import *  # noqa: F401,F403
from mmnas_amd.utils.ops_adapter import OpsAdapter  # noqa: F401
