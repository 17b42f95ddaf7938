from mmnas_amd.model.mixed import *  # noqa: F401,F403
from mmnas_amd.model import mixed as _impl
__all__ = [n for n in dir(_impl) if not n.startswith("_")]
