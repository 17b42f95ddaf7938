from mmnas_amd.model.full_itm import *  # noqa: F401,F403
from mmnas_amd.model import full_itm as _impl
__all__ = [n for n in dir(_impl) if not n.startswith("_")]
