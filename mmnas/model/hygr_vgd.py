from mmnas_amd.model.hygr_vgd import *  # noqa: F401,F403
from mmnas_amd.model import hygr_vgd as _impl
__all__ = [n for n in dir(_impl) if not n.startswith("_")]
