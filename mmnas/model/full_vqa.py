from mmnas_amd.model.full_vqa import *  # noqa: F401,F403
from mmnas_amd.model import full_vqa as _impl
__all__ = [n for n in dir(_impl) if not n.startswith("_")]
