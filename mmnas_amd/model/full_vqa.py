"""Fixed-architecture network for VQA -- exported under the reference's module path and class names
(mmnas/model/full_vqa.py); implementation shared in nets.py."""
from .nets import Cell_Full, Backbone_Full, NetFullBase


class Net_Full(NetFullBase):
    TASK = 'vqa'
