"""Fixed-architecture network for image-text matching -- exported under the reference's module path and class names
(mmnas/model/full_itm.py); implementation shared in nets.py."""
from .nets import Cell_Full, Backbone_Full, NetFullBase


class Net_Full(NetFullBase):
    TASK = 'itm'
