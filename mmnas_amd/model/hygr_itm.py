"""Supernet for image-text matching -- exported under the reference's module path and class names
(mmnas/model/hygr_itm.py); implementation shared in nets.py."""
from .nets import Cell_Search, Backbone_Search, NetSearchBase


class Net_Search(NetSearchBase):
    TASK = 'itm'
