"""mmnas/utils/itm_loss.py: the ITM training losses (train_itm.py:248 `loss_fn = BCE_Loss(__C)`; both names are
imported at search_itm.py:23 / train_itm.py:20)."""
import torch.nn as nn

from ..harness import BCE_Loss  # noqa: F401


class Margin_Loss(nn.Module):
    """mmnas/utils/itm_loss.py:27-37: hinge on the score differences with margin 0.2, summed over the batch:
    sum(max(0, 0.2 + s_negc - s_pos)) + sum(max(0, 0.2 + s_negi - s_pos))."""

    def __init__(self, __C=None):
        super().__init__()
        self.margin = 0.2

    def forward(self, scores_pos, scores_negc, scores_negi):
        cost_c = (self.margin + scores_negc - scores_pos).clamp(min=0)
        cost_i = (self.margin + scores_negi - scores_pos).clamp(min=0)
        return cost_c.sum() + cost_i.sum()
