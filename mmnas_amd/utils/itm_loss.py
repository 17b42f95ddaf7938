"""mmnas/utils/itm_loss.py: the ITM training loss (train_itm.py:248 `loss_fn = BCE_Loss(__C)`)."""
from ..harness import BCE_Loss  # noqa: F401
