from mmnas_amd.utils.itm_loss import BCE_Loss, Margin_Loss  # noqa: F401
