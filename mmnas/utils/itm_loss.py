from mmnas_amd.utils.itm_loss import BCE_Loss  # noqa: F401
