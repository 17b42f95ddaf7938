from mmnas_amd.utils.optimizer import WarmupOptimizer  # noqa: F401
