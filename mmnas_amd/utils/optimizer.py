"""mmnas/utils/optimizer.py: WarmupOptimizer (search_vqa.py:179-191); wraps FlatAdam or any torch optimizer."""
from ..optim import WarmupOptimizer  # noqa: F401
