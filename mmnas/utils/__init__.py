"""`mmnas.utils`: ops_adapter / optimizer / itm_loss are this repository's; the other modules of the reference's
`mmnas/utils/` (sampler, vqa, vqaEval, answer_punct, bbox_transform, bbox, overlaps) come from the checkout the parent
package found (mmnas/__init__.py)."""
import os as _os

import mmnas as _mmnas

__path__ = [__path__[0]] + [_os.path.join(d, 'utils') for d in _mmnas.__path__[1:] if _os.path.isdir(_os.path.join(d, 'utils'))]
