"""Alias package: the `mmnas.*` import paths of the reference that belong to the operator hot path
(`mmnas.model.*`, `mmnas.utils.{ops_adapter,optimizer,itm_loss}`) resolve to mmnas_amd, the MI355X implementation.

Everything else under the reference's `mmnas/` (a PEP-420 namespace package: `mmnas.loader.*`,
`mmnas.utils.{sampler,vqa,vqaEval,answer_punct,bbox_transform,bbox,overlaps}`; search_vqa.py:17-24,
train_vgd.py:15-21) is out of scope here and must keep resolving to the integrator's own checkout.  A regular
package hides a namespace package of the same name whatever the sys.path order (also when the scripts are run from
the checkout, sys.path[0] = the checkout), so this package WINS for the modules it ships and then extends its
`__path__` with every other `mmnas/` directory on sys.path (and under $MMNAS_REFERENCE_ROOT), after its own.
"""
import os as _os
import sys as _sys


def _other_portions(own, *parts):
    """`<root>/<parts...>` directories of every other sys.path root (and $MMNAS_REFERENCE_ROOT), in sys.path order."""
    roots = list(_sys.path)
    if _os.environ.get('MMNAS_REFERENCE_ROOT'):
        roots.append(_os.environ['MMNAS_REFERENCE_ROOT'])
    own_real, out = _os.path.realpath(own), []
    for r in roots:
        d = _os.path.join(r or _os.getcwd(), *parts)
        if _os.path.isdir(d) and _os.path.realpath(d) != own_real and d not in out:
            out.append(d)
    return out


__path__ = [__path__[0]] + _other_portions(__path__[0], 'mmnas')
