"""torchreid -- MI355X-native drop-in for the hot path of weleen/AGRL.pytorch.

This package supplies ``torchreid.models`` (``vmgn``), ``torchreid.metrics``, ``torchreid.losses``,
``torchreid.samplers`` and the two ``torchreid.utils`` helpers the model needs. Everything else the
reference's driver imports (data managers, transforms, optimisers, loggers ...) is outside the hot
path and is NOT re-implemented: set ``AGRL_REFERENCE_ROOT`` to a checkout of the reference and the
remaining ``torchreid.*`` sub-modules resolve from there, so ``train_vidreid_xent_htri.py`` runs
unchanged with this directory first on ``PYTHONPATH``.
"""
import os as _os

__version__ = '0.1.0'

_ref_root = _os.environ.get('AGRL_REFERENCE_ROOT', '')
if _ref_root:
    _ref_pkg = _os.path.join(_ref_root, 'torchreid')
    if _os.path.isdir(_ref_pkg) and _ref_pkg not in __path__:
        __path__.append(_ref_pkg)  # searched AFTER this package: ours win, the rest falls through
    _ref_utils = _os.path.join(_ref_pkg, 'utils')
else:
    _ref_utils = ''
