import os as _os

# fall through to the reference's torchreid/utils for the helpers outside the hot path (see torchreid/__init__.py)
_ref_root = _os.environ.get('AGRL_REFERENCE_ROOT', '')
if _ref_root:
    _ref_utils = _os.path.join(_ref_root, 'torchreid', 'utils')
    if _os.path.isdir(_ref_utils) and _ref_utils not in __path__:
        __path__.append(_ref_utils)
