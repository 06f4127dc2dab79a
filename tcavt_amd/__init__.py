"""Importable alias of the product package.

The product lives in the directory the build contract names
(``traffic-context-augmented-vehicle-trajectory-prediction-framework-using-multimodal-llm_amd/``),
which is not a valid Python identifier; this shim makes its modules importable as
``tcavt_amd.<module>`` by pointing ``__path__`` at that directory.
"""
import os as _os

_PKG_DIR = _os.path.join(
    _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))),
    "traffic-context-augmented-vehicle-trajectory-prediction-framework-using-multimodal-llm_amd",
)
if not _os.path.isdir(_PKG_DIR):  # pragma: no cover
    raise ImportError(f"product package directory missing: {_PKG_DIR}")
__path__ = [_PKG_DIR]
PACKAGE_DIR = _PKG_DIR
