"""Import alias: ``import f3dgaus_amd`` -> the package in ./f3d-gaus_amd (whose directory name is not an identifier)."""
import importlib
import os
import sys

_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)
_pkg = importlib.import_module("f3d-gaus_amd")
sys.modules[__name__] = _pkg
