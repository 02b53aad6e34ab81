"""Import alias: ``import f3dgaus_amd`` -> the package in ./f3d-gaus_amd (whose directory name is not an identifier).

Every ``f3dgaus_amd.<sub>`` import resolves to the ONE module object ``f3d-gaus_amd.<sub>`` (a meta-path finder maps the names);
without it ``from f3dgaus_amd.diff_gof_rasterization.backward import ...`` would execute the submodules a second time under the
alias name and leave two copies of their module state (workspace caches, classes) in the process."""
import importlib
import importlib.abc
import importlib.util
import os
import sys

_REAL, _ALIAS = "f3d-gaus_amd", __name__
_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)


class _AliasFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path=None, target=None):
        if fullname != _ALIAS and not fullname.startswith(_ALIAS + "."):
            return None
        return importlib.util.spec_from_loader(fullname, self)

    def create_module(self, spec):
        module = importlib.import_module(_REAL + spec.name[len(_ALIAS):])    # the real module object, imported once
        self._real_spec = getattr(self, "_real_spec", {})
        self._real_spec[id(module)] = module.__spec__
        return module

    def exec_module(self, module):
        # importlib has just replaced module.__spec__ by the alias spec (its __package__ is still the real name): put the real one
        # back, or relative imports inside the module warn "__package__ != __spec__.parent" and importlib.reload breaks
        real = getattr(self, "_real_spec", {}).pop(id(module), None)
        if real is not None:
            module.__spec__ = real


if not any(isinstance(f, _AliasFinder) or type(f).__name__ == "_AliasFinder" for f in sys.meta_path):
    sys.meta_path.insert(0, _AliasFinder())
_pkg = importlib.import_module(_REAL)
for _name, _mod in list(sys.modules.items()):
    if _name.startswith(_REAL + "."):
        sys.modules[_ALIAS + _name[len(_REAL):]] = _mod
sys.modules[__name__] = _pkg
