"""Imports the PYTHON half of the reference (/root/reference) on CPU torch, in the build container only.

The reference hard-codes device="cuda" and imports packages that are not installed here (omegaconf, prettytable,
torchvision, the CUDA rasterizer extension); this harness injects empty stand-in MODULE OBJECTS for those names
into sys.modules (no reference code is copied or re-implemented) and rewrites device='cuda' -> 'cpu' through a
TorchFunctionMode. Used ONLY by tests/tools/gen_golden.py to produce the committed fixtures under tests/golden/.
"""
import sys
import types

import torch
from torch.overrides import TorchFunctionMode

REF = "/root/reference"


class _Any:
    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return _Any()

    def __getattr__(self, n):
        return _Any()


class _AnyModule(types.ModuleType):
    def __getattr__(self, n):
        return _Any()


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def install(rasterizer_settings=None, rasterizer_cls=None):
    if REF not in sys.path:
        sys.path.insert(0, REF)
    _stub('omegaconf', DictConfig=dict, OmegaConf=object)
    _stub('prettytable', PrettyTable=object)
    tv = _stub('torchvision')
    tr = _AnyModule('torchvision.transforms')
    sys.modules['torchvision.transforms'] = tr
    tr.functional = _AnyModule('torchvision.transforms.functional')
    sys.modules['torchvision.transforms.functional'] = tr.functional
    tu = _AnyModule('torchvision.utils')
    sys.modules['torchvision.utils'] = tu
    tv.transforms, tv.utils = tr, tu
    _stub('diff_gof_rasterization', GaussianRasterizationSettings_GOF=rasterizer_settings or object,
          GaussianRasterizer_GOF=rasterizer_cls or object)


class Cuda2Cpu(TorchFunctionMode):
    def __torch_function__(self, func, types_, args=(), kwargs=None):
        kwargs = dict(kwargs or {})
        if 'device' in kwargs and str(kwargs['device']).startswith('cuda'):
            kwargs['device'] = 'cpu'
        if getattr(func, '__name__', '') == 'cuda':
            return args[0]
        return func(*args, **kwargs)
