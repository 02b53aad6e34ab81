"""Deterministic, formula-defined network weights shared by the fixture generator (applied to the REFERENCE module in
the build container) and the tests (applied to this build's module): no 185 MB checkpoint has to travel."""
import math

import torch


def formula_state_dict(shapes, keep=None):
    """shapes: {key: shape}. Values depend only on (sorted key index, element index): a quasi-random but
    well-conditioned pattern -- conv weights ~ U(-a, a) with a = sqrt(3 / fan_in), norm weights near 1, biases small."""
    keep = keep or {}
    out = {}
    for idx, key in enumerate(sorted(shapes)):
        if key in keep:
            out[key] = keep[key].clone()
            continue
        shape = tuple(shapes[key])
        n = 1
        for s in shape:
            n *= s
        i = torch.arange(n, dtype=torch.float64)
        u = torch.frac(torch.sin(i * 12.9898 + (idx + 1) * 78.233) * 43758.5453)      # in (-1, 1)
        if key.endswith("weight") and len(shape) == 4:
            fan_in = shape[1] * shape[2] * shape[3]
            v = u * math.sqrt(3.0 / fan_in)
        elif key.endswith("weight") and len(shape) == 1:
            v = 1.0 + 0.1 * u
        else:
            v = 0.05 * u
        out[key] = v.reshape(shape).float()
    return out
