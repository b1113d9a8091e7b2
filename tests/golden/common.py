"""Helpers shared by the fixture generator (make_golden.py) and the tests that consume the
fixtures: name-seeded weights (so multi-megabyte state dicts need not be stored)."""
import zlib

import numpy as np
import torch


def deterministic_init(module, salt=0):
    """Fill every floating parameter / buffer from a generator seeded by its NAME."""
    with torch.no_grad():
        for name, p in sorted(list(module.named_parameters()) + list(module.named_buffers())):
            if not p.dtype.is_floating_point:
                continue
            g = torch.Generator().manual_seed((zlib.crc32(name.encode()) + salt) % (2 ** 31))
            r = torch.randn(p.shape, generator=g)
            if name.endswith("running_var"):
                p.copy_(1.0 + 0.2 * r.abs())
            elif name.endswith("running_mean"):
                p.copy_(0.1 * r)
            elif p.dim() >= 2:
                fan_in = int(np.prod(p.shape[1:]))
                p.copy_(r / np.sqrt(fan_in))
            elif "bn" in name and name.endswith("weight"):
                p.copy_(1.0 + 0.3 * r)
            else:
                p.copy_(0.1 * r)
    return module
