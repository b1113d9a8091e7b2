"""src/guard.py of the reference: clamp-then-exp / clamp-then-sqrt (same numeric guards)."""
import torch


def guard_exp(x, max_value=75, min_value=-75):
    return torch.exp(torch.clamp(x, max=max_value, min=min_value))


def guard_sqrt(x, minimum=1e-5):
    return torch.sqrt(torch.clamp(x, min=minimum))
