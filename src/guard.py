"""src/guard.py of the reference: the clamp-then-exp / clamp-then-sqrt guards, re-exported from
the module that uses them."""
from parsenet_codebase_amd.fitting import guard_exp, guard_sqrt  # noqa: F401
