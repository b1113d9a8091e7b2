"""Tensor-library operators of ONE cfg5 step by the exact package line that called them (forward and host-side
code; operators of the autograd engine's backward pass have no Python frame and are counted by their name):
python tools/probes/op_lines.py"""
import collections
import os
import sys
import traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from torch.utils._python_dispatch import TorchDispatchMode
from parsenet_codebase_amd import dp, workloads

dp.limit_host_threads()
dev = torch.device("cuda:0")
step = workloads.ParsenetE2EStep(dev, batch=4, num_points=10000, pretrain_steps=2000, pool=16, pretrain_pool=64)
step.warm_paths()
for _ in range(3):
    step.step()
torch.cuda.synchronize()
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
VIEWS = ("view", "reshape", "permute", "transpose", "expand", "slice", "select", "unsqueeze", "squeeze", "detach",
         "alias", "as_strided", "t.default", "unbind", "split", "_unsafe_view", "narrow", "unfold", "lift_fresh",
         "empty", "_local_scalar", "is_", "size", "stride", "numel", "dim", "set_", "record_stream", "_to_copy")


class Lines(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.by_line = collections.Counter()
        self.ops = collections.defaultdict(collections.Counter)

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func).replace("aten.", "")
        if not any(name.startswith(v) for v in VIEWS):
            site = "(autograd engine / no package frame)"
            for fr in reversed(traceback.extract_stack(limit=30)):
                if "parsenet_codebase_amd" in fr.filename and "probes" not in fr.filename:
                    site = "%s:%d %s" % (fr.filename.replace(root + "/parsenet_codebase_amd/", ""), fr.lineno, fr.name)
                    break
            self.by_line[site] += 1
            self.ops[site][name] += 1
        return func(*args, **(kwargs or {}))


with Lines() as m:
    step.step()
    torch.cuda.synchronize()
print("operators (views, allocations and size queries left out) of one step: %d" % sum(m.by_line.values()))
for site, n in m.by_line.most_common(120):
    print("%4d  %-60s %s" % (n, site, ", ".join("%s x%d" % kv for kv in m.ops[site].most_common(12))))
