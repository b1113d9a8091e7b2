"""Child process of tests/test_rccl_world1_gpu.py: three cfg5 (or cfg4) optimizer steps on cuda:0 from fixed seeds;
prints one JSON line with the loss bits and a SHA-256 of all parameters after every step.  Run twice by the test:
once plain, once under PARSENET_FORCE_COLLECTIVE=1 with torchrun-style environment variables (WORLD_SIZE=1), so
that init_process_group("nccl", device_id=...), the gloo side group next to it, the broadcast of the pre-trained
state, the all-reduce of the HIP gradient bucket, barrier and destroy_process_group all execute on the one GPU."""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from parsenet_codebase_amd import dp, workloads as W  # noqa: E402


def digest(model):
    h = hashlib.sha256()
    for p in model.parameters():
        h.update(p.detach().cpu().numpy().tobytes())
    return h.hexdigest()


def main():
    workload = sys.argv[1]
    rank, world, device = dp.init_from_env()
    info = {"initialized": dist.is_initialized(), "backend": dist.get_backend() if dist.is_initialized() else None,
            "multi_rank": dp.multi_rank()}
    np.random.seed(99)
    if workload == "cfg5":
        step = W.ParsenetE2EStep(device, batch=2, num_points=2500, pretrain_steps=25, pool=4, pretrain_pool=4)
    else:
        step = W.ParsenetSegStep(device, batch=2, num_points=2000, pool=4)
    rec = []
    for s in range(3):
        np.random.seed(1000 + s)
        loss = step.step()
        rec.append([loss.detach().cpu().numpy().tobytes().hex(), digest(step.model)])
    if dp.multi_rank():
        dist.barrier()
        torch.cuda.synchronize()
    info["side_group"] = step.bucket._side is not None
    info["steps"] = rec
    info["skipped"] = getattr(step, "skipped_steps", 0)
    print(json.dumps(info))
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
