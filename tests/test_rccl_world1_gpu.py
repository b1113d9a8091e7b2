"""The RCCL code path of the data-parallel step on the ONE GPU a test box has.

No multi-GPU node is available to this suite, so no scaling curve is measured here; what these tests pin is that the
collective path EXECUTES on a HIP device — `init_process_group("nccl", device_id=...)`, the host-side gloo group
beside the NCCL world, the rank-0 pre-training + flat broadcast, the all-reduce of the HIP gradient bucket, barrier,
`destroy_process_group` — and that a step through it equals the non-collective step bit for bit.
Reference: torch.nn.DataParallel in /root/reference/train_parsenet.py:90-91,183 (mean over replicas)."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = os.path.join(ROOT, "tests", "helpers", "rccl_world1_child.py")


def _free_port():
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def _env(forced):
    env = dict(os.environ)
    env.pop("PARSENET_FORCE_COLLECTIVE", None)
    env["PARSENET_HOST_THREADS"] = "1"
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if forced:
        env.update(PARSENET_FORCE_COLLECTIVE="1", WORLD_SIZE="1", RANK="0", LOCAL_RANK="0",
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    return env


def _run(cmd, forced, timeout=900):
    r = subprocess.run(cmd, env=_env(forced), capture_output=True, text=True, timeout=timeout, cwd=ROOT)
    assert r.returncode == 0, "child failed:\n%s\n%s" % (r.stdout[-3000:], r.stderr[-3000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert lines, r.stdout[-2000:] + r.stderr[-2000:]
    return json.loads(lines[-1])


@pytest.mark.gpu
@pytest.mark.parametrize("workload", ["cfg4", "cfg5"])
def test_forced_collective_step_equals_the_plain_step(gpu, workload):
    plain = _run([sys.executable, CHILD, workload], False)
    forced = _run([sys.executable, CHILD, workload], True)
    assert not plain["initialized"] and not plain["multi_rank"]
    assert forced["initialized"] and forced["backend"] == "nccl" and forced["multi_rank"]
    if workload == "cfg5":
        # the step status was agreed upon on the host-side gloo group created beside the NCCL world
        assert forced["side_group"]
    assert forced["skipped"] == plain["skipped"] == 0
    assert forced["steps"] == plain["steps"]          # loss bits and the SHA-256 of all parameters, every step


@pytest.mark.gpu
def test_bench_line_through_the_forced_collective_path(gpu):
    """`bench.py --workload cfg4 --steps 3` with WORLD_SIZE=1 and the collectives forced: barrier-bracketed timing,
    MAX over ranks with all-reduces on the device, the profiled steps with their all-reduce, process-group teardown."""
    out = _run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "cfg4", "--steps", "3", "--warmup", "1",
                "--no-cpu-baseline", "--profile-steps", "1"], True)
    assert out["n_gpus"] == 1 and out["world_size_observed"] == 1
    assert out["value"] > 0 and out["steps"] == 3
    assert out["config"].get("collective") == "forced on one rank (PARSENET_FORCE_COLLECTIVE=1)"
