"""Data parallelism for the hot path: one process per GPU, shapes sharded across ranks, and
exactly ONE RCCL all-reduce of a flat fp32 gradient bucket per optimizer step (the reference
uses torch.nn.DataParallel: train_parsenet.py:90-91).  Nothing finer crosses
GPUs: kNN graphs, GroupNorm statistics, clustering and fitting are all per shape."""
import os

import torch
import torch.distributed as dist

BUCKET_GATHER = os.environ.get("PARSENET_BUCKET_GATHER", "1") != "0"


def collective_forced():
    """PARSENET_FORCE_COLLECTIVE=1: run the data-parallel machinery — process group, rank-0 broadcast, status
    agreement, gradient all-reduce, barriers — even with ONE rank.  A one-GPU box can then execute the RCCL code
    path the 8-GPU job takes (tests/test_rccl_world1_gpu.py); the result equals the plain step bit for bit (a sum
    over one rank, divided by 1)."""
    return os.environ.get("PARSENET_FORCE_COLLECTIVE", "0") == "1"


def multi_rank():
    """True when the step's collectives must run: a process group with more than one rank, or a forced one."""
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size() > 1 or collective_forced()


def usable_cpus():
    """CPUs this process may actually use: the smaller of the visible cores and the cgroup's CFS quota.
    The boxes of this pool show 256 cores and grant 16 (``cpu.max`` = 1600000 100000)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as f:
                parts = f.read().split()
            if path.endswith("cpu.max"):
                quota, period = parts[0], float(parts[1])
            else:
                quota = parts[0]
                with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                    period = float(f.read().split()[0])
            if quota != "max" and float(quota) > 0:
                n = min(n, max(1, int(float(quota) / period)))
            break
        except (OSError, ValueError, IndexError):
            continue
    return max(1, n)


def limit_host_threads(n=None):
    """torch's intra-op CPU pool down to ``n`` threads (default 1; PARSENET_HOST_THREADS overrides, 0 = leave
    torch's default of one thread per visible core).  Returns the count it found.

    The product computes nothing on the CPU: the host thread queues launches and runs the small
    host steps of the reference (Hungarian matching, a batched 3x3 geev).  With the default pool
    every one of those steps that enters an OpenMP region wakes ~128 threads that spin afterwards;
    under a CFS quota (round 4: 16 CPUs on a 256-core box) the process is throttled for tens of
    milliseconds and the GPU idles: measured on one box 70.4 / 75.3 shapes/s with the default pool against
    94.4 / 92.6 with one thread (cfg5, `profiles/r04_host_threads_ab.txt`; `nr_throttled` +20 per
    run against +2), and the batched geev step alone 0.15 ms against a bimodal 0.5 / 4.7 ms.  With
    one process per GPU the default pool is also 8 x oversubscribed on a node."""
    before = torch.get_num_threads()
    want = os.environ.get("PARSENET_HOST_THREADS")
    if n is None:
        n = int(want) if want not in (None, "") else 1
    if n > 0 and n != before:
        torch.set_num_threads(n)
    return before


def init_from_env(backend=None):
    """Initialise torch.distributed from torchrun's environment; returns (rank, world, device).
    Also caps the host's intra-op thread pool (``limit_host_threads``)."""
    limit_host_threads()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if torch.cuda.is_available():
        torch.cuda.set_device(local)
        device = torch.device("cuda", local)
    else:
        device = torch.device("cpu")
    if (world > 1 or collective_forced()) and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if device.type == "cuda" else "gloo"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        kw = {}
        if backend == "nccl":
            kw["device_id"] = device   # bind the RCCL communicator to this rank's GPU up front
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return rank, world, device


def shard_range(num_items, rank, world):
    """Contiguous shard [lo, hi) of ``num_items`` shapes owned by ``rank``."""
    per = num_items // world
    extra = num_items % world
    lo = rank * per + min(rank, extra)
    return lo, lo + per + (1 if rank < extra else 0)


class FitStatusError(RuntimeError):
    """The fit status of a step says "drop it": a degenerate segment (no full-rank ridge system in ``lstsq``,
    a non-finite design matrix or residual).  The reference's loop catches the exception such a segment raises
    and skips the batch (train_parsenet_e2e.py:243-257); ``FlatGradBucket.finish_or_skip`` treats THIS class —
    and nothing else — as a skipped step."""


class FlatGradBucket:
    """All parameter gradients live in one contiguous fp32 buffer (``p.grad`` are views), so
    the data-parallel reduction is a single collective on a pre-flattened bucket.  Parameters
    that never receive a gradient (the reference's unused ``encoder.bn4/bn5``) stay zero."""

    def __init__(self, params):
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("no trainable parameters")
        dev, dt = self.params[0].device, self.params[0].dtype
        self.numel = sum(p.numel() for p in self.params)
        self.flat = torch.zeros(self.numel, dtype=dt, device=dev)
        self.collective = True      # False: steps that only one rank runs (workloads.train_on_rank0_then_broadcast)
        self._side = None           # host-side (gloo) group for the step-status flag, created on first use
        self.views = []
        self._offsets = []
        o = 0
        for p in self.params:
            n = p.numel()
            self.views.append(self.flat[o:o + n].view_as(p))
            self._offsets.append(o)
            p.grad = self.views[-1]
            o += n
        self._written = set()        # parameters whose slot holds a gradient of the previous gather()

    def zero(self):
        """Zero the bucket and (re-)attach the views: backward passes then ACCUMULATE into it (several
        micro-batches per optimizer step: trainer.accumulate_or_skip)."""
        self.flat.zero_()
        for p, v in zip(self.params, self.views):
            p.grad = v
        # autograd now ACCUMULATES into the views: gather() cannot see which slots that fills, so a later
        # begin() / gather() step must treat every slot as written (a parameter without a gradient then gets zeros)
        self._written = set(range(len(self.params)))

    def begin(self):
        """Start of a step with ONE backward pass: detach the parameters from the bucket.  autograd then hands
        every gradient over as a tensor of its own instead of adding it into a zeroed view — an in-place add per
        parameter, ~75 small launches per cfg5 step (profiles/r04_cfg5_torch_sites.txt: workloads.py step) —
        and ``gather()`` moves them into the bucket with one multi-tensor copy.
        (PARSENET_BUCKET_GATHER=0, developer A/B: the old form — zero the bucket, autograd adds into the views.)"""
        if not BUCKET_GATHER:
            self.zero()
            return
        for p in self.params:
            p.grad = None

    def gather(self):
        """After the backward pass of a ``begin()`` step: all gradients into the flat bucket (one multi-tensor
        copy), ``p.grad`` are its views again.  A parameter that received no gradient keeps a zero slot."""
        if not BUCKET_GATHER:
            return self.flat
        dst, src, now = [], [], set()
        for i, (p, v) in enumerate(zip(self.params, self.views)):
            g = p.grad
            if g is not None and g.data_ptr() != v.data_ptr():
                dst.append(i)
                src.append(g.detach())
                now.add(i)
        for i in self._written - now:            # had a gradient last time, none now: its slot must not keep the old one
            self.views[i].zero_()
        if dst:
            self._gather(dst, src)
        self._written = now
        for p, v in zip(self.params, self.views):
            p.grad = v
        return self.flat

    def _gather(self, slots, grads):
        """``grads[j]`` into slot ``slots[j]`` of the flat buffer.  On the GPU: ONE launch for up to 64 tensors
        (pn_gather_flat_f32; torch._foreach_copy_ decomposes into a device-to-device copy per parameter on this
        stack — 46 launches for the segmentation network)."""
        if self.flat.is_cuda and self.flat.dtype == torch.float32:
            import ctypes
            from . import _lib
            grads = [g if (g.is_contiguous() and g.dtype == torch.float32) else g.contiguous().float() for g in grads]
            n = len(grads)
            srcs = (ctypes.c_void_p * n)(*[g.data_ptr() for g in grads])
            offs = (ctypes.c_longlong * n)(*[self._offsets[i] for i in slots])
            ns = (ctypes.c_longlong * n)(*[g.numel() for g in grads])
            with _lib.on_device(self.flat.device):
                rc = _lib.load().pn_gather_flat_f32(srcs, offs, ns, n, _lib.ptr(self.flat),
                                                    _lib.current_stream(self.flat.device))
            _lib.check(rc, "pn_gather_flat_f32")
            return
        torch._foreach_copy_([self.views[i] for i in slots], grads)

    def _multi(self):
        return self.collective and multi_rank()

    def any_rank_failed(self, failed):
        """True if ``failed`` is set on ANY rank.  A rank whose fitting stage raises must not skip the
        gradient all-reduce alone (the others would sit in RCCL until the watchdog fires) and the
        others must not enter it without that rank: the status is agreed upon with a one-element
        all-reduce on a HOST-side gloo group — no device synchronisation, the backward pass keeps
        running underneath — and every rank then takes the same branch.  A collective: all ranks of
        a step call it, in the same place.  Single rank / collective off: returns ``failed``."""
        if not self._multi():
            return bool(failed)
        if self._side is None:
            # new_group is itself collective: every rank reaches its first status check together
            self._side = dist.group.WORLD if dist.get_backend() == "gloo" else dist.new_group(backend="gloo")
        t = torch.tensor([1 if failed else 0], dtype=torch.int32)
        dist.all_reduce(t, group=self._side)
        return bool(int(t.item()) > 0)

    def finish_or_skip(self, finish, optimizer):
        """Tail of a data-parallel step whose status arrives late (ParsenetE2EStep: the fit status of
        the batched fitting stage rides in a deferred download; train_parsenet_e2e.py:243-257 drops
        the batch on such an exception).  Runs ``finish()``; if it raised on ANY rank, nobody reduces
        and nobody moves the weights; otherwise ONE gradient all-reduce and the optimizer step
        (returns (finish's result, None, True)).

        Only a ``FitStatusError`` is a skipped step (returns (None, the error or None, False)).  Anything
        else — out of memory, a HIP launch error, a programming error — still goes through the status
        agreement first, so that no rank is left waiting in a collective, and is then RE-RAISED on the rank
        it happened on: it must not be recorded as a degenerate segment and silently drop every step that
        follows (the other ranks return a skipped step for this one and see the failure as the collective
        error / process exit it then is)."""
        err, out = None, None
        try:
            out = finish()
        except Exception as e:          # agreed upon below; a FitStatusError is the step's status
            err = e
        if self.any_rank_failed(err is not None):
            if err is not None and not isinstance(err, FitStatusError):
                raise err
            return None, err, False
        self.all_reduce_mean()
        optimizer.step()
        return out, None, True

    def all_reduce_mean(self):
        """Average gradients over ranks with one all-reduce (no-op on a single rank)."""
        if self._multi():
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
            self.flat.div_(dist.get_world_size())
        return self.flat
