"""Adam on one flat parameter buffer: the optimizer of every training script of the reference
(train_parsenet.py:96, train_parsenet_e2e.py:88, train_open_splines.py:81, train_closed_control_points.py:75:
``optim.Adam(model.parameters(), lr=...)`` with torch's defaults) as ONE kernel launch per step.

torch.optim.Adam(fused=True) groups the model's ~40 tensors on the host at every step (0.6 ms between its two
launches under the profiler, 0.16 ms without: profiles/r05_cfg5_step_gaps.txt) and reads each tensor's pointer
from a table.  Here the parameters are views of ONE fp32 buffer in the order of dp.FlatGradBucket — whose flat
gradient buffer is the second operand — and the two moments are flat buffers of the same layout:
``pn_adam_flat_f32`` (csrc/fused.hip) updates all of them with 16-byte lanes.  ``state_dict`` /
``load_state_dict`` keep torch.optim.Adam's format (per-parameter ``step`` / ``exp_avg`` / ``exp_avg_sq``), so
checkpoints interchange with it."""
import torch

from . import _lib
from ._lib import check, current_stream, ptr


class FlatAdam(torch.optim.Optimizer):
    """``bucket``: the model's dp.FlatGradBucket (gradients of ``bucket.params`` in ``bucket.flat``).  The
    parameters are MOVED into one flat buffer (``p.data`` become views of it; values unchanged)."""

    def __init__(self, bucket, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        params = list(bucket.params)
        if not params or not params[0].is_cuda:
            raise RuntimeError("FlatAdam runs on the GPU (no CPU path)")
        if any(p.dtype != torch.float32 for p in params):
            raise TypeError("FlatAdam: fp32 parameters only")
        # torch.optim.Adam's param-group keys, at its defaults (the ones that change the rule are refused in step()):
        # a state_dict of either optimizer loads into the other
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=0, amsgrad=False, maximize=False,
                                      foreach=None, capturable=False, differentiable=False, fused=None,
                                      decoupled_weight_decay=False))
        self.bucket = bucket
        dev = params[0].device
        self.flat_p = torch.empty(bucket.numel, dtype=torch.float32, device=dev)
        self.flat_m = torch.zeros_like(self.flat_p)
        self.flat_v = torch.zeros_like(self.flat_p)
        self.steps = 0
        o = 0
        self._slices = []
        with torch.no_grad():
            for p in params:
                n = p.numel()
                view = self.flat_p[o:o + n].view_as(p)
                view.copy_(p.data)
                p.data = view                      # same values, now inside the flat buffer
                self._slices.append((o, n))
                o += n
        self._attach_state()

    def _attach_state(self):
        for p, (o, n) in zip(self.bucket.params, self._slices):
            self.state[p] = {"step": torch.tensor(float(self.steps)),
                             "exp_avg": self.flat_m[o:o + n].view_as(p),
                             "exp_avg_sq": self.flat_v[o:o + n].view_as(p)}

    def _parameters_still_flat(self):
        base = self.flat_p.data_ptr()
        return all(p.data_ptr() == base + 4 * o for p, (o, _) in zip(self.bucket.params, self._slices))

    @torch.no_grad()
    def step(self, closure=None):
        if closure is not None:
            raise NotImplementedError("FlatAdam: no closure")
        if not self._parameters_still_flat():
            raise RuntimeError("FlatAdam: a parameter left the flat buffer (module.to() / a new .data after the "
                               "optimizer was built); build the optimizer last")
        g = self.param_groups[0]
        if g.get("weight_decay", 0) or g.get("amsgrad", False) or g.get("maximize", False):
            raise NotImplementedError("FlatAdam: plain Adam only (no weight decay / amsgrad / maximize)")
        self.steps += 1
        with _lib.on_device(self.flat_p.device):
            rc = _lib.load().pn_adam_flat_f32(ptr(self.flat_p), ptr(self.bucket.flat), ptr(self.flat_m),
                                              ptr(self.flat_v), self.flat_p.numel(), float(g["lr"]),
                                              float(g["betas"][0]), float(g["betas"][1]), float(g["eps"]), self.steps,
                                              current_stream(self.flat_p.device))
        check(rc, "pn_adam_flat_f32")

    def state_dict(self):
        for st in self.state.values():
            st["step"] = torch.tensor(float(self.steps))
        return super().state_dict()

    def load_state_dict(self, state_dict):
        """torch.optim.Adam's format; the loaded moments are copied INTO the flat buffers."""
        super().load_state_dict(state_dict)
        steps = 0
        with torch.no_grad():
            for p, (o, n) in zip(self.bucket.params, self._slices):
                st = self.state.get(p, {})
                if "exp_avg" in st:
                    self.flat_m[o:o + n].copy_(st["exp_avg"].reshape(-1))
                    self.flat_v[o:o + n].copy_(st["exp_avg_sq"].reshape(-1))
                    steps = max(steps, int(float(st["step"])))
                else:
                    self.flat_m[o:o + n].zero_()
                    self.flat_v[o:o + n].zero_()
        self.steps = steps
        self._attach_state()
