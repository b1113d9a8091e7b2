"""Run-to-run reproducibility of the training steps on the GPU (round 4, verdict item 1).

Two instances of a workload are built from the same seed in ONE process; every step's loss and the
parameters after every optimizer step are compared bit for bit.  With --warn the run is made under
torch.use_deterministic_algorithms(True, warn_only=True) and the tensor-library operations that
have no deterministic implementation are listed with the source line that issued them.

  python tools/determinism_probe.py --workload cfg5 --pretrain 40 --steps 3 [--points 10000] [--warn]
"""
import argparse
import os
import sys
import warnings

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402


def build(workload, dev, args):
    from parsenet_codebase_amd import workloads as W
    np.random.seed(99)
    if workload == "cfg5":
        return W.ParsenetE2EStep(dev, batch=args.batch, num_points=args.points, pretrain_steps=args.pretrain,
                                 pool=args.batch * 2, pretrain_pool=8)
    if workload == "cfg4":
        return W.ParsenetSegStep(dev, batch=args.batch, num_points=args.points, pool=args.batch * 2)
    return W.SplineNetStep(dev, closed=(workload == "cfg3"))


def fingerprint(model):
    return {k: v.detach().clone() for k, v in model.state_dict().items()}


def diff(a, b, what):
    bad = [(k, float((a[k].double() - b[k].double()).abs().max())) for k in a if not torch.equal(a[k], b[k])]
    if bad:
        print("  %s: %d / %d tensors differ; first: %s" % (what, len(bad), len(a), bad[:4]))
    return not bad


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="cfg5")
    ap.add_argument("--pretrain", type=int, default=40)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--points", type=int, default=10000)
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--warn", action="store_true")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    if args.warn:
        torch.use_deterministic_algorithms(True, warn_only=True)
        warnings.simplefilter("always")
    runs = []
    ok = True
    for rep in range(2):
        with warnings.catch_warnings(record=True) as caught:
            warnings.simplefilter("always")
            step = build(args.workload, dev, args)
            rec = {"init": fingerprint(step.model), "loss": [], "after": [], "grads": []}
            for s in range(args.steps):
                np.random.seed(1000 + s)
                loss = step.step()
                rec["loss"].append(loss.detach().clone())
                rec["grads"].append({n: p.grad.detach().clone() for n, p in step.model.named_parameters()
                                     if p.grad is not None})
                rec["after"].append(fingerprint(step.model))
            torch.cuda.synchronize()
        if args.warn and rep == 0:
            seen = set()
            for w in caught:
                msg = str(w.message).split("\n")[0][:160]
                if "deterministic" in msg and msg not in seen:
                    seen.add(msg)
                    print("WARN", msg, "@", w.filename, w.lineno)
        runs.append(rec)
        print("run %d: pretrain_loss %r losses %s" % (rep, getattr(step, "pretrain_loss", None),
                                                       [float(x) for x in rec["loss"]]))
    a, b = runs
    ok &= diff(a["init"], b["init"], "state after pre-training")
    for s in range(args.steps):
        same = torch.equal(a["loss"][s], b["loss"][s])
        print("step %d: loss %s" % (s, "identical" if same else "DIFFERS %.9g vs %.9g" % (float(a["loss"][s]),
                                                                                          float(b["loss"][s]))))
        ok &= same
        ok &= diff(a["grads"][s], b["grads"][s], "gradients of step %d" % s)
        ok &= diff(a["after"][s], b["after"][s], "parameters after step %d" % s)
    print("BIT-REPRODUCIBLE" if ok else "NOT REPRODUCIBLE")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
