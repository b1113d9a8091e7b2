"""ParSeNet end-to-end training (reference: train_parsenet_e2e.py) on the MI355X hot path.

    python train_parsenet_e2e.py [config file in the reference's configs/*.yml format]
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 train_parsenet_e2e.py cfg.yml

The loop itself lives in parsenet_codebase_amd/trainer.py (train_parsenet_e2e)."""
import sys

from parsenet_codebase_amd.trainer import TrainConfig, train_parsenet_e2e

if __name__ == "__main__":
    cfg = TrainConfig.from_file(sys.argv[1]) if len(sys.argv) > 1 else TrainConfig()
    for rec in train_parsenet_e2e(cfg):
        pass
