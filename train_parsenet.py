"""ParSeNet segmentation-only training (reference: train_parsenet.py) on the MI355X hot path.

    python train_parsenet.py [config file in the reference's configs/*.yml format]
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 train_parsenet.py cfg.yml

The loop itself lives in parsenet_codebase_amd/trainer.py (train_parsenet)."""
import sys

from parsenet_codebase_amd.trainer import TrainConfig, train_parsenet

if __name__ == "__main__":
    cfg = TrainConfig.from_file(sys.argv[1]) if len(sys.argv) > 1 else TrainConfig()
    for rec in train_parsenet(cfg):
        pass
