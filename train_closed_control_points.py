"""SplineNet training (reference: train_closed_control_points.py) on the MI355X hot path.

    python train_closed_control_points.py [config file in the reference's configs/*.yml format]

The loop itself lives in parsenet_codebase_amd/trainer.py (train_splinenet)."""
import sys

from parsenet_codebase_amd.trainer import TrainConfig, train_splinenet

if __name__ == "__main__":
    cfg = TrainConfig.from_file(sys.argv[1]) if len(sys.argv) > 1 else TrainConfig(batch_size=32, lr=1e-3)
    train_splinenet(cfg, closed=True)
