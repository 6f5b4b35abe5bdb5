"""CONFIG of the MNIST baseline (mnist_test/01_baseline_causal_vae/config.py:6-23); models read M_DIM/T_DIM/Z_DIM at
construction time exactly like the reference."""
import torch

CONFIG = dict(BATCH_SIZE=128, EPOCHS=100, LR=1e-3, Z_DIM=10, M_DIM=12, T_DIM=10, SEED=42, BETA=1.0, LAMBDA_ADV=10.0,
              DEVICE=torch.device("cuda" if torch.cuda.is_available() else "cpu"))

FEATURE_NAMES = ("Area Perimeter Thickness MajorAxis Eccentricity Orientation Solidity Extent AspectRatio Euler "
                 "H_Symmetry V_Symmetry").split()
