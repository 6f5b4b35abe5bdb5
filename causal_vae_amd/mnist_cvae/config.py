"""CONFIG of the measurement-approach experiment (mnist_test/03_measurement_approach/config.py:6-17): Z_DIM / T_DIM are read by
ConditionalVAE at construction time exactly like the reference."""
import torch

CONFIG = dict(BATCH_SIZE=128, EPOCHS=30, LR=1e-3, Z_DIM=10, M_DIM=16, T_DIM=10, SEED=42, BETA=1.0, LAMBDA_ADV=10.0,
              DEVICE=torch.device("cuda" if torch.cuda.is_available() else "cpu"))
