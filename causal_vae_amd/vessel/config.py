"""Hyper-parameters of the 2D vessel recipe (vessel_analysis/00_core/config.py:10-27); the author's absolute paths are not mirrored."""
CONFIG = {
    "EPOCHS": 150, "BATCH_SIZE": 8, "LEARNING_RATE": 1e-4, "BETA": 0.5, "LAMBDA_MORPH": 10000,
    "IMG_HEIGHT": 768, "IMG_WIDTH": 1280, "T_DIM": 19, "M_DIM": 12, "Z_DIM": 128,
}
