"""Blendshape-coefficient I/O (reference: said/util/blendshape.py:36-84 and the
32 ARKit class names of script/dataset/dataset_voca.py:99-132)."""
from __future__ import annotations

from typing import List

import numpy as np
import torch

DEFAULT_BLENDSHAPE_CLASSES: List[str] = [
    "jawForward", "jawLeft", "jawRight", "jawOpen", "mouthClose", "mouthFunnel", "mouthPucker", "mouthLeft",
    "mouthRight", "mouthSmileLeft", "mouthSmileRight", "mouthFrownLeft", "mouthFrownRight", "mouthDimpleLeft",
    "mouthDimpleRight", "mouthStretchLeft", "mouthStretchRight", "mouthRollLower", "mouthRollUpper",
    "mouthShrugLower", "mouthShrugUpper", "mouthPressLeft", "mouthPressRight", "mouthLowerDownLeft",
    "mouthLowerDownRight", "mouthUpperUpLeft", "mouthUpperUpRight", "cheekPuff", "cheekSquintLeft",
    "cheekSquintRight", "noseSneerLeft", "noseSneerRight",
]


def load_blendshape_coeffs(coeffs_path: str) -> torch.FloatTensor:
    """(T_b, num_classes) coefficients from a CSV with a header row."""
    import pandas as pd

    return torch.FloatTensor(pd.read_csv(coeffs_path).values)


def save_blendshape_coeffs(coeffs: np.ndarray, classes: List[str], output_path: str) -> None:
    """CSV with a header of class names, one row per frame, no index column."""
    import pandas as pd

    pd.DataFrame(coeffs, columns=classes).to_csv(output_path, index=False)


def save_blendshape_coeffs_image(coeffs: np.ndarray, output_path: str) -> None:
    from PIL import Image

    Image.fromarray((255 * coeffs.transpose()).round()).convert("L").save(output_path)
