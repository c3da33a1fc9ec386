"""Deterministic, name-seeded synthetic weights.

No pretrained weights are reachable offline (torchvision's VGG16-BN / VGG19 checkpoints and the
released Slice3D checkpoints live on external hosts), so every parity check in this repository runs
on synthetic weights that depend ONLY on (state_dict key, tensor shape, seed).  The same generator
feeds the imported reference model (tests/golden/make_golden.py, run in the authoring container) and
this package's model, so goldens never depend on torch's global RNG or on module construction order.

Statistics are deliberately non-trivial (BN running stats != identity, non-zero biases) so that
BN-folding / bias bugs cannot hide.  See SURVEY.md section 8(c).
"""
import zlib

import numpy as np
import torch


def seeded_array(key: str, shape, seed: int = 0) -> np.ndarray:
    """Value for state_dict entry `key` of shape `shape` (float32, or int64 for counters)."""
    shape = tuple(int(s) for s in shape)
    rng = np.random.default_rng((zlib.crc32(key.encode()) ^ (seed * 0x9E3779B1)) & 0xFFFFFFFF)
    leaf = key.rsplit(".", 1)[-1]
    if leaf in ("mean", "std") and shape == (1, 3, 1, 1):  # VGGPerceptualLoss ImageNet constants
        v = (0.485, 0.456, 0.406) if leaf == "mean" else (0.229, 0.224, 0.225)
        return np.asarray(v, dtype=np.float32).reshape(shape)
    if leaf == "num_batches_tracked":
        return np.zeros(shape, dtype=np.int64)
    if leaf == "running_mean":
        return rng.normal(0.0, 0.1, shape).astype(np.float32)
    if leaf == "running_var":
        return rng.uniform(0.75, 1.25, shape).astype(np.float32)
    if len(shape) <= 1:
        if leaf == "weight":  # BatchNorm / LayerNorm gain
            return rng.uniform(0.75, 1.25, shape).astype(np.float32)
        return rng.normal(0.0, 0.1, shape).astype(np.float32)  # biases
    fan_in = int(np.prod(shape[1:]))
    gain = 2.0
    if ".outc." in key:  # keep the tanh output head out of saturation so image parity stays sensitive
        gain = 0.05
    return rng.normal(0.0, np.sqrt(gain / fan_in), shape).astype(np.float32)


def seeded_state_dict(template, seed: int = 0):
    """Build a state_dict with the keys/shapes of `template` (a module or a state_dict)."""
    sd = template.state_dict() if hasattr(template, "state_dict") else template
    out = {}
    for k, v in sd.items():
        out[k] = torch.from_numpy(seeded_array(k, v.shape, seed)).to(v.dtype)
    return out


def load_seeded(module, seed: int = 0):
    module.load_state_dict(seeded_state_dict(module, seed), strict=True)
    return module
