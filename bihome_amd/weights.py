"""Deterministic synthetic weights keyed on state-dict names.

There is no network on the build or GPU box, so the ImageNet ResNet-34 weights the reference pulls
(src/backbones/Rethinking.py:178-183, src/heads/PerceptualHead.py:22) are unobtainable.  Parity and
the benchmark use random-init weights that are a pure function of (seed, parameter name, shape):
numpy PCG64 streams, stable across machines and torch versions, so the reference (when the golden
vectors are made), the CPU oracle and the HIP modules all see bit-identical values.
"""
import zlib

import numpy as np
import torch


def _rng(seed, key):
    return np.random.Generator(np.random.PCG64([seed, zlib.crc32(key.encode())]))


def synth_tensor(seed, key, shape, kind):
    g = _rng(seed, key)
    if kind == "conv":          # He-normal on fan_in
        fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else shape[0]
        a = g.standard_normal(shape) * np.sqrt(2.0 / max(fan_in, 1))
    elif kind == "convT":       # ConvTranspose2d weight is [Cin, Cout, kh, kw]; fan_in = Cin (stride == kernel)
        a = g.standard_normal(shape) * np.sqrt(2.0 / shape[0])
    elif kind == "gamma":
        a = 1.0 + 0.1 * g.standard_normal(shape)
    elif kind == "beta" or kind == "bias":
        a = 0.1 * g.standard_normal(shape)
    else:
        raise ValueError(kind)
    return a.astype(np.float32)


def synthetic_state_dict(module, seed=0):
    """Return a state dict for `module` (any nn.Module built from Conv2d / ConvTranspose2d /
    BatchNorm2d / Linear leaves) filled deterministically; buffers are reset to their defaults."""
    out = {}
    kinds = {}
    for name, m in module.named_modules():
        p = name + "." if name else ""
        if isinstance(m, torch.nn.ConvTranspose2d):
            kinds[p + "weight"], kinds[p + "bias"] = "convT", "bias"
        elif isinstance(m, (torch.nn.Conv2d, torch.nn.Linear)):
            kinds[p + "weight"], kinds[p + "bias"] = "conv", "bias"
        elif isinstance(m, torch.nn.BatchNorm2d):
            kinds[p + "weight"], kinds[p + "bias"] = "gamma", "beta"
    for key, val in module.state_dict().items():
        if key.endswith("running_mean"):
            out[key] = torch.zeros_like(val)
        elif key.endswith("running_var"):
            out[key] = torch.ones_like(val)
        elif key.endswith("num_batches_tracked"):
            out[key] = torch.zeros_like(val)
        else:
            out[key] = torch.from_numpy(synth_tensor(seed, key, tuple(val.shape), kinds[key]))
    return out


def load_synthetic(module, seed=0):
    """Fill `module` in place.  Call it on the backbone and on `head.auxiliary_resnet` separately
    (not on the nn.Sequential that aliases the backbone under '1.backbone.*', train.py:696) so the
    key names - and therefore the values - do not depend on how the modules are wrapped."""
    module.load_state_dict(synthetic_state_dict(module, seed))
    return module
