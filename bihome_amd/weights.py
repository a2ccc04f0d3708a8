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


# -----------------------------------------------------------------------------------------------
# ImageNet ResNet-34 initialisation (PRETRAINED_RESNET: True upstream).  Upstream downloads
# resnet34-333f7ec4.pth (Rethinking.py:178-183; torchvision inside ResNet34.py:15); there is no network here, so the
# caller hands over that state dict (a path or a dict) and these functions place it.
# -----------------------------------------------------------------------------------------------
_UNIT = {"conv1": "upper_branch.0", "bn1": "upper_branch.1", "conv2": "upper_branch.3", "bn2": "upper_branch.4",
         "downsample": "lower_branch"}


def zeng_keys_from_torchvision(tv_state):
    """torchvision resnet34 `layer{1,2,3}.<i>.<conv1|bn1|conv2|bn2|downsample.<j>>.<p>` -> the Zeng backbone's
    `layer{2,3,4}.<i>.<upper_branch.{0,1,3,4}|lower_branch.<j>>.<p>` (the placement Rethinking.py:189-282 performs; stem,
    layer4 and fc of the ImageNet net are not used by it)."""
    out = {}
    for key, value in tv_state.items():
        parts = key.split(".")
        if parts[0] not in ("layer1", "layer2", "layer3") or parts[2] not in _UNIT:
            continue
        stage = "layer%d" % (int(parts[0][5:]) + 1)
        out[".".join([stage, parts[1], _UNIT[parts[2]]] + parts[3:])] = value
    return out


def load_imagenet_resnet34(backbone, tv_state):
    """Place torchvision ImageNet resnet34 weights into a bihome_amd backbone (Rethinking.Model or ResNet34.Model).
    tv_state: state dict or path to resnet34-333f7ec4.pth.  Returns the list of keys that were loaded.  Shapes are
    checked (a mismatch raises, where upstream only prints); for the ResNet-34 regressor the 2-channel conv1 and the
    8-way fc keep their fresh initialisation exactly as ResNet34.py:15-19 replaces them after loading."""
    import torch
    if isinstance(tv_state, (str, bytes)):
        tv_state = torch.load(tv_state, map_location="cpu")
    own = backbone.state_dict()
    if hasattr(backbone, "resnet34"):
        mapped = {"resnet34." + k: v for k, v in tv_state.items() if not (k.startswith("conv1.") or k.startswith("fc."))}
    else:
        mapped = zeng_keys_from_torchvision(tv_state)
    for k, v in mapped.items():
        if k not in own:
            raise KeyError("ImageNet key %s has no counterpart in %s" % (k, type(backbone).__name__))
        if tuple(own[k].shape) != tuple(v.shape):
            raise ValueError("ImageNet tensor %s has shape %s, the backbone expects %s" % (k, tuple(v.shape), tuple(own[k].shape)))
    backbone.load_state_dict(mapped, strict=False)
    return sorted(mapped)
