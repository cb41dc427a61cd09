"""Deterministic parameter fill keyed on ``state_dict`` names -- TEST INFRASTRUCTURE.

The same closed-form fill is applied to the imported reference (build container, when the
golden fixtures are generated) and to the product's modules (GPU box, when they are checked),
so no RNG state, construction order or checkpoint file has to travel.
Values depend only on (key, shape): numpy ``default_rng`` seeded with crc32(key).
"""
import zlib

import numpy as np
import torch


def _rng(key):
    return np.random.default_rng(zlib.crc32(key.encode("utf-8")))


def fill_value(key, shape):
    """Returns a float32 ndarray (or int64 scalar for ``num_batches_tracked``)."""
    r = _rng(key)
    shape = tuple(shape)
    if key.endswith("num_batches_tracked"):
        return np.asarray(0, dtype=np.int64)
    if key.endswith("running_mean"):
        return r.uniform(-0.2, 0.2, shape).astype(np.float32)
    if key.endswith("running_var"):
        return r.uniform(0.6, 1.4, shape).astype(np.float32)
    if key.endswith("bn.bn.weight"):
        return r.uniform(0.8, 1.2, shape).astype(np.float32)
    if key.endswith("bn.bn.bias"):
        return r.uniform(-0.1, 0.1, shape).astype(np.float32)
    if key.endswith("conv.bias"):
        return r.uniform(-0.05, 0.05, shape).astype(np.float32)
    if key.endswith("conv.weight"):
        fan_out, fan_in = shape[0], int(np.prod(shape[1:]))
        a = float(np.sqrt(6.0 / (fan_in + fan_out)))  # xavier-uniform bound, like the reference init
        return r.uniform(-a, a, shape).astype(np.float32)
    raise KeyError(f"params.fill_value: unexpected state_dict key {key!r}")


def fill_state_dict(sd):
    """In-place fill of a ``state_dict`` (tensors keep device/dtype).  Returns ``sd``."""
    with torch.no_grad():
        for k, v in sd.items():
            val = torch.from_numpy(np.array(fill_value(k, v.shape))).reshape(v.shape)
            v.copy_(val.to(v.dtype))
    return sd


def make_state_dict(shapes):
    """Build a CPU float32 state_dict from ``{key: shape}`` (tests/golden/state_shapes.json)."""
    out = {}
    for k, shp in shapes.items():
        out[k] = torch.from_numpy(np.array(fill_value(k, tuple(shp)))).reshape(tuple(shp))
    return out


def fill_generic(key, shape):
    """Closed-form fill for modules whose keys do not follow the SharedMLP naming (plain ``nn.Sequential`` of
    Conv2d / BatchNorm2d / Linear, e.g. ``mlps.0.1.running_var``): decided by suffix and rank only."""
    r = _rng("generic:" + key)
    shape = tuple(shape)
    if key.endswith("num_batches_tracked"):
        return np.asarray(0, dtype=np.int64)
    if key.endswith("running_mean"):
        return r.uniform(-0.2, 0.2, shape).astype(np.float32)
    if key.endswith("running_var"):
        return r.uniform(0.6, 1.4, shape).astype(np.float32)
    if len(shape) >= 2:
        fan_out, fan_in = shape[0], int(np.prod(shape[1:]))
        a = float(np.sqrt(6.0 / (fan_in + fan_out)))
        return r.uniform(-a, a, shape).astype(np.float32)
    if key.endswith("weight"):
        return r.uniform(0.8, 1.2, shape).astype(np.float32)
    if key.endswith("bias"):
        return r.uniform(-0.1, 0.1, shape).astype(np.float32)
    raise KeyError(f"params.fill_generic: unexpected state_dict key {key!r}")


def fill_module_generic(module):
    """In-place ``fill_generic`` of every entry of ``module.state_dict()``.  Returns the module."""
    with torch.no_grad():
        for k, v in module.state_dict().items():
            v.copy_(torch.from_numpy(np.array(fill_generic(k, v.shape))).reshape(v.shape).to(v.dtype))
    return module
