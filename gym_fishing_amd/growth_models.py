"""The reference's module-level growth functions on the device (gym_fishing/envs/growth_models.py:208-269).

    from gym_fishing_amd.growth_models import allen, beverton_holt, may, myers, ricker, population_model
    x1 = ricker(x, {"r": 0.3, "K": 1, "sigma": 0.1})

Same call as the reference's: `x` a scalar, an ndarray of any shape or a device tensor of populations, `params` the
model's dict (a missing key raises KeyError, as there).  Each returns max(0, lognormal(mu(x), sigma)) elementwise through
`fishing_population_draw_*` (include/fishing_hip.h) -- float64 arithmetic unless `dtype=torch.float32`; a tensor comes back
as a tensor on its device, anything else as float64 NumPy like the reference's result.

Noise: the reference draws np.random.lognormal(mu, sigma) = exp(mu + sigma z) with one legacy standard normal z per
element of x from NumPy's global stream, whatever sigma is.  With `noise=None` the same normals are consumed here
(np.random.normal(0, 1, shape)), so np.random.seed(s) reproduces the reference's numbers; `noise=` takes the z explicitly
(array or tensor of x's shape), e.g. from a device generator.
"""
import numpy as np
import torch

from . import _capi
from ._capi import MODEL_V5, MODEL_V6, MODEL_V7, MODEL_V8, MODEL_V9

# (model id of the fishing-vN env built on the function, the keys the reference's function reads)
_SPEC = {
    "allen": (MODEL_V5, ("r", "K", "C", "sigma")),                       # :208-217
    "beverton_holt": (MODEL_V6, ("r", "K", "sigma")),                    # :220-226
    "may": (MODEL_V7, ("r", "M", "a", "q", "b", "sigma")),               # :229-242
    "myers": (MODEL_V8, ("r", "theta", "M", "sigma")),                   # :247-255
    "ricker": (MODEL_V9, ("r", "K", "sigma")),                           # :258-261
}


def _draw(name, x, params, noise, dtype, device):
    model, keys = _SPEC[name]
    cp = _capi.FishingParams()
    cp.model, cp.Tmax, cp.n_actions = model, 100, 100
    cp.K, cp.C, cp.x0 = 1.0, 0.5, 0.75
    for k in keys:
        setattr(cp, k, float(params[k]))
    is_tensor = isinstance(x, torch.Tensor)
    if device is None:
        device = x.device if is_tensor and x.is_cuda else "cuda"
    device = torch.device(device)
    if not torch.cuda.is_available() or device.type != "cuda":
        raise _capi.FishingLibraryError("the growth functions run on a HIP device; there is no CPU backend")
    if device.index is None:
        device = torch.device("cuda", torch.cuda.current_device())
    dtype = torch.float64 if dtype is None else dtype
    if dtype not in (torch.float32, torch.float64):
        raise ValueError("dtype must be torch.float32 or torch.float64")
    shape = tuple(x.shape) if is_tensor else np.shape(x)
    xt = (x if is_tensor else torch.as_tensor(np.asarray(x, dtype=np.float64))).to(device=device, dtype=dtype).reshape(-1).contiguous()
    if noise is None:
        noise = np.random.normal(0, 1, shape)
    zt = torch.as_tensor(noise).to(device=device, dtype=dtype).reshape(-1).contiguous()
    if zt.numel() != xt.numel():
        raise ValueError("noise needs one standard normal per population (%d), got %d" % (xt.numel(), zt.numel()))
    out = torch.empty_like(xt)
    if xt.numel():
        fn = getattr(_capi.lib(), "fishing_population_draw_" + ("f32" if dtype == torch.float32 else "f64"))
        with torch.cuda.device(device):
            stream = torch.cuda.current_stream(device).cuda_stream
            rc = fn(cp, xt.numel(), xt.data_ptr(), zt.data_ptr(), None, None, None, out.data_ptr(), stream)
        _capi.check(rc, "fishing_population_draw")
    if is_tensor:
        return out.reshape(shape)
    res = out.cpu().numpy().astype(np.float64).reshape(shape)
    return res if shape else np.float64(res)


def allen(x, params, noise=None, dtype=None, device=None):
    """growth_models.py:208-217: mu = log x + r (1 - x / K)(1 - C) / K."""
    return _draw("allen", x, params, noise, dtype, device)


def beverton_holt(x, params, noise=None, dtype=None, device=None):
    """growth_models.py:220-226: mu = log A + log x - log(1 + x / B), A = clip(r) + 1, B = clip(K) / clip(r)."""
    return _draw("beverton_holt", x, params, noise, dtype, device)


def may(x, params, noise=None, dtype=None, device=None):
    """growth_models.py:229-242: mu = log(x + x r (1 - x / M) - a x^q / (x^q + b^q))."""
    return _draw("may", x, params, noise, dtype, device)


def myers(x, params, noise=None, dtype=None, device=None):
    """growth_models.py:247-255: mu = log(r + 1) + theta log x - log(1 + x^theta / M)."""
    return _draw("myers", x, params, noise, dtype, device)


def ricker(x, params, noise=None, dtype=None, device=None):
    """growth_models.py:258-261: mu = log x + r (1 - x / K)."""
    return _draw("ricker", x, params, noise, dtype, device)


# growth_models.py:262-268
population_model = {"allen": allen, "beverton_holt": beverton_holt, "myers": myers, "may": may, "ricker": ricker}
