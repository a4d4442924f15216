"""Synthetic benchmark fields: white noise N(0,1), smoothed by `passes`
applications of the separable [1,2,1]/4 kernel per axis (periodic), normalised to unit variance,
outermost 2 samples forced below the minimum ("closed interior").

The smoothing is applied in Fourier space -- `passes` applications of [1,2,1]/4 along one axis are
exactly a multiplication by cos(w/2)^(2*passes) -- so large pass counts cost one FFT pair.
Works on a HIP device through torch (fields are generated straight into HBM) or on the CPU.
"""
import math

import numpy as np


def smooth_noise_torch(shape, seed, passes, device, slab=None, dtype=None):
    """fp32 tensor of `shape` on `device`.  slab=(i0, i1): return only planes i0:i1 of axis 0
    (the field itself is always generated whole so that slabs of different ranks agree)."""
    import torch
    g = torch.Generator(device="cpu")
    g.manual_seed(int(seed))
    n0, n1, n2 = (int(n) for n in shape)
    # counter-based, device independent: draw on the CPU generator in planes, move to the device
    x = torch.empty((n0, n1, n2), dtype=torch.float32, device=device)
    for i in range(n0):
        x[i] = torch.randn((n1, n2), generator=g, dtype=torch.float32).to(device)
    X = torch.fft.rfftn(x)
    del x
    for axis, n in enumerate((n0, n1, n2)):
        m = X.shape[axis]
        w = 2.0 * math.pi * torch.arange(m, device=device, dtype=torch.float64) / n
        filt = torch.cos(w / 2.0).abs().pow(2 * passes).to(torch.float32)
        view = [1, 1, 1]
        view[axis] = m
        X *= filt.view(view)
    y = torch.fft.irfftn(X, s=(n0, n1, n2))
    del X
    y -= y.mean()
    y /= y.std()
    lo = float(y.min()) - 1.0
    for axis in range(3):
        for idx in (0, 1, -1, -2):
            y.select(axis, idx).fill_(lo)
    if slab is not None:
        y = y[slab[0]:slab[1]].contiguous()
    return y.contiguous()


def smooth_noise_numpy(shape, seed, passes):
    """same recipe on the CPU with numpy (different random stream than the torch version)."""
    rng = np.random.RandomState(seed)
    x = rng.standard_normal(shape).astype(np.float32)
    X = np.fft.rfftn(x)
    for axis, n in enumerate(shape):
        m = X.shape[axis]
        w = 2.0 * np.pi * np.arange(m) / n
        filt = np.abs(np.cos(w / 2.0)) ** (2 * passes)
        view = [1, 1, 1]
        view[axis] = m
        X = X * filt.reshape(view)
    y = np.fft.irfftn(X, s=shape)
    y = (y - y.mean()) / y.std()
    lo = float(y.min()) - 1.0
    for axis in range(3):
        sl = [slice(None)] * 3
        for idx in (0, 1, -1, -2):
            sl[axis] = idx
            y[tuple(sl)] = lo
    return np.ascontiguousarray(y, dtype=np.float32)
