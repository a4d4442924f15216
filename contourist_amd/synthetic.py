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


CONFIG4_VALUE = 0.5     # isovalue of BASELINE config 4 on moving_blobs_torch


def moving_blobs_torch(shape, seed, device):
    """BASELINE config 4's field: a 4-D array A[x][y][z][t] of two Gaussian blobs that move (and pass each other) over t,
    plus 5 % smooth noise, zero on the two outermost samples of every axis (closed interior)."""
    import torch
    n0, n1, n2, n3 = (int(n) for n in shape)
    g = torch.Generator(device="cpu")
    g.manual_seed(int(seed))
    ax = [torch.arange(n, device=device, dtype=torch.float32) for n in (n0, n1, n2, n3)]
    X, Y, Z, T = torch.meshgrid(*ax, indexing="ij")
    s = T / max(n3 - 1, 1)
    c1 = (0.30 + 0.35 * s, 0.35 + 0.2 * s, 0.5 + 0.0 * s)
    c2 = (0.70 - 0.30 * s, 0.65 - 0.2 * s, 0.45 + 0.1 * s)

    def blob(c, w):
        return torch.exp(-(((X / n0 - c[0]) ** 2 + (Y / n1 - c[1]) ** 2 + (Z / n2 - c[2]) ** 2) / (2 * w * w)))
    A = blob(c1, 0.12) + blob(c2, 0.10)
    del X, Y, Z, T
    noise = torch.randn((n0 // 8 + 1, n1 // 8 + 1, n2 // 8 + 1, n3 // 8 + 1), generator=g).to(device)
    up = torch.nn.functional.interpolate(noise.permute(3, 0, 1, 2).unsqueeze(1), size=(n0, n1, n2), mode="trilinear", align_corners=True)
    up = torch.nn.functional.interpolate(up.squeeze(1).permute(1, 2, 3, 0).reshape(1, n0 * n1 * n2, -1), size=n3, mode="linear", align_corners=True)
    A = A + 0.05 * up.reshape(n0, n1, n2, n3)
    for axis in range(4):
        for idx in (0, 1, -1, -2):
            A.select(axis, idx).fill_(0.0)
    return A.contiguous()
