"""Synthetic benchmark fields: white noise N(0,1), smoothed by `passes`
applications of the separable [1,2,1]/4 kernel per axis (periodic), normalised to unit variance,
outermost 2 samples forced below the minimum ("closed interior").

The smoothing is applied in Fourier space -- `passes` applications of [1,2,1]/4 along one axis are
exactly a multiplication by cos(w/2)^(2*passes) -- so large pass counts cost one FFT pair.
The bench field is computed on the HOST (pocketfft) and copied to the device, so that it is bit for bit the same
in every environment (plain run, under rocprofv3, another box): see smooth_noise_host.
"""
import math

import numpy as np


_FIELD_CACHE = {}      # (shape, seed, passes) -> fp32 numpy array; one entry (a 512^3 field is 537 MB)


def smooth_noise_host(shape, seed, passes):
    """The bench field as an fp32 numpy array, computed entirely on the host: the noise from torch's CPU generator (plane by
    plane, as every round has drawn it), the filter with scipy's pocketfft (every 1-D transform is computed by one thread:
    the result does not depend on the worker count), mean and deviation accumulated in float64.  Nothing here depends on
    a GPU library's kernel selection.  (Rounds 1-3 ran the two FFTs on the device, torch.fft = rocFFT: under rocprofv3 the
    forward transform came out with different roundings -- tools/field_check.py: same noise checksum, different spectrum
    checksum -- so that ~100 of the 134 M samples within 1e-6 of the isovalue changed and the profiled mesh had 12 441 978
    vertices where the plain run had 12 441 984.)"""
    import scipy.fft as sfft
    import torch
    key = (tuple(int(n) for n in shape), int(seed), int(passes))
    hit = _FIELD_CACHE.get(key)
    if hit is not None:
        return hit
    n0, n1, n2 = key[0]
    g = torch.Generator(device="cpu")
    g.manual_seed(int(seed))
    # ONE call draws the same stream as the plane-by-plane draws of rounds 1-3 (torch's CPU generator fills serially,
    # whatever the thread count: checked at 512^3, checksum of the noise -2240853796164619 either way)
    x = torch.randn((n0, n1, n2), generator=g, dtype=torch.float32).numpy()
    import os
    workers = max(1, min(16, os.cpu_count() or 1))
    X = sfft.rfftn(x, workers=workers)
    del x
    for axis, n in enumerate((n0, n1, n2)):
        m = X.shape[axis]
        w = 2.0 * np.pi * np.arange(m, dtype=np.float64) / n
        filt = (np.abs(np.cos(w / 2.0)) ** (2 * passes)).astype(np.float32)
        view = [1, 1, 1]
        view[axis] = m
        X *= filt.reshape(view)
    y = sfft.irfftn(X, s=(n0, n1, n2), workers=workers)
    del X
    y = np.ascontiguousarray(y, dtype=np.float32)
    mean = float(y.mean(dtype=np.float64))
    y -= np.float32(mean)
    std = float(np.sqrt(_sumsq64(y) / y.size))
    y /= np.float32(std)
    lo = np.float32(float(y.min()) - 1.0)
    for axis in range(3):
        sl = [slice(None)] * 3
        for idx in (0, 1, -1, -2):
            sl[axis] = idx
            y[tuple(sl)] = lo
    _FIELD_CACHE.clear()
    _FIELD_CACHE[key] = y
    return y


def _sumsq64(y):
    "sum of squares in float64, plane by plane (no 1 GB temporary)"
    acc = 0.0
    for i in range(y.shape[0]):
        acc += float(np.square(y[i], dtype=np.float64).sum())     # numpy's pairwise summation: one fixed order
    return acc


def field_checksum(a):
    """order-free checksum of a field: the sum of its fp32 bit patterns as signed 32-bit integers, in 64 bits (numpy array
    or torch tensor on any device).  Equal fields have equal checksums in every environment."""
    if type(a).__module__.split(".")[0] == "torch":
        import torch
        return int(a.contiguous().view(torch.int32).to(torch.int64).sum().item())
    return int(np.ascontiguousarray(a, dtype=np.float32).view(np.int32).astype(np.int64).sum())


def smooth_noise_torch(shape, seed, passes, device, slab=None, dtype=None):
    """fp32 tensor of `shape` on `device`.  slab=(i0, i1): return only planes i0:i1 of axis 0
    (the field itself is always generated whole so that slabs of different ranks agree).  Generated on the host
    (smooth_noise_host: bit-identical in every environment), then copied to the device."""
    import torch
    y = smooth_noise_host(shape, seed, passes)
    if slab is not None:
        y = y[slab[0]:slab[1]]
    return torch.from_numpy(y).to(device).contiguous()


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
