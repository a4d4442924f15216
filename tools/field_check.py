#!/usr/bin/env python3
"""(The stages printed first are the DEVICE FFT recipe of rounds 1-3, kept to show where it diverges; `field` is what the
bench uses today: contourist_amd.synthetic.smooth_noise_host.)
Where do two runs of the bench field differ?  Prints order-free checksums of every stage of
contourist_amd.synthetic.smooth_noise_torch (CPU noise, forward FFT, filtered spectrum, inverse FFT, normalised field),
the number of samples within 1e-6 of the isovalue, and the HIP Level-0 counts of that very grid.  Run it plain and under
rocprofv3 (program directly after `--`) and compare the lines: the first stage whose checksum differs is the cause of
the 12 441 984 / 12 441 978 vertex split ("same seed, two meshes")."""
import json, math, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from contourist_amd import _ffi, synthetic


def csum(t):
    """sum of the bit patterns (int64, wraps) -- order-free, exact"""
    if t.is_complex():
        t = torch.view_as_real(t)
    return int(t.contiguous().view(torch.int32).to(torch.int64).sum().item())


def main():
    size = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    seed, passes = 1235, 1400 if size >= 512 else 700
    dev = torch.device("cuda", 0)
    out = {"size": size}
    g = torch.Generator(device="cpu"); g.manual_seed(seed)
    x = torch.empty((size,) * 3, dtype=torch.float32, device=dev)
    for i in range(size):
        x[i] = torch.randn((size, size), generator=g, dtype=torch.float32).to(dev)
    out["noise"] = csum(x)
    X = torch.fft.rfftn(x); del x
    out["rfftn"] = csum(X)
    for axis, n in enumerate((size,) * 3):
        m = X.shape[axis]
        w = 2.0 * math.pi * torch.arange(m, device=dev, dtype=torch.float64) / n
        filt = torch.cos(w / 2.0).abs().pow(2 * passes).to(torch.float32)
        view = [1, 1, 1]; view[axis] = m
        X *= filt.view(view)
    out["filtered"] = csum(X)
    y = torch.fft.irfftn(X, s=(size,) * 3); del X
    out["irfftn"] = csum(y)
    out["mean"], out["std"] = float(y.mean()), float(y.std())
    del y
    A = synthetic.smooth_noise_torch((size,) * 3, seed, passes, dev)
    out["field"] = csum(A)
    out["field_host"] = synthetic.field_checksum(synthetic.smooth_noise_host((size,) * 3, seed, passes))
    out["near_zero_1e-6"] = int((A.abs() < 1e-6).sum().item())
    out["near_zero_1e-7"] = int((A.abs() < 1e-7).sum().item())
    ctx = _ffi.Context(0, stream=torch.cuda.current_stream().cuda_stream)
    ctx.adopt_device_grid(A.data_ptr(), tuple(A.shape), keepalive=A)
    c = ctx.extract3d(0.0, 1)
    out["counts"] = {k: int(v) for k, v in c.items() if k.startswith("n_")}
    if len(sys.argv) > 2:      # save the near-zero samples' positions and bits for a diff
        idx = torch.nonzero(A.abs() < 1e-6)
        out["near"] = [[int(a) for a in r] + [float(A[tuple(r)])] for r in idx.cpu().tolist()[:64]]
    ctx.close()
    print(json.dumps(out))


main()
