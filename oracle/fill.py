"""Deterministic, construction-order-independent weight / input fill (TEST
INFRASTRUCTURE).  The golden generator applies it to the reference's modules,
the tests apply it to the oracle and to the product, so no weights have to be
stored in the fixtures: every tensor is a function of (its state_dict key, its
shape, a seed) only."""
from __future__ import annotations

import zlib

import torch


def _gen(key: str, seed: int) -> torch.Generator:
    g = torch.Generator(device="cpu")
    g.manual_seed((zlib.crc32(key.encode()) + 7919 * seed) % (2 ** 31))
    return g


def fill_tensor(key: str, shape, seed: int = 0) -> torch.Tensor:
    """Value for one state_dict entry (fp32, CPU)."""
    shape = tuple(shape)
    g = _gen(key, seed)
    leaf = key.rsplit(".", 1)[-1]
    if leaf == "num_batches_tracked":
        return torch.zeros(shape, dtype=torch.long)
    if leaf == "running_var":
        return 0.5 + torch.rand(shape, generator=g)
    if leaf == "running_mean":
        return 0.1 * torch.randn(shape, generator=g)
    if len(shape) >= 2:                       # conv / linear / lstm matrices
        fan_in = 1
        for d in shape[1:]:
            fan_in *= d
        if "upsampling" in key:               # ConvTranspose2d layout [in,out,kh,kw]
            fan_in = shape[0] * shape[2] * shape[3] // 4
        return torch.randn(shape, generator=g) * (1.0 / max(fan_in, 1)) ** 0.5
    if leaf == "weight":                      # every 1-D weight is a norm scale
        return 1.0 + 0.1 * torch.randn(shape, generator=g)
    return 0.1 * torch.randn(shape, generator=g)


def fill_state_dict(shapes: dict, seed: int = 0) -> dict:
    """shapes: {key: shape}. Returns {key: tensor}."""
    return {k: fill_tensor(k, s, seed) for k, s in shapes.items()}


def weight_checksum(sd: dict) -> float:
    tot = 0.0
    for k in sorted(sd):
        t = sd[k]
        if t.dtype.is_floating_point:
            tot += float(t.double().abs().sum())
    return tot


def synth_sources(B: int, L: int, seed: int = 0, n_sources: int = 2, sr: int = 16000):
    """Synthetic 2-speaker mixtures: band-limited, slowly amplitude-modulated
    noise, peak 0.3.  Pure torch, deterministic."""
    g = _gen("synth_sources", seed)
    outs = []
    t = torch.arange(L, dtype=torch.float64) / sr
    for k in range(n_sources):
        w = torch.randn(B, L, generator=g, dtype=torch.float64)
        W = torch.fft.rfft(w, dim=-1)
        f = torch.fft.rfftfreq(L, 1.0 / sr).double()
        band = 1.0 / (1.0 + (80.0 / f.clamp(min=1e-3)) ** 8) / (1.0 + (f / 7000.0) ** 8)
        s = torch.fft.irfft(W * band.sqrt(), n=L, dim=-1)
        rate = 2.0 + 6.0 * torch.rand(B, 1, generator=g, dtype=torch.float64)
        ph = 6.283185307179586 * torch.rand(B, 1, generator=g, dtype=torch.float64)
        env = 0.5 - 0.5 * torch.cos(6.283185307179586 * rate * t.unsqueeze(0) + ph)
        s = s * env
        s = 0.3 * s / s.abs().amax(dim=-1, keepdim=True).clamp(min=1e-9)
        outs.append(s.float())
    return outs


def synth_video(B: int, T_v: int, seed: int = 0, H: int = 88, W: int = 88):
    g = _gen("synth_video", seed)
    v = torch.rand(B, 2, T_v, H, W, generator=g)
    return (v - 0.421) / 0.165


def stable_stft(x, n_fft, hop):
    """Machine-portable STFT features for fixtures: float64 DFT, then imaginary
    parts that are pure rounding noise (|im| < 1e-9 |re|, e.g. the reflect-padded
    first frame, which is symmetric) are snapped to +0 so that ``angle`` does not
    flip between -pi and +pi from one CPU's FFT rounding to another's.  Returns
    fp32 (mag, phase) [B, F, T]; same recipe as oracle.stft_mag_phase otherwise."""
    import math
    import torch.nn.functional as F
    xd = x.double()
    pad = n_fft // 2
    xp = F.pad(xd.unsqueeze(1), (pad, pad), mode="reflect").squeeze(1)
    frames = xp.unfold(-1, n_fft, hop)
    n = torch.arange(n_fft, dtype=torch.float64)
    win = 0.5 - 0.5 * torch.cos(2.0 * math.pi * n / n_fft)
    spec = torch.fft.rfft(frames * win, dim=-1).transpose(1, 2)
    re, im = spec.real.clone(), spec.imag.clone()
    scale = frames.abs().amax(dim=-1).unsqueeze(1) * n_fft          # [B,1,T] magnitude scale of a frame
    im = torch.where(im.abs() < 1e-9 * (re.abs() + 1e-3 * scale), torch.zeros_like(im), im)
    re = torch.where(re.abs() < 1e-12 * scale, torch.zeros_like(re), re)
    mag = torch.sqrt(re * re + im * im).float()
    phase = torch.atan2(im, re).float()
    return mag, phase
