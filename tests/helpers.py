"""Shared test helpers: seeded batches / weights identical to the ones the
golden generator (tests/golden/make_golden.py) fed to the reference."""
import torch

import oracle
from oracle.fill import fill_state_dict, synth_sources, weight_checksum, stable_stft  # noqa: F401


def make_batch(B, L, n_fft, hop, seed):
    s1, s2 = synth_sources(B, L, seed)
    mix = s1 + s2
    spec, phase = stable_stft(mix, n_fft, hop)
    sources = []
    for s in (s1, s2):
        sp, ph = stable_stft(s, n_fft, hop)
        sources.append({"audio": s, "spectrogram": sp, "phase": ph})
    return {"audio": mix, "spectrogram": spec, "phase": phase, "sources": sources,
            "audio_length": L}


def to_dev(batch, dev):
    out = {}
    for k, v in batch.items():
        if torch.is_tensor(v):
            out[k] = v.to(dev)
        elif isinstance(v, list):
            out[k] = [{kk: vv.to(dev) for kk, vv in d.items()} for d in v]
        else:
            out[k] = v
    return out


def rel_err(a, b):
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    return float((a - b).norm() / b.norm().clamp(min=1e-30))


def grad_close(a, b, rtol, atol=2e-6):
    """||a-b|| <= rtol*||b|| + atol*sqrt(n): gradients that are mathematically
    zero (bias ahead of a normalisation, key bias under softmax) hold only
    rounding noise on both sides and must not be compared relatively."""
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    return float((a - b).norm()) <= rtol * float(b.norm()) + atol * (b.numel() ** 0.5)
