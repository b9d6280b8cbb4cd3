"""Batch-dict producer on the device: the step right before the hot path.

The reference's data loader computes one STFT per item on CPU workers
(base_dataset.py:88-107,154-167: mix and every source) and ``collate_fn``
(collate.py:4-83) zero-pads audio, spectrograms and phases to the longest item
of the batch and stacks the two mouth crops.  Here the items carry waveforms
only: they are padded and uploaded once, the STFTs run on the GPU (DFT GEMM,
``ops.stft_mag_phase``) one launch group per distinct length -- every item
keeps its own reflect padding and frame count, as in the reference, and the
frames beyond an item's length are zero -- and the mouth crops travel as
``uint8`` and are scaled on the device (torchvision's ``ConvertImageDtype``:
x / 255, base_dataset.py:113-116).  The returned dict has the keys and shapes
of ``collate_fn``."""
from __future__ import annotations

import torch

from . import ops


def _to_1d(a):
    a = torch.as_tensor(a)
    return a.reshape(-1).float()


def collate_on_device(dataset_items, device, n_fft=1024, hop_length=256):
    """dataset_items: list of dicts with ``audio`` [1,L] (mix), ``sources`` (list of dicts with ``audio``),
    and optionally ``audio_original``, ``audio_path``, ``mouth1`` / ``mouth2`` ([T_v,H,W] uint8 or float)."""
    B = len(dataset_items)
    n_src = len(dataset_items[0]["sources"])
    lens = [int(_to_1d(it["audio"]).numel()) for it in dataset_items]
    Lmax = max(lens)
    Tn = [1 + L // hop_length for L in lens]
    Tmax, Fq = max(Tn), n_fft // 2 + 1
    # one pinned staging buffer for every waveform of the batch: [1 + n_src, B, Lmax]
    stage = torch.zeros(1 + n_src, B, Lmax, dtype=torch.float32).pin_memory()
    for b, it in enumerate(dataset_items):
        stage[0, b, :lens[b]] = _to_1d(it["audio"])
        for s in range(n_src):
            a = _to_1d(it["sources"][s]["audio"])
            if a.numel() != lens[b]:
                raise ValueError("collate_on_device: a source and its mixture differ in length")
            stage[1 + s, b, :lens[b]] = a
    wav = stage.to(device, non_blocking=True)
    mag = torch.zeros(1 + n_src, B, Fq, Tmax, dtype=torch.float32, device=device)
    ph = torch.zeros(1 + n_src, B, Fq, Tmax, dtype=torch.float32, device=device)
    groups = {}
    for b, L in enumerate(lens):
        groups.setdefault(L, []).append(b)
    for L, idx in groups.items():
        if L <= n_fft // 2:
            raise ValueError(f"collate_on_device: {L} samples are too few for reflect padding at n_fft {n_fft}")
        T = 1 + L // hop_length
        if len(groups) == 1:
            m, p = ops.stft_mag_phase(wav.reshape(-1, Lmax), n_fft, hop_length)
            mag, ph = m.view(1 + n_src, B, Fq, T), p.view(1 + n_src, B, Fq, T)
            break
        sel = torch.tensor(idx, device=device)
        x = wav[:, sel, :L].reshape(-1, L)
        m, p = ops.stft_mag_phase(x, n_fft, hop_length)
        mag[:, sel, :, :T] = m.view(1 + n_src, len(idx), Fq, T)
        ph[:, sel, :, :T] = p.view(1 + n_src, len(idx), Fq, T)
    batch = {
        "spectrogram": mag[0], "phase": ph[0],
        "spectrogram_length": torch.tensor(Tn),
        "audio": wav[0],
        "audio_lengths": lens,
        "sources": [{"audio": wav[1 + s], "spectrogram": mag[1 + s], "phase": ph[1 + s]} for s in range(n_src)],
        "audio_original": [it.get("audio_original") for it in dataset_items],
        "audio_path": [it.get("audio_path") for it in dataset_items],
        "audio_length": Lmax,
    }
    if dataset_items[0].get("mouth1") is not None or dataset_items[0].get("mouth2") is not None:
        vids = []
        for it in dataset_items:
            m1, m2 = it.get("mouth1"), it.get("mouth2")
            m1 = None if m1 is None else torch.as_tensor(m1)
            m2 = None if m2 is None else torch.as_tensor(m2)
            if m1 is None:
                m1 = torch.zeros_like(m2)
            if m2 is None:
                m2 = torch.zeros_like(m1)
            vids.append(torch.stack([m1, m2]))
        v = torch.stack(vids)
        if v.dtype == torch.uint8:                            # 4x fewer bytes over PCIe; scaled on the device
            batch["video"] = v.pin_memory().to(device, non_blocking=True).float().div_(255.0)
        else:
            batch["video"] = v.float().to(device, non_blocking=True)
    return batch


def spec_augment_on_device(x, p=0.9, time_drop_width=16, time_stripes_num=2, freq_drop_width=16, freq_stripes_num=2,
                           generator=None):
    """SpecAugment stripes on a batch already on the GPU.

    Reference: src/transforms/spec_augs/spec_augmentation.py:7-17 wraps torchlibrosa.augmentation.SpecAugmentation
    (torchlibrosa is a third-party dependency, not vendored and not installed here: its published algorithm is
    restated, parity unpinned) and the data loader applies it per item, independently to ``spectrogram`` and
    ``phase`` (instance_transforms/baseline.yaml:49-64), with probability ``p``.  torchlibrosa's DropStripes draws,
    per item and stripe, ``distance = randint(0, drop_width)`` and ``bgn = randint(0, width - distance)`` and zeroes
    ``[bgn, bgn + distance)``; its "time" axis is dim 2 and its "freq" axis dim 3 of a [B, 1, *, *] tensor, which for
    the reference's [1, F, T] items are F and T respectively (kept as is: parity is with the reference's behaviour).

    x [B, F, T] -> augmented copy.  All draws come from ``generator`` on x's device: no host synchronisation."""
    B, Fq, T = x.shape
    dev = x.device

    def keep_mask(width, drop_width, stripes):
        keep = torch.ones(B, width, dtype=torch.bool, device=dev)
        pos = torch.arange(width, device=dev).unsqueeze(0)
        for _ in range(stripes):
            dist = torch.floor(torch.rand(B, 1, device=dev, generator=generator) * drop_width)          # [0, drop_width)
            bgn = torch.floor(torch.rand(B, 1, device=dev, generator=generator) * (width - dist).clamp(min=1))
            keep &= ~((pos >= bgn) & (pos < bgn + dist))
        return keep

    apply = torch.rand(B, 1, 1, device=dev, generator=generator) <= p
    kf = keep_mask(Fq, time_drop_width, time_stripes_num).unsqueeze(2)        # torchlibrosa's dim 2 = our F axis
    kt = keep_mask(T, freq_drop_width, freq_stripes_num).unsqueeze(1)         # torchlibrosa's dim 3 = our T axis
    return torch.where(apply & ~(kf & kt), torch.zeros((), dtype=x.dtype, device=dev), x)
