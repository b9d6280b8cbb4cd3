"""SI-SNR / SI-SNRi metrics with the reference's API (src/metrics/{base_metric,
si_snr,si_snr_i,si_snr_PIT,si_snr_i_PIT}.py) on the fused HIP SI-SNR reduction:
one launch set computes SI-SNR of every (estimate, source) pair, mixture
included, for the whole batch."""
from __future__ import annotations

from itertools import permutations

import torch

from . import ops
from ._lib import DttError


class BaseMetric:
    def __init__(self, name=None, *args, **kwargs):
        self.name = name if name is not None else type(self).__name__

    def __call__(self, **batch):
        raise NotImplementedError()


class SiSnr(BaseMetric):
    """si_snr.py:6-34 (no mean removal, eps=1e-5 in both denominators)."""

    def __init__(self, *args, **kwargs):
        super().__init__()
        self.eps = 1e-5

    def __call__(self, pred_audio_batch, true_audio_batch, mix_audio_batch=None):
        return ops.si_snr_matrix(pred_audio_batch.unsqueeze(0), [true_audio_batch], self.eps)[0, 0]


class SiSnrI(BaseMetric):
    """si_snr_i.py:7-31."""

    def __init__(self, *args, **kwargs):
        super().__init__()
        self.si_snr = SiSnr()

    def __call__(self, pred_audio_batch, true_audio_batch, mix_audio_batch):
        m = ops.si_snr_matrix(torch.stack([pred_audio_batch, mix_audio_batch]), [true_audio_batch], self.si_snr.eps)
        return m[0, 0] - m[1, 0]


def _pair_matrix(audio, preds, sources, improvement):
    """[n_pred, n_src, B] metric of every (pred, source) pair."""
    if not audio.is_cuda:
        raise DttError("dtt_amd metrics run on an MI355X only (no CPU path)")
    est = torch.stack([p["audio"].detach() for p in preds] + [audio])
    m = ops.si_snr_matrix(est, [s["audio"] for s in sources], 1e-5)
    n = len(preds)
    return m[:n] - m[n:n + 1] if improvement else m[:n]


def PIT_wrapper_training(improvement, audio, preds, sources, batch_permuts, **batch):
    """base_metric.py:28-42: perm[pred] = source from the loss."""
    m = _pair_matrix(audio, preds, sources, improvement)             # [P,S,B]
    B = audio.shape[0]
    idx = batch_permuts.t().unsqueeze(1)                             # [P,1,B]
    score = m.gather(1, idx).squeeze(1).sum(0)                       # [B]
    out = (score / len(sources)).detach().cpu().mean()
    ops.handoff_check(audio.device)                                  # the .cpu() above synchronised: read the status too
    return out


def PIT_wrapper_inference(improvement, audio, preds, sources, **batch):
    """base_metric.py:45-67: perm[src] = pred, argmax (first maximum wins)."""
    m = _pair_matrix(audio, preds, sources, improvement)
    n = len(sources)
    rows = []
    for perm in permutations(range(n), n):
        tot = 0
        for s_i, p_i in enumerate(perm):
            tot = tot + m[p_i, s_i]
        rows.append(tot)
    scores = torch.stack(rows)
    best = scores.argmax(dim=0)
    out = (scores.gather(0, best.unsqueeze(0)).squeeze(0) / n).mean().cpu()
    ops.handoff_check(audio.device)
    return out


class SiSnrPITTraining(BaseMetric):
    def __call__(self, audio, preds, sources, batch_permuts, **batch):
        return PIT_wrapper_training(False, audio, preds, sources, batch_permuts)


class SiSnrPITInference(BaseMetric):
    def __call__(self, audio, preds, sources, **batch):
        return PIT_wrapper_inference(False, audio, preds, sources)


class SiSnrIPITTraining(BaseMetric):
    def __call__(self, audio, preds, sources, batch_permuts, **batch):
        return PIT_wrapper_training(True, audio, preds, sources, batch_permuts)


class SiSnrIPITInference(BaseMetric):
    def __call__(self, audio, preds, sources, **batch):
        return PIT_wrapper_inference(True, audio, preds, sources)
