"""Checkpoint interchange with the reference.

The reference saves ``{"arch", "epoch", "state_dict", "optimizer", "lr_scheduler",
"monitor_best", "config"}`` (base_trainer.py:596-625) and loads either that
dict or a bare state_dict (HF ``model.pth``), after stripping wrapper prefixes
(utils.py:77-86) and dropping keys that are unknown or of another shape
(model_loader.py:199-239).  State-dict names and shapes of the dtt_amd models
equal the reference's, so its files load directly.  Files are read with
``weights_only=True``: nothing in the file is executed (the reference itself
uses ``weights_only=False``)."""
from __future__ import annotations

import torch

_PREFIXES = ("module.", "model.", "compile.", "_orig_mod.")


def align_state_dict_keys(state_dict):
    """utils.py:77-86: remove DDP / torch.compile / wrapper prefixes wherever they occur in a key."""
    out = {}
    for k, v in state_dict.items():
        for pre in _PREFIXES:
            k = k.replace(pre, "")
        out[k] = v
    return out


def load_reference_checkpoint(model, checkpoint, map_location=None):
    """model_loader.py:199-239.  ``checkpoint``: a path or an already loaded object.  Returns
    ``(missing, unexpected, skipped)``: ``skipped`` lists the keys dropped for an unknown name or
    a shape mismatch (the reference drops them silently)."""
    if isinstance(checkpoint, (str, bytes)) or hasattr(checkpoint, "__fspath__"):
        if map_location is None:
            map_location = next(model.parameters()).device
        checkpoint = torch.load(checkpoint, map_location=map_location, weights_only=True)
    sd = checkpoint["state_dict"] if isinstance(checkpoint, dict) and "state_dict" in checkpoint else checkpoint
    sd = align_state_dict_keys(sd)
    own = model.state_dict()
    kept, skipped = {}, []
    for k, v in sd.items():
        if k in own and tuple(v.shape) == tuple(own[k].shape):
            kept[k] = v
        else:
            skipped.append(k)
    missing, unexpected = model.load_state_dict(kept, strict=False)
    return list(missing), list(unexpected), skipped


def save_reference_checkpoint(path, model, epoch=0, optimizer=None, lr_scheduler=None, monitor_best=0.0, config=None):
    """base_trainer.py:596-625: the dict the reference's trainer writes (and ``_resume_checkpoint`` reads)."""
    dev = next((p.device for p in model.parameters() if p.is_cuda), None)
    if dev is not None:
        from . import ops
        ops.handoff_check(dev)      # never write weights that a timed-out hand-off launch may have poisoned
    state = {
        "arch": type(model).__name__,
        "epoch": epoch,
        "state_dict": {k: v.detach().cpu() for k, v in model.state_dict().items()},
        "optimizer": optimizer.state_dict() if optimizer is not None else None,
        "lr_scheduler": lr_scheduler.state_dict() if lr_scheduler is not None else None,
        "monitor_best": monitor_best,
        "config": config,
    }
    torch.save(state, path)
    return state


def write_wav_float32(path, audio, sample_rate=16000):
    """One mono/multi-channel float32 WAV (IEEE float, what ``torchaudio.save`` writes for a float32 tensor):
    audio [L] or [channels, L]."""
    import struct
    a = torch.as_tensor(audio).detach().to("cpu", torch.float32)
    if a.dim() == 1:
        a = a.unsqueeze(0)
    ch, n = a.shape
    data = a.t().contiguous().numpy().tobytes()                # interleaved frames
    fmt = struct.pack("<HHIIHH", 3, ch, sample_rate, sample_rate * ch * 4, ch * 4, 32)
    fact = struct.pack("<I", n)
    body = b"WAVE" + b"fmt " + struct.pack("<I", len(fmt)) + fmt + b"fact" + struct.pack("<I", 4) + fact \
        + b"data" + struct.pack("<I", len(data)) + data
    with open(path, "wb") as f:
        f.write(b"RIFF" + struct.pack("<I", len(body)) + body)


def save_separated(batch, save_dir, part="test", sample_rate=16000):
    """Per-item dump of the separated sources (Inferencer.process_batch, src/trainer/inferencer.py:169-181):
    ``<save_dir>/<part>/source_<k>/<stem of audio_path>.wav`` for every item and prediction.  One device-to-host
    copy per prediction tensor (the reference copies item by item).  Returns the written paths."""
    import os
    out = []
    preds = [p["audio"].detach().to("cpu", torch.float32) for p in batch["preds"]]
    for i, src_path in enumerate(batch["audio_path"]):
        stem = os.path.splitext(os.path.basename(str(src_path)))[0]
        for k, a in enumerate(preds):
            d = os.path.join(str(save_dir), part, f"source_{k}")
            os.makedirs(d, exist_ok=True)
            path = os.path.join(d, stem + ".wav")
            write_wav_float32(path, a[i], sample_rate)
            out.append(path)
    return out
