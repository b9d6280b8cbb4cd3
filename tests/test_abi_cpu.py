"""CPU: the C-ABI library loads and exports every symbol include/dtt_hip.h declares
(no compute calls), the ctypes struct matches the header, and the product refuses
to run without a GPU (no CPU fallback)."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    txt = open(os.path.join(ROOT, "include", "dtt_hip.h")).read()
    return sorted(set(re.findall(r"\b(dtt_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    from dtt_amd._lib import lib, DttGemm
    L = lib()
    names = _declared()
    assert len(names) >= 25
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/dtt_hip.h but not exported"
    assert L.dtt_version() >= 1
    # struct layout agrees with what a C compiler makes of the header
    import subprocess
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "s.c")
        open(src, "w").write('#include <stdio.h>\n#include <stddef.h>\n#include "dtt_hip.h"\nint main(){printf("%zu %zu %zu",'
                             'sizeof(DttGemm), offsetof(DttGemm, M), offsetof(DttGemm, sh_edge));return 0;}')
        exe = os.path.join(d, "s")
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), src, "-o", exe])
        size, off_m, off_e = map(int, subprocess.check_output([exe]).split())
    assert ctypes.sizeof(DttGemm) == size
    assert DttGemm.M.offset == off_m and DttGemm.sh_edge.offset == off_e


def test_header_compiles_as_c():
    import subprocess
    subprocess.check_call(["gcc", "-std=c99", "-fsyntax-only", "-x", "c", os.path.join(ROOT, "include", "dtt_hip.h")])


def test_no_cpu_fallback():
    import dtt_amd
    from dtt_amd._lib import DttError
    m = dtt_amd.DTTNetModel(g=8, n_layers=3, n_idp_layers=1, n_fft=64, hop_length=16)
    with pytest.raises(DttError):
        m(spectrogram=torch.zeros(1, 33, 5), phase=torch.zeros(1, 33, 5), audio_length=64)
    with pytest.raises(DttError):
        dtt_amd.stft_mag_phase(torch.zeros(1, 800), 64, 16)


def test_state_dict_contract():
    import dtt_amd
    from conftest import load_manifest
    man = load_manifest("full_audio")
    m = dtt_amd.DTTNetModel(g=32, n_sources=2, n_layers=3, n_idp_layers=4, n_fft=1024, hop_length=256, n_heads=2)
    sd = m.state_dict()
    assert set(sd) == set(man)
    for k, v in sd.items():
        assert tuple(v.shape) == man[k], k
    assert sum(p.numel() for p in m.parameters()) == 15166699


def test_reference_checkpoint_interchange(tmp_path):
    """The reference's checkpoint schema (base_trainer.py:596-625) and its loader's key / shape
    filtering (utils.py:77-86, model_loader.py:199-239), read with weights_only=True."""
    import dtt_amd
    torch.manual_seed(0)
    kw = dict(g=8, n_sources=2, n_layers=3, n_idp_layers=1, n_fft=64, hop_length=16, n_heads=2)
    a = dtt_amd.DTTNetModel(**kw)
    b = dtt_amd.DTTNetModel(**kw)
    opt = torch.optim.AdamW(a.parameters(), lr=1e-3)
    path = tmp_path / "checkpoint-epoch3.pth"
    st = dtt_amd.save_reference_checkpoint(path, a, epoch=3, optimizer=opt, monitor_best=1.5, config={"trainer": {"n_epochs": 4}})
    assert set(st) == {"arch", "epoch", "state_dict", "optimizer", "lr_scheduler", "monitor_best", "config"}
    assert st["arch"] == "DTTNetModel"
    missing, unexpected, skipped = dtt_amd.load_reference_checkpoint(b, path, map_location="cpu")
    assert not missing and not unexpected and not skipped
    for (k, v), (_, w) in zip(a.state_dict().items(), b.state_dict().items()):
        assert torch.equal(v, w), k
    # a DDP / torch.compile style state_dict (HF model.pth), one foreign key, one tensor of another shape
    c = dtt_amd.DTTNetModel(**kw)
    raw = {"module._orig_mod." + k: v.clone() for k, v in a.state_dict().items()}
    raw["module.foreign_head.weight"] = torch.zeros(3)
    key = "encoder.init_conv.weight"
    raw["module._orig_mod." + key] = torch.zeros(5, 5)
    before = c.state_dict()[key].clone()
    missing, unexpected, skipped = dtt_amd.load_reference_checkpoint(c, raw)
    assert missing == [key] and not unexpected and sorted(skipped) == sorted([key, "foreign_head.weight"])
    assert torch.equal(c.state_dict()[key], before)
    assert torch.equal(c.state_dict()["decoder.out_conv.weight"], a.state_dict()["decoder.out_conv.weight"])


def test_per_item_wav_writer(tmp_path):
    """Inferencer.process_batch's dump (inferencer.py:169-181): <dir>/<part>/source_<k>/<stem>.wav, float32 WAV."""
    import dtt_amd
    from scipy.io import wavfile
    g = torch.Generator().manual_seed(0)
    preds = [{"audio": torch.randn(3, 800, generator=g) * 0.1} for _ in range(2)]
    batch = {"preds": preds, "audio_path": ["/data/a/mix_0.wav", "mix_1.flac", "dir/x.y.wav"]}
    paths = dtt_amd.save_separated(batch, tmp_path, part="val", sample_rate=16000)
    assert len(paths) == 6
    for k in range(2):
        for i, stem in enumerate(["mix_0", "mix_1", "x.y"]):
            sr, data = wavfile.read(tmp_path / "val" / f"source_{k}" / f"{stem}.wav")
            assert sr == 16000 and data.dtype == "float32" and data.shape == (800,)
            assert torch.equal(torch.from_numpy(data.copy()), preds[k]["audio"][i])
