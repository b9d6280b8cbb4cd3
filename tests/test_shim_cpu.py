"""The drop-in boundary as the reference's callers see it: Hydra resolves ``_target_: src.model.DTTNetModel`` etc.
by importing ``src.model`` / ``src.loss`` / ``src.metrics`` (src/model/__init__.py:1-9, src/loss/__init__.py:1-5,
src/metrics/__init__.py:1-4) and calling the class with the YAML's keys.  With ``speech-separation_amd/`` first on
sys.path those import paths must resolve to the MI355X-native classes, and the YAML keys of
configs/model/{dttnet,dttnet_plus,dttnet_vision}.yaml, configs/train.yaml:32-35 and configs/metrics/*.yaml must be
accepted unchanged -- including the YAML-only ``freeze_params`` / ``unfreeze_params`` of dttnet_plus.yaml:12-13.
Construction needs no GPU (the modules are parameter containers); running them without one raises (no CPU path)."""
import importlib
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "speech-separation_amd")

# what hydra.utils.instantiate does with a `_target_` string, in four lines (hydra itself is not installed here)
def _instantiate(cfg):
    cfg = dict(cfg)
    mod, _, cls = cfg.pop("_target_").rpartition(".")
    kw = {k: (_instantiate(v) if isinstance(v, dict) and "_target_" in v else v) for k, v in cfg.items()}
    return getattr(importlib.import_module(mod), cls)(**kw)


YAML_MODELS = {
    # configs/model/dttnet.yaml
    "dttnet": dict(_target_="src.model.DTTNetModel", g=32, n_sources=2, n_layers=3, n_idp_layers=4, n_heads=2,
                   use_checkpoints=True, n_fft=1024, hop_length=256),
    # configs/model/dttnet_plus.yaml (freeze_params / unfreeze_params exist only in the YAML)
    "dttnet_plus": dict(_target_="src.model.DTTNetPlusModel", g=32, n_sources=2, n_layers=3, n_idp_layers=4, n_heads=2,
                        video_embed_dim=512, use_checkpoints=False, video_enc_weights_path=None, n_fft=1024,
                        hop_length=256, freeze_params=None, unfreeze_params=None),
    # configs/model/dttnet_vision.yaml, and the fusion the AV baselines were trained with
    "dttnet_vision": dict(_target_="src.model.DTTNetVisionModel", g=32, n_sources=2, n_layers=3, n_idp_layers=4,
                          n_heads=2, video_channels=512, fusion_method="temporal_alignment", use_checkpoints=True,
                          n_fft=1024, hop_length=256),
    "dttnet_vision_xattn": dict(_target_="src.model.DTTNetVisionModel", g=32, n_sources=2, n_layers=3, n_idp_layers=4,
                                n_heads=2, video_channels=512, fusion_method="cross_attention_positional",
                                use_checkpoints=True, n_fft=1024, hop_length=256),
}
N_PARAMS = {"dttnet": 15166699, "dttnet_plus": None, "dttnet_vision": None, "dttnet_vision_xattn": 60667307}


@pytest.fixture()
def shim_first(monkeypatch):
    monkeypatch.syspath_prepend(PKG)
    for name in [m for m in sys.modules if m == "src" or m.startswith("src.")]:
        monkeypatch.delitem(sys.modules, name)
    yield
    for name in [m for m in sys.modules if m == "src" or m.startswith("src.")]:
        sys.modules.pop(name, None)


@pytest.mark.parametrize("name", sorted(YAML_MODELS))
def test_model_targets_resolve_and_accept_the_yaml_keys(shim_first, name):
    import dtt_amd
    m = _instantiate(YAML_MODELS[name])
    assert type(m).__module__ == "dtt_amd.model" and isinstance(m, torch.nn.Module)
    assert sys.modules["src.model"].__file__.startswith(PKG)
    if N_PARAMS[name] is not None:
        assert sum(p.numel() for p in m.parameters()) == N_PARAMS[name]
    if name == "dttnet_plus":       # frozen lip-reading encoder, trainable rest (SURVEY: 16 545 133 trainable)
        assert sum(p.numel() for p in m.parameters() if p.requires_grad) == 16545133
    # forward signatures are the reference's (called as model(**batch)); without a GPU the product refuses loudly
    spec = torch.zeros(1, 513, 126)
    kw = dict(spectrogram=spec, phase=spec, audio_length=32000, audio=torch.zeros(1, 32000), epoch=0)
    if name != "dttnet":
        kw["video"] = torch.zeros(1, 2, 50, 88, 88)
    if not torch.cuda.is_available():
        with pytest.raises(dtt_amd.DttError):
            m(**kw)


def test_loss_and_metric_targets(shim_first):
    import dtt_amd
    # configs/train.yaml:32-35
    crit = _instantiate({"_target_": "src.loss.DTTNetLossWrapper", "loss_function": {"_target_": "src.loss.DTTNetLoss"}})
    assert isinstance(crit, dtt_amd.DTTNetLossWrapper) and isinstance(crit.loss_function, dtt_amd.DTTNetLoss)
    pit = _instantiate({"_target_": "src.loss.PITLossWrapper", "loss_function": {"_target_": "src.loss.DTTNetLoss"}})
    assert isinstance(pit, dtt_amd.PITLossWrapper)
    # configs/metrics/*.yaml: name is a constructor key, exposed as .name (base_metric.py:12-25)
    for cls, nm in (("SiSnrPITTraining", "SI-SNR"), ("SiSnrIPITTraining", "SI-SNRi"), ("SiSnrPITInference", "SI-SNR"),
                    ("SiSnrIPITInference", "SI-SNRi")):
        met = _instantiate({"_target_": f"src.metrics.{cls}", "name": nm})
        assert met.name == nm and type(met).__module__ == "dtt_amd.metrics"
    default = importlib.import_module("src.metrics").SiSnrIPITInference()
    assert default.name == "SiSnrIPITInference"


def test_unknown_fusion_method_is_the_references_error(shim_first):
    with pytest.raises(ValueError, match="not supported"):          # model_dttnet_vision.py:47-48
        _instantiate(dict(YAML_MODELS["dttnet_vision"], fusion_method="concat"))


@pytest.mark.gpu
def test_shim_classes_run_the_hip_path(shim_first):
    """Through the reference's import paths on the GPU: tiny model, loss and metric as Trainer.process_batch calls
    them (trainer.py:44-58,70-71) against the reference fixture of the same weights."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from conftest import load_golden, load_manifest
    from helpers import make_batch, to_dev, rel_err
    from oracle.fill import fill_state_dict
    fx = load_golden("tiny_audio")
    model = _instantiate(dict(_target_="src.model.DTTNetModel", g=8, n_sources=2, n_layers=3, n_idp_layers=1, n_heads=2,
                              use_checkpoints=True, n_fft=64, hop_length=16)).to("cuda")
    model.load_state_dict(fill_state_dict(load_manifest("tiny_audio"), 1))
    crit = _instantiate({"_target_": "src.loss.PITLossWrapper", "loss_function": {"_target_": "src.loss.DTTNetLoss"}})
    met = _instantiate({"_target_": "src.metrics.SiSnrIPITTraining", "name": "SI-SNRi"})
    batch = to_dev(make_batch(2, 800, 64, 16, seed=7), "cuda")
    batch.update(model(**batch))                     # precision "auto": fp32 outside autocast
    batch.update(crit(**batch))
    batch["loss"].backward()
    assert torch.equal(batch["batch_permuts"].cpu(), fx["batch_permuts"])
    assert rel_err(batch["loss"], fx["loss"]) < 1e-5
    v = met(**batch)
    assert v.device.type == "cpu" and v.dim() == 0
    assert abs(float(v) - float(fx["metric.SiSnrIPITTraining"])) < 0.05
