"""BASELINE.json configs[2] and configs[3] at their OWN hyper-parameters with RAW lip video, end to end against
fixtures the reference itself produced (tests/golden/make_golden.py r3):

* cfg3 -- the reference's real ``DTTNetPlusModel`` as configs/model/dttnet_plus.yaml builds it (g = 32, n_fft 1024,
  hop 256, 4 IDP layers, video_embed_dim = 512; ShuffleNetV2 lip front end frozen, FiLM fusion), B = 2 x 2 s, video
  [2, 2, 50, 88, 88], train mode (DTTNetPlus/model_dttnet.py:47-54);
* cfg4 -- ``DTTNetVisionModel.forward`` (model_dttnet_vision.py:79-108) with the ResNet-18 lip encoder IN the
  forward and backward (B = 1, 2 x 50 frames, train mode: batch statistics), incl. gradient slices of the lip
  encoder's own weights and its BatchNorm running statistics after the step.

fp32 against the reference's tensors (depth-dependent tolerance, SURVEY.md Appendix A.3: the reference is ~5e-4
from its own fp64 end to end); bf16 -- the mode bench.py times -- inside the reference's own bf16 envelope."""
import pytest
import torch

from oracle.fill import fill_state_dict, synth_video
from conftest import load_golden, load_manifest
from helpers import make_batch, rel_err, to_dev

pytestmark = pytest.mark.gpu
DEV = "cuda"
SL = slice(0, None, 4001)


def _slice(t):
    return t.reshape(t.shape[0], -1)[:, SL]


def _grad_slice(g):
    g = g.reshape(-1)
    return g[:: max(1, g.numel() // 997)][:997]


def _run(model, fx, batch, video, taps_tol, precision):
    from dtt_amd.model import _Tap
    import dtt_amd
    _Tap.store = {}
    try:
        out = model(video=video, **batch)
    finally:
        taps, _Tap.store = _Tap.store, None
    errs = {}
    for n, tol in taps_tol:
        if f"tap.{n}.full" in fx:                    # small taps are stored whole (fp16)
            errs[n] = rel_err(taps[n], fx[f"tap.{n}.full"].float())
        else:
            errs[n] = rel_err(_slice(taps[n]), fx[f"tap.{n}.slice"])
        assert errs[n] < tol, (n, errs[n], tol)
    batch.update(out)
    res = dtt_amd.PITLossWrapper(dtt_amd.DTTNetLoss())(**batch)
    res["loss"].backward()
    batch.update(res)
    return taps, res, errs


def _film_model(precision):
    import dtt_amd
    # the YAML passes freeze_params / unfreeze_params to the constructor (configs/model/dttnet_plus.yaml:12-13)
    model = dtt_amd.DTTNetPlusModel(g=32, n_sources=2, n_layers=3, n_idp_layers=4, n_heads=2, video_embed_dim=512,
                                    use_checkpoints=False, video_enc_weights_path=None, n_fft=1024, hop_length=256,
                                    freeze_params=None, unfreeze_params=None, precision=precision).to(DEV)
    man = load_manifest("film_full")
    assert set(man) == set(model.state_dict().keys())          # reference keys minus the never-run encoder.tcn.* head
    model.load_state_dict(fill_state_dict(man, 61))
    model.train()
    return model


def test_cfg3_film_full_size_raw_video_fp32_against_reference():
    import dtt_amd
    fx = load_golden("film_full")
    model = _film_model("fp32")
    assert sum(p.numel() for p in model.parameters() if p.requires_grad) == int(fx["n_trainable"]) == 16545133
    b = to_dev(make_batch(2, 32000, 1024, 256, seed=62), DEV)
    vid = synth_video(2, 50, seed=63).to(DEV)
    # measured: 2.1e-4 (lip features: 20 train-mode BatchNorms over 200 frames, MIOpen vs the reference's CPU convs),
    # 1e-5 at every latent tap, 7e-5 at out_conv
    tol = (("video_encoder", 1e-3), ("latent.tfc_tdf", 1e-4), ("latent.fusion_block", 1e-4),
           ("latent.idp_modules.0", 1e-4), ("latent", 1e-4), ("decoder.out_conv", 5e-4))
    _, res, errs = _run(model, fx, b, vid, tol, "fp32")
    print("cfg3 fp32 tap errors", {k: f"{v:.1e}" for k, v in errs.items()})
    assert torch.equal(res["batch_permuts"].cpu(), fx["batch_permuts"])
    assert abs(float(res["loss"].detach()) / float(fx["loss"]) - 1) < 2e-4
    for name in ("SiSnrIPITTraining", "SiSnrIPITInference"):
        v = getattr(dtt_amd, name)()(**b)
        assert abs(float(v) - float(fx["metric." + name])) < 0.05, (name, float(v))
    for i in range(2):
        assert rel_err(_slice(b["preds"][i]["audio"]), fx[f"pred{i}.audio.slice"]) < 2e-2, i
        assert abs(float(b["preds"][i]["spectrogram"].double().abs().sum()) / float(fx[f"pred{i}.spectrogram.abs_sum"]) - 1) < 2e-3
    params = dict(model.named_parameters())
    for k in [k[5:] for k in fx if k.startswith("grad.") and not k.endswith((".slice", ".abs_sum"))]:
        assert rel_err(params[k].grad, fx["grad." + k]) < 3e-2, k
    for k in [k[5:-6] for k in fx if k.startswith("grad.") and k.endswith(".slice")]:
        e = rel_err(_grad_slice(params[k].grad), fx[f"grad.{k}.slice"])
        assert e < 3e-2, (k, e)
    assert all(p.grad is None for n, p in params.items() if n.startswith("video_encoder.encoder."))


def test_cfg3_film_full_size_raw_video_bf16_envelope():
    fx = load_golden("film_full")
    model = _film_model("bf16")
    b = to_dev(make_batch(2, 32000, 1024, 256, seed=62), DEV)
    vid = synth_video(2, 50, seed=63).to(DEV)
    # lip features: 20 train-mode BatchNorms deep in bf16 (0.14 measured on 8 frames, test_lipreading_front_end);
    # U-Net depths: the reference's own bf16-vs-fp32 deviation x ~1.5
    tol = (("video_encoder", 0.3), ("latent.tfc_tdf", 0.3), ("latent.fusion_block", 0.3), ("latent", 0.3))
    _, res, errs = _run(model, fx, b, vid, tol, "bf16")
    print("cfg3 bf16 tap errors", {k: f"{v:.1e}" for k, v in errs.items()})
    assert abs(float(res["loss"].detach()) / float(fx["loss"]) - 1) < 0.1
    for k, p in model.named_parameters():
        if p.requires_grad:
            assert p.grad is not None and torch.isfinite(p.grad).all(), k


def _xattn_model(precision):
    import dtt_amd
    model = dtt_amd.DTTNetVisionModel(g=32, n_sources=2, n_layers=3, n_idp_layers=4, n_fft=1024, hop_length=256,
                                      n_heads=2, video_channels=512, fusion_method="cross_attention_positional",
                                      use_checkpoints=True, precision=precision).to(DEV)
    man = load_manifest("xattn_video")
    assert set(man) == set(model.state_dict().keys())
    model.load_state_dict(fill_state_dict(man, 71))
    model.train()
    return model


def test_cfg4_xattn_raw_video_fp32_against_reference():
    import dtt_amd
    fx = load_golden("xattn_video")
    model = _xattn_model("fp32")
    assert sum(p.numel() for p in model.parameters()) == int(fx["n_params"])
    b = to_dev(make_batch(1, 32000, 1024, 256, seed=72), DEV)
    vid = synth_video(1, 50, seed=73).to(DEV)
    # measured: 2.1e-4 (ResNet-18 on own kernels, train mode), 5e-6 fusion, 1e-5 latent, 2e-5 compression, 7e-5 out_conv
    tol = (("video_encoder", 1e-3), ("fusion", 1e-4), ("latent", 1e-4), ("channel_compression", 2e-4),
           ("decoder.out_conv", 5e-4))
    _, res, errs = _run(model, fx, b, vid, tol, "fp32")
    print("cfg4 fp32 tap errors", {k: f"{v:.1e}" for k, v in errs.items()})
    assert torch.equal(res["batch_permuts"].cpu(), fx["batch_permuts"])
    assert abs(float(res["loss"].detach()) / float(fx["loss"]) - 1) < 2e-4
    for name in ("SiSnrIPITTraining", "SiSnrIPITInference"):
        v = getattr(dtt_amd, name)()(**b)
        assert abs(float(v) - float(fx["metric." + name])) < 0.05, (name, float(v))
    assert rel_err(_slice(b["preds"][0]["audio"]), fx["pred0.audio.slice"]) < 2e-2
    params = dict(model.named_parameters())
    worst = {}
    for k in [k[5:] for k in fx if k.startswith("grad.") and not k.endswith((".slice", ".abs_sum"))]:
        worst[k] = rel_err(params[k].grad, fx["grad." + k])
    for k in [k[5:-6] for k in fx if k.startswith("grad.") and k.endswith(".slice")]:
        worst[k] = rel_err(_grad_slice(params[k].grad), fx[f"grad.{k}.slice"])
    print("cfg4 fp32 gradient errors", {k: f"{v:.1e}" for k, v in worst.items()})
    for k, e in worst.items():
        assert e < 3e-2, (k, e)        # sign() gradients of the L1 loss behind a random-weight network: see tiny_audio
    # one train-mode forward moved the BatchNorm running statistics exactly as the reference's did (momentum 0.1)
    bufs = dict(model.named_buffers())
    for k in [k[4:] for k in fx if k.startswith("buf.")]:
        assert rel_err(bufs[k], fx["buf." + k]) < 1e-3, k


def test_cfg4_xattn_raw_video_bf16_envelope():
    fx = load_golden("xattn_video")
    model = _xattn_model("bf16")
    b = to_dev(make_batch(1, 32000, 1024, 256, seed=72), DEV)
    vid = synth_video(1, 50, seed=73).to(DEV)
    tol = (("video_encoder", 0.15), ("fusion", 0.15), ("latent", 0.3), ("channel_compression", 0.4))
    _, res, errs = _run(model, fx, b, vid, tol, "bf16")
    print("cfg4 bf16 tap errors", {k: f"{v:.1e}" for k, v in errs.items()})
    assert abs(float(res["loss"].detach()) / float(fx["loss"]) - 1) < 0.1
    for k, p in model.named_parameters():
        assert p.grad is not None and torch.isfinite(p.grad).all(), k
    # a lip-encoder gradient keeps its direction through the whole bf16 network behind it (B = 1, L1's sign()
    # gradients: the reference's own bf16 run is 0.9 away from its fp32 run at out_conv; 0.49 measured)
    g = model.video_encoder.trunk.layer4[1].bn2.bias.grad.double().cpu()
    r = fx["grad.video_encoder.trunk.layer4.1.bn2.bias"].double()
    cos = float(torch.dot(g, r) / (g.norm() * r.norm()))
    assert cos > 0.3, cos
