#!/usr/bin/env python3
"""Generate the golden fixtures in this directory FROM THE REFERENCE ITSELF.

Runs only in the build container (needs /root/reference); the reference never
travels to the GPU box, the small ``.npz`` fixtures written here do.

Import recipe: the reference's package ``__init__``s pull in
gdown / hydra / torchmetrics, which are not installed, so ``src``, ``src.model``,
``src.loss`` and ``src.metrics`` are registered as bare namespace packages and
``hydra.utils.instantiate`` / ``gdown`` get inert stubs (gdown.download raises).

Weights and inputs come from ``oracle/fill.py`` (a pure function of key, shape,
seed), so fixtures hold outputs + a weight checksum only.

torchaudio (the STFT) is not installed: STFT fixtures come from ``torch.stft``
with the arguments torchaudio's Spectrogram defaults resolve to -- parity at
that boundary is therefore "unpinned" against torchaudio proper.
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)

from oracle.fill import fill_tensor, weight_checksum, synth_sources, synth_video, stable_stft  # noqa: E402


def _install_reference():
    for name, sub in (("src", "src"), ("src.model", "src/model"), ("src.loss", "src/loss"),
                      ("src.metrics", "src/metrics")):
        m = types.ModuleType(name)
        m.__path__ = [os.path.join(REF, sub)]
        sys.modules[name] = m
    hydra = types.ModuleType("hydra")
    hu = types.ModuleType("hydra.utils")
    hu.instantiate = lambda *a, **k: (_ for _ in ()).throw(RuntimeError("stub"))
    hydra.utils = hu
    sys.modules["hydra"], sys.modules["hydra.utils"] = hydra, hu
    gd = types.ModuleType("gdown")

    def _no_download(*a, **k):
        raise RuntimeError("network fetch attempted")
    gd.download = _no_download
    sys.modules["gdown"] = gd
    sys.path.insert(0, REF)


MANIFEST = {}


def refill(module, seed, name=None):
    sd = module.state_dict()
    if name is not None:
        MANIFEST[name] = {k: list(v.shape) for k, v in sd.items()}
    new = {k: fill_tensor(k, v.shape, seed).to(v.dtype) for k, v in sd.items()}
    module.load_state_dict(new)
    return weight_checksum(new)


def torch_stft(x, n_fft, hop):
    s = torch.stft(x, n_fft, hop, n_fft, torch.hann_window(n_fft), center=True,
                   pad_mode="reflect", normalized=False, onesided=True, return_complex=True)
    return s.abs(), s.angle()


def make_batch(B, L, n_fft, hop, seed):
    s1, s2 = synth_sources(B, L, seed)
    mix = s1 + s2
    # model inputs: machine-portable STFT features (oracle/fill.py:stable_stft); the
    # torch.stft known-answer vectors live in stft_kat.npz
    spec, phase = stable_stft(mix, n_fft, hop)
    sources = []
    for s in (s1, s2):
        sp, ph = stable_stft(s, n_fft, hop)
        sources.append({"audio": s, "spectrogram": sp, "phase": ph})
    return {"audio": mix, "spectrogram": spec, "phase": phase, "sources": sources,
            "audio_length": L}


def hook_taps(model, names):
    taps = {}
    mods = dict(model.named_modules())
    hs = []
    for n in names:
        def mk(n):
            def f(_m, _i, o):
                taps[n] = (o[0] if isinstance(o, tuple) else o).detach().clone()
            return f
        hs.append(mods[n].register_forward_hook(mk(n)))
    return taps, hs


def npify(d):
    out = {}
    for k, v in d.items():
        if torch.is_tensor(v):
            out[k] = v.detach().cpu().numpy()
        else:
            out[k] = np.asarray(v)
    return out


GRAD_KEYS_AUDIO = [
    "encoder.init_conv.weight", "encoder.init_conv.bias",
    "encoder.encoder_layers.0.tfc_tdf.conv1.0.0.weight",
    "encoder.encoder_layers.0.tfc_tdf.conv1.0.0.bias",
    "encoder.encoder_layers.0.tfc_tdf.fc.0.weight",
    "encoder.encoder_layers.0.tfc_tdf.fc.1.bias",
    "encoder.encoder_layers.1.tfc_tdf.skip_conv.weight",
    "encoder.encoder_layers.2.downsampling.downsampling.0.weight",
    "latent.tfc_tdf.conv2.2.0.weight",
    "latent.idp_modules.0.tc_rnn.norm.weight", "latent.idp_modules.0.tc_rnn.norm.bias",
    "latent.idp_modules.0.tc_rnn.blstm.weight_ih_l0",
    "latent.idp_modules.0.tc_rnn.blstm.weight_hh_l0_reverse",
    "latent.idp_modules.0.tc_rnn.blstm.bias_hh_l0",
    "latent.idp_modules.0.fc_rnn.blstm.weight_hh_l0",
    "latent.idp_modules.0.fc_rnn.fc.weight",
    "decoder.decoder_layers.0.upsampling.upsampling.0.weight",
    "decoder.decoder_layers.2.tfc_tdf.fc.1.weight",
    "decoder.out_conv.weight", "decoder.out_conv.bias",
]


def run_model_case(model, batch, loss_mod, metric_mods, tap_names, grad_keys, extra_inputs=None,
                   full_taps=True, slices=None, also_full=()):
    """Forward + PIT loss + backward through the REFERENCE; returns fixture dict."""
    fx = {}
    taps, hs = hook_taps(model, tap_names)
    kwargs = dict(spectrogram=batch["spectrogram"], phase=batch["phase"],
                  audio_length=batch["audio_length"])
    if extra_inputs:
        kwargs.update(extra_inputs)
    out = model(**kwargs)
    for h in hs:
        h.remove()
    res = loss_mod(audio=batch["audio"], sources=batch["sources"], preds=out["preds"])
    res["loss"].backward()
    fx["loss"] = res["loss"].detach()
    fx["batch_permuts"] = res["batch_permuts"]
    for i, p in enumerate(out["preds"]):
        for k in ("audio", "spectrogram", "phase"):
            t = p[k].detach()
            if full_taps:
                fx[f"pred{i}.{k}"] = t
            else:
                fx[f"pred{i}.{k}.abs_sum"] = t.double().abs().sum()
                fx[f"pred{i}.{k}.slice"] = t.reshape(t.shape[0], -1)[:, slices].clone()
    for n, t in taps.items():
        if full_taps:
            fx["tap." + n] = t
        else:
            fx[f"tap.{n}.abs_sum"] = t.double().abs().sum()
            fx[f"tap.{n}.slice"] = t.reshape(t.shape[0], -1)[:, slices].clone()
            if n in also_full:
                fx[f"tap.{n}.full"] = t.half()          # small taps whole (fp16: 2e-4 relative, far below the bars)
    params = dict(model.named_parameters())
    for k in grad_keys:
        g = params[k].grad
        if full_taps or g.numel() <= 4096:
            fx["grad." + k] = g.detach().clone()
        else:
            fx[f"grad.{k}.abs_sum"] = g.double().abs().sum()
            fx[f"grad.{k}.slice"] = g.reshape(-1)[:: max(1, g.numel() // 997)][:997].clone()
    with torch.no_grad():
        for name, m in metric_mods.items():
            if "Training" in name:
                fx["metric." + name] = m(audio=batch["audio"], preds=out["preds"],
                                         sources=batch["sources"],
                                         batch_permuts=res["batch_permuts"])
            else:
                fx["metric." + name] = m(audio=batch["audio"], preds=out["preds"],
                                         sources=batch["sources"])
    return fx


def main():
    torch.set_num_threads(8)
    _install_reference()
    from src.model.DTTNet.model_dttnet import DTTNetModel
    from src.model.DTTNet.modules.encoder import Encoder
    from src.model.DTTNet.modules.decoder import Decoder
    from src.model.DTTNet.modules.latent import LatentModule
    from src.model.DTTNet import model_dttnet_vision as mdv
    from src.model.VideoAudioFusions.cross_attention_positional import CrossAttentionPositional
    from src.model.VideoAudioFusions.temporal_alignment import TemporalAligner
    from src.model.VideoEncoders import resnet_video_encoder as rve
    from src.model.VideoEncoders.resnet import ResNet, BasicBlock, Swish
    from src.model.DTTNetPlus.blocks.block_fusion import FusionBlock
    from src.model.DTTNetPlus.modules.latent import LatentModule as LatentModulePlus
    from src.loss.dttnet_loss import DTTNetLoss
    from src.loss.pit_loss_wrapper import PITLossWrapper
    from src.loss.dttnet_loss_wrapper import DTTNetLossWrapper
    from src.metrics.si_snr import SiSnr
    from src.metrics.si_snr_i import SiSnrI
    from src.metrics.base_metric import PIT_wrapper_training, PIT_wrapper_inference

    class _M:  # the PIT metric classes live in files that import torchmetrics siblings
        def __init__(self, fn, training):
            self.fn, self.training = fn, training

        def __call__(self, **b):
            if self.training:
                return PIT_wrapper_training(self.fn, **b)
            return PIT_wrapper_inference(self.fn, **b)

    metrics = {"SiSnrPITTraining": _M(SiSnr(), True), "SiSnrPITInference": _M(SiSnr(), False),
               "SiSnrIPITTraining": _M(SiSnrI(), True), "SiSnrIPITInference": _M(SiSnrI(), False)}
    pit = PITLossWrapper(DTTNetLoss())

    # ---------------- STFT / iSTFT known-answer vectors ----------------------
    fx = {}
    x = synth_sources(2, 800, seed=3)[0]
    m, p = torch_stft(x, 64, 16)
    fx["x64"], fx["mag64"], fx["phase64"] = x, m, p
    s = torch.stft(x, 64, 16, 64, torch.hann_window(64), center=True, pad_mode="reflect",
                   normalized=False, onesided=True, return_complex=True)
    fx["re64"], fx["im64"] = s.real, s.imag
    g = torch.Generator().manual_seed(11)
    sp = torch.randn(2, 33, 51, generator=g)           # negative "magnitudes" on purpose
    ph = 3.0 * torch.randn(2, 33, 51, generator=g)
    fx["ispec64"], fx["iphase64"] = sp, ph
    fx["iaudio64"] = torch.istft(torch.polar(sp, ph), n_fft=64, hop_length=16, length=800)
    fx["iaudio64_short"] = torch.istft(torch.polar(sp, ph), n_fft=64, hop_length=16, length=777)
    xl = synth_sources(1, 32000, seed=4)[0]
    ml, pl = torch_stft(xl, 1024, 256)
    fx["mag1024_slice"] = ml[0, ::37, ::11]
    fx["phase1024_slice"] = pl[0, ::37, ::11]
    fx["mag1024_abs_sum"] = ml.double().abs().sum()
    rt = torch.istft(torch.polar(ml, pl), n_fft=1024, hop_length=256, length=32000)
    fx["roundtrip1024_head"] = rt[0, :600]
    fx["roundtrip1024_tail"] = rt[0, -600:]
    fx["roundtrip1024_mid_ratio_err"] = (rt[0, 256:-256] - 0.5 * xl[0, 256:-256]).abs().max()
    np.savez_compressed(os.path.join(HERE, "stft_kat.npz"), **npify(fx))
    print("stft_kat", {k: tuple(v.shape) for k, v in fx.items()})

    # ---------------- PIT / SI-SNR known-answer tests -------------------------
    fx = {}
    b = make_batch(3, 800, 64, 16, seed=5)
    srcs = b["sources"]
    # (a) preds == sources swapped  -> [1,0]; (b) identical sources -> tie -> [0,1]
    swapped = [dict(srcs[1]), dict(srcs[0])]
    r = pit(audio=b["audio"], sources=srcs, preds=swapped)
    fx["swap.loss"], fx["swap.permuts"] = r["loss"], r["batch_permuts"]
    same = [dict(srcs[0]), dict(srcs[0])]
    noisy = [{k: v + 0.01 for k, v in srcs[1].items()}, {k: v - 0.02 for k, v in srcs[0].items()}]
    r = pit(audio=b["audio"], sources=same, preds=noisy)
    fx["tie.loss"], fx["tie.permuts"] = r["loss"], r["batch_permuts"]
    r2 = DTTNetLossWrapper(DTTNetLoss())(audio=b["audio"], sources=srcs, preds=noisy)
    fx["fixed.loss"], fx["fixed.permuts"] = r2["loss"], r2["batch_permuts"]
    r3 = pit(audio=b["audio"], sources=srcs, preds=noisy)
    fx["noisy.loss"], fx["noisy.permuts"] = r3["loss"], r3["batch_permuts"]
    for name, mm in metrics.items():
        if "Training" in name:
            fx["noisy." + name] = mm(audio=b["audio"], preds=noisy, sources=srcs,
                                     batch_permuts=r3["batch_permuts"])
            fx["swap." + name] = mm(audio=b["audio"], preds=swapped, sources=srcs,
                                    batch_permuts=fx["swap.permuts"])
        else:
            fx["noisy." + name] = mm(audio=b["audio"], preds=noisy, sources=srcs)
            fx["swap." + name] = mm(audio=b["audio"], preds=swapped, sources=srcs)
    sis = SiSnr()
    fx["sisnr.vec"] = sis(noisy[0]["audio"], srcs[0]["audio"])
    fx["sisnri.vec"] = SiSnrI()(noisy[0]["audio"], srcs[0]["audio"], b["audio"])
    fx["sisnr.silence"] = sis(torch.zeros(2, 100) + 1e-4, torch.zeros(2, 100) + 2e-4)
    o = torch.zeros(1, 4)
    o[0, 0] = 1.0
    t = torch.zeros(1, 4)
    t[0, 1] = 1.0
    fx["sisnr.orthogonal"] = sis(o, t)
    np.savez_compressed(os.path.join(HERE, "loss_metric_kat.npz"), **npify(fx))
    print("loss_metric_kat ok", fx["swap.permuts"].tolist(), fx["tie.permuts"].tolist())

    # ---------------- tiny audio-only model, full tensors ----------------------
    cfg = dict(g=8, n_sources=2, n_layers=3, n_idp_layers=1, n_fft=64, hop_length=16, n_heads=2)
    model = DTTNetModel(**cfg)
    ck = refill(model, seed=1, name='tiny_audio')
    batch = make_batch(2, 800, 64, 16, seed=7)
    tap_names = ["encoder.init_conv", "encoder.encoder_layers.0.tfc_tdf.conv1.0",
                 "encoder.encoder_layers.0.tfc_tdf.conv1", "encoder.encoder_layers.0.tfc_tdf",
                 "encoder.encoder_layers.0.downsampling", "encoder.encoder_layers.1",
                 "encoder.encoder_layers.2", "latent.tfc_tdf", "latent.idp_modules.0", "latent",
                 "decoder.decoder_layers.0.upsampling", "decoder.decoder_layers.0",
                 "decoder.decoder_layers.1", "decoder.decoder_layers.2", "decoder.out_conv"]
    fx = run_model_case(model, batch, pit, metrics, tap_names, GRAD_KEYS_AUDIO)
    fx["weight_checksum"] = ck
    fx["in.audio"] = batch["audio"]
    fx["in.spectrogram"], fx["in.phase"] = batch["spectrogram"], batch["phase"]
    for i, s_ in enumerate(batch["sources"]):
        fx[f"in.source{i}.audio"] = s_["audio"]
    np.savez_compressed(os.path.join(HERE, "tiny_audio.npz"), **npify(fx))
    print("tiny_audio loss", float(fx["loss"]), "ck", ck)

    # ---------------- TC_FC_RNN / IDP alone (teacher-forced, bigger C') -------
    from src.model.DTTNet.modules.latent import IDPModule
    idp = IDPModule(2, 64)
    ck = refill(idp, seed=2, name='idp')
    g = torch.Generator().manual_seed(21)
    xin = torch.randn(2, 64, 5, 7, generator=g, requires_grad=True)
    y = idp(xin)
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)
    fx = {"x": xin.detach(), "y": y.detach(), "gy": gy, "gx": xin.grad, "weight_checksum": ck}
    for k, p_ in idp.named_parameters():
        fx["grad." + k] = p_.grad
    np.savez_compressed(os.path.join(HERE, "idp.npz"), **npify(fx))

    # ---------------- TFC-TDF block alone --------------------------------------
    from src.model.DTTNet.blocks.block_tfc_tdf import TFC_TDF_Block
    blk = TFC_TDF_Block(16, 17)
    ck = refill(blk, seed=3, name='tfc_tdf')
    xin = torch.randn(2, 16, 17, 13, generator=g, requires_grad=True)
    y = blk(xin)
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)
    fx = {"x": xin.detach(), "y": y.detach(), "gy": gy, "gx": xin.grad, "weight_checksum": ck}
    for k, p_ in blk.named_parameters():
        fx["grad." + k] = p_.grad
    np.savez_compressed(os.path.join(HERE, "tfc_tdf.npz"), **npify(fx))

    # ---------------- FiLM fusion block + FiLM latent --------------------------
    fb = FusionBlock(audio_channels=16, audio_freq=5, video_channels=24)
    ck = refill(fb, seed=4, name='film')
    a = torch.randn(2, 16, 5, 7, generator=g, requires_grad=True)
    v = torch.randn(2, 11, 24, generator=g, requires_grad=True)
    y = fb(a, v)
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)
    fx = {"a": a.detach(), "v": v.detach(), "y": y.detach(), "gy": gy, "ga": a.grad, "gv": v.grad,
          "weight_checksum": ck}
    for k, p_ in fb.named_parameters():
        fx["grad." + k] = p_.grad
    lat = LatentModulePlus(fc_dim=5, n_heads=2, n_channels=32, n_layers=1, video_embed_dim=24)
    fx["latent_weight_checksum"] = refill(lat, seed=5, name='film_latent')
    a2 = torch.randn(2, 32, 5, 7, generator=g)
    fx["lat_in"] = a2
    fx["lat_out"] = lat(a2, v.detach()).detach()
    np.savez_compressed(os.path.join(HERE, "film.npz"), **npify(fx))

    # ---------------- cross-attention / temporal-aligner fusion ----------------
    ca = CrossAttentionPositional(audio_channels=32, video_channels=48, n_heads=2)
    ck = refill(ca, seed=6, name='xattn')
    a = torch.randn(2, 32, 5, 7, generator=g, requires_grad=True)
    v = torch.randn(2, 2, 24, 11, generator=g, requires_grad=True)
    y = ca(a, v)
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)
    fx = {"a": a.detach(), "v": v.detach(), "y": y.detach(), "gy": gy, "ga": a.grad, "gv": v.grad,
          "weight_checksum": ck}
    for k, p_ in ca.named_parameters():
        fx["grad." + k] = p_.grad
    ta = TemporalAligner(audio_channels=32, video_channels=48)
    fx["ta_weight_checksum"] = refill(ta, seed=7, name='temporal_aligner')
    fx["ta_y"] = ta(a.detach(), v.detach()).detach()
    np.savez_compressed(os.path.join(HERE, "xattn.npz"), **npify(fx))

    # ---------------- ResNet-18 lip encoder (reference forward, no download) ---
    enc = rve.VideoEncoder.__new__(rve.VideoEncoder)
    torch.nn.Module.__init__(enc)
    enc.frontend_nout, enc.backend_out = 64, 512
    enc.trunk = ResNet(BasicBlock, [2, 2, 2, 2], relu_type="swish")
    enc.frontend3D = torch.nn.Sequential(
        torch.nn.Conv3d(1, 64, kernel_size=(5, 7, 7), stride=(1, 2, 2), padding=(2, 3, 3), bias=False),
        torch.nn.BatchNorm3d(64), Swish(),
        torch.nn.MaxPool3d(kernel_size=(1, 3, 3), stride=(1, 2, 2), padding=(0, 1, 1)))
    ck = refill(enc, seed=8, name='resnet18')
    vid = synth_video(1, 3, seed=9)
    enc.eval()                      # eval first: the train pass updates the running stats
    y_eval = enc(vid).detach()
    enc.train()
    y_train = enc(vid).detach()
    np.savez_compressed(os.path.join(HERE, "resnet18.npz"),
                        **npify({"y_train": y_train, "y_eval": y_eval, "weight_checksum": ck}))
    print("resnet18", tuple(y_train.shape))

    # ---------------- tiny AV cross-attention model through the REAL forward ---
    def build_vision(g_, n_idp, n_fft, hop, n_heads, fusion_method, vch):
        m = mdv.DTTNetVisionModel.__new__(mdv.DTTNetVisionModel)
        torch.nn.Module.__init__(m)
        m.n_sources, m.n_fft, m.hop_length = 2, n_fft, hop
        fc = n_fft // 2 + 1
        m.audio_dim = g_ * 8
        m.fusion_method = fusion_method
        m.video_encoder = torch.nn.Identity()           # features fed in directly
        if fusion_method == "cross_attention_positional":
            m.fusion = CrossAttentionPositional(m.audio_dim, 2 * vch, n_heads)
        else:
            m.fusion = TemporalAligner(m.audio_dim, 2 * vch)
        m.encoder = Encoder(fc, 2, g_, 3, False)
        m.decoder = Decoder(fc, g_, 2, 3, False)
        m.latent = LatentModule((fc + 8) // 8, n_heads, g_ * 16, n_idp)
        m.channel_compression = torch.nn.Sequential(
            torch.nn.Conv2d(m.audio_dim * 2, m.audio_dim, 1),
            torch.nn.InstanceNorm2d(m.audio_dim), torch.nn.GELU())
        return m

    vm = build_vision(8, 1, 64, 16, 2, "cross_attention_positional", 24)
    ck = refill(vm, seed=10, name='tiny_xattn')
    batch = make_batch(2, 800, 64, 16, seed=12)
    vf = torch.randn(2, 2, 24, 9, generator=g)
    taps_v = ["fusion", "latent", "channel_compression", "decoder.out_conv"]
    gk = ["fusion.q_audio.weight", "fusion.cross_attn.in_proj_weight", "fusion.video_proj.0.weight",
          "fusion.cross_ffn.2.bias", "fusion.audio_gate.0.weight", "fusion.inp_layernorm.weight",
          "channel_compression.0.weight", "encoder.init_conv.weight",
          "latent.idp_modules.0.fc_rnn.blstm.weight_ih_l0_reverse"]
    fx = run_model_case(vm, batch, pit, metrics, taps_v, gk, extra_inputs={"video": vf})
    fx["weight_checksum"] = ck
    fx["in.video_features"] = vf
    fx["in.audio"] = batch["audio"]
    for i, s_ in enumerate(batch["sources"]):
        fx[f"in.source{i}.audio"] = s_["audio"]
    np.savez_compressed(os.path.join(HERE, "tiny_xattn.npz"), **npify(fx))
    print("tiny_xattn loss", float(fx["loss"]))

    # ---------------- full-size configs: scalars, slices, checksums ------------
    sl = slice(0, None, 4001)
    model = DTTNetModel(g=32, n_sources=2, n_layers=3, n_idp_layers=4, n_fft=1024, hop_length=256,
                        n_heads=2)
    ck = refill(model, seed=1, name='full_audio')
    batch = make_batch(2, 32000, 1024, 256, seed=7)
    taps_f = ["encoder.init_conv", "encoder.encoder_layers.0.tfc_tdf", "encoder.encoder_layers.2",
              "latent.tfc_tdf", "latent.idp_modules.0", "latent", "decoder.decoder_layers.2",
              "decoder.out_conv"]
    fx = run_model_case(model, batch, pit, metrics, taps_f, GRAD_KEYS_AUDIO, full_taps=False,
                        slices=sl)
    fx["weight_checksum"] = ck
    fx["n_params"] = sum(p_.numel() for p_ in model.parameters())
    np.savez_compressed(os.path.join(HERE, "full_audio.npz"), **npify(fx))
    print("full_audio loss", float(fx["loss"]), "params", int(fx["n_params"]))

    vm = build_vision(32, 4, 1024, 256, 2, "cross_attention_positional", 512)
    ck = refill(vm, seed=10, name='full_xattn')
    batch = make_batch(1, 32000, 1024, 256, seed=12)
    vf = torch.randn(1, 2, 512, 50, generator=g).half().float()   # stored as fp16, exactly
    fx = run_model_case(vm, batch, pit, metrics, taps_v, gk, extra_inputs={"video": vf},
                        full_taps=False, slices=sl)
    fx["weight_checksum"] = ck
    fx["in.video_features.abs_sum"] = vf.double().abs().sum()
    fx["in.video_features"] = vf.half()
    fx["n_params"] = sum(p_.numel() for p_ in vm.parameters())
    np.savez_compressed(os.path.join(HERE, "full_xattn.npz"), **npify(fx))
    print("full_xattn loss", float(fx["loss"]), "params", int(fx["n_params"]))
    import json
    with open(os.path.join(HERE, "manifest.json"), "w") as f:
        json.dump(MANIFEST, f, indent=0, sort_keys=True)


def main_r2():
    """Round-2 fixtures (own generators: the round-1 fixtures above are not regenerated and stay bit-identical).
    ``python make_golden.py r2 [lipreading] [film_av] [xattn_g64]``."""
    torch.set_num_threads(8)
    _install_reference()
    import json
    which = set(sys.argv[2:]) or {"lipreading", "film_av", "xattn_g64"}
    mpath = os.path.join(HERE, "manifest.json")
    with open(mpath) as f:
        MANIFEST.update(json.load(f))
    from src.model.lipreading.model import Lipreading
    from src.loss.dttnet_loss import DTTNetLoss
    from src.loss.pit_loss_wrapper import PITLossWrapper
    pit = PITLossWrapper(DTTNetLoss())
    tcn_options = {"num_layers": 4, "kernel_size": [3, 5, 7], "dropout": 0.2, "dwpw": True, "width_mult": 1}

    def lipreading():      # exactly the constructor call of DTTNetPlus/modules/video_encoder.py:22-31
        return Lipreading(modality="video", num_classes=500, tcn_options=tcn_options, densetcn_options={},
                          backbone_type="shufflenet", relu_type="relu", width_mult=1.0, use_boundary=False,
                          extract_feats=True)

    if "lipreading" in which:
        # ---- frozen lip-reading front end alone: eval (running stats) and train (batch stats) mode ----
        enc = lipreading()
        sd = enc.state_dict()
        keep = {k: v for k, v in sd.items() if not k.startswith("tcn.")}
        MANIFEST["lipreading"] = {k: list(v.shape) for k, v in keep.items()}
        new = {k: (fill_tensor(k, v.shape, 31).to(v.dtype) if k in keep else v) for k, v in sd.items()}
        enc.load_state_dict(new)
        ck = weight_checksum({k: new[k] for k in keep})
        vid = synth_video(1, 4, seed=32)                      # [1,2,4,88,88]
        x = vid.reshape(2, 1, 4, 88, 88)
        enc.eval()
        with torch.no_grad():
            y_eval = enc(x, 4).reshape(1, 2, 4, 1024).transpose(1, 2).flatten(-2, -1)
            enc.train()
            y_train = enc(x, 4).reshape(1, 2, 4, 1024).transpose(1, 2).flatten(-2, -1)
        np.savez_compressed(os.path.join(HERE, "lipreading.npz"),
                            **npify({"y_eval": y_eval, "y_train": y_train, "weight_checksum": ck,
                                     "n_params_no_tcn": sum(v.numel() for k, v in keep.items()
                                                            if v.dtype.is_floating_point and "running" not in k)}))
        print("lipreading", tuple(y_train.shape), float(y_train.abs().mean()))

    if "film_av" in which:
        # ---- the composed FiLM model through the REAL DTTNetPlusModel.forward with raw video (tiny config) ----
        from src.model.DTTNetPlus.model_dttnet import DTTNetPlusModel
        m = DTTNetPlusModel(g=8, n_sources=2, n_layers=3, n_idp_layers=1, n_fft=64, hop_length=16, n_heads=2,
                            video_embed_dim=24, use_checkpoints=False, video_enc_weights_path=None)
        sd = m.state_dict()
        keep = {k: v for k, v in sd.items() if ".encoder.tcn." not in k}
        MANIFEST["film_av"] = {k: list(v.shape) for k, v in keep.items()}
        new = {k: (fill_tensor(k, v.shape, 33).to(v.dtype) if k in keep else v) for k, v in sd.items()}
        m.load_state_dict(new)
        m.train()
        batch = make_batch(2, 800, 64, 16, seed=34)
        vid = synth_video(2, 2, seed=35)                      # T_v = 2 frames for 0.05 s
        out = m(spectrogram=batch["spectrogram"], phase=batch["phase"], audio_length=800, video=vid)
        res = pit(audio=batch["audio"], sources=batch["sources"], preds=out["preds"])
        res["loss"].backward()
        fx = {"loss": res["loss"].detach(), "batch_permuts": res["batch_permuts"],
              "weight_checksum": weight_checksum({k: new[k] for k in keep}),
              "n_trainable": sum(p.numel() for p in m.parameters() if p.requires_grad)}
        for i, p in enumerate(out["preds"]):
            for k in ("audio", "spectrogram", "phase"):
                fx[f"pred{i}.{k}"] = p[k].detach()
        params = dict(m.named_parameters())
        for k in ("video_encoder.post_processor.weight", "latent.fusion_block.w_gamma_freq.weight",
                  "latent.fusion_block.w_beta_channels.bias", "encoder.init_conv.weight",
                  "latent.idp_modules.0.tc_rnn.blstm.weight_hh_l0", "decoder.out_conv.weight"):
            fx["grad." + k] = params[k].grad.detach().clone()
        frozen = [k for k, p in params.items() if k.startswith("video_encoder.encoder.")]
        assert all(params[k].grad is None for k in frozen)
        np.savez_compressed(os.path.join(HERE, "film_av.npz"), **npify(fx))
        print("film_av loss", float(fx["loss"]), "trainable", int(fx["n_trainable"]))

    if "xattn_g64" in which:
        # ---- cfg5: AV cross-attention 200M (g = 64), 6 s, B = 1, features fed in: scalars / slices ----
        from src.model.DTTNet.modules.encoder import Encoder
        from src.model.DTTNet.modules.decoder import Decoder
        from src.model.DTTNet.modules.latent import LatentModule
        from src.model.DTTNet import model_dttnet_vision as mdv
        from src.model.VideoAudioFusions.cross_attention_positional import CrossAttentionPositional
        from src.metrics.si_snr_i import SiSnrI
        from src.metrics.base_metric import PIT_wrapper_inference
        g_, n_fft, hop, L = 64, 1024, 256, 96000
        m = mdv.DTTNetVisionModel.__new__(mdv.DTTNetVisionModel)
        torch.nn.Module.__init__(m)
        m.n_sources, m.n_fft, m.hop_length = 2, n_fft, hop
        fc = n_fft // 2 + 1
        m.audio_dim = g_ * 8
        m.fusion_method = "cross_attention_positional"
        m.video_encoder = torch.nn.Identity()
        m.fusion = CrossAttentionPositional(m.audio_dim, 1024, 2)
        m.encoder = Encoder(fc, 2, g_, 3, False)
        m.decoder = Decoder(fc, g_, 2, 3, False)
        m.latent = LatentModule((fc + 8) // 8, 2, g_ * 16, 4)
        m.channel_compression = torch.nn.Sequential(torch.nn.Conv2d(m.audio_dim * 2, m.audio_dim, 1),
                                                    torch.nn.InstanceNorm2d(m.audio_dim), torch.nn.GELU())
        ck = refill(m, seed=41, name="full_xattn_g64")
        batch = make_batch(1, L, n_fft, hop, seed=42)
        gg = torch.Generator().manual_seed(43)
        vf = torch.randn(1, 2, 512, 150, generator=gg).half().float()
        sl = slice(0, None, 4001)
        taps_v = ["fusion", "latent.idp_modules.0", "latent", "channel_compression", "decoder.out_conv"]
        gk = ["fusion.q_audio.weight", "channel_compression.0.weight", "encoder.init_conv.weight",
              "latent.idp_modules.0.fc_rnn.blstm.weight_ih_l0_reverse", "latent.idp_modules.3.tc_rnn.blstm.weight_hh_l0",
              "decoder.out_conv.weight"]
        metrics = {"SiSnrIPITInference": type("M", (), {"__call__": lambda self, **b: PIT_wrapper_inference(SiSnrI(), **b)})()}
        fx = run_model_case(m, batch, pit, metrics, taps_v, gk, extra_inputs={"video": vf}, full_taps=False, slices=sl)
        fx["weight_checksum"] = ck
        fx["in.video_features"] = vf.half()
        fx["n_params"] = sum(p_.numel() for p_ in m.parameters())
        np.savez_compressed(os.path.join(HERE, "full_xattn_g64.npz"), **npify(fx))
        print("full_xattn_g64 loss", float(fx["loss"]), "params", int(fx["n_params"]))

    with open(mpath, "w") as f:
        json.dump(MANIFEST, f, indent=0, sort_keys=True)


def main_r3():
    """Round-3 fixtures: the two configurations whose end-to-end composition was untested at their own
    hyper-parameters.  ``python make_golden.py r3 [film_full] [xattn_video]``.

    film_full   -- BASELINE.json configs[2] (cfg3): the reference's REAL ``DTTNetPlusModel`` at
                   configs/model/dttnet_plus.yaml (g = 32, n_fft 1024, hop 256, 4 IDPs, video_embed_dim = 512,
                   incl. the YAML-only kwargs freeze_params / unfreeze_params), B = 2 x 2 s, raw lip video
                   [2, 2, 50, 88, 88], train mode.
    xattn_video -- configs[3] (cfg4) with RAW video: the reference's ``VideoEncoder`` (built with ``__new__``: its
                   ``__init__`` downloads a checkpoint) wired into the ``DTTNetVisionModel`` composition of ``main``,
                   forward through the real ``DTTNetVisionModel.forward`` (model_dttnet_vision.py:79-108), B = 1,
                   T_v = 50, train mode; gradient slices of the lip encoder's weights included."""
    torch.set_num_threads(8)
    _install_reference()
    import json
    which = set(sys.argv[2:]) or {"film_full", "xattn_video"}
    mpath = os.path.join(HERE, "manifest.json")
    with open(mpath) as f:
        MANIFEST.update(json.load(f))
    from src.loss.dttnet_loss import DTTNetLoss
    from src.loss.pit_loss_wrapper import PITLossWrapper
    from src.metrics.si_snr_i import SiSnrI
    from src.metrics.base_metric import PIT_wrapper_inference, PIT_wrapper_training
    pit = PITLossWrapper(DTTNetLoss())
    sl = slice(0, None, 4001)

    class _M:
        def __init__(self, training):
            self.training = training

        def __call__(self, **b):
            return (PIT_wrapper_training if self.training else PIT_wrapper_inference)(SiSnrI(), **b)

    metrics = {"SiSnrIPITTraining": _M(True), "SiSnrIPITInference": _M(False)}

    if "film_full" in which:
        from src.model.DTTNetPlus.model_dttnet import DTTNetPlusModel
        m = DTTNetPlusModel(g=32, n_sources=2, n_layers=3, n_idp_layers=4, n_heads=2, video_embed_dim=512,
                            use_checkpoints=False, video_enc_weights_path=None, n_fft=1024, hop_length=256,
                            freeze_params=None, unfreeze_params=None)
        sd = m.state_dict()
        keep = {k: v for k, v in sd.items() if ".encoder.tcn." not in k}
        MANIFEST["film_full"] = {k: list(v.shape) for k, v in keep.items()}
        new = {k: (fill_tensor(k, v.shape, 61).to(v.dtype) if k in keep else v) for k, v in sd.items()}
        m.load_state_dict(new)
        m.train()
        batch = make_batch(2, 32000, 1024, 256, seed=62)
        vid = synth_video(2, 50, seed=63)
        taps = ["video_encoder", "latent.tfc_tdf", "latent.fusion_block", "latent.idp_modules.0", "latent",
                "decoder.out_conv"]
        gk = ["video_encoder.post_processor.weight", "video_encoder.post_processor.bias",
              "latent.fusion_block.w_gamma_freq.weight", "latent.fusion_block.w_beta_channels.bias",
              "latent.fusion_block.w_gamma_channels.weight", "encoder.init_conv.weight",
              "latent.idp_modules.0.tc_rnn.blstm.weight_hh_l0", "latent.idp_modules.3.fc_rnn.fc.weight",
              "decoder.out_conv.weight"]
        fx = run_model_case(m, batch, pit, metrics, taps, gk, extra_inputs={"video": vid}, full_taps=False, slices=sl,
                            also_full=("video_encoder",))
        fx["weight_checksum"] = weight_checksum({k: new[k] for k in keep})
        fx["n_trainable"] = sum(p.numel() for p in m.parameters() if p.requires_grad)
        fx["n_params_no_tcn"] = sum(v.numel() for k, v in keep.items() if v.dtype.is_floating_point and "running" not in k)
        params = dict(m.named_parameters())
        assert all(p.grad is None for k, p in params.items() if k.startswith("video_encoder.encoder."))
        np.savez_compressed(os.path.join(HERE, "film_full.npz"), **npify(fx))
        print("film_full loss", float(fx["loss"]), "trainable", int(fx["n_trainable"]), "SI-SNRi",
              float(fx["metric.SiSnrIPITInference"]))

    if "xattn_video" in which:
        from src.model.DTTNet.modules.encoder import Encoder
        from src.model.DTTNet.modules.decoder import Decoder
        from src.model.DTTNet.modules.latent import LatentModule
        from src.model.DTTNet import model_dttnet_vision as mdv
        from src.model.VideoAudioFusions.cross_attention_positional import CrossAttentionPositional
        from src.model.VideoEncoders import resnet_video_encoder as rve
        from src.model.VideoEncoders.resnet import ResNet, BasicBlock, Swish
        g_, n_fft, hop = 32, 1024, 256
        enc = rve.VideoEncoder.__new__(rve.VideoEncoder)          # the constructor's body minus the download
        torch.nn.Module.__init__(enc)
        enc.frontend_nout, enc.backend_out = 64, 512
        enc.trunk = ResNet(BasicBlock, [2, 2, 2, 2], relu_type="swish")
        enc.frontend3D = torch.nn.Sequential(
            torch.nn.Conv3d(1, 64, kernel_size=(5, 7, 7), stride=(1, 2, 2), padding=(2, 3, 3), bias=False),
            torch.nn.BatchNorm3d(64), Swish(),
            torch.nn.MaxPool3d(kernel_size=(1, 3, 3), stride=(1, 2, 2), padding=(0, 1, 1)))
        m = mdv.DTTNetVisionModel.__new__(mdv.DTTNetVisionModel)
        torch.nn.Module.__init__(m)
        m.n_sources, m.n_fft, m.hop_length = 2, n_fft, hop
        fc = n_fft // 2 + 1
        m.audio_dim = g_ * 8
        m.fusion_method = "cross_attention_positional"
        m.video_encoder = enc
        m.fusion = CrossAttentionPositional(m.audio_dim, 1024, 2)
        m.encoder = Encoder(fc, 2, g_, 3, False)
        m.decoder = Decoder(fc, g_, 2, 3, False)
        m.latent = LatentModule((fc + 8) // 8, 2, g_ * 16, 4)
        m.channel_compression = torch.nn.Sequential(torch.nn.Conv2d(m.audio_dim * 2, m.audio_dim, 1),
                                                    torch.nn.InstanceNorm2d(m.audio_dim), torch.nn.GELU())
        ck = refill(m, seed=71, name="xattn_video")
        m.train()
        batch = make_batch(1, 32000, n_fft, hop, seed=72)
        vid = synth_video(1, 50, seed=73)
        taps = ["video_encoder", "fusion", "latent", "channel_compression", "decoder.out_conv"]
        gk = ["video_encoder.frontend3D.0.weight", "video_encoder.frontend3D.1.weight",
              "video_encoder.trunk.layer1.0.conv1.weight", "video_encoder.trunk.layer2.0.downsample.0.weight",
              "video_encoder.trunk.layer3.0.conv1.weight", "video_encoder.trunk.layer4.1.conv2.weight",
              "video_encoder.trunk.layer4.1.bn2.bias", "fusion.video_proj.0.weight", "fusion.q_audio.weight",
              "channel_compression.0.weight", "encoder.init_conv.weight", "decoder.out_conv.weight"]
        fx = run_model_case(m, batch, pit, metrics, taps, gk, extra_inputs={"video": vid}, full_taps=False, slices=sl,
                            also_full=("video_encoder",))
        fx["weight_checksum"] = ck
        fx["n_params"] = sum(p_.numel() for p_ in m.parameters())
        # BatchNorm running statistics after this ONE train-mode forward (momentum 0.1 from the filled buffers)
        bufs = dict(m.named_buffers())
        for k in ("video_encoder.frontend3D.1.running_mean", "video_encoder.trunk.layer4.1.bn2.running_var"):
            fx["buf." + k] = bufs[k].detach().clone()
        np.savez_compressed(os.path.join(HERE, "xattn_video.npz"), **npify(fx))
        print("xattn_video loss", float(fx["loss"]), "params", int(fx["n_params"]), "SI-SNRi",
              float(fx["metric.SiSnrIPITInference"]))

    with open(mpath, "w") as f:
        json.dump(MANIFEST, f, indent=0, sort_keys=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "r2":
        main_r2()
    elif len(sys.argv) > 1 and sys.argv[1] == "r3":
        main_r3()
    else:
        main()
