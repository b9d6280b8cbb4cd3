"""CPU: pin the oracle (oracle/dtt_oracle.py) against fixtures generated FROM
THE REFERENCE (tests/golden/make_golden.py).  fp32 tolerance 1e-5 relative
(BASELINE.json north_star "1e-5 in fp32 mode"); permutations bit-exact."""
import math

import pytest
import torch

import oracle
from oracle.fill import fill_state_dict, weight_checksum, synth_sources, synth_video
from conftest import load_golden, load_manifest
from helpers import make_batch, rel_err, grad_close

TOL = 1e-5


def _sd(name, seed):
    sd = fill_state_dict(load_manifest(name), seed)
    return sd


def test_stft_istft_known_answers():
    fx = load_golden("stft_kat")
    re_im = oracle.stft_complex(fx["x64"], 64, 16)
    assert rel_err(re_im.real, fx["re64"]) < TOL
    assert rel_err(re_im.imag, fx["im64"]) < TOL
    mag, ph = oracle.stft_mag_phase(fx["x64"], 64, 16)
    assert rel_err(mag, fx["mag64"]) < TOL
    # wrap-aware phase comparison, weighted by magnitude (phase of ~0 bins is noise)
    d = torch.remainder(ph - fx["phase64"] + math.pi, 2 * math.pi) - math.pi
    assert float((d.abs() * fx["mag64"]).max()) < 1e-4
    # DC / Nyquist rows: imaginary part is +0 so angle is exactly 0 or +pi
    for row in (0, 32):
        assert torch.all((ph[:, row] == 0) | (ph[:, row] == math.pi))
        assert torch.equal(ph[:, row], fx["phase64"][:, row])
    y = oracle.istft_rect(fx["ispec64"], fx["iphase64"], 64, 16, 800)
    assert rel_err(y, fx["iaudio64"]) < TOL
    y = oracle.istft_rect(fx["ispec64"], fx["iphase64"], 64, 16, 777)
    assert rel_err(y, fx["iaudio64_short"]) < TOL
    # full-size n_fft=1024 / hop=256 slices and the 0.5x round trip
    xl = synth_sources(1, 32000, seed=4)[0]
    ml, pl = oracle.stft_mag_phase(xl, 1024, 256)
    assert ml.shape == (1, 513, 126)
    assert rel_err(ml[0, ::37, ::11], fx["mag1024_slice"]) < TOL
    assert abs(float(ml.double().abs().sum()) / float(fx["mag1024_abs_sum"]) - 1) < 1e-6
    rt = oracle.istft_rect(ml, pl, 1024, 256, 32000)
    assert float((rt[0, :600] - fx["roundtrip1024_head"]).abs().max()) < 2e-6
    assert float((rt[0, -600:] - fx["roundtrip1024_tail"]).abs().max()) < 2e-6
    assert float((rt[0, 256:-256] - 0.5 * xl[0, 256:-256]).abs().max()) < 2e-6
    # envelope 3 on the first hop: ratio ramps 0.5 -> 2/3
    assert abs(float(rt[0, 0] / xl[0, 0]) - 2.0 / 3.0) < 0.2


def test_loss_and_metric_known_answers():
    fx = load_golden("loss_metric_kat")
    b = make_batch(3, 800, 64, 16, seed=5)
    srcs = b["sources"]
    swapped = [dict(srcs[1]), dict(srcs[0])]
    r = oracle.pit_loss(b["audio"], srcs, swapped)
    assert torch.equal(r["batch_permuts"], fx["swap.permuts"])
    assert r["batch_permuts"].dtype == torch.int64
    assert abs(float(r["loss"]) - float(fx["swap.loss"])) < 1e-6
    same = [dict(srcs[0]), dict(srcs[0])]
    noisy = [{k: v + 0.01 for k, v in srcs[1].items()}, {k: v - 0.02 for k, v in srcs[0].items()}]
    r = oracle.pit_loss(b["audio"], same, noisy)
    assert torch.equal(r["batch_permuts"], fx["tie.permuts"])       # tie -> first permutation
    assert rel_err(r["loss"], fx["tie.loss"]) < TOL
    r = oracle.fixed_order_loss(b["audio"], srcs, noisy)
    assert torch.equal(r["batch_permuts"], fx["fixed.permuts"])
    assert rel_err(r["loss"], fx["fixed.loss"]) < TOL
    r3 = oracle.pit_loss(b["audio"], srcs, noisy)
    assert torch.equal(r3["batch_permuts"], fx["noisy.permuts"])
    assert rel_err(r3["loss"], fx["noisy.loss"]) < TOL
    fns = {"SiSnr": oracle.si_snr, "SiSnrI": oracle.si_snr_i}
    for base, fn in fns.items():
        v = oracle.pit_metric_training(fn, b["audio"], noisy, srcs, r3["batch_permuts"])
        assert abs(float(v) - float(fx[f"noisy.{base}PITTraining"])) < 1e-3
        v = oracle.pit_metric_inference(fn, b["audio"], noisy, srcs)
        assert abs(float(v) - float(fx[f"noisy.{base}PITInference"])) < 1e-3
        v = oracle.pit_metric_training(fn, b["audio"], swapped, srcs, fx["swap.permuts"])
        assert abs(float(v) - float(fx[f"swap.{base}PITTraining"])) < 2e-2   # ~100 dB regime
        v = oracle.pit_metric_inference(fn, b["audio"], swapped, srcs)
        assert abs(float(v) - float(fx[f"swap.{base}PITInference"])) < 2e-2
    assert rel_err(oracle.si_snr(noisy[0]["audio"], srcs[0]["audio"]), fx["sisnr.vec"]) < 1e-5
    assert rel_err(oracle.si_snr_i(noisy[0]["audio"], srcs[0]["audio"], b["audio"]),
                   fx["sisnri.vec"]) < 1e-5
    v = oracle.si_snr(torch.zeros(2, 100) + 1e-4, torch.zeros(2, 100) + 2e-4)
    assert rel_err(v, fx["sisnr.silence"]) < 1e-5
    o = torch.zeros(1, 4)
    o[0, 0] = 1.0
    t = torch.zeros(1, 4)
    t[0, 1] = 1.0
    assert torch.isinf(oracle.si_snr(o, t)).all() and torch.isinf(fx["sisnr.orthogonal"]).all()


def _grads_of(sd, keys):
    for k in keys:
        sd[k].requires_grad_(True)


def test_tiny_audio_model_layers_loss_grads():
    fx = load_golden("tiny_audio")
    sd = _sd("tiny_audio", 1)
    assert abs(weight_checksum(sd) / float(fx["weight_checksum"]) - 1) < 1e-9
    gkeys = [k[5:] for k in fx if k.startswith("grad.")]
    _grads_of(sd, gkeys)
    b = make_batch(2, 800, 64, 16, seed=7)
    assert rel_err(b["spectrogram"], fx["in.spectrogram"]) < TOL
    taps = {}
    out = oracle.dttnet_forward(sd, b["spectrogram"], b["phase"], 800, n_fft=64, hop=16, taps=taps)
    name_map = {
        "encoder.init_conv": "encoder.init_conv",
        "encoder.encoder_layers.0.tfc_tdf.conv1.0": "encoder.encoder_layers.0.tfc_tdf.conv1.0",
        "encoder.encoder_layers.0.tfc_tdf.conv1": "encoder.encoder_layers.0.tfc_tdf.conv1.2",
        "encoder.encoder_layers.0.tfc_tdf": "encoder.encoder_layers.0.tfc_tdf",
        "encoder.encoder_layers.1": "encoder.encoder_layers.1",
        "encoder.encoder_layers.2": "encoder.encoder_layers.2",
        "latent.tfc_tdf": "latent.tfc_tdf",
        "latent.idp_modules.0": "latent.idp_modules.0",
        "decoder.decoder_layers.0": "decoder.decoder_layers.0",
        "decoder.decoder_layers.1": "decoder.decoder_layers.1",
        "decoder.decoder_layers.2": "decoder.decoder_layers.2",
        "decoder.out_conv": "decoder.out_conv",
    }
    for ref_name, tap_name in name_map.items():
        e = rel_err(taps[tap_name], fx["tap." + ref_name])
        # fp32 rounding noise doubles per TFC-TDF block
        assert e < (1e-4 if ref_name.startswith("decoder") else 2e-5), (ref_name, e)
    for i in range(2):
        for k in ("audio", "spectrogram", "phase"):
            e = rel_err(out["preds"][i][k], fx[f"pred{i}.{k}"])
            assert e < 1e-4, (i, k, e)
    r = oracle.pit_loss(b["audio"], b["sources"], out["preds"])
    assert torch.equal(r["batch_permuts"], fx["batch_permuts"])
    assert rel_err(r["loss"], fx["loss"]) < TOL
    r["loss"].backward()
    for k in gkeys:
        if k.endswith("conv1.0.0.bias"):
            # bias ahead of InstanceNorm: the true gradient is 0, both sides hold rounding noise
            assert float(sd[k].grad.abs().max()) < 1e-5 and float(fx["grad." + k].abs().max()) < 1e-5
            continue
        e = rel_err(sd[k].grad, fx["grad." + k])
        # L1 loss => sign() gradients; fp32 noise flips a few signs, and it grows with depth
        assert e < 2e-3, (k, e)
    with torch.no_grad():
        for base, fn in (("SiSnr", oracle.si_snr), ("SiSnrI", oracle.si_snr_i)):
            v = oracle.pit_metric_training(fn, b["audio"], out["preds"], b["sources"],
                                           r["batch_permuts"])
            assert abs(float(v) - float(fx[f"metric.{base}PITTraining"])) < 1e-3
            v = oracle.pit_metric_inference(fn, b["audio"], out["preds"], b["sources"])
            assert abs(float(v) - float(fx[f"metric.{base}PITInference"])) < 1e-3


def test_idp_and_tfc_tdf_blocks():
    fx = load_golden("idp")
    sd = _sd("idp", 2)
    for v in sd.values():
        v.requires_grad_(True)
    x = fx["x"].clone().requires_grad_(True)
    y = oracle.idp(sd, "", x, 2)
    assert rel_err(y, fx["y"]) < TOL
    y.backward(fx["gy"])
    assert rel_err(x.grad, fx["gx"]) < 5e-5
    for k in sd:
        assert rel_err(sd[k].grad, fx["grad." + k]) < 5e-5, k
    fx = load_golden("tfc_tdf")
    sd = _sd("tfc_tdf", 3)
    for v in sd.values():
        v.requires_grad_(True)
    x = fx["x"].clone().requires_grad_(True)
    y = oracle.tfc_tdf(sd, "", x)
    assert rel_err(y, fx["y"]) < TOL
    y.backward(fx["gy"])
    assert rel_err(x.grad, fx["gx"]) < 5e-5
    for k in sd:
        if k.startswith("conv") and k.endswith("bias"):      # bias ahead of InstanceNorm: true grad 0
            assert float(sd[k].grad.abs().max()) < 1e-4
            continue
        assert rel_err(sd[k].grad, fx["grad." + k]) < 5e-5, k


def test_film_and_cross_attention_fusion():
    fx = load_golden("film")
    sd = _sd("film", 4)
    for v in sd.values():
        v.requires_grad_(True)
    a = fx["a"].clone().requires_grad_(True)
    v = fx["v"].clone().requires_grad_(True)
    y = oracle.film(sd, "", a, v)
    assert rel_err(y, fx["y"]) < TOL
    y.backward(fx["gy"])
    assert rel_err(a.grad, fx["ga"]) < TOL and rel_err(v.grad, fx["gv"]) < 5e-5
    for k in sd:
        assert rel_err(sd[k].grad, fx["grad." + k]) < 5e-5, k
    sdl = _sd("film_latent", 5)
    out = oracle.latent(sdl, "", fx["lat_in"], 2, video=fx["v"])
    assert rel_err(out, fx["lat_out"]) < TOL

    fx = load_golden("xattn")
    sd = _sd("xattn", 6)
    for t in sd.values():
        t.requires_grad_(True)
    a = fx["a"].clone().requires_grad_(True)
    v = fx["v"].clone().requires_grad_(True)
    y = oracle.cross_attention_positional(sd, "", a, v, 2)
    assert rel_err(y, fx["y"]) < TOL
    y.backward(fx["gy"])
    assert rel_err(a.grad, fx["ga"]) < 5e-5 and rel_err(v.grad, fx["gv"]) < 5e-5
    for k in sd:
        assert grad_close(sd[k].grad, fx["grad." + k], 1e-4), k
    sdt = _sd("temporal_aligner", 7)
    assert rel_err(oracle.temporal_aligner(sdt, "", fx["a"], fx["v"]), fx["ta_y"]) < TOL


def test_resnet18_video_encoder():
    fx = load_golden("resnet18")
    sd = _sd("resnet18", 8)
    assert abs(weight_checksum(sd) / float(fx["weight_checksum"]) - 1) < 1e-9
    vid = synth_video(1, 3, seed=9)
    assert rel_err(oracle.resnet18_video_encoder(sd, "", vid, training=True), fx["y_train"]) < 1e-4
    assert rel_err(oracle.resnet18_video_encoder(sd, "", vid, training=False), fx["y_eval"]) < 1e-4


def test_lipreading_front_end_and_film_av_model():
    """cfg3: the frozen ShuffleNetV2 lip-reading front end (fixture from the reference's own Lipreading object,
    train and eval BatchNorm modes) and the composed FiLM model through the reference's REAL
    DTTNetPlusModel.forward with raw video [B,2,T_v,88,88]."""
    import torch.nn.functional as F
    fx = load_golden("lipreading")
    sd = _sd("lipreading", 31)
    assert abs(weight_checksum(sd) / float(fx["weight_checksum"]) - 1) < 1e-9
    vid = synth_video(1, 4, seed=32)
    assert rel_err(oracle.lipreading_video_features(sd, "", vid, training=True), fx["y_train"]) < 2e-5
    assert rel_err(oracle.lipreading_video_features(sd, "", vid, training=False), fx["y_eval"]) < 2e-5
    fx = load_golden("film_av")
    sd = _sd("film_av", 33)
    assert abs(weight_checksum(sd) / float(fx["weight_checksum"]) - 1) < 1e-9
    gkeys = [k[5:] for k in fx if k.startswith("grad.")]
    _grads_of(sd, gkeys)
    b = make_batch(2, 800, 64, 16, seed=34)
    with torch.no_grad():
        feats = oracle.lipreading_video_features(sd, "video_encoder.encoder.", synth_video(2, 2, seed=35), training=True)
    venc = F.linear(feats, sd["video_encoder.post_processor.weight"], sd["video_encoder.post_processor.bias"])
    out = oracle.dttnet_film_forward(sd, b["spectrogram"], b["phase"], 800, venc, n_fft=64, hop=16)
    for i in range(2):
        for k in ("audio", "spectrogram", "phase"):
            assert rel_err(out["preds"][i][k], fx[f"pred{i}.{k}"]) < 3e-4, (i, k)   # end-to-end fp32 envelope
    r = oracle.pit_loss(b["audio"], b["sources"], out["preds"])
    assert torch.equal(r["batch_permuts"], fx["batch_permuts"])
    assert rel_err(r["loss"], fx["loss"]) < TOL
    r["loss"].backward()
    for k in gkeys:
        e = rel_err(sd[k].grad, fx["grad." + k])
        assert e < 2e-3, (k, e)    # sign() gradients of the L1 loss, see tiny_audio test


def test_tiny_xattn_model():
    fx = load_golden("tiny_xattn")
    sd = _sd("tiny_xattn", 10)
    gkeys = [k[5:] for k in fx if k.startswith("grad.")]
    _grads_of(sd, gkeys)
    b = make_batch(2, 800, 64, 16, seed=12)
    taps = {}
    out = oracle.dttnet_vision_forward(sd, b["spectrogram"], b["phase"], 800,
                                       fx["in.video_features"], "cross_attention_positional",
                                       n_fft=64, hop=16, taps=taps)
    assert rel_err(taps["fusion"], fx["tap.fusion"]) < 2e-5
    assert rel_err(taps["channel_compression"], fx["tap.channel_compression"]) < 5e-5
    assert rel_err(taps["decoder.out_conv"], fx["tap.decoder.out_conv"]) < 5e-5
    r = oracle.pit_loss(b["audio"], b["sources"], out["preds"])
    assert torch.equal(r["batch_permuts"], fx["batch_permuts"])
    assert rel_err(r["loss"], fx["loss"]) < TOL
    r["loss"].backward()
    for k in gkeys:
        e = rel_err(sd[k].grad, fx["grad." + k])
        assert e < 2e-3, (k, e)    # sign() gradients of the L1 loss, see tiny_audio test


@pytest.mark.parametrize("case", ["full_audio"])
def test_full_size_audio_model_scalars(case):
    """cfg1/cfg2 hyper-parameters (g=32, n_fft=1024, hop=256, 4 IDP layers), B=2.
    End-to-end fp32 tolerance is depth dependent (the
    reference is only ~5e-4 from its own fp64), so 2e-3 here."""
    fx = load_golden(case)
    sd = _sd(case, 1)
    assert abs(weight_checksum(sd) / float(fx["weight_checksum"]) - 1) < 1e-9
    assert sum(v.numel() for v in sd.values()) == int(fx["n_params"]) == 15166699
    b = make_batch(2, 32000, 1024, 256, seed=7)
    taps = {}
    with torch.no_grad():
        out = oracle.dttnet_forward(sd, b["spectrogram"], b["phase"], 32000, taps=taps)
        r = oracle.pit_loss(b["audio"], b["sources"], out["preds"])
    sl = slice(0, None, 4001)
    for n in ("encoder.init_conv", "encoder.encoder_layers.0.tfc_tdf", "latent.tfc_tdf",
              "decoder.out_conv"):
        t = taps[n]
        e = rel_err(t.reshape(t.shape[0], -1)[:, sl], fx[f"tap.{n}.slice"])
        assert e < 2e-3, (n, e)
    assert torch.equal(r["batch_permuts"], fx["batch_permuts"])
    assert abs(float(r["loss"]) / float(fx["loss"]) - 1) < 1e-4
    v = oracle.pit_metric_inference(oracle.si_snr_i, b["audio"], out["preds"], b["sources"])
    assert abs(float(v) - float(fx["metric.SiSnrIPITInference"])) < 0.05


def test_cfg3_and_cfg4_with_raw_video_at_full_size():
    """Round 3: the oracle's composition of lip encoder + U-Net at BASELINE.json configs[2] / configs[3]
    hyper-parameters with RAW video, against what the reference's real DTTNetPlusModel / DTTNetVisionModel.forward
    produced (make_golden.py r3): lip features whole, deep taps by slice, loss, permutation, SI-SNRi."""
    import torch.nn.functional as F
    sl = slice(0, None, 4001)
    # cfg3: ShuffleNetV2 front end (train-mode batch statistics over 2 x 2 x 50 frames) -> post_processor -> FiLM model
    fx = load_golden("film_full")
    sd = _sd("film_full", 61)
    assert abs(weight_checksum(sd) / float(fx["weight_checksum"]) - 1) < 1e-9
    b = make_batch(2, 32000, 1024, 256, seed=62)
    taps = {}
    with torch.no_grad():
        feats = oracle.lipreading_video_features(sd, "video_encoder.encoder.", synth_video(2, 50, seed=63), training=True)
        venc = F.linear(feats, sd["video_encoder.post_processor.weight"], sd["video_encoder.post_processor.bias"])
        assert rel_err(venc, fx["tap.video_encoder.full"].float()) < 5e-4          # stored as fp16
        out = oracle.dttnet_film_forward(sd, b["spectrogram"], b["phase"], 32000, venc, taps=taps)
        r = oracle.pit_loss(b["audio"], b["sources"], out["preds"])
    for n in ("latent.tfc_tdf", "decoder.out_conv"):
        t = taps[n]
        assert rel_err(t.reshape(t.shape[0], -1)[:, sl], fx[f"tap.{n}.slice"]) < 2e-3, n
    assert torch.equal(r["batch_permuts"], fx["batch_permuts"])
    assert abs(float(r["loss"]) / float(fx["loss"]) - 1) < 1e-4
    v = oracle.pit_metric_inference(oracle.si_snr_i, b["audio"], out["preds"], b["sources"])
    assert abs(float(v) - float(fx["metric.SiSnrIPITInference"])) < 0.05
    # cfg4: ResNet-18 lip encoder (train mode) -> cross-attention fusion -> 2C latent -> compression -> decoder
    fx = load_golden("xattn_video")
    sd = _sd("xattn_video", 71)
    assert abs(weight_checksum(sd) / float(fx["weight_checksum"]) - 1) < 1e-9
    b = make_batch(1, 32000, 1024, 256, seed=72)
    taps = {}
    with torch.no_grad():
        vf = oracle.resnet18_video_encoder(sd, "video_encoder.", synth_video(1, 50, seed=73), training=True)
        assert rel_err(vf, fx["tap.video_encoder.full"].float()) < 5e-4
        out = oracle.dttnet_vision_forward(sd, b["spectrogram"], b["phase"], 32000, vf, "cross_attention_positional", taps=taps)
        r = oracle.pit_loss(b["audio"], b["sources"], out["preds"])
    for n in ("fusion", "channel_compression", "decoder.out_conv"):
        t = taps[n]
        assert rel_err(t.reshape(t.shape[0], -1)[:, sl], fx[f"tap.{n}.slice"]) < 2e-3, n
    assert torch.equal(r["batch_permuts"], fx["batch_permuts"])
    assert abs(float(r["loss"]) / float(fx["loss"]) - 1) < 1e-4
    v = oracle.pit_metric_inference(oracle.si_snr_i, b["audio"], out["preds"], b["sources"])
    assert abs(float(v) - float(fx["metric.SiSnrIPITInference"])) < 0.05
