"""Functional CPU restatement of the DTTNet separation path (TEST ORACLE).

Every function takes plain tensors and a ``state_dict`` whose key names and
shapes are the reference's, so the reference's own
weights can be fed straight in.  Layout is the reference's NCHW = [B, C, F, T].
All arithmetic runs in the dtype of the inputs (fp32 or fp64) on CPU.

``path:line`` citations are relative to ``/root/reference/``.  Nothing here is
imported by the product package.
"""
from __future__ import annotations

import math
from itertools import permutations

import torch
import torch.nn.functional as F

__all__ = [
    "stft_complex", "stft_mag_phase", "istft_rect", "conv_norm_gelu", "tfc_tdf",
    "encoder", "lstm_dir", "lstm_bidir", "tc_fc_rnn", "idp", "latent", "decoder", "mask_recon",
    "dttnet_forward", "film", "cross_attention_positional", "temporal_aligner",
    "resnet18_video_encoder", "lipreading_video_features", "dttnet_vision_forward", "dttnet_film_forward",
    "dttnet_l1", "pit_loss", "fixed_order_loss", "si_snr", "si_snr_i",
    "pit_metric_training", "pit_metric_inference", "sinusoids", "collate",
]


# --------------------------------------------------------------------------- #
# STFT front end / iSTFT back end
# --------------------------------------------------------------------------- #
def stft_complex(x, n_fft=1024, hop=256):
    """torchaudio ``Spectrogram(n_fft, hop_length=hop, power=None)`` restated.

    Follows src/configs/transforms/instance_transforms/baseline.yaml:2-6 and
    src/datasets/base_dataset.py:154-167 (torchaudio 2.8.0 is a third-party
    dependency, not vendored; its defaults resolve to: periodic Hann window,
    center=True with reflect padding, onesided, no normalisation).
    x: [B, L] -> complex [B, n_fft//2+1, 1 + L//hop].
    """
    pad = n_fft // 2
    xp = F.pad(x.unsqueeze(1), (pad, pad), mode="reflect").squeeze(1)
    frames = xp.unfold(-1, n_fft, hop)                       # [B, T, n_fft]
    n = torch.arange(n_fft, dtype=x.dtype)
    win = 0.5 - 0.5 * torch.cos(2.0 * math.pi * n / n_fft)   # periodic Hann
    spec = torch.fft.rfft(frames * win, dim=-1)              # [B, T, F]
    return spec.transpose(1, 2)


def stft_mag_phase(x, n_fft=1024, hop=256):
    """``abs`` / ``angle`` of the STFT (base_dataset.py:165-166)."""
    s = stft_complex(x, n_fft, hop)
    return s.abs(), s.angle()


def istft_rect(spec, phase, n_fft, hop, length):
    """``torch.istft(torch.polar(spec, phase), n_fft, hop, length=...)`` with
    window=None (rectangular), as called at
    src/model/DTTNet/model_dttnet.py:77-85, written out as an explicit
    irFFT + overlap-add + envelope division.  spec may be negative.
    spec, phase: [B, F, T] -> audio [B, length].
    """
    B, Fq, T = spec.shape
    # torch.polar on purpose: its autograd formula is grad_abs = Re(conj(g) * sgn(result)),
    # i.e. the mathematical d/d(abs) times sign(abs) (0 where abs == 0).  The reference
    # feeds NEGATIVE "magnitudes" into it (model_dttnet.py:73-78), so the reference's own
    # gradients carry that sign factor and parity is against them.
    frames = torch.fft.irfft(torch.polar(spec, phase).transpose(1, 2), n=n_fft, dim=-1)  # [B,T,n_fft]
    total = n_fft + hop * (T - 1)
    out = spec.new_zeros(B, total)
    env = spec.new_zeros(total)
    for t in range(T):
        out[:, t * hop:t * hop + n_fft] += frames[:, t]
        env[t * hop:t * hop + n_fft] += 1.0
    start = n_fft // 2
    out = out[:, start:start + length] / env[start:start + length]
    if out.shape[1] < length:
        out = F.pad(out, (0, length - out.shape[1]))
    return out


# --------------------------------------------------------------------------- #
# U-Net pieces
# --------------------------------------------------------------------------- #
def _gelu(x):
    return 0.5 * x * (1.0 + torch.erf(x / math.sqrt(2.0)))


def _inorm(x, eps=1e-5):
    mu = x.mean(dim=(2, 3), keepdim=True)
    var = x.var(dim=(2, 3), unbiased=False, keepdim=True)
    return (x - mu) / torch.sqrt(var + eps)


def conv_norm_gelu(x, w, b, stride=1, transposed=False):
    """conv3x3 -> InstanceNorm2d(affine=False, eps=1e-5) -> exact GELU.

    block_tfc_tdf.py:12-17, block_sampling.py:8-12 and :23-28."""
    if transposed:
        y = F.conv_transpose2d(x, w, b, stride=2, padding=1, output_padding=1)
    else:
        y = F.conv2d(x, w, b, stride=stride, padding=1)
    return _gelu(_inorm(y))


def tfc_tdf(sd, p, x, taps=None):
    """TFC_TDF_Block.forward_block (block_tfc_tdf.py:40-47)."""
    skip = F.conv2d(x, sd[p + "skip_conv.weight"], sd[p + "skip_conv.bias"], padding=1)
    h = x
    for i in range(3):
        h = conv_norm_gelu(h, sd[f"{p}conv1.{i}.0.weight"], sd[f"{p}conv1.{i}.0.bias"])
        if taps is not None:
            taps[f"{p}conv1.{i}"] = h
    z = h.transpose(-1, -2)                                   # [B,C,T,F]
    z = F.linear(z, sd[p + "fc.0.weight"], sd[p + "fc.0.bias"])
    z = F.linear(z, sd[p + "fc.1.weight"], sd[p + "fc.1.bias"])
    h = z.transpose(-1, -2) + h
    if taps is not None:
        taps[p + "tdf"] = h
    for i in range(3):
        h = conv_norm_gelu(h, sd[f"{p}conv2.{i}.0.weight"], sd[f"{p}conv2.{i}.0.bias"])
    out = h + skip
    if taps is not None:
        taps[p[:-1]] = out
    return out


def _n_levels(sd, p, stem):
    n = 0
    while f"{p}{stem}.{n}.tfc_tdf.skip_conv.weight" in sd:
        n += 1
    return n


def encoder(sd, p, spec, phase, taps=None):
    """Encoder.forward (modules/encoder.py:25-46). Returns (x, skips deepest-first)."""
    x = torch.stack([spec, phase], dim=1)
    x = F.conv2d(x, sd[p + "init_conv.weight"], sd[p + "init_conv.bias"])
    if taps is not None:
        taps[p + "init_conv"] = x
    skips = []
    for l in range(_n_levels(sd, p, "encoder_layers")):
        q = f"{p}encoder_layers.{l}."
        s = tfc_tdf(sd, q + "tfc_tdf.", x, taps)
        skips.insert(0, s)
        x = conv_norm_gelu(s, sd[q + "downsampling.downsampling.0.weight"],
                           sd[q + "downsampling.downsampling.0.bias"], stride=2)
        if taps is not None:
            taps[q[:-1]] = x
    return x, skips


def lstm_dir(xin, w_hh, rev):
    """One direction of the recurrence on pre-computed input projections: xin [N, L, 4H] = x W_ih^T + b_ih + b_hh,
    gate order i,f,g,o, zero initial state (modules/latent.py:87-92) -> h [N, L, H]."""
    N, L, _ = xin.shape
    H = w_hh.shape[1]
    h = xin.new_zeros(N, H)
    c = xin.new_zeros(N, H)
    ys = [None] * L
    order = range(L - 1, -1, -1) if rev else range(L)
    for t in order:
        g = xin[:, t] + F.linear(h, w_hh)
        i_, f_, g_, o_ = g.chunk(4, dim=1)
        c = torch.sigmoid(f_) * c + torch.sigmoid(i_) * torch.tanh(g_)
        h = torch.sigmoid(o_) * torch.tanh(c)
        ys[t] = h
    return torch.stack(ys, dim=1)


def lstm_bidir(x, sd, p):
    """Single-layer bidirectional ``nn.LSTM(batch_first=True)`` with zero
    initial state, gate order i,f,g,o (modules/latent.py:87-92), as an explicit
    time loop.  x: [N, L, I] -> [N, L, 2H]."""
    outs = []
    for suffix, rev in (("", False), ("_reverse", True)):
        w_ih = sd[f"{p}weight_ih_l0{suffix}"]
        w_hh = sd[f"{p}weight_hh_l0{suffix}"]
        bias = sd[f"{p}bias_ih_l0{suffix}"] + sd[f"{p}bias_hh_l0{suffix}"]
        outs.append(lstm_dir(F.linear(x, w_ih, bias), w_hh, rev))      # xin [N,L,4H]
    return torch.cat(outs, dim=-1)


def tc_fc_rnn(sd, p, x):
    """TC_FC_RNN.forward (modules/latent.py:98-104). x: [N, L, C']."""
    Cp = x.shape[-1]
    groups = max(1, Cp // 16)
    y = F.group_norm(x.transpose(-1, -2), groups, sd[p + "norm.weight"], sd[p + "norm.bias"],
                     eps=1e-5).transpose(-1, -2)
    y = lstm_bidir(y, sd, p + "blstm.")
    y = F.linear(y, sd[p + "fc.weight"], sd[p + "fc.bias"])
    return y + x


def idp(sd, p, x, n_heads, taps=None):
    """IDPModule.forward incl. split/merge heads (modules/latent.py:39-74)."""
    B, C, Fq, T = x.shape
    Cp = C // n_heads
    z = x.view(B, n_heads, Cp, Fq, T).permute(0, 1, 3, 4, 2).reshape(B * n_heads * Fq, T, Cp)
    z = tc_fc_rnn(sd, p + "tc_rnn.", z)
    if taps is not None:
        taps[p + "tc_rnn"] = z.reshape(B, n_heads, Fq, T, Cp).permute(0, 1, 4, 2, 3).reshape(B, C, Fq, T)
    z = z.reshape(B * n_heads, Fq, T, Cp).permute(0, 2, 1, 3).reshape(B * n_heads * T, Fq, Cp)
    z = tc_fc_rnn(sd, p + "fc_rnn.", z)
    # merge_heads (latent.py:68-74) VIEWS the [B', T, F, C'] result as
    # [B, heads, F, T, C'] without permuting T and F back: the (t, f) plane is
    # reinterpreted, flat index t*F + f -> f'*T + t'.  Parity is against that.
    z = z.reshape(B, n_heads, Fq, T, Cp).permute(0, 1, 4, 2, 3).reshape(B, C, Fq, T)
    if taps is not None:
        taps[p[:-1]] = z
    return z


def film(sd, p, x, video):
    """FusionBlock.forward (DTTNetPlus/blocks/block_fusion.py:16-37).
    x: [B,C,F,T]; video: [B,T_v,D]."""
    T = x.shape[-1]
    v = F.interpolate(video.transpose(-1, -2), size=T, mode="linear").transpose(-1, -2)  # [B,T,D]
    gc = F.linear(v, sd[p + "w_gamma_channels.weight"], sd[p + "w_gamma_channels.bias"])
    bc = F.linear(v, sd[p + "w_beta_channels.weight"], sd[p + "w_beta_channels.bias"])
    gf = F.linear(v, sd[p + "w_gamma_freq.weight"], sd[p + "w_gamma_freq.bias"])
    bf = F.linear(v, sd[p + "w_beta_freq.weight"], sd[p + "w_beta_freq.bias"])
    gamma = gc.permute(0, 2, 1).unsqueeze(2) + gf.permute(0, 2, 1).unsqueeze(1)
    beta = bc.permute(0, 2, 1).unsqueeze(2) + bf.permute(0, 2, 1).unsqueeze(1)
    return (1 + gamma) * x + beta


def latent(sd, p, x, n_heads, video=None, taps=None):
    """LatentModule.forward (modules/latent.py:21-26; FiLM variant
    DTTNetPlus/modules/latent.py:29-35)."""
    x = tfc_tdf(sd, p + "tfc_tdf.", x, taps)
    if video is not None:
        x = film(sd, p + "fusion_block.", x, video)
        if taps is not None:
            taps[p + "fusion_block"] = x
    i = 0
    while f"{p}idp_modules.{i}.tc_rnn.norm.weight" in sd:
        x = idp(sd, f"{p}idp_modules.{i}.", x, n_heads, taps)
        i += 1
    return x


def decoder(sd, p, x, skips, taps=None):
    """Decoder.forward / DecoderBlock (modules/decoder.py:26-52)."""
    for k in range(_n_levels(sd, p, "decoder_layers")):
        q = f"{p}decoder_layers.{k}."
        y = conv_norm_gelu(x, sd[q + "upsampling.upsampling.0.weight"],
                           sd[q + "upsampling.upsampling.0.bias"], transposed=True)
        s = skips[k]
        y = y[:, :, :s.shape[2], :s.shape[3]]                # negative F.pad == crop
        x = tfc_tdf(sd, q + "tfc_tdf.", y, taps) * s
        if taps is not None:
            taps[q[:-1]] = x
    x = F.conv2d(x, sd[p + "out_conv.weight"], sd[p + "out_conv.bias"])
    if taps is not None:
        taps[p + "out_conv"] = x
    return x


def mask_recon(x, spec, phase, n_sources, n_fft, hop, audio_length):
    """postprocess / recon_signal_spectral / recon_audio (model_dttnet.py:47-85)."""
    B, _, Fq, T = x.shape
    m = x.view(B, n_sources, 2, Fq, T)
    preds = []
    for i in range(n_sources):
        ms, mp = m[:, i, 0], m[:, i, 1]
        s = spec * ms - phase * mp
        ph = spec * mp + phase * ms
        preds.append({"audio": istft_rect(s, ph, n_fft, hop, audio_length),
                      "spectrogram": s, "phase": ph})
    return preds


def dttnet_forward(sd, spec, phase, audio_length, n_sources=2, n_heads=2, n_fft=1024,
                   hop=256, taps=None):
    """DTTNetModel.forward (model_dttnet.py:40-45)."""
    x, skips = encoder(sd, "encoder.", spec, phase, taps)
    x = latent(sd, "latent.", x, n_heads, taps=taps)
    x = decoder(sd, "decoder.", x, skips, taps)
    return {"preds": mask_recon(x, spec, phase, n_sources, n_fft, hop, audio_length)}


def dttnet_film_forward(sd, spec, phase, audio_length, video_encoded, n_sources=2, n_heads=2,
                        n_fft=1024, hop=256, taps=None):
    """DTTNetPlusModel.forward (DTTNetPlus/model_dttnet.py:47-77) downstream of
    the (frozen, third-party) lip-reading encoder: ``video_encoded`` is its
    post-processed output [B, T_v, video_embed_dim]."""
    x, skips = encoder(sd, "encoder.", spec, phase, taps)
    x = latent(sd, "latent.", x, n_heads, video=video_encoded, taps=taps)
    x = decoder(sd, "decoder.", x, skips, taps)
    return {"preds": mask_recon(x, spec, phase, n_sources, n_fft, hop, audio_length)}


# --------------------------------------------------------------------------- #
# Audio-visual fusion (cross-attention / temporal alignment) and ResNet-18
# --------------------------------------------------------------------------- #
def sinusoids(length, dim, dtype=torch.float32):
    """_build_sinusoids (VideoAudioFusions/cross_attention_positional.py:6-12)."""
    pos = torch.arange(length, dtype=dtype).unsqueeze(1)
    div = torch.exp(torch.arange(0, dim, 2, dtype=dtype) * (-math.log(10000.0) / dim))
    pe = torch.zeros(length, dim, dtype=dtype)
    pe[:, 0::2] = torch.sin(pos * div)
    pe[:, 1::2] = torch.cos(pos * div)
    return pe


def _layernorm(x, w, b, eps=1e-5):
    mu = x.mean(-1, keepdim=True)
    var = x.var(-1, unbiased=False, keepdim=True)
    return (x - mu) / torch.sqrt(var + eps) * w + b


def _mha(q, k, v, w_in, b_in, w_out, b_out, n_heads):
    """nn.MultiheadAttention(batch_first=True, dropout=0) forward."""
    C = q.shape[-1]
    qp = F.linear(q, w_in[:C], b_in[:C])
    kp = F.linear(k, w_in[C:2 * C], b_in[C:2 * C])
    vp = F.linear(v, w_in[2 * C:], b_in[2 * C:])
    B, Tq, _ = qp.shape
    Tk = kp.shape[1]
    dh = C // n_heads
    qh = qp.view(B, Tq, n_heads, dh).transpose(1, 2)
    kh = kp.view(B, Tk, n_heads, dh).transpose(1, 2)
    vh = vp.view(B, Tk, n_heads, dh).transpose(1, 2)
    att = torch.softmax(qh @ kh.transpose(-1, -2) / math.sqrt(dh), dim=-1)
    o = (att @ vh).transpose(1, 2).reshape(B, Tq, C)
    return F.linear(o, w_out, b_out)


def cross_attention_positional(sd, p, audio, video, n_heads):
    """CrossAttentionPositional.forward (cross_attention_positional.py:51-80).
    audio [B,C,F,T]; video [B,C_v,E_v,T_v] -> [B,2C,F,T]."""
    B, C, Fq, T = audio.shape
    _, Cv, Ev, Tv = video.shape
    a = audio.mean(dim=2).transpose(1, 2)                     # [B,T,C]
    v = video.permute(0, 3, 1, 2).reshape(B, Tv, Cv * Ev)
    v = _gelu(F.linear(v, sd[p + "video_proj.0.weight"], sd[p + "video_proj.0.bias"]))
    a = a + sinusoids(T, C, a.dtype).unsqueeze(0)
    v = v + sinusoids(Tv, C, a.dtype).unsqueeze(0)
    lw, lb = sd[p + "inp_layernorm.weight"], sd[p + "inp_layernorm.bias"]
    q = F.linear(_layernorm(a, lw, lb), sd[p + "q_audio.weight"], sd[p + "q_audio.bias"])
    k = F.linear(_layernorm(v, lw, lb), sd[p + "k_video.weight"], sd[p + "k_video.bias"])
    vv = F.linear(_layernorm(v, lw, lb), sd[p + "v_video.weight"], sd[p + "v_video.bias"])
    att = _mha(q, k, vv, sd[p + "cross_attn.in_proj_weight"], sd[p + "cross_attn.in_proj_bias"],
               sd[p + "cross_attn.out_proj.weight"], sd[p + "cross_attn.out_proj.bias"], n_heads)
    a = _layernorm(a + att, sd[p + "attn_layernorm.weight"], sd[p + "attn_layernorm.bias"])
    h = _gelu(F.linear(a, sd[p + "cross_ffn.0.weight"], sd[p + "cross_ffn.0.bias"]))
    a = a + F.linear(h, sd[p + "cross_ffn.2.weight"], sd[p + "cross_ffn.2.bias"])
    a = a * torch.sigmoid(F.linear(a, sd[p + "audio_gate.0.weight"], sd[p + "audio_gate.0.bias"]))
    enh = a.transpose(1, 2).unsqueeze(2).expand(B, C, Fq, T)
    return torch.cat([audio, enh], dim=1)


def temporal_aligner(sd, p, audio, video):
    """TemporalAligner.forward (VideoAudioFusions/temporal_alignment.py:12-25)."""
    B, C, Fq, T = audio.shape
    _, Cv, Ev, Tv = video.shape
    v = video.reshape(B, Cv * Ev, 1, Tv)
    v = F.conv2d(v, sd[p + "video_projection.weight"], sd[p + "video_projection.bias"])
    v = v.repeat(1, 1, Fq, 1)
    if Tv != T:
        v = F.interpolate(v, size=(Fq, T), mode="bilinear", align_corners=False)
    a = F.conv2d(audio, sd[p + "audio_projection.weight"], sd[p + "audio_projection.bias"])
    return torch.cat([a, v], dim=1)


def _bn(x, sd, p, training, eps=1e-5):
    return F.batch_norm(x, None if training else sd[p + "running_mean"],
                        None if training else sd[p + "running_var"],
                        sd[p + "weight"], sd[p + "bias"], training=training, eps=eps)


def _swish(x):
    return x * torch.sigmoid(x)


def resnet18_video_encoder(sd, p, video, training=True):
    """VideoEncoder.forward (VideoEncoders/resnet_video_encoder.py:62-72) over
    ResNet-18 BasicBlocks with Swish (VideoEncoders/resnet.py:36-137).
    video [B,N,T,H,W] -> [B,N,512,T]."""
    B, N, T, H, W = video.shape
    x = video.reshape(B * N, 1, T, H, W)
    x = F.conv3d(x, sd[p + "frontend3D.0.weight"], None, stride=(1, 2, 2), padding=(2, 3, 3))
    x = F.batch_norm(x, None if training else sd[p + "frontend3D.1.running_mean"],
                     None if training else sd[p + "frontend3D.1.running_var"],
                     sd[p + "frontend3D.1.weight"], sd[p + "frontend3D.1.bias"], training=training)
    x = _swish(x)
    x = F.max_pool3d(x, (1, 3, 3), (1, 2, 2), (0, 1, 1))
    x = x.transpose(1, 2).reshape(B * N * T, x.shape[1], x.shape[3], x.shape[4])
    for li, stride in ((1, 1), (2, 2), (3, 2), (4, 2)):
        for bi in range(2):
            q = f"{p}trunk.layer{li}.{bi}."
            s = stride if bi == 0 else 1
            r = x
            y = F.conv2d(x, sd[q + "conv1.weight"], None, stride=s, padding=1)
            y = _swish(_bn(y, sd, q + "bn1.", training))
            y = F.conv2d(y, sd[q + "conv2.weight"], None, padding=1)
            y = _bn(y, sd, q + "bn2.", training)
            if q + "downsample.0.weight" in sd:
                r = F.conv2d(x, sd[q + "downsample.0.weight"], None, stride=s)
                r = _bn(r, sd, q + "downsample.1.", training)
            x = _swish(y + r)
    x = x.mean(dim=(2, 3))                                    # AdaptiveAvgPool2d(1)
    return x.view(B, N, T, -1).transpose(2, 3)


def lipreading_video_features(sd, p, video, training=True):
    """The frozen lip-reading encoder of the FiLM model as VideoEncoderModule.forward runs it
    (DTTNetPlus/modules/video_encoder.py:41-47): Lipreading.forward with extract_feats=True
    (lipreading/model.py:164-185) = Conv3d stem + BatchNorm3d + ReLU + MaxPool3d (model.py:128-132), then the
    ShuffleNetV2 1.0x units, the 1x1 conv_last and the 3x3 average pool (lipreading/models/shufflenetv2.py:42-104,
    118-150).  ``p`` is the prefix of the Lipreading object ("video_encoder.encoder.").
    video [B,2,T,H,W] -> [B,T,2048] (the input of post_processor)."""
    B, N, T, H, W = video.shape

    def bn(x, q):
        return F.batch_norm(x, None if training else sd[q + "running_mean"], None if training else sd[q + "running_var"],
                            sd[q + "weight"], sd[q + "bias"], training=training, eps=1e-5)

    def pw(x, q, i):                     # 1x1 conv, BN, ReLU at indices i, i+1, i+2 of the Sequential q
        return torch.relu(bn(F.conv2d(x, sd[f"{q}{i}.weight"]), f"{q}{i + 1}."))

    def dw(x, q, i, stride):             # depthwise 3x3, BN (no activation)
        w = sd[f"{q}{i}.weight"]
        return bn(F.conv2d(x, w, None, stride=stride, padding=1, groups=w.shape[0]), f"{q}{i + 1}.")

    x = video.reshape(B * N, 1, T, H, W)
    x = F.conv3d(x, sd[p + "frontend3D.0.weight"], None, stride=(1, 2, 2), padding=(2, 3, 3))
    x = torch.relu(bn(x, p + "frontend3D.1."))
    x = F.max_pool3d(x, (1, 3, 3), (1, 2, 2), (0, 1, 1))
    x = x.transpose(1, 2).reshape(B * N * T, x.shape[1], x.shape[3], x.shape[4])
    u = 0
    while f"{p}trunk.0.{u}.banch2.0.weight" in sd:
        q = f"{p}trunk.0.{u}."
        if q + "banch1.0.weight" in sd:                       # stride-2 unit: both branches see the whole input
            a = pw(dw(x, q + "banch1.", 0, 2), q + "banch1.", 2)
            b = pw(dw(pw(x, q + "banch2.", 0), q + "banch2.", 3, 2), q + "banch2.", 5)
        else:                                                 # stride-1 unit: first half passes through
            half = x.shape[1] // 2
            a = x[:, :half]
            b = pw(dw(pw(x[:, half:], q + "banch2.", 0), q + "banch2.", 3, 1), q + "banch2.", 5)
        out = torch.cat((a, b), 1)
        n, c, h, w = out.shape                                # channel_shuffle(out, 2)
        x = out.view(n, 2, c // 2, h, w).transpose(1, 2).reshape(n, c, h, w)
        u += 1
    x = pw(x, p + "trunk.1.", 0)
    x = F.avg_pool2d(x, 3).reshape(B, N, T, -1)
    return x.transpose(1, 2).flatten(-2, -1)


def dttnet_vision_forward(sd, spec, phase, audio_length, video_features, fusion_method,
                          n_sources=2, n_heads=2, n_fft=1024, hop=256, taps=None):
    """DTTNetVisionModel.forward (model_dttnet_vision.py:79-108) downstream of
    the video encoder: ``video_features`` = [B, 2, 512, T_v]."""
    x, skips = encoder(sd, "encoder.", spec, phase, taps)
    if fusion_method == "cross_attention_positional":
        x = cross_attention_positional(sd, "fusion.", x, video_features, n_heads)
    elif fusion_method == "temporal_alignment":
        x = temporal_aligner(sd, "fusion.", x, video_features)
    else:
        raise ValueError(f"{fusion_method} audio-video fusion method is not supported yet")
    if taps is not None:
        taps["fusion"] = x
    x = latent(sd, "latent.", x, n_heads, taps=taps)
    x = F.conv2d(x, sd["channel_compression.0.weight"], sd["channel_compression.0.bias"])
    x = _gelu(_inorm(x))
    if taps is not None:
        taps["channel_compression"] = x
    x = decoder(sd, "decoder.", x, skips, taps)
    return {"preds": mask_recon(x, spec, phase, n_sources, n_fft, hop, audio_length)}


# --------------------------------------------------------------------------- #
# Loss and metrics
# --------------------------------------------------------------------------- #
def dttnet_l1(pred, src):
    """DTTNetLoss.forward (src/loss/dttnet_loss.py:28-34) -> [B]."""
    spec = (src["spectrogram"] - pred["spectrogram"]).abs() + (src["phase"] - pred["phase"]).abs()
    wave = (src["audio"] - pred["audio"]).abs()
    return spec.mean(dim=(1, 2)) + wave.mean(dim=-1)


def pit_loss(audio, sources, preds):
    """PITLossWrapper.forward (src/loss/pit_loss_wrapper.py:41-71):
    perm[pred_idx] = source_idx, itertools order, first minimum wins."""
    n = len(sources)
    perms = list(permutations(range(n), n))
    rows = []
    for perm in perms:
        acc = torch.zeros(audio.shape[0], dtype=audio.dtype)
        for i, pred in enumerate(preds):
            acc = acc + dttnet_l1(pred, sources[perm[i]])
        rows.append(acc)
    vals, idx = torch.min(torch.stack(rows), dim=0)
    bp = torch.tensor([perms[int(i)] for i in idx], dtype=torch.long)
    return {"loss": vals.mean(), "batch_permuts": bp}


def fixed_order_loss(audio, sources, preds):
    """DTTNetLossWrapper.forward (src/loss/dttnet_loss_wrapper.py:41-62)."""
    B, n = audio.shape[0], len(sources)
    acc = torch.zeros(B, dtype=audio.dtype)
    for i, pred in enumerate(preds):
        acc = acc + dttnet_l1(pred, sources[i])
    return {"loss": acc.mean(), "batch_permuts": torch.arange(n).unsqueeze(0).repeat(B, 1)}


def si_snr(pred, true, mix=None, eps=1e-5):
    """SiSnr.__call__ (src/metrics/si_snr.py:26-34): no mean removal."""
    scale = (pred * true).sum(1, keepdim=True) / (true.pow(2).sum(1, keepdim=True) + eps)
    tgt = scale * true
    noise = pred - tgt
    return 10 * torch.log10(tgt.pow(2).sum(1) / (noise.pow(2).sum(1) + eps))


def si_snr_i(pred, true, mix):
    """SiSnrI.__call__ (src/metrics/si_snr_i.py:29-31)."""
    return si_snr(pred, true) - si_snr(mix, true)


def pit_metric_training(metric_fn, audio, preds, sources, batch_permuts):
    """PIT_wrapper_training (src/metrics/base_metric.py:28-42)."""
    B, n = audio.shape[0], len(sources)
    stacked = torch.stack([s["audio"] for s in sources])
    acc = torch.zeros(B, dtype=audio.dtype)
    for i, pred in enumerate(preds):
        acc = acc + metric_fn(pred["audio"], stacked[batch_permuts[:, i], torch.arange(B)], audio)
    return (acc / n).mean()


def pit_metric_inference(metric_fn, audio, preds, sources):
    """PIT_wrapper_inference (src/metrics/base_metric.py:45-67):
    perm[src_idx] = pred_idx, argmax (first maximum wins)."""
    B, n = audio.shape[0], len(sources)
    perms = list(permutations(range(n), n))
    rows = []
    for perm in perms:
        acc = torch.zeros(B, dtype=audio.dtype)
        for s_i, p_i in enumerate(perm):
            acc = acc + metric_fn(preds[p_i]["audio"], sources[s_i]["audio"], audio)
        rows.append(acc)
    scores = torch.stack(rows)
    best = scores.argmax(dim=0)
    return (scores[best, torch.arange(B)] / n).mean()


# --------------------------------------------------------------------------- #
# Data-loader boundary (the step before the path)
# --------------------------------------------------------------------------- #
def collate(dataset_items):
    """collate_fn (src/datasets/collate.py:4-83) on items shaped as BaseDataset.__getitem__ makes them
    (base_dataset.py:72-138): audio [1,L], spectrogram / phase [1,F,T]; zero padding to the longest item."""
    from torch.nn.utils.rnn import pad_sequence

    def pad_spec(key, src=None):
        xs = [(it if src is None else it["sources"][src])[key].squeeze(0).transpose(0, 1) for it in dataset_items]
        return pad_sequence(xs, batch_first=True, padding_value=0.0).transpose(-2, -1)

    def pad_audio(src=None):
        xs = [(it if src is None else it["sources"][src])["audio"].squeeze(0) for it in dataset_items]
        return pad_sequence(xs, batch_first=True, padding_value=0.0)

    audio = pad_audio()
    out = {
        "spectrogram": pad_spec("spectrogram"), "phase": pad_spec("phase"),
        "spectrogram_length": torch.tensor([it["spectrogram"].shape[-1] for it in dataset_items]),
        "audio": audio,
        "audio_lengths": [it["audio"].shape[-1] for it in dataset_items],
        "sources": [{"audio": pad_audio(s), "spectrogram": pad_spec("spectrogram", s), "phase": pad_spec("phase", s)}
                    for s in range(len(dataset_items[0]["sources"]))],
        "audio_length": audio.shape[-1],
    }
    if dataset_items[0].get("mouth1") is not None or dataset_items[0].get("mouth2") is not None:
        vids = []
        for it in dataset_items:
            m1, m2 = it.get("mouth1"), it.get("mouth2")
            m1 = torch.zeros_like(torch.as_tensor(m2)) if m1 is None else torch.as_tensor(m1)
            m2 = torch.zeros_like(m1) if m2 is None else torch.as_tensor(m2)
            vids.append(torch.stack([m1, m2]))
        out["video"] = torch.stack(vids)
    return out
