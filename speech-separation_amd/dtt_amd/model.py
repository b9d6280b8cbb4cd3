"""MI355X-native DTTNet models with the reference's constructor / forward /
state_dict contract.

Module tree and parameter names are the reference's:
stock ``nn.Conv2d / nn.Linear / nn.LSTM / nn.GroupNorm`` objects are used as
PARAMETER CONTAINERS only (same names, shapes, default initialisation); the
arithmetic runs in the HIP kernels through ``dtt_amd.ops``.

Reference: src/model/DTTNet/{model_dttnet,model_dttnet_vision}.py,
src/model/DTTNet/modules/{encoder,decoder,latent}.py,
src/model/DTTNet/blocks/{block_tfc_tdf,block_sampling}.py,
src/model/DTTNetPlus/*, src/model/VideoAudioFusions/*.
"""
from __future__ import annotations

import torch
from torch import nn

from . import ops
from ._lib import DttError


def _nchw(x):
    return x.detach().permute(0, 3, 1, 2).float().contiguous()


class PredList(list):
    """The reference's ``preds`` list (model_dttnet.py:57-68) that also remembers the stacked tensors its
    entries are views of, so that the fused loss does not have to re-stack them.  It IS a plain list of
    dicts for every caller; a copy of it (or of its entries) simply loses the shortcut."""
    stacked = None


class _Tap:
    """Optional per-module output recorder (NCHW fp32 copies) for the parity tests."""
    store = None

    @classmethod
    def put(cls, name, x):
        if cls.store is not None:
            cls.store[name] = _nchw(x)


def _conv_block(channels):
    return nn.Sequential(nn.Conv2d(channels, channels, 3, padding="same"), nn.InstanceNorm2d(channels), nn.GELU())


class TFC_TDF_Block(nn.Module):
    """block_tfc_tdf.py:7-47."""

    def __init__(self, channels, fc_dim, bf=2, use_checkpoints=False):
        super().__init__()
        self.use_checkpoints = use_checkpoints      # accepted for API parity; 288 GB HBM: no recompute
        self.conv1 = nn.Sequential(*[_conv_block(channels) for _ in range(3)])
        self.fc = nn.Sequential(nn.Linear(fc_dim, fc_dim // bf), nn.Linear(fc_dim // bf, fc_dim))
        self.conv2 = nn.Sequential(*[_conv_block(channels) for _ in range(3)])
        self.skip_conv = nn.Conv2d(channels, channels, 3, padding="same")

    def forward(self, x, name=""):
        h = x
        # x feeds conv1[0] and skip_conv: their two partial gradients meet inside conv1[0]'s data-gradient kernel
        # (ops.GradSlot) instead of an autograd add over the whole tensor
        slot = ops.new_slot(x)
        for i in range(3):
            c = self.conv1[i][0]
            h = ops.ConvNormGelu.apply(h, c.weight, c.bias, "s1", None, slot if i == 0 else None)
            _Tap.put(f"{name}.conv1.{i}", h)
        h = ops.TDF.apply(h, self.fc[0].weight, self.fc[0].bias, self.fc[1].weight, self.fc[1].bias)
        for i in range(3):
            c = self.conv2[i][0]
            h = ops.ConvNormGelu.apply(h, c.weight, c.bias, "s1", None)
        out = ops.ConvAdd.apply(x, self.skip_conv.weight, self.skip_conv.bias, h, slot)
        _Tap.put(name, out)
        return out


class DownSamplingBlock(nn.Module):
    """block_sampling.py:4-16."""

    def __init__(self, in_channels):
        super().__init__()
        self.downsampling = nn.Sequential(nn.Conv2d(in_channels, 2 * in_channels, 3, stride=2, padding=1),
                                          nn.InstanceNorm2d(2 * in_channels), nn.GELU())

    def forward(self, x, slot=None):
        c = self.downsampling[0]
        return ops.ConvNormGelu.apply(x, c.weight, c.bias, "down", None, slot)


class UpSamplingBlock(nn.Module):
    """block_sampling.py:19-32 (+ the crop of decoder.py:46-52 folded in)."""

    def __init__(self, out_chanels):
        super().__init__()
        self.upsampling = nn.Sequential(
            nn.ConvTranspose2d(out_chanels * 2, out_chanels, 3, stride=2, padding=1, output_padding=1),
            nn.InstanceNorm2d(out_chanels), nn.GELU())

    def forward(self, x, crop):
        c = self.upsampling[0]
        return ops.ConvNormGelu.apply(x, c.weight, c.bias, "up", crop)


class EncoderBlock(nn.Module):
    def __init__(self, fc_dim, channels=32, use_checkpoints=False):
        super().__init__()
        self.tfc_tdf = TFC_TDF_Block(channels, fc_dim, bf=2, use_checkpoints=use_checkpoints)
        self.downsampling = DownSamplingBlock(channels)

    def forward(self, x, name=""):
        skip = self.tfc_tdf(x, name + ".tfc_tdf")
        # the skip feeds this down-sampling conv and, much later, the decoder's multiply (decoder.py:43): the slot
        # rides on the tensor object so that DecoderBlock finds it (ops.GradSlot)
        slot = ops.new_slot(skip)
        skip._dtt_grad_slot = slot
        out = self.downsampling(skip, slot)
        _Tap.put(name, out)
        return out, skip


class Encoder(nn.Module):
    """modules/encoder.py:8-46."""

    def __init__(self, fc_dim, in_channels=2, out_channels=32, n_layers=3, use_checkpoints=False):
        super().__init__()
        self.init_conv = nn.Conv2d(in_channels, out_channels, 1)
        self.encoder_layers = nn.ModuleList()
        for l in range(n_layers):
            enc_dim = (fc_dim + 2 ** l) if l > 0 else fc_dim
            self.encoder_layers.append(EncoderBlock(enc_dim // 2 ** l, out_channels * 2 ** l, use_checkpoints))

    def forward(self, spectrogram, phase, dt):
        x = ops.InitConv.apply(spectrogram, phase, self.init_conv.weight, self.init_conv.bias, dt)
        _Tap.put("encoder.init_conv", x)
        skips = []
        for l, layer in enumerate(self.encoder_layers):
            x, s = layer(x, f"encoder.encoder_layers.{l}")
            skips.insert(0, s)
        return x, skips


class TC_FC_RNN(nn.Module):
    """modules/latent.py:77-104 (parameter container; see ops.DualPathRNN)."""

    def __init__(self, n_heads, in_chanels):
        super().__init__()
        self.norm = nn.GroupNorm(max(1, in_chanels // 16), in_chanels)
        self.blstm = nn.LSTM(in_chanels, 2 * in_chanels, batch_first=True, bidirectional=True)
        self.fc = nn.Linear(in_chanels * 4, in_chanels)

    def run(self, x, heads, along_f):
        l = self.blstm
        return ops.DualPathRNN.apply(
            x, self.norm.weight, self.norm.bias, l.weight_ih_l0, l.weight_hh_l0, l.bias_ih_l0, l.bias_hh_l0,
            l.weight_ih_l0_reverse, l.weight_hh_l0_reverse, l.bias_ih_l0_reverse, l.bias_hh_l0_reverse,
            self.fc.weight, self.fc.bias, heads, along_f)


class IDPModule(nn.Module):
    """modules/latent.py:29-74.  Head split / merge are pure addressing here."""

    def __init__(self, n_heads, in_channels):
        super().__init__()
        self.n_heads = n_heads
        ch = int(in_channels // n_heads)
        if ch % 16 != 0 and ch > 16:
            raise DttError("channels per head must be a multiple of 16 (GroupNorm groups)")
        H = 2 * ch                                          # hidden width of the BiLSTMs (latent.py:87-92)
        if not (H % 32 == 0 and (32 <= H <= 256 or H in (512, 1024))):
            raise DttError(f"IDPModule: in_channels / n_heads = {ch} gives LSTM hidden width {H}; the recurrent kernels "
                           "cover H = 32, 64, ..., 256, 512 and 1024 (channels per head 16 ... 128 in steps of 16, "
                           "256 or 512)")
        self.tc_rnn = TC_FC_RNN(n_heads, ch)
        self.fc_rnn = TC_FC_RNN(n_heads, ch)

    def forward(self, x, name=""):
        x = self.tc_rnn.run(x, self.n_heads, False)
        _Tap.put(name + ".tc_rnn", x)
        x = self.fc_rnn.run(x, self.n_heads, True)
        _Tap.put(name, x)
        return x


class FusionBlock(nn.Module):
    """FiLM: DTTNetPlus/blocks/block_fusion.py:6-37."""

    def __init__(self, audio_channels, audio_freq, video_channels=256):
        super().__init__()
        self.w_gamma_channels = nn.Linear(video_channels, audio_channels)
        self.w_beta_channels = nn.Linear(video_channels, audio_channels)
        self.w_gamma_freq = nn.Linear(video_channels, audio_freq)
        self.w_beta_freq = nn.Linear(video_channels, audio_freq)

    def forward(self, x, video):
        from .fusion import film_apply
        return film_apply(self, x, video)


class LatentModule(nn.Module):
    """modules/latent.py:7-26; with ``video_embed_dim`` the FiLM variant of
    DTTNetPlus/modules/latent.py:9-35."""

    def __init__(self, fc_dim, n_heads=1, n_channels=256, n_layers=3, video_embed_dim=None):
        super().__init__()
        assert n_channels % n_heads == 0, "n_channels must be divisible by n_heads"
        self.tfc_tdf = TFC_TDF_Block(channels=n_channels, fc_dim=fc_dim)
        if video_embed_dim is not None:
            self.fusion_block = FusionBlock(n_channels, fc_dim, video_embed_dim)
        self.idp_modules = nn.ModuleList([IDPModule(n_heads, n_channels) for _ in range(n_layers)])

    def forward(self, x, video_encoded=None):
        x = self.tfc_tdf(x, "latent.tfc_tdf")
        if video_encoded is not None:
            x = self.fusion_block(x, video_encoded)
            _Tap.put("latent.fusion_block", x)
        for i, m in enumerate(self.idp_modules):
            x = m(x, f"latent.idp_modules.{i}")
        _Tap.put("latent", x)
        return x


class DecoderBlock(nn.Module):
    def __init__(self, fc_dim, channels=32, use_checkpoints=False):
        super().__init__()
        self.tfc_tdf = TFC_TDF_Block(channels, fc_dim, bf=2, use_checkpoints=use_checkpoints)
        self.upsampling = UpSamplingBlock(channels)

    def forward(self, x, skip, name=""):
        out = self.upsampling(x, (skip.shape[1], skip.shape[2]))
        _Tap.put(name + ".upsampling", out)
        out = self.tfc_tdf(out, name + ".tfc_tdf")
        out = ops.MulSkip.apply(out, skip, getattr(skip, "_dtt_grad_slot", None))
        _Tap.put(name, out)
        return out


class Decoder(nn.Module):
    """modules/decoder.py:9-52; out_conv is applied by ops.MaskISTFT (fused with the mask)."""

    def __init__(self, fc_dim, out_channels=32, n_sources=2, n_layers=3, use_checkpoints=False):
        super().__init__()
        self.out_conv = nn.Conv2d(out_channels, n_sources * 2, 1)
        self.decoder_layers = nn.ModuleList()
        for l in range(n_layers):
            dec_dim = (fc_dim + 2 ** l) if l > 0 else fc_dim
            self.decoder_layers.insert(0, DecoderBlock(dec_dim // 2 ** l, out_channels * 2 ** l, use_checkpoints))

    def forward(self, x, skips):
        for k, layer in enumerate(self.decoder_layers):
            x = layer(x, skips[k], f"decoder.decoder_layers.{k}")
        return x


class _SeparatorBase(nn.Module):
    """Shared tail: fused out_conv + mask + polar + iSTFT (model_dttnet.py:47-85)."""

    precision = "auto"

    def compute_dtype(self):
        p = self.precision
        if p == "auto":
            p = "bf16" if torch.is_autocast_enabled() else "fp32"
        if p == "bf16":
            return torch.bfloat16
        if p == "fp32":
            return torch.float32
        raise ValueError(f"precision must be 'auto', 'bf16' or 'fp32', got {p!r}")

    def _check(self, spectrogram):
        if not spectrogram.is_cuda:
            raise DttError("dtt_amd models run on an MI355X only (no CPU path); move the batch to cuda")

    def postprocess(self, x, spectrogram, phase, audio_length):
        oc = self.decoder.out_conv
        if _Tap.store is not None:
            w = oc.weight.reshape(oc.weight.shape[0], -1)
            _Tap.store["decoder.out_conv"] = (torch.einsum("bftc,kc->bkft", x.float(), w) +
                                              oc.bias[None, :, None, None]).detach()
        pspec, pphase, audio = ops.MaskISTFT.apply(x, oc.weight, oc.bias, spectrogram, phase, self.n_sources,
                                                   self.n_fft, self.hop_length, int(audio_length))
        preds = PredList({"audio": audio[i], "spectrogram": pspec[i], "phase": pphase[i]}
                         for i in range(self.n_sources))
        preds.stacked = (pspec, pphase, audio)
        return {"preds": preds}


class DTTNetModel(_SeparatorBase):
    """src/model/DTTNet/model_dttnet.py:9-85."""

    def __init__(self, g=32, n_sources=2, n_layers=3, n_idp_layers=3, n_fft=512, hop_length=128, n_heads=2,
                 use_checkpoints=False, precision="auto"):
        super().__init__()
        self.n_sources, self.n_fft, self.hop_length, self.precision = n_sources, n_fft, hop_length, precision
        fc_dim = n_fft // 2 + 1
        self.encoder = Encoder(fc_dim, 2, g, n_layers, use_checkpoints)
        self.decoder = Decoder(fc_dim, g, n_sources, n_layers, use_checkpoints)
        self.latent = LatentModule((fc_dim + 2 ** n_layers) // 2 ** n_layers, n_heads, g * 2 ** n_layers,
                                   n_idp_layers)

    def forward(self, spectrogram, phase, audio_length, **batch):
        self._check(spectrogram)
        ops.prepare_packs()
        x, skips = self.encoder(spectrogram, phase, self.compute_dtype())
        x = self.latent(x)
        x = self.decoder(x, skips)
        return self.postprocess(x, spectrogram, phase, audio_length)


class DTTNetVisionModel(_SeparatorBase):
    """src/model/DTTNet/model_dttnet_vision.py:12-123 (audio-visual, fusion ahead of a
    2x-wide latent, 1x1 channel compression after it)."""

    def __init__(self, g=32, n_sources=2, n_layers=3, n_idp_layers=3, n_fft=512, hop_length=128, n_heads=2,
                 video_channels=512, fusion_method="temporal_alignment", use_checkpoints=False, precision="auto",
                 video_checkpoint_path=None):
        super().__init__()
        from .fusion import CrossAttentionPositional, TemporalAligner
        from .video import VideoEncoder
        self.n_sources, self.n_fft, self.hop_length, self.precision = n_sources, n_fft, hop_length, precision
        fc_dim = n_fft // 2 + 1
        self.audio_dim = g * 2 ** n_layers
        self.fusion_method = fusion_method
        self.video_encoder = VideoEncoder(video_checkpoint_path)
        if fusion_method == "temporal_alignment":
            self.fusion = TemporalAligner(self.audio_dim, 2 * video_channels)
        elif fusion_method == "cross_attention_positional":
            self.fusion = CrossAttentionPositional(self.audio_dim, 2 * video_channels, n_heads)
        else:
            raise ValueError(f"{self.fusion_method} audio-video fusion method is not supported yet")
        self.encoder = Encoder(fc_dim, 2, g, n_layers, use_checkpoints)
        self.decoder = Decoder(fc_dim, g, n_sources, n_layers, use_checkpoints)
        self.latent = LatentModule((fc_dim + 2 ** n_layers) // 2 ** n_layers, n_heads, g * 2 ** n_layers * 2,
                                   n_idp_layers)
        self.channel_compression = nn.Sequential(nn.Conv2d(self.audio_dim * 2, self.audio_dim, 1),
                                                 nn.InstanceNorm2d(self.audio_dim), nn.GELU())

    def encode_video(self, video):
        if self.compute_dtype() == torch.bfloat16:
            with torch.autocast("cuda", dtype=torch.bfloat16):
                return self.video_encoder(video)
        return self.video_encoder(video)

    def forward(self, spectrogram, video, phase, audio_length, video_features=None, **batch):
        self._check(spectrogram)
        ops.prepare_packs()
        if video_features is None:                      # (tests may feed encoder outputs directly)
            # the lip encoder on a stream of its own, beside the audio encoder (ops.video_lane)
            with ops.video_stream(spectrogram.device) as vs:
                video_features = self.encode_video(video)
            x, skips = self.encoder(spectrogram, phase, self.compute_dtype())
            video_features = vs.join(video_features)
            if _Tap.store is not None:
                _Tap.store["video_encoder"] = video_features.detach().float()
        else:
            x, skips = self.encoder(spectrogram, phase, self.compute_dtype())
        x = self.fusion(x, video_features)
        _Tap.put("fusion", x)
        x = self.latent(x)
        cc = self.channel_compression[0]
        x = ops.Conv1x1NormGelu.apply(x, cc.weight, cc.bias)
        _Tap.put("channel_compression", x)
        x = self.decoder(x, skips)
        return self.postprocess(x, spectrogram, phase, audio_length)


class DTTNetPlusModel(_SeparatorBase):
    """src/model/DTTNetPlus/model_dttnet.py:10-77: FiLM fusion between the latent TFC-TDF and the IDPs."""

    def __init__(self, g=32, n_sources=2, n_layers=3, n_idp_layers=3, n_fft=1024, hop_length=128, n_heads=2,
                 video_embed_dim=256, use_checkpoints=False, video_enc_weights_path=None, precision="auto",
                 *args, **kwargs):
        super().__init__()
        self.n_sources, self.n_fft, self.hop_length, self.precision = n_sources, n_fft, hop_length, precision
        fc_dim = n_fft // 2 + 1
        from .video import VideoEncoderModule
        self.encoder = Encoder(fc_dim, 2, g, n_layers, use_checkpoints)
        self.video_encoder = VideoEncoderModule(video_enc_weights_path, video_embed_dim)
        self.decoder = Decoder(fc_dim, g, n_sources, n_layers, use_checkpoints)
        self.latent = LatentModule((fc_dim + 2 ** n_layers) // 2 ** n_layers, n_heads, g * 2 ** n_layers,
                                   n_idp_layers, video_embed_dim=video_embed_dim)

    def forward(self, spectrogram, phase, audio_length, video=None, video_encoded=None, video_features=None,
                **batch):
        """``video`` [B,2,T_v,88,88] as in the reference (DTTNetPlus/model_dttnet.py:47-52).  Fast paths for
        callers that cache the frozen encoder's output: ``video_features`` [B,T_v,2048] (ahead of the trainable
        post_processor) or ``video_encoded`` [B,T_v,D] (behind it)."""
        self._check(spectrogram)
        ops.prepare_packs()
        if video_encoded is None and video_features is None:
            # the frozen lip front end on a stream of its own, beside the audio encoder (ops.video_lane)
            with ops.video_stream(spectrogram.device) as vs:
                video_features = self.video_encoder.features(video, self.compute_dtype() == torch.bfloat16)
            x, skips = self.encoder(spectrogram, phase, self.compute_dtype())
            video_features = vs.join(video_features)
        else:
            x, skips = self.encoder(spectrogram, phase, self.compute_dtype())
        if video_encoded is None:
            if video_features.dim() != 3 or video_features.shape[-1] != 2048:
                raise DttError("video_features must be [B, T_v, 2048]")
            with torch.autocast("cuda", enabled=False):
                pp = self.video_encoder.post_processor          # trainable Linear(2048 -> D), fp32, through dtt_gemm
                video_encoded = ops.Linear.apply(video_features.float(), pp.weight, pp.bias, False)
            if _Tap.store is not None:
                _Tap.store["video_encoder"] = video_encoded.detach().float()
        x = self.latent(x, video_encoded)
        x = self.decoder(x, skips)
        return self.postprocess(x, spectrogram, phase, audio_length)
