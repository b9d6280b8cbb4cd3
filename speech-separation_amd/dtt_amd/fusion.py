"""Audio-visual fusion blocks on channels-last activations.

Big-tensor work (frequency-mean tokens, FiLM apply, broadcast + concat, 1x1
projection) runs in the HIP kernels; the token-level network (<= 50 tokens per
item, 0.08-0.26 GFLOP/sample) runs its Linear layers through
dtt_gemm (ops.Linear, exact-fp32 MFMA), its attention core on attn.hip and the
resampling of the video embedding on dtt_lerp_time; LayerNorm, GELU, concatenation
and residual adds are elementwise torch ops on [B, <= 150, C] tensors.

Reference: src/model/VideoAudioFusions/{cross_attention_positional,
temporal_alignment}.py, src/model/DTTNetPlus/blocks/block_fusion.py.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F
from torch import nn

from . import ops


def _build_sinusoids(length, dim, device):
    """cross_attention_positional.py:6-12."""
    pos = torch.arange(length, device=device, dtype=torch.float32).unsqueeze(1)
    div = torch.exp(torch.arange(0, dim, 2, device=device, dtype=torch.float32) * (-math.log(10000.0) / dim))
    pe = torch.zeros(length, dim, device=device)
    pe[:, 0::2] = torch.sin(pos * div)
    pe[:, 1::2] = torch.cos(pos * div)
    return pe


class CrossAttentionPositional(nn.Module):
    """cross_attention_positional.py:15-80; input/outputs channels-last:
    audio [B,F,T,C], video [B,C_v,E_v,T_v] (as the video encoder emits it) -> [B,F,T,2C]."""

    def __init__(self, audio_channels, video_channels, n_heads=2):
        super().__init__()
        self.audio_channels, self.video_channels, self.n_heads = audio_channels, video_channels, n_heads
        self.video_proj = nn.Sequential(nn.Linear(video_channels, audio_channels), nn.GELU())
        self.q_audio = nn.Linear(audio_channels, audio_channels)
        self.k_video = nn.Linear(audio_channels, audio_channels)
        self.v_video = nn.Linear(audio_channels, audio_channels)
        self.inp_layernorm = nn.LayerNorm(audio_channels)
        self.attn_layernorm = nn.LayerNorm(audio_channels)
        self.cross_attn = nn.MultiheadAttention(audio_channels, n_heads, dropout=0.0, batch_first=True)
        self.cross_ffn = nn.Sequential(nn.Linear(audio_channels, 4 * audio_channels), nn.GELU(),
                                       nn.Linear(4 * audio_channels, audio_channels))
        self.audio_gate = nn.Sequential(nn.Linear(audio_channels, audio_channels), nn.Sigmoid())

    def _tokens(self, a, video):
        """Token-level network in fp32 (outside autocast: 16 + 50 tokens per item).  Every Linear runs through
        dtt_gemm (ops.Linear: exact-fp32 MFMA, fused bias / sigmoid), the attention core on the MFMA tile kernel
        (ops.CrossAttnCore); LayerNorm, GELU and the residual adds are elementwise torch ops on [B, <= 150, C]."""
        lin = ops.Linear.apply
        B, T, Cc = a.shape
        _, Cv, Ev, Tv = video.shape
        v = video.float().permute(0, 3, 1, 2).reshape(B, Tv, Cv * Ev)
        v = F.gelu(lin(v, self.video_proj[0].weight, self.video_proj[0].bias, False))
        a = a + _build_sinusoids(T, Cc, a.device).unsqueeze(0)
        v = v + _build_sinusoids(Tv, Cc, a.device).unsqueeze(0)
        ln = self.inp_layernorm
        q = lin(F.layer_norm(a, (Cc,), ln.weight, ln.bias, ln.eps), self.q_audio.weight, self.q_audio.bias, False)
        vn = F.layer_norm(v, (Cc,), ln.weight, ln.bias, ln.eps)
        k = lin(vn, self.k_video.weight, self.k_video.bias, False)
        vv = lin(vn, self.v_video.weight, self.v_video.bias, False)
        ca = self.cross_attn
        qp = lin(q, ca.in_proj_weight[:Cc], ca.in_proj_bias[:Cc], False)
        kp = lin(k, ca.in_proj_weight[Cc:2 * Cc], ca.in_proj_bias[Cc:2 * Cc], False)
        vp = lin(vv, ca.in_proj_weight[2 * Cc:], ca.in_proj_bias[2 * Cc:], False)
        dh = Cc // self.n_heads
        if dh % 32 == 0 and T <= 32 and Tv <= 160:
            o = ops.CrossAttnCore.apply(qp, kp, vp, self.n_heads)
        else:                                   # head widths / token counts the tile kernel does not cover
            H = self.n_heads
            att = torch.softmax(qp.view(B, T, H, dh).transpose(1, 2) @ kp.view(B, Tv, H, dh).permute(0, 2, 3, 1)
                                / math.sqrt(dh), dim=-1)
            o = (att @ vp.view(B, Tv, H, dh).transpose(1, 2)).transpose(1, 2).reshape(B, T, Cc)
        o = lin(o, ca.out_proj.weight, ca.out_proj.bias, False)
        al = self.attn_layernorm
        a = F.layer_norm(a + o, (Cc,), al.weight, al.bias, al.eps)
        f = lin(F.gelu(lin(a, self.cross_ffn[0].weight, self.cross_ffn[0].bias, False)), self.cross_ffn[2].weight,
                self.cross_ffn[2].bias, False)
        a = a + f
        return a * lin(a, self.audio_gate[0].weight, self.audio_gate[0].bias, True)

    def forward(self, audio_features, video_features):
        # audio_features feeds the token mean and the concat: one pass forms the sum of their gradients (ops.GradSlot)
        slot = ops.new_slot(audio_features)
        tok = ops.FreqMeanTokens.apply(audio_features, slot)           # [B,T,C] fp32
        with torch.autocast("cuda", enabled=False):
            enh = self._tokens(tok, video_features)
        return ops.BroadcastConcat.apply(audio_features, enh, slot)


class TemporalAligner(nn.Module):
    """temporal_alignment.py:6-25.  The video projection is constant over F, so the
    bilinear resize (align_corners=False) reduces to linear interpolation over time."""

    def __init__(self, audio_channels, video_channels):
        super().__init__()
        self.audio_projection = nn.Conv2d(audio_channels, audio_channels, 1)
        self.video_projection = nn.Conv2d(video_channels, audio_channels, 1)

    def forward(self, audio_features, video_features):
        B, Fq, T, Cc = audio_features.shape
        _, Cv, Ev, Tv = video_features.shape
        with torch.autocast("cuda", enabled=False):
            v = video_features.float().reshape(B, Cv * Ev, Tv).transpose(1, 2)
            w = self.video_projection.weight.reshape(Cc, Cv * Ev)
            vp = ops.Linear.apply(v, w, self.video_projection.bias, False)          # [B,Tv,C]
            if Tv != T:
                vp = ops.LerpTime.apply(vp, T)                                      # [B,T,C]
        a = ops.Conv1x1.apply(audio_features, self.audio_projection.weight, self.audio_projection.bias)
        return ops.BroadcastConcat.apply(a, vp)


def film_apply(block, x, video):
    """FusionBlock.forward (block_fusion.py:16-37): x [B,F,T,C] channels-last, video [B,T_v,D]."""
    T = x.shape[2]
    with torch.autocast("cuda", enabled=False):
        v = video.float()
        v = ops.LerpTime.apply(v, T) if v.shape[1] != T else v                                      # [B,T,D]
        lin = ops.Linear.apply
        gc, bc, gf, bf = (lin(v, m.weight, m.bias, False) for m in
                          (block.w_gamma_channels, block.w_beta_channels, block.w_gamma_freq, block.w_beta_freq))
        gbc = torch.cat([gc, bc], dim=-1)                                                           # [B,T,2C]
        gbf = torch.cat([gf, bf], dim=-1)                                                           # [B,T,2F]
    return ops.Film.apply(x, gbc, gbf)
