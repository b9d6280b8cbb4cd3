"""Where a step's time goes, by C-ABI entry point and shape, measured live.

``with CallTimer() as ct: step()`` brackets every ``libdtt_hip`` call of an EAGERLY issued step with two HIP events
on the launch stream (the stream the C ABI is handed, ``_lib.stream_ptr()``) and groups the calls by (entry point,
shape): launches per step, average duration, the ALGORITHMIC work of the shape (flops of the contraction or bytes that
must cross HBM once -- SURVEY.md section 8(d): reference op group -> work) and the fraction of the roofline that bounds
it (dense bf16 MFMA 2.5 PFLOP/s, HBM 8 TB/s: MI355X_MICROARCH.md).  An entry whose arithmetic intensity is below the
ridge (312 flop/byte) is priced against HBM, the rest against MFMA.  bench.py prints the top of this table, so the
reported roofline follows the time instead of a kernel chosen beforehand.

A bracket holds every kernel the entry point launches (a conv and its statistics merge, an LSTM launch and its
counter reset): it is the cost of the ENTRY, which is what a caller of the C ABI pays.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib

PEAK_BF16_TFLOPS = 2500.0
PEAK_FP32_MFMA_TFLOPS = 157.3      # dense fp32 MFMA (the exact-fp32 DFT / token GEMMs)
PEAK_HBM_GBS = 8000.0
RIDGE = PEAK_BF16_TFLOPS * 1e12 / (PEAK_HBM_GBS * 1e9)      # flop per byte


def _v(a):
    if isinstance(a, int):
        return a
    if a is None:
        return 0
    return getattr(a, "value", None)


def _es(d):
    return 2 if d == 1 else 4


def _gemm(args):
    g = args[0]._obj
    es = _es(g.dtype)
    batch = max(1, g.batch)
    flops = 2.0 * g.M * g.N * g.K * batch
    if g.a_mode == 1:                                     # implicit-GEMM conv rows: x is read once, not once per tap
        a_bytes = g.cB * g.cFi * g.cTi * g.cCi * es
        kind = "conv-rows" + (f" s{g.cstride}" if g.cstride != 1 else "") + (" T" if g.ctrans else "") + \
               (f" ph{g.cphase}" if g.cphase else "")
    else:
        a_bytes = g.M * g.K * es * (batch if g.a_bs else 1)
        kind = "A^T" if g.a_ms == 1 and g.a_ks != 1 else "A"
    if g.b_mode == 1:                                     # conv weight gradient: the gathered side is read once
        b_bytes = g.cB * g.cFi * g.cTi * g.cCi * es
        kind += " x conv-gather"
    else:
        b_bytes = g.N * g.K * es * (batch if g.b_bs else 1)
        kind += " x " + ("B^T" if g.b_ks == 1 else "B") + (" lstm-shift" if g.b_mode == 2 else "")
    c_bytes = g.M * g.N * (4 if g.out_f32 else es) * (batch if g.c_bs else 1)
    dims = dict(M=g.M, N=g.N, K=g.K, batch=batch, splitk=g.splitk, form=kind, dtype="bf16" if g.dtype == 1 else "fp32")
    return dims, flops, a_bytes + b_bytes + c_bytes


def _conv3(i0, dt_idx):
    def f(args):
        B, F, T, Ci, Co = (_v(args[i0 + k]) for k in range(5))
        es = _es(_v(args[dt_idx]) if dt_idx is not None else 1)
        return dict(B=B, F=F, T=T, Ci=Ci, Co=Co), 2.0 * B * F * T * 9 * Ci * Co, B * F * T * (Ci + Co) * es
    return f


def _lstm(i0, fused=False):
    def f(args):
        N, L, heads, H = (_v(args[i0 + k]) for k in range(4))
        flops = 2.0 * N * L * 2 * H * 4 * H
        d = dict(N=N, L=L, H=H)
        if fused:
            Cp = _v(args[i0 + 4])
            flops += 2.0 * N * L * 2 * Cp * 4 * H
            d["Cp"] = Cp
        # Priced against the MFMA peak like the contraction it replaces (SURVEY 8(d) "LSTM input + recurrent GEMMs";
        # MFMA_PRICED below), although the saved activations of the stationary kernels -- per row and direction xg /
        # dG 4H, gates 4H, y / dy H bf16 and c H fp32: 44 H bytes per row -- put the intensity (186 flop / byte at
        # H = 512) below the ridge: the row carries that fraction too (`hbm_frac`).
        return d, flops, (0.0 if fused else float(N) * L * 44 * H)
    return f


MFMA_PRICED = ("dtt_lstm_rs_fwd", "dtt_lstm_rs_bwd", "dtt_lstm_cluster_fwd", "dtt_lstm_cluster_bwd")


def _ew(idx, ntensors, dt_idx):
    """elementwise / normalisation passes: `ntensors` tensors of prod(args[idx]) elements cross HBM once."""
    def f(args):
        n = 1
        dims = {}
        for k, i in idx.items():
            dims[k] = _v(args[i])
            n *= dims[k]
        return dims, 0.0, n * ntensors * _es(_v(args[dt_idx]))
    return f


def _dwconv(args):
    N, H, W, Cc, stride = (_v(args[i]) for i in range(6, 11))
    es = _es(_v(args[11]))
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    return dict(N=N, H=H, W=W, C=Cc, stride=stride), 0.0, N * Cc * (H * W + Ho * Wo) * es


def _shuffle(args):
    P, h, hp = (_v(args[i]) for i in range(11, 14))
    es = _es(_v(args[14]))
    return dict(P=P, h=h, hp=hp), 0.0, P * (2 * h + 2 * hp) * es


def _avgpool(args):
    NF, S, Cc = (_v(args[i]) for i in range(5, 8))
    return dict(NF=NF, S=S, C=Cc), 0.0, NF * S * Cc * _es(_v(args[8])) + NF * Cc * 4


SPECS = {
    "dtt_dwconv3x3_bnrelu": _dwconv, "dtt_bn_relu_shuffle": _shuffle, "dtt_bn_relu_avgpool": _avgpool,
    "dtt_gemm": _gemm,
    "dtt_conv3x3_s1": _conv3(7, 13),
    "dtt_conv3x3_wgrad": _conv3(3, None),
    "dtt_conv3x3_wgrad_ws": _conv3(5, None), "dtt_conv3x3_wgrad_ws_param": _conv3(5, None),
    "dtt_lstm_rs_fwd": _lstm(6), "dtt_lstm_cluster_fwd": _lstm(6),
    "dtt_lstm_rs_bwd": _lstm(8), "dtt_lstm_cluster_bwd": _lstm(8),
    "dtt_lstm_fwd": _lstm(5), "dtt_lstm_bwd": _lstm(5), "dtt_lstm_fwd_fused": _lstm(7, True), "dtt_lstm_bwd_blk": _lstm(6),
    "dtt_in_gelu_fwd": _ew(dict(B=3, F=4, T=5, C=8), 2, 9),            # read y, write z
    "dtt_in_gelu_bwd": _ew(dict(B=6, F=7, T=8, C=11), 3, 13),          # read y, dz; write dy (its two sweeps re-read)
    "dtt_in_stats": _ew(dict(B=3, P=4, C=5), 1, 8),
    "dtt_bn_stats": _ew(dict(P=3, C=4), 1, 6), "dtt_bn_stats_running": _ew(dict(P=3, C=4), 1, 6),
    "dtt_bn_act_fwd": _ew(dict(P=6, C=7), 2, 9),
    "dtt_bn_act_bwd": _ew(dict(P=11, C=12), 3, 15),
    "dtt_bn_act_pool_fwd": _ew(dict(N=6, H=7, W=8, C=9), 1.25, 11),
    "dtt_maxpool3x3s2_bwd": _ew(dict(N=3, H=4, W=5, C=6), 1.25, 7),
}


# queries and switches: no launch behind them
_NO_LAUNCH = ("_bytes", "_words", "_chunk", "_chunks", "_geometry", "_supported", "_covers", "_version", "_last_error",
              "_drop_workgroups")


class _Proxy:
    def __init__(self, real, timer):
        self._real, self._timer = real, timer

    def __getattr__(self, name):
        fn = getattr(self._real, name)
        if not name.startswith("dtt_") or not self._timer.active or name.endswith(_NO_LAUNCH):
            return fn
        timer = self._timer

        def call(*args):
            if torch.cuda.is_current_stream_capturing():
                return fn(*args)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            rc = fn(*args)
            e1.record()
            spec = SPECS.get(name)
            try:
                dims, flops, nbytes = spec(args) if spec else ({}, 0.0, 0.0)
            except Exception:                                  # noqa: BLE001  (a signature the table does not know)
                dims, flops, nbytes = {}, 0.0, 0.0
            timer.calls.append((name, tuple(sorted(dims.items())), flops, nbytes, e0, e1))
            return rc
        return call


class CallTimer:
    """Context manager; ``table(nsteps)`` after the block (synchronises)."""

    def __init__(self):
        self.calls, self.active = [], False

    def __enter__(self):
        _lib.lib()
        self._real = _lib._lib
        _lib._lib = _Proxy(self._real, self)
        self.active = True
        self.t0, self.t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        self.t0.record()
        return self

    def __exit__(self, *exc):
        self.t1.record()
        self.active = False
        _lib._lib = self._real
        return False

    def table(self, nsteps=1):
        torch.cuda.synchronize()
        span_ms = self.t0.elapsed_time(self.t1) / nsteps
        rows = {}
        for name, dims, flops, nbytes, e0, e1 in self.calls:
            r = rows.setdefault((name, dims), dict(entry=name, shape=dict(dims), n=0, ms=0.0, flops=flops, bytes=nbytes))
            r["n"] += 1
            r["ms"] += e0.elapsed_time(e1)
        out = []
        for r in rows.values():
            n, tot = r["n"], r["ms"]
            avg_s = tot / n / 1e3
            row = dict(entry=r["entry"], shape=r["shape"], launches_per_step=round(n / nsteps, 2),
                       ms_per_step=round(tot / nsteps, 4), avg_us=round(avg_s * 1e6, 1))
            if r["flops"] > 0 or r["bytes"] > 0:
                fp32 = r["shape"].get("dtype") == "fp32"
                # (the ridge of the type the entry computes in: the exact-fp32 DFT products sit far above theirs)
                ridge = RIDGE * PEAK_FP32_MFMA_TFLOPS / PEAK_BF16_TFLOPS if fp32 else RIDGE
                mfma = r["entry"] in MFMA_PRICED or r["bytes"] <= 0 or r["flops"] / max(r["bytes"], 1.0) >= ridge
                if mfma:
                    peak = PEAK_FP32_MFMA_TFLOPS if fp32 else PEAK_BF16_TFLOPS
                    ach = r["flops"] / avg_s / 1e12
                    row.update(bound="mfma", achieved=round(ach, 1), peak=peak, unit="TFLOP/s", frac=round(ach / peak, 4),
                               flops_per_launch=r["flops"])
                    if r["bytes"] > 0 and r["flops"] / r["bytes"] < ridge:      # below the ridge by its own bytes
                        row.update(bytes_per_launch=r["bytes"], hbm_frac=round(r["bytes"] / avg_s / 1e9 / PEAK_HBM_GBS, 4))
                else:
                    ach = r["bytes"] / avg_s / 1e9
                    row.update(bound="hbm", achieved=round(ach, 1), peak=PEAK_HBM_GBS, unit="GB/s",
                               frac=round(ach / PEAK_HBM_GBS, 4), bytes_per_launch=r["bytes"])
            out.append(row)
        out.sort(key=lambda r: -r["ms_per_step"])
        ours = sum(r["ms_per_step"] for r in out)
        return dict(span_ms_per_step=round(span_ms, 3), entry_points_ms_per_step=round(ours, 3),
                    outside_entry_points_ms_per_step=round(span_ms - ours, 3), rows=out)


def format_table(tab, top=40):
    lines = [f"eager step {tab['span_ms_per_step']:.2f} ms; inside C-ABI entry points {tab['entry_points_ms_per_step']:.2f} ms; "
             f"outside (torch glue, launch gaps) {tab['outside_entry_points_ms_per_step']:.2f} ms",
             f"{'entry point':28s} {'n/step':>6s} {'ms/step':>8s} {'avg us':>8s} {'achieved':>14s} {'frac':>6s}  shape"]
    for r in tab["rows"][:top]:
        ach = f"{r['achieved']:.0f} {r['unit']}" if "achieved" in r else ""
        frac = f"{r['frac']:.3f}" if "frac" in r else ""
        shape = " ".join(f"{k}={v}" for k, v in r["shape"].items())
        if "hbm_frac" in r:
            shape += f"  (by its {r['bytes_per_launch'] / 1e6:.0f} MB of saved activations: {r['hbm_frac']:.3f} of HBM)"
        lines.append(f"{r['entry'][4:]:28s} {r['launches_per_step']:6.1f} {r['ms_per_step']:8.3f} {r['avg_us']:8.1f} {ach:>14s} {frac:>6s}  {shape}")
    return "\n".join(lines)
