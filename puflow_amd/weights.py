"""Deterministic synthetic weights for the discrete PointInterpFlow (806 103 parameters).

The key names / shapes restate the reference `state_dict` layout
(`modules/discrete/interpflow.py:262-290`), so a dict made here
loads into the reference module and a reference checkpoint loads into ours.  Values come
from numpy's PCG64 so the same seed gives the same bits on every box (no weight blob has
to travel to the GPU machine).  Statistics are chosen "trained-like": BN running stats away
from (0,1), ActNorm / inv1x1 non-trivial, the zero-initialised last conditioner layers
(`interpflow.py:26-28`) made non-zero so every coupling is exercised.
"""
from __future__ import annotations

from collections import OrderedDict
from typing import List, Tuple

import numpy as np
import torch

NUM_BLOCKS = 6
FEAT_CHANNELS = [3, 32, 64, 128, 128, 128, 128]
GROWTH = [8, 16, 32, 32, 32, 32]
COND_CHANNELS = [32, 64, 128, 128, 128, 128]
HDIM = 64


def _edgeconv_spec(pfx: str, cin: int, odim: int, g: int) -> List[Tuple[str, tuple, str]]:
    spec = []
    nconv = odim // g
    for t in range(nconv):
        ic = 3 * cin + g * t
        spec += [(f"{pfx}.convs.{t}.0.weight", (g, ic, 1, 1), "w"),
                 (f"{pfx}.convs.{t}.0.bias", (g,), "b"),
                 (f"{pfx}.convs.{t}.1.weight", (g,), "bn_w"),
                 (f"{pfx}.convs.{t}.1.bias", (g,), "b"),
                 (f"{pfx}.convs.{t}.1.running_mean", (g,), "bn_m"),
                 (f"{pfx}.convs.{t}.1.running_var", (g,), "bn_v"),
                 (f"{pfx}.convs.{t}.1.num_batches_tracked", (), "nbt")]
    spec += [(f"{pfx}.conv_out.weight", (odim, 3 * cin + g * nconv, 1, 1), "w"),
             (f"{pfx}.conv_out.bias", (odim,), "b")]
    return spec


def _lin_a1d_spec(pfx: str, cin: int, cout: int) -> List[Tuple[str, tuple, str]]:
    return [(f"{pfx}.layers.0.weight", (HDIM, cin), "w"),
            (f"{pfx}.layers.2.weight", (HDIM, HDIM), "w"),
            (f"{pfx}.layers.2.bias", (HDIM,), "b"),
            (f"{pfx}.layers.4.weight", (cout, HDIM), "w_last"),
            (f"{pfx}.layers.4.bias", (cout,), "b_last")]


def _mlp_bn_spec(pfx: str, dims: List[int]) -> List[Tuple[str, tuple, str]]:
    """Conv,BN,LReLU,Conv,BN,LReLU,Conv  -> sequential indices 0,1,3,4,6."""
    spec = []
    for li, a in enumerate((0, 3, 6)):
        spec += [(f"{pfx}.{a}.weight", (dims[li + 1], dims[li], 1, 1), "w"),
                 (f"{pfx}.{a}.bias", (dims[li + 1],), "b")]
        if a != 6:
            c = dims[li + 1]
            spec += [(f"{pfx}.{a + 1}.weight", (c,), "bn_w"), (f"{pfx}.{a + 1}.bias", (c,), "b"),
                     (f"{pfx}.{a + 1}.running_mean", (c,), "bn_m"),
                     (f"{pfx}.{a + 1}.running_var", (c,), "bn_v"),
                     (f"{pfx}.{a + 1}.num_batches_tracked", (), "nbt")]
    return spec


def state_dict_spec() -> List[Tuple[str, tuple, str]]:
    """(key, shape, kind) in the reference's registration order (408 entries)."""
    spec: List[Tuple[str, tuple, str]] = []
    spec += _mlp_bn_spec("interp.knn_context.distance_encoder.mlp", [10, 64, 64, 128])
    spec += _edgeconv_spec("interp.knn_context.feat_conv", 3, 128, 16)
    spec += _mlp_bn_spec("interp.weight_unit.mlp", [256, 128, 64, 32])
    for i in range(NUM_BLOCKS):
        spec += _edgeconv_spec(f"feat_convs.{i}", FEAT_CHANNELS[i], FEAT_CHANNELS[i + 1], GROWTH[i])
    for i in range(NUM_BLOCKS):
        o = FEAT_CHANNELS[i + 1]
        spec += [(f"merge_convs.{i}.conv1.weight", (o // 2, o), "w"),
                 (f"merge_convs.{i}.conv1.bias", (o // 2,), "b"),
                 (f"merge_convs.{i}.conv2.weight", (COND_CHANNELS[i], o // 2), "w")]
    for i in range(NUM_BLOCKS):
        p = f"flow_blocks.{i}"
        tdim = 1 if i % 2 == 0 else 2
        cdim = COND_CHANNELS[i]
        spec += [(p + ".actnorm.logs", (1, 1, 3), "an_logs"), (p + ".actnorm.bias", (1, 1, 3), "an_bias"),
                 (p + ".permutate1.permutater.W", (3, 3), "inv1x1"),
                 (p + ".permutate2.permutater.direct_idx", (3,), "rev"),
                 (p + ".permutate2.permutater.inverse_idx", (3,), "rev")]
        spec += _lin_a1d_spec(p + ".coupling1.bias_net", tdim + cdim, 3 - tdim)
        spec += _lin_a1d_spec(p + ".coupling2.bias_net", cdim, 3)
        spec += _lin_a1d_spec(p + ".coupling2.scale_net", cdim, 3)
    return spec


def synth_state_dict(seed: int = 2021, style: str = "unit") -> "OrderedDict[str, torch.Tensor]":
    """style "unit": fan-in scaled weights, O(1) activations everywhere (default: benchmarks, most tests).
    style "trained": the wide dynamic range of the reference's pretrained checkpoints (weight std ~0.17
    independent of fan-in with a heavy tail, BN gamma ~N(1, 0.6), BN beta ~N(-0.3, 0.35): features reach
    |h| ~ 50, BN variances ~1e3) - BN running statistics and ActNorm must then be calibrated on data
    (tests do it with the oracle's train-mode forward) to obtain a self-consistent model."""
    rng = np.random.Generator(np.random.PCG64(seed))
    sd: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    if style == "trained":
        for key, shape, kind in state_dict_spec():
            if kind == "w":
                fan_in = int(np.prod(shape[1:]))
                gain = 2.5 if ("conv_out" in key or "feat_convs" in key) else 1.0     # un-normalised layers grow the range
                v = rng.standard_t(5, shape) * (gain * 0.77 / np.sqrt(fan_in))         # heavy tail (t5: std 1.29)
            elif kind == "w_last":                                       # zero-init in the reference: stays small when trained
                v = rng.standard_normal(shape) * 0.002
            elif kind == "b":
                v = rng.standard_normal(shape) * 0.2 - (0.3 if ".1.bias" in key or ".4.bias" in key else 0.0)
            elif kind == "b_last":
                v = rng.standard_normal(shape) * 0.05
            elif kind == "bn_w":
                v = np.clip(rng.standard_normal(shape) * 0.5 + 1.04, 0.05, None)
            elif kind == "bn_m":
                v = np.zeros(shape)                                      # calibrated later
            elif kind == "bn_v":
                v = np.ones(shape)
            elif kind == "an_logs":
                v = rng.standard_normal(shape) * 0.3 + 0.3
            elif kind == "an_bias":
                v = rng.standard_normal(shape) * 0.06
            elif kind == "inv1x1":
                q, _ = np.linalg.qr(rng.standard_normal((3, 3)))
                v = q * rng.uniform(0.7, 1.4, (1, 3))
            elif kind == "rev":
                sd[key] = torch.tensor([2, 1, 0], dtype=torch.int64)
                continue
            elif kind == "nbt":
                sd[key] = torch.tensor(100, dtype=torch.int64)
                continue
            else:
                raise KeyError(kind)
            sd[key] = torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32))
        return sd
    for key, shape, kind in state_dict_spec():
        if kind == "w":
            fan_in = int(np.prod(shape[1:]))
            v = rng.standard_normal(shape) * (1.0 / np.sqrt(fan_in))
        elif kind == "w_last":
            v = rng.standard_normal(shape) * (0.4 / np.sqrt(shape[1]))
        elif kind == "b":
            v = rng.standard_normal(shape) * 0.1
        elif kind == "b_last":
            v = rng.standard_normal(shape) * 0.05
        elif kind == "bn_w":
            v = rng.uniform(0.6, 1.4, shape)
        elif kind == "bn_m":
            v = rng.standard_normal(shape) * 0.2
        elif kind == "bn_v":
            v = rng.uniform(0.5, 1.5, shape)
        elif kind == "an_logs":
            v = rng.standard_normal(shape) * 0.2
        elif kind == "an_bias":
            v = rng.standard_normal(shape) * 0.2
        elif kind == "inv1x1":
            q, _ = np.linalg.qr(rng.standard_normal((3, 3)))
            v = q * rng.uniform(0.8, 1.3, (1, 3))        # |det| != 1 so log|det W| != 0
        elif kind == "rev":
            sd[key] = torch.tensor([2, 1, 0], dtype=torch.int64)
            continue
        elif kind == "nbt":
            sd[key] = torch.tensor(100, dtype=torch.int64)
            continue
        else:
            raise KeyError(kind)
        sd[key] = torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32))
    return sd


def synth_patches(batch: int, npoint: int, seed: int = 2021, surface: bool = True) -> torch.Tensor:
    """Synthetic input patches [B,N,3] fp32, normalised like `PatchHelper.normalize_pc`
    (`modules/utils/patch.py:168-178`: centre, divide by max norm).  `surface=True` samples a
    bumpy 2-manifold + small jitter (patch-like); otherwise U(-1,1)^3."""
    rng = np.random.Generator(np.random.PCG64(seed + 7919))
    if surface:
        uv = rng.uniform(-1, 1, (batch, npoint, 2))
        a = rng.uniform(0.5, 2.0, (batch, 1, 2))
        ph = rng.uniform(0, 2 * np.pi, (batch, 1, 2))
        h = 0.3 * np.sin(a[..., 0] * np.pi * uv[..., 0] + ph[..., 0]) * np.cos(a[..., 1] * np.pi * uv[..., 1] + ph[..., 1])
        pts = np.concatenate([uv, h[..., None]], axis=-1) + rng.standard_normal((batch, npoint, 3)) * 0.005
    else:
        pts = rng.uniform(-1, 1, (batch, npoint, 3))
    pts = pts.astype(np.float32)
    pts = pts - pts.mean(axis=1, keepdims=True, dtype=np.float32)
    scale = np.sqrt((pts ** 2).sum(-1, keepdims=True)).max(axis=1, keepdims=True)
    return torch.from_numpy((pts / scale).astype(np.float32))


# ---------------------------------------------------------------------------------------
# Continuous (CNF) variant: modules/continuous/interpflow.py.  Shares the interp / feat_convs / merge_convs
# keys with the discrete model (288 of them); flow_blocks.{i} holds a CNF (17 entries each, 390 keys in total -
# the key list of pretrain/puflow-x4-cnf-pu1k.pt).
# ---------------------------------------------------------------------------------------
CNF_COND_CHANNELS = [32, 64, 128, 128, 128, 128]


def cnf_state_dict_spec():
    """[(key, shape, kind)] in the reference's registration order."""
    spec = [e for e in state_dict_spec() if not e[0].startswith("flow_blocks.")]
    for i, cd in enumerate(CNF_COND_CHANNELS):
        p = f"flow_blocks.{i}.cnf"
        spec.append((p + ".sqrt_end_time", (), "T"))
        spec.append((p + ".odefunc._num_evals", (), "nfe"))
        for j, (din, dout) in enumerate([(3, 64), (64, 64), (64, 3)]):
            q = f"{p}.odefunc.diffeq.layers.{j}"
            spec.append((q + "._layer.weight", (dout, din), "w"))
            spec.append((q + "._layer.bias", (dout,), "b"))
            spec.append((q + "._hyper_bias.weight", (dout, 1 + cd), "w"))
            spec.append((q + "._hyper_gate.weight", (dout, 1 + cd), "w"))
            spec.append((q + "._hyper_gate.bias", (dout,), "b"))
    return spec


# Integration end times T = sqrt_end_time^2 of the six CNF blocks of the reference's trained checkpoint
# (pretrain/puflow-x4-cnf-pu1k.pt, `flow_blocks.{i}.cnf.sqrt_end_time`; cnf.py:41 initialises every block to 0.5 and trains
# it): the last two blocks integrate 10x / 100x longer than the first four.  Six numbers describing the checkpoint; the
# synthetic benchmark workload uses them so that dopri5 sees a trained model's mix of short stiff and long smooth blocks.
CNF_PU1K_END_TIMES = (0.330, 0.372, 0.310, 0.195, 2.92, 36.3)
CNF_PU1K_DYNAMICS = 1.7     # with the end times above: 462 evaluations / 59 accepted / 14 rejected steps per 1 x 2048 forward
                            # (the trained checkpoint: 462 / 62 / 11), and a map as well-conditioned as the trained one
                            # (fp32 vs fp64 oracle: 3e-4 of max|x|; the old `dynamics=5, T=0.5` workload: 1e-2)


def synth_cnf_state_dict(seed: int = 2021, dynamics: float = 1.0, end_times=None) -> "OrderedDict[str, torch.Tensor]":
    """Random-init weights of the continuous model: the shared extractor / interpolation part is the discrete
    generator's, the ODE nets get nn.Linear-style fan-in scaling (large enough that dopri5 takes real steps).
    dynamics > 1 scales the main path of the ODE nets (`_layer` weights): a stiffer right-hand side, more and rejected steps;
    end_times: the six integration end times (default: cnf.py:41's initial 0.5 everywhere).  bench.py --mode cnf uses
    (CNF_PU1K_DYNAMICS, CNF_PU1K_END_TIMES) to give the solver the workload it has on the reference's pretrained checkpoint."""
    base = synth_state_dict(seed)
    rng = np.random.Generator(np.random.PCG64(seed + 104729))
    sd: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    for key, shape, kind in cnf_state_dict_spec():
        if not key.startswith("flow_blocks."):
            sd[key] = base[key].clone()
        elif kind == "T":
            T_end = 0.5 if end_times is None else float(end_times[int(key.split(".")[1])])       # cnf.py:41: T = 0.5 at init
            sd[key] = torch.tensor(float(np.sqrt(T_end)), dtype=torch.float32)
        elif kind == "nfe":
            sd[key] = torch.tensor(0.0, dtype=torch.float32)
        else:
            fan_in = shape[1] if len(shape) == 2 else 64
            scale = (1.6 * dynamics if "_layer" in key else 1.0) / np.sqrt(fan_in)
            sd[key] = torch.from_numpy((rng.uniform(-1, 1, shape) * scale).astype(np.float32))
    return sd
