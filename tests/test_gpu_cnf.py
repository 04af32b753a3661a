"""Continuous (CNF) variant on the GPU against oracle/cnf_ref.py.  PARITY UNPINNED against the
reference (torchdiffeq absent): the oracle is a from-text restatement, see its header."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import cnf_ref as C
from oracle import ref_cpu as O
from puflow_amd.weights import cnf_state_dict_spec, synth_cnf_state_dict, synth_patches

DEV = "cuda:0"


def _net(sd):
    from puflow_amd.cnf import PointInterpFlow
    net = PointInterpFlow(3)
    missing, unexpected = net.load_state_dict(sd, strict=True)
    assert not missing and not unexpected
    return net.to(DEV).eval()


def test_state_dict_surface():
    from puflow_amd.cnf import PointInterpFlow
    keys = list(PointInterpFlow(3).state_dict().keys())
    assert keys == [k for k, _, _ in cnf_state_dict_spec()] and len(keys) == 390


@pytest.mark.parametrize("block,R,reverse", [(0, 1, False), (3, 1, False), (5, 4, True), (2, 4, True)])
def test_rhs_matches_autograd_oracle(block, R, reverse):
    """One right-hand side (network + Hutchinson term by explicit VJP) against the oracle's autograd evaluation."""
    sd = synth_cnf_state_dict(7)
    net = _net(sd)
    eng = net._engine(4)
    g = torch.Generator().manual_seed(block)
    T = 200
    cd = sd[f"flow_blocks.{block}.cnf.odefunc.diffeq.layers.0._hyper_gate.weight"].shape[1] - 1
    c = torch.randn(T, cd, generator=g) * 0.7
    e = torch.randn(T, 3, generator=g)
    rows = T * R
    state = torch.randn(rows, 4, generator=g) * 0.8
    t = 0.137
    cr = torch.repeat_interleave(c, R, dim=0)
    er = torch.repeat_interleave(e, R, dim=0)
    ref = C.rhs(sd, block, t, state, cr, er)
    if reverse:
        ref = -ref
    ctx = eng.context(block, c.to(DEV))
    out = torch.empty(rows, 4, device=DEV)
    yd = state.to(DEV)
    eng._rhs(block, yd, yd, [], 0.0, t, -1.0 if reverse else 1.0, ctx, e.to(DEV), out, None, rows, R)
    assert (out.cpu() - ref).abs().max() < 2e-5 * max(1.0, float(ref.abs().max()))


@pytest.mark.parametrize("B,N,seed", [(1, 256, 0), (2, 200, 1)])
def test_forward_matches_oracle(B, N, seed):
    sd = synth_cnf_state_dict(seed)
    xyz = synth_patches(B, N, seed=seed + 10)
    g = torch.Generator().manual_seed(seed)
    noise = [torch.randn(B, N, 3, generator=g) for _ in range(6)]
    ref = C.forward(sd, xyz, 4, noise=noise, stages=True)
    net = _net(sd)
    st = net(xyz.to(DEV), 4, noise=[n.to(DEV) for n in noise], stages=True)
    assert torch.equal(st["idx16"].cpu().long(), ref["idx16"])
    # same step sequence (the controller sees the same norms up to fp32 rounding) -> same function-evaluation count
    assert st["nfe"] == ref["nfe"] and st["accepted"] == ref["accepted"] and st["rejected"] == ref["rejected"]
    # tolerance = the solver's own (atol = rtol = 1e-5 per step, 12 integrations chained)
    assert (st["z"].cpu() - ref["z"]).abs().max() < 1e-4
    assert (st["x"].cpu() - ref["x"]).abs().max() < 1e-4
    assert ((st["ldj"].cpu() - ref["ldj"]).abs() / ref["ldj"].abs().clamp_min(1.0)).max() < 1e-4
    assert abs(float(st["logp"]) - float(ref["logp"])) / abs(float(ref["logp"])) < 1e-4


def test_flow_is_invertible_and_batch_independent_given_steps():
    """f then g with R = 1 on the same latents returns the input within the solver tolerance."""
    sd = synth_cnf_state_dict(3)
    net = _net(sd)
    xyz = synth_patches(2, 256, seed=5).to(DEV)
    eng = net._engine(1)
    base = eng.base
    idx16 = base.knn(xyz)
    cs, _, _ = base.features(xyz, idx16, want_cs=True)
    T = 512
    e = torch.randn(T, 3, device=DEV)
    p = xyz.reshape(T, 3)
    ctxs = [eng.context(i, cs[i].reshape(T, -1)) for i in range(6)]
    for i in range(6):
        p = eng.integrate(i, p, ctxs[i], e, 1, False, 0, 0.0)[:, :3].contiguous()
    for i in reversed(range(6)):
        p = eng.integrate(i, p, ctxs[i], e, 1, True, 0, 0.0)[:, :3].contiguous()
    assert (p.view(2, 256, 3) - xyz).abs().max() < 2e-4


def test_cli_continuous(tmp_path):
    """`python -m puflow_amd.upsample_cnf` (modules/continuous/upsample.py): .xyz in -> 4x .xyz out, on the surface."""
    from puflow_amd import upsample_cnf
    src, dst = tmp_path / "in", tmp_path / "out"
    src.mkdir()
    pts = synth_patches(1, 512, seed=11)[0].numpy()
    np.savetxt(src / "a.xyz", pts, fmt="%.6f")
    ck = tmp_path / "cnf.pt"
    torch.save(synth_cnf_state_dict(5), ck)
    upsample_cnf.main(["--source", str(src), "--target", str(dst), "--checkpoint", str(ck), "--up_ratio", "4"])
    out = np.loadtxt(dst / "a.xyz", dtype=np.float32)
    assert out.shape == (2048, 3) and np.isfinite(out).all()
    d = O.pairwise_sqdist(torch.from_numpy(out)[None], torch.from_numpy(pts)[None]).min(-1)[0].sqrt()
    assert float(d.max()) < 0.5          # random weights: a sanity bound, the cloud stays around the input surface


def test_rhs_kernel_matches_reference_golden(golden_dir):
    """PINNED: `pf_cnf_rhs` against the REFERENCE's own ODEfunc.forward outputs (tools/make_golden_cnf.py)."""
    import os
    g = np.load(os.path.join(golden_dir, "cnf_rhs.npz"))
    sd = synth_cnf_state_dict(int(g["meta_seed"]))
    net = _net(sd)
    eng = net._engine(4)
    for block, R in ((0, 1), (3, 1), (5, 4), (2, 4)):
        tag = f"b{block}_R{R}"
        y, c, e = (torch.from_numpy(g[f"{tag}_{k}"]) for k in ("y", "c", "e"))
        B, NR, _ = y.shape
        T = c.shape[0] * c.shape[1]
        rows = B * NR
        state = torch.cat([y.reshape(rows, 3), torch.zeros(rows, 1)], dim=-1).to(DEV)
        ctx = eng.context(block, c.reshape(T, -1).to(DEV).contiguous())
        out = torch.empty(rows, 4, device=DEV)
        eng._rhs(block, state, state, [], 0.0, float(g[f"{tag}_t"]), 1.0, ctx, e.reshape(T, 3).to(DEV).contiguous(), out, None,
                 rows, R)
        out = out.cpu()
        assert (out[:, :3] - torch.from_numpy(g[f"{tag}_dy"]).reshape(-1, 3)).abs().max() < 1e-5
        assert (out[:, 3] - torch.from_numpy(g[f"{tag}_ndiv"]).reshape(-1)).abs().max() < 1e-5


# ---- the reference's TRAINED continuous checkpoint (tests/golden/pretrained_cnf.npz, tools/make_golden_cnf_pretrained.py) ----
def _pretrained(golden_dir):
    import os
    g = np.load(os.path.join(golden_dir, "pretrained_cnf.npz"))
    sd = {k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd/")}
    return g, sd


def test_pretrained_checkpoint_rhs_matches_reference_golden(golden_dir):
    """PINNED, trained weights: `pf_cnf_rhs` against what the REFERENCE's own ODEfunc.forward returns with
    pretrain/puflow-x4-cnf-pu1k.pt, every block, forward (R = 1) and inverse pass (R = 4).  The trained nets have a
    Hutchinson term of up to ~70 per row: the bound is 1e-5 of each output's scale."""
    g, sd = _pretrained(golden_dir)
    net = _net(sd)
    eng = net._engine(4)
    noise = torch.from_numpy(g["noise"])
    T = g["xyz"].shape[0] * g["xyz"].shape[1]
    for block in range(6):
        # the conditioning features ODEfunc saw (the fixture's): this test is about the right-hand-side kernel alone
        ctx = eng.context(block, torch.from_numpy(g[f"rhs/b{block}_c"]).reshape(T, -1).to(DEV).contiguous())
        e = noise[block].reshape(T, 3).to(DEV).contiguous()
        for R in (1, 4):
            tag = f"rhs/b{block}_R{R}"
            y = torch.from_numpy(g[tag + "_y"]).reshape(-1, 3)
            rows = y.shape[0]
            state = torch.cat([y, torch.zeros(rows, 1)], dim=-1).to(DEV)
            out = torch.empty(rows, 4, device=DEV)
            eng._rhs(block, state, state, [], 0.0, float(g[tag + "_t"]), 1.0, ctx, e, out, None, rows, R)
            out = out.cpu()
            dy, nd = torch.from_numpy(g[tag + "_dy"]).reshape(-1, 3), torch.from_numpy(g[tag + "_ndiv"]).reshape(-1)
            assert (out[:, :3] - dy).abs().max() < 1e-5 * max(1.0, float(dy.abs().max())), tag
            assert (out[:, 3] - nd).abs().max() < 1e-5 * max(1.0, float(nd.abs().max())), tag


def _oracle_cache(golden_dir_, prefix, fp_key, xyz, noise, keys32, keys64):
    """Cached outputs of the CPU oracle (tools/make_golden_cnf_oracle.py -> tests/golden/cnf_oracle_cache.npz) for these exact
    inputs, or None when the fixture is missing or was made from other inputs (the caller then runs the oracle)."""
    import os
    path = os.path.join(golden_dir_, "cnf_oracle_cache.npz")
    if not os.path.exists(path):
        return None
    g = np.load(path)
    fp = np.array([float(xyz.double().sum()), float(xyz.double().abs().sum())] + [float(n.double().sum()) for n in noise])
    if fp_key not in g.files or g[fp_key].shape != fp.shape or not np.allclose(g[fp_key], fp, rtol=0, atol=1e-9):
        return None

    def unpack(pref, keys):
        d = {}
        for k in keys:
            v = g[f"{pref}_{k}"]
            d[k] = int(v) if v.ndim == 0 else torch.from_numpy(v)
        return d
    return unpack(prefix + "_o32", keys32), unpack(prefix + "_o64", keys64), g


def _anchor_report(got, o32, o64):
    """max |a - fp64 oracle| for the HIP path and for the fp32 oracle, per output."""
    rep = {}
    for k in ("z", "x", "ldj"):
        ref = o64[k]
        scale = 1.0 if k != "ldj" else float(ref.abs().max())
        rep[k] = (float((got[k].cpu().double() - ref).abs().max()) / scale, float((o32[k].double() - ref).abs().max()) / scale)
    return rep


def test_pretrained_checkpoint_forward_against_the_fp64_anchor(golden_dir):
    """The integrated path with the reference's trained weights, 1 x 256.  No reference output exists for it (torchdiffeq is
    absent), and with trained weights the map is ill-conditioned at the fp32 level: the fp32 CPU oracle and the same oracle in
    float64 differ by ~5e-3 in z, ~6e-4 in x, and even in the NUMBER of steps (462 vs 468 evaluations, 10 vs 11 rejected: an
    error ratio next to 1.0 falls on either side).  So the HIP path is held to the float64 anchor with the fp32 oracle's own
    distance as the yardstick: no further from fp64 than 4x what plain fp32 arithmetic is, and a step sequence within two
    attempts of the fp32 oracle's."""
    g, sd = _pretrained(golden_dir)
    xyz = torch.from_numpy(g["xyz"])
    noise = [torch.from_numpy(n) for n in g["noise"]]
    cached = _oracle_cache(golden_dir, "pre", "pre_fp", xyz, noise, ("x", "z", "ldj", "idx16", "nfe", "accepted", "rejected"),
                           ("x", "z", "ldj", "nfe", "accepted", "rejected"))
    if cached is not None:                                                  # the oracle's fp32 and float64 results, made on the CPU once
        o32, o64, _ = cached
    else:
        o32 = C.forward(sd, xyz, 4, noise=noise, stages=True)
        o64 = C.forward(sd, xyz, 4, noise=noise, stages=True, dtype=torch.float64)
    net = _net(sd)
    got = net(xyz.to(DEV), 4, noise=[n.to(DEV) for n in noise], stages=True)
    assert torch.equal(got["idx16"].cpu().long(), o32["idx16"])
    assert o32["rejected"] >= 5                                             # the real thing: rejected steps, ~460 evaluations
    assert abs(got["accepted"] - o32["accepted"]) <= 2 and abs(got["rejected"] - o32["rejected"]) <= 2, (got["nfe"], o32["nfe"], o64["nfe"])
    rep = _anchor_report(got, o32, o64)
    print("pretrained CNF vs fp64 anchor (hip, fp32 oracle):", rep, "nfe", got["nfe"], o32["nfe"], o64["nfe"])
    for k, (e_hip, e_o32) in rep.items():
        assert e_hip <= 4.0 * e_o32 + 1e-4, (k, e_hip, e_o32)
    assert torch.isfinite(got["x"]).all() and float(got["x"].abs().max()) < 2.0          # a trained model: the cloud stays on the patch


def test_full_size_properties_32x2048():
    """BASELINE configs[4] at its real size (32 x 2048 -> 8192) on the bench's synthetic workload (the trained checkpoint's
    end times, ODE nets scaled until dopri5 works like on that checkpoint: ~460 evaluations, rejected steps): finite; the two
    first items run alone take the same number of evaluations as the CPU oracle on them and sit as close to the float64
    anchor as the fp32 oracle does; the full batch agrees with the items run alone as well as the ODE's solution is
    determined at rtol = 1e-5 at all - other items in the batch change the solver's step sequence (its RMS norm runs over the
    whole batch), not the model, and that sensitivity is MEASURED here with the float64 oracle (an item alone vs the same
    item in a pair); f then g with R = 1 on all 65 536 rows returns the input within the solver's tolerance."""
    from puflow_amd.weights import CNF_PU1K_DYNAMICS, CNF_PU1K_END_TIMES
    sd = synth_cnf_state_dict(2021, dynamics=CNF_PU1K_DYNAMICS, end_times=CNF_PU1K_END_TIMES)
    net = _net(sd)
    B, N = 32, 2048
    xyz_cpu = synth_patches(B, N, seed=2021)
    xyz = xyz_cpu.to(DEV)
    g = torch.Generator().manual_seed(0)
    noise_cpu = [torch.randn(B, N, 3, generator=g) for _ in range(6)]
    noise = [n.to(DEV) for n in noise_cpu]
    full = net(xyz, 4, noise=noise, stages=True)
    assert full["x"].shape == (B, 4 * N, 3) and torch.isfinite(full["x"]).all() and torch.isfinite(full["z"]).all()
    assert torch.isfinite(full["logp"]) and full["rejected"] >= 5 and full["nfe"] >= 400
    two = net(xyz[:2], 4, noise=[n[:2] for n in noise], stages=True)
    import os
    cached = _oracle_cache(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"), "full", "full_fp", xyz_cpu[:2],
                           [n[:2] for n in noise_cpu], ("x", "z", "ldj", "nfe", "accepted", "rejected"),
                           ("x", "z", "ldj", "nfe", "accepted", "rejected"))
    if cached is not None:      # the CPU oracle's results on these two items (fp32, float64) and on item 0 alone (float64): a minute of
        o32, o64, gc = cached   # CPU time per run otherwise
        o64_1 = {"x": torch.from_numpy(gc["full_o64_1_x"])}
    else:
        o32 = C.forward(sd, xyz_cpu[:2], 4, noise=[n[:2] for n in noise_cpu], stages=True)
        o64 = C.forward(sd, xyz_cpu[:2], 4, noise=[n[:2] for n in noise_cpu], stages=True, dtype=torch.float64)
        o64_1 = C.forward(sd, xyz_cpu[:1], 4, noise=[n[:1] for n in noise_cpu], stages=True, dtype=torch.float64)
    assert (two["nfe"], two["accepted"], two["rejected"]) == (o32["nfe"], o32["accepted"], o32["rejected"])
    rep = _anchor_report(two, o32, o64)
    print("pu1k-like synthetic CNF, 2 x 2048, vs fp64 anchor (hip, fp32 oracle):", rep)
    for k, (e_hip, e_o32) in rep.items():
        assert e_hip <= 4.0 * e_o32 + 1e-4, (k, e_hip, e_o32)
    # batch independence of the MODEL, against the solver's own step-sequence sensitivity (float64: no rounding involved):
    # item 0 alone vs item 0 inside the pair above
    sens = (o64["x"][:1] - o64_1["x"]).abs().max(-1)[0].flatten()
    diff = (full["x"][:2] - two["x"]).abs().max(-1)[0].flatten().cpu().double()
    print(f"step-sequence sensitivity of x (fp64 oracle, item 0 alone vs in a pair): max {float(sens.max()):.3e} median "
          f"{float(sens.median()):.3e}; HIP batch of 32 vs the 2 items alone: max {float(diff.max()):.3e} median {float(diff.median()):.3e}")
    assert float(diff.median()) <= 8.0 * float(sens.median()) + 1e-4
    assert float(diff.max()) <= 8.0 * float(sens.max()) + 1e-3
    # invertibility at full size (R = 1, all rows, the engine's own integrate calls)
    eng = net._engine(1)
    T = B * N
    cs = full["cs"]
    e = noise[0].reshape(T, 3).contiguous()
    ctxs = [eng.context(i, cs[i].reshape(T, -1)) for i in range(6)]
    p = xyz.reshape(T, 3)
    for i in range(6):
        p = eng.integrate(i, p, ctxs[i], e, 1, False, 0, 0.0)[:, :3].contiguous()
    for i in reversed(range(6)):
        p = eng.integrate(i, p, ctxs[i], e, 1, True, 0, 0.0)[:, :3].contiguous()
    err = (p.view(B, N, 3) - xyz).abs().max()
    zmax = float(full["z"].abs().max())
    print("g(f(x)) - x at 32 x 2048:", float(err), "max|z|", zmax)
    assert err < 2e-4 * max(1.0, zmax)           # the solver's rtol = 1e-5 per step on latents of this size, 12 chained integrations


def test_deferred_forward_equals_the_look_per_batch_loop():
    """The forward enqueued without a look at the step controller (one read of the twelve final controller states) gives the
    result of the look-per-batch loop bit for bit, with the same evaluation / accept / reject counts - also when an
    integration does not finish inside its blind attempts and the forward falls back."""
    from puflow_amd.weights import CNF_PU1K_DYNAMICS, CNF_PU1K_END_TIMES
    sd = synth_cnf_state_dict(2021, dynamics=CNF_PU1K_DYNAMICS, end_times=CNF_PU1K_END_TIMES)
    net = _net(sd)
    xyz = synth_patches(2, 512, seed=4).to(DEV)
    g = torch.Generator().manual_seed(1)
    noise = [torch.randn(2, 512, 3, generator=g).to(DEV) for _ in range(6)]
    eng = net._engine(4)
    res = {}
    eng.async_attempts = 0
    res["loop"] = net(xyz, 4, noise=noise, stages=True)                 # also leaves the attempts each integration took
    eng.async_attempts = 1
    assert eng.hint is not None and len(eng.hint) == 12 and max(eng.hint) > 2 * min(eng.hint)     # the T = 36 block needs the most
    res["deferred"] = net(xyz, 4, noise=noise, stages=True)
    good = list(eng.hint)
    eng.hint = [1] * 12                                                 # a guess that is too small: the blind part ends at the first
    res["fallback"] = net(xyz, 4, noise=noise, stages=True)             # integration that needs more, the loop takes over there
    assert eng.hint is not None and max(eng.hint) > 3
    eng.hint = good[:6] + [1] + good[7:]                                # ... and at the start of the inverse pass (its first
    res["resume_in_g"] = net(xyz, 4, noise=noise, stages=True)          # integration is the T = 36 block: 3 attempts are not enough)
    assert good[6] > 3 and eng.hint[6] == good[6] and eng.hint[:6] == good[:6]
    assert res["loop"]["rejected"] >= 3 and res["loop"]["nfe"] > 300
    for name in ("deferred", "fallback", "resume_in_g"):
        assert torch.equal(res[name]["x"], res["loop"]["x"]) and torch.equal(res[name]["z"], res["loop"]["z"])
        assert torch.equal(res[name]["ldj"], res["loop"]["ldj"])
        assert (res[name]["nfe"], res[name]["accepted"], res[name]["rejected"]) == (res["loop"]["nfe"], res["loop"]["accepted"], res["loop"]["rejected"])


def test_split_gate_kernel_agrees_with_the_plain_one(golden_dir):
    """PF_CNF_SPLIT_GATES (the inverse pass's step kernel computes 2^(gt t + gc) once per point and step and 2^(gt alpha h) once
    per stage instead of one v_exp_f32 per gate and evaluation): same step sequence as the plain kernel on the trained
    checkpoint, and a result no farther from it than the fp32 oracle is from its own float64 evaluation."""
    g, sd = _pretrained(golden_dir)
    net = _net(sd)
    xyz = torch.from_numpy(g["xyz"]).to(DEV)
    noise = [n.to(DEV) for n in torch.from_numpy(g["noise"])]
    eng = net._engine(4)
    assert eng.split == [1] * 6
    res = {}
    for name, flags in (("split", [1] * 6), ("plain", [0] * 6)):
        eng.split = flags
        res[name] = net(xyz, 4, noise=noise, stages=True)
    a, b = res["split"], res["plain"]
    assert abs(a["accepted"] - b["accepted"]) <= 1 and abs(a["rejected"] - b["rejected"]) <= 1
    scale = float(b["x"].abs().max())
    # the trained ODEs amplify one rounding difference: on this patch the fp32 CPU oracle is 2.2e-3 (of scale 0.76) from its own
    # float64 evaluation (fp64 anchor test above); two fp32 evaluations that differ in the rounding of their gates stay inside that
    assert float((a["x"] - b["x"]).abs().max()) <= 2e-3 * scale
    assert float((a["ldj"] - b["ldj"]).abs().max()) <= 2e-3 * max(1.0, float(b["ldj"].abs().max()))


@pytest.mark.parametrize("block", [0, 1, 4])
def test_context_gemm_matches_float64(golden_dir, block):
    """pf_cnf_context (the split-fp16 streaming GEMM, cd = 32 / 64 / 128) against c Hc^T + hb in float64 on the trained
    checkpoint's hyper-network weights: fp32-grade (the products keep 22+ bits)."""
    g, sd = _pretrained(golden_dir)
    eng = _net(sd)._engine(4)
    cd = eng.Hc[block].shape[1]
    gen = torch.Generator().manual_seed(block)
    c = (torch.randn(1000, cd, generator=gen) * 0.8).to(DEV)           # 1000 rows: a ragged last tile
    ctx = eng.context(block, c)
    ref = c.double().cpu() @ eng.Hc[block].double().cpu().T + eng.hb[block].double().cpu()
    err = (ctx.double().cpu() - ref).abs().max()
    assert float(err) <= 2e-6 * max(1.0, float(ref.abs().max())), float(err)


def test_features_only_mode_of_the_conditioner_stage():
    """pf_cond_all with st = cp = NULL (the continuous model's call) writes the same conditioning features as the full stage."""
    sd = synth_cnf_state_dict(5)
    net = _net(sd)
    base = net._engine(4).base
    xyz = synth_patches(2, 300, seed=8).to(DEV)
    idx16 = base.knn(xyz)
    full, cp, st = base.features(xyz, idx16, want_cs=True)
    only, cp0, st0 = base.features(xyz, idx16, want_cs=True, cs_only=True)
    assert cp0 is None and st0 is None and cp is not None
    for a, b in zip(full, only):
        assert torch.equal(a, b)


@pytest.mark.parametrize("R", [2, 3, 8])
def test_forward_other_up_ratios_match_oracle(R):
    """Up-ratios other than 4: R = 8 keeps the inverse pass on the LDS-context kernel (8 rows per point), R = 2 / 3 send it
    through the forward pass's kernel with several rows per point (compact gate columns, rows -> points by division)."""
    sd = synth_cnf_state_dict(11)
    xyz = synth_patches(1, 200, seed=21)
    g = torch.Generator().manual_seed(R)
    noise = [torch.randn(1, 200, 3, generator=g) for _ in range(6)]
    ref = C.forward(sd, xyz, R, noise=noise, stages=True)
    net = _net(sd)
    st = net(xyz.to(DEV), R, noise=[n.to(DEV) for n in noise], stages=True)
    assert tuple(st["x"].shape) == (1, 200 * R, 3)
    assert st["nfe"] == ref["nfe"] and st["accepted"] == ref["accepted"] and st["rejected"] == ref["rejected"]
    assert (st["x"].cpu() - ref["x"]).abs().max() < 1e-4 and (st["z"].cpu() - ref["z"]).abs().max() < 1e-4


def test_scaled_sumsq_vector_and_scalar_paths():
    """pf_scaled_sumsq: the 16-byte-load path (aligned, n % 4 == 0) and the element path (a view that starts one float in)
    against the float64 sum, with and without the second state of the maximum and the subtrahend."""
    import ctypes
    from puflow_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(3)
    n = 4 * 50_001
    base = [torch.randn(n + 4, generator=g).to(DEV) for _ in range(4)]
    ws = torch.empty(256, dtype=torch.float64, device=DEV)
    out = torch.empty(1, dtype=torch.float64, device=DEV)
    w0 = (ctypes.c_float * 1)(0.0)
    for off, nn_ in ((0, n), (1, n), (0, n - 2)):
        a, b, s0, s1 = (t[off:off + nn_] for t in base)
        for use_b, use_s1 in ((False, False), (True, True)):
            _lib.check(lib.pf_scaled_sumsq(a.data_ptr(), b.data_ptr() if use_b else None, s0.data_ptr(),
                                           s1.data_ptr() if use_s1 else None, None, w0, 0, 0.0, 1e-5, 1e-5, nn_, ws.data_ptr(),
                                           out.data_ptr(), torch.cuda.current_stream().cuda_stream), "pf_scaled_sumsq")
            v = (a - b) if use_b else a
            m = torch.maximum(s0.abs(), s1.abs()) if use_s1 else s0.abs()
            r = (v / (1e-5 + 1e-5 * m)).double()            # the kernel divides in fp32 (its denominator is one fma: an ulp from
            ref = float((r * r).sum())                      # torch's mul + add), squares and sums in double
            assert abs(float(out) - ref) <= 1e-7 * ref, (off, nn_, use_b, use_s1)
