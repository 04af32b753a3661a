"""Host logic without a GPU: the C-ABI library loads and exports every symbol declared in
include/puflow_hip.h; the module surface matches the reference state_dict; packing invariants."""
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built_lib():
    from puflow_amd import build
    return build.build(verbose=False)


def test_library_exports_every_declared_symbol(built_lib):
    from puflow_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "puflow_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(pf_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    assert declared == set(_lib.SIGNATURES), (declared ^ set(_lib.SIGNATURES))
    lib = _lib.load()
    for name in declared:
        assert hasattr(lib, name)
    assert lib.pf_version() >= 100
    assert lib.pf_error_string(-2) == b"shape precondition violated"


def test_argument_validation_without_gpu(built_lib):
    """Entry points reject bad arguments before touching the device."""
    from puflow_amd import _lib
    lib = _lib.load()
    assert lib.pf_knn(None, None, 1, 16, 16, 16, None, None, None) == -1
    assert lib.pf_knn(8, 8, 1, 16, 16, 17, 8, None, None) == -2        # K > M
    assert lib.pf_knn(8, 8, 1, 16, 16, 5, 8, None, None) == -3         # K not built
    assert lib.pf_edgeconv(99, 8, 8, 8, 8, 8, 1, 64, None) == -3
    assert lib.pf_interp(8, 8, 8, 8, _lib.offsets([0] * 15), 8, 1, 64, 33, None) == -3


def test_training_entry_points_validate_their_descriptors(built_lib):
    """The fused training entry points (PfEcTrain / PfMlpTrain / PfBnMlpTrain descriptors, the flow-block ops, the fused
    optimizer, the neighbour-list transpose) reject incomplete or unsupported descriptors before any launch."""
    import ctypes
    from puflow_amd import _lib
    lib = _lib.load()
    ec = _lib.PfEcTrain()
    assert lib.pf_ec_train_ws_floats(ctypes.byref(ec)) == -1                      # all-zero shape
    ec.B, ec.N, ec.K, ec.C, ec.growth, ec.nconv, ec.odim, ec.pooling = 2, 64, 16, 32, 16, 4, 64, 1
    assert lib.pf_ec_train_ws_floats(ctypes.byref(ec)) > 0
    assert lib.pf_ec_train_fwd(ctypes.byref(ec), None) == -1                      # no buffers
    ec.growth = 12
    assert lib.pf_ec_train_ws_floats(ctypes.byref(ec)) == -1                      # growth must be 8 / 16 / 32
    ec.growth, ec.K = 16, 8
    assert lib.pf_ec_train_fwd(ctypes.byref(ec), None) == -3                      # pooled units need K = 16
    ec.K, ec.nconv = 16, 3
    assert lib.pf_ec_train_fwd(ctypes.byref(ec), None) == -3                      # growth * nconv must be 32 / 64 / 128
    assert lib.pf_ec_train_fwd(None, None) == -1 and lib.pf_ec_train_bwd(None, None) == -1
    assert lib.pf_ec_train_fold_batch(None, 1, None) == -1 and lib.pf_ec_train_fold_batch(ctypes.byref(ec), 9, None) == -2
    m = _lib.PfMlpTrain()
    assert lib.pf_mlp_train_ws_floats(ctypes.byref(m)) == -1
    m.rows, m.nl, m.td, m.cc, m.cdiv = 64, 3, 1, 48, 1
    m.width[0], m.width[1], m.width[2] = 64, 64, 2
    assert lib.pf_mlp_train_fwd(ctypes.byref(m), None) == -3                      # cc must be 16 / 32 / 64 / 128
    m.cc = 64
    assert lib.pf_mlp_train_fwd(ctypes.byref(m), None) == -1                      # weights missing
    m.cdiv = 3
    assert lib.pf_mlp_train_fwd(ctypes.byref(m), None) == -3                      # replica count must be a power of two <= 16
    assert lib.pf_mlp_train_fwd_batch(None, 2, None, None) == -1
    bm = _lib.PfBnMlpTrain()
    assert lib.pf_bnmlp_train_ws_floats(ctypes.byref(bm)) == -1
    bm.rows, bm.nl, bm.kin0a, bm.kin0b = 256, 3, 10, 0
    bm.width[0], bm.width[1], bm.width[2] = 64, 64, 100
    assert lib.pf_bnmlp_train_fwd(ctypes.byref(bm), None) == -3                   # widths are multiples of 16
    assert lib.pf_flow_affine_fwd(None, None, 1, None, None, None, 0, 10, None, None) == -1
    assert lib.pf_flow_affine_fwd(8, None, 5, 8, 8, 8, 0, 10, 8, None) == -2      # td out of range
    assert lib.pf_couple_inject2_fwd(8, 8, 8, 8, 0, 10, 8, 8, 8, 8, None) == -2   # td must be 1 or 2
    assert lib.pf_inject_inv2_fwd(8, 8, 8, 4, 10, 8, None) == -2                  # rows not a multiple of the replica count
    assert lib.pf_clip_adam(None, None, None, None, None, 4, None, None, 0.9, 0.999, 1e-8, 1e-2, None, None, None, None) == -1
    assert lib.pf_knn_csr(None, 1, 16, 4, None, None, None, None) == -1
    assert lib.pf_knn_csr(8, 0, 16, 4, 8, 8, 8, None) == -2
    assert lib.pf_sum_n(None, 2, None, 4, None) == -1 and lib.pf_copy_n(None, None, None, 1, None) == -1
    assert lib.pf_knn_csr_pair(None, 1, 16, 4, 2, None, None, None, None, None, None) == -1
    assert lib.pf_knn_csr_pair(8, 1, 16, 4, 5, 8, 8, 8, 8, 8, None) == -2          # K2 > K
    assert lib.pf_fps_grouped(None, 1, 16, 4, 0, None, None, None) == -1 and lib.pf_fps_grouped(8, 1, 16, 4, -1, 8, 8, None) == -2
    assert lib.pf_fold_wu_fwd(*([None] * 6), 128, 64, 137, *([None] * 5)) == -1
    assert lib.pf_fold_wu_bwd(*([8] * 5), 0, 64, 137, *([8] * 10), None) == -2
    assert lib.pf_knn_csr_sort(None, None, 4, None) == -1 and lib.pf_knn_csr_sort(8, 8, 0, None) == -2
    assert lib.pf_cnf_steps(None, None, None, None, None, None, None, None, None, 1e-5, 1e-5, 16, 1, 1, None, 0, None) == -1


def test_reduced_precision_library_has_the_same_surface(built_lib):
    """libpuflow_hip_f16.so (bench.py's secondary line) exports every symbol of the main library."""
    import ctypes
    from puflow_amd import _lib, build
    f16 = ctypes.CDLL(build.build_f16(verbose=False))
    for name in _lib.SIGNATURES:
        assert hasattr(f16, name), name


def test_module_state_dict_matches_reference_census(golden_dir):
    import json
    from puflow_amd.interpflow import PointInterpFlow
    from puflow_amd.weights import synth_state_dict
    with open(os.path.join(golden_dir, "state_dict_census.json")) as f:
        census = json.load(f)
    net = PointInterpFlow(3)
    sd = net.state_dict()
    assert [k for k, _, _ in census] == list(sd.keys())
    for k, shape, dt in census:
        assert list(sd[k].shape) == shape and str(sd[k].dtype) == dt, k
    res = net.load_state_dict(synth_state_dict(5))
    assert not res.missing_keys and not res.unexpected_keys
    # zero-initialised last conditioner layers, like the reference (interpflow.py:26-28)
    fresh = PointInterpFlow(3)
    assert float(fresh.flow_blocks[0].coupling1.bias_net.layers[4].weight.abs().sum()) == 0.0


def test_product_fails_loudly_on_cpu():
    from puflow_amd import _lib
    from puflow_amd.interpflow import PointInterpFlow
    net = PointInterpFlow(3).eval()
    net.set_to_initialized_state()
    with pytest.raises(_lib.PuflowHipError):
        net(torch.zeros(1, 64, 3))
    net.train()
    with pytest.raises(_lib.PuflowHipError):
        net(torch.zeros(1, 64, 3))


def test_pack_plan_layout():
    from puflow_amd.packing import FLOW_REC, fold_state_dict, frag_unpack, pack_plan
    from puflow_amd.weights import synth_state_dict
    plan = fold_state_dict(synth_state_dict(0))
    pk = pack_plan(plan)
    blob = pk["blob"]
    assert blob.dtype == np.float32 and blob.size % 64 == 0
    for offs in [pk["ec_w"], pk["interp"], [pk["flow"], pk["ec_tab0"]]] + pk["post"]:
        assert all(o % 4 == 0 for o in offs)                   # 16-byte aligned float4 loads
    # unit 3 growth weights: G1 (2x2 frags) first
    g1 = frag_unpack(blob[pk["ec_w"][3]:pk["ec_w"][3] + 4 * 256].reshape(2, 2, 64, 4), 32, 32)
    np.testing.assert_array_equal(g1, plan["units"][3]["G1"])
    # flow record: natural-scale split-fp16 images of 2^s W2 then the replicated 2^s W4 rows, with 2^-s stored behind:
    # (hi + lo) 2^-s reproduces fp32 to ~2^-22 (the scale puts max |W| at 2^13: every lo that matters is a normal fp16)
    from puflow_amd.packing import frag_unpack_f16n
    rec = blob[pk["flow"] + 2 * FLOW_REC: pk["flow"] + 3 * FLOW_REC]
    W2 = plan["flows"][2]["c1_W2"]
    inv2, inv4 = float(rec[5352]), float(rec[5353])
    assert np.log2(inv2) == np.round(np.log2(inv2)) and 2.0 ** 13 <= np.abs(W2).max() / inv2 < 2.0 ** 14
    np.testing.assert_allclose(frag_unpack_f16n(rec[:4096], 64, 64) * inv2, W2, rtol=2.0 ** -21, atol=1e-12)
    w4 = frag_unpack_f16n(rec[4096:5120], 16, 64) * inv4
    np.testing.assert_allclose(w4[4:6], plan["flows"][2]["c1_W4"], rtol=2.0 ** -21, atol=1e-12)
    np.testing.assert_allclose(rec[5120:5184] * inv2, plan["flows"][2]["c1_b2"], rtol=1e-6)
    np.testing.assert_array_equal(rec[5328:5337].reshape(3, 3), plan["flows"][2]["A"])
    assert FLOW_REC == 5360 and rec.size == FLOW_REC
    # EdgeConv unit 3, f16n image: G1 [32, 32] = 2 ob x 1 pair, columns scaled by a_1 / a_0 = 4 (packing.ec4_scales)
    g1n = frag_unpack_f16n(blob[pk["ec4_w"][3]:pk["ec4_w"][3] + 2 * 512], 32, 32)
    np.testing.assert_allclose(g1n / 4.0, plan["units"][3]["G1"], rtol=2.0 ** -21, atol=2.0 ** -26)      # natural-scale lo: absolute floor 2^-25 of the stored (x4) value
    with pytest.raises(ValueError):
        pack_plan(plan, "bf16x3")


def test_cli_file_sharding_covers_every_file_once():
    from puflow_amd.upsample import shard_paths
    files = [f"c{i:02d}.xyz" for i in (5, 1, 9, 3, 0, 7, 2)]
    for world in (1, 2, 3, 8):
        parts = [shard_paths(files, r, world) for r in range(world)]
        assert sorted(sum(parts, [])) == sorted(files)
        assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1


def test_cnf_module_state_dict_matches_reference_census(golden_dir):
    """The continuous model's parameter names / shapes / dtypes are those of the reference's pretrained CNF checkpoint
    (census made by tools/make_golden.py from pretrain/puflow-x4-cnf-pu1k.pt)."""
    import json
    from puflow_amd.cnf import PointInterpFlow
    with open(os.path.join(golden_dir, "state_dict_census_cnf.json")) as f:
        census = json.load(f)
    sd = PointInterpFlow(3).state_dict()
    assert len(census) == 390 and [k for k, _, _ in census] == list(sd.keys())
    for k, shape, dt in census:
        assert list(sd[k].shape) == shape and str(sd[k].dtype) == dt, k


def test_save_xyz_writes_the_bytes_of_savetxt(tmp_path):
    """upsample.py:57 writes with np.savetxt(fmt='%.6f'); the library's formatter (pf_format_xyz) must produce the same file:
    random clouds, values on and next to rounding ties of the 6th decimal, negative values that round to zero, large and
    tiny magnitudes, non-finite values."""
    import numpy as np
    from puflow_amd.upsample import save_xyz
    rng = np.random.default_rng(3)
    a = ((rng.random((20000, 3)) - 0.5) * 7.0).astype(np.float32)
    a[0] = [0.0, -0.0, 1e-7]
    a[1] = [123456.789, -1e-3, 0.9999995]
    a[2] = [-4e-7, -5e-7, -6e-7]
    a[3] = [0.5, 0.25, 0.125]                                  # exact binary fractions
    a[4] = [2.5e-6, 3.5e-6, -2.5e-6]
    a[5] = [1e9, -3.9e9, 4.1e9]                                # around the fast path's limit
    a[6] = [3.4e38, -1e20, 1e-30]
    a[7] = [np.inf, -np.inf, np.nan]
    a[8] = [-np.nan, np.float32(-0.0), np.float32(1.17549435e-38)]
    ties = (np.arange(0, 3000, dtype=np.float64) + 0.5) * 1e-6           # nearest float32 of k + 0.5 millionths
    a[100:1100] = ties.astype(np.float32).reshape(1000, 3)
    a[1100:2100] = np.nextafter(ties.astype(np.float32), np.float32(1)).reshape(1000, 3)
    a[2100:3100] = -np.nextafter(ties.astype(np.float32), np.float32(0)).reshape(1000, 3)
    a[3100:4100] = (rng.integers(0, 2 ** 22, (1000, 3)) / 2.0 ** 16).astype(np.float32)   # exact ties of the 6th decimal exist here: k / 65536
    np.savetxt(tmp_path / "ref.xyz", a, fmt="%.6f")
    save_xyz(tmp_path / "got.xyz", a)
    assert (tmp_path / "got.xyz").read_bytes() == (tmp_path / "ref.xyz").read_bytes()
    one = a[:1]
    np.savetxt(tmp_path / "ref1.xyz", one, fmt="%.6f")
    save_xyz(tmp_path / "got1.xyz", one)
    assert (tmp_path / "got1.xyz").read_bytes() == (tmp_path / "ref1.xyz").read_bytes()
    d = a[:50].astype(np.float64) * 1.000000123                # not float32: the pure-Python path
    np.savetxt(tmp_path / "ref2.xyz", d, fmt="%.6f")
    save_xyz(tmp_path / "got2.xyz", d)
    assert (tmp_path / "got2.xyz").read_bytes() == (tmp_path / "ref2.xyz").read_bytes()


def test_load_xyz_reads_what_loadtxt_reads(tmp_path):
    """upsample.py:42 reads with np.loadtxt(dtype=float32); the library's parser (pf_parse_xyz: fast path for plain decimals,
    strtod for the rest) must return the same float32 bits - signs, leading / trailing dots, exponents, long digit strings,
    nan / inf, comments, blank lines - and hand anything it does not understand to numpy."""
    import random
    import numpy as np
    from puflow_amd.upsample import load_xyz
    random.seed(5)
    toks = []
    for _ in range(30000):
        k = random.random()
        digits = lambda a, b: "".join(random.choice("0123456789") for _ in range(random.randint(a, b)))
        if k < 0.5:
            t = "%s%d.%s" % (random.choice(["", "-", "+"]), random.randint(0, 999), digits(0, 14))
        elif k < 0.7:
            t = "%s%d.%s" % (random.choice(["", "-"]), random.randint(0, 9), digits(10, 25))
        elif k < 0.85:
            t = "%.6e" % ((random.random() - 0.5) * 10 ** random.randint(-30, 30))
        elif k < 0.9:
            t = str(random.randint(-10 ** 9, 10 ** 9))
        elif k < 0.95:
            t = "." + digits(1, 9)
        else:
            t = random.choice(["nan", "inf", "-inf", "1e5", "-0.0", "0", "5.", "-.5", "123456789012345", "0.000001"])
        toks.append(t)
    rows = ["\t".join(toks[i:i + 3]) if i % 5 == 0 else "  ".join(toks[i:i + 3]) for i in range(0, len(toks), 3)]
    text = "# header\n" + "\n".join(r + ("   # note" if i % 7 == 0 else "") for i, r in enumerate(rows)) + "\n\n"
    (tmp_path / "a.xyz").write_text(text)
    ref = np.loadtxt(tmp_path / "a.xyz", dtype=np.float32)
    got = load_xyz(tmp_path / "a.xyz")
    assert got.dtype == np.float32 and got.shape == ref.shape
    assert (got.view(np.uint32) == ref.view(np.uint32)).all()                      # bit for bit, -0.0 and NaN included
    (tmp_path / "b.xyz").write_text("1.5 2.5 3.5\n")                              # a single row: 1-d, like loadtxt
    assert load_xyz(tmp_path / "b.xyz").shape == np.loadtxt(tmp_path / "b.xyz", dtype=np.float32).shape == (3,)
    (tmp_path / "c.xyz").write_text("1,2,3\n4,5,6\n")                            # not whitespace-separated: numpy's call
    import pytest
    with pytest.raises(ValueError):
        load_xyz(tmp_path / "c.xyz")
    (tmp_path / "d.xyz").write_text("1 2 3\n4 5\n")                              # ragged: numpy raises
    with pytest.raises(ValueError):
        load_xyz(tmp_path / "d.xyz")


def test_param_fan_sums_the_gradients_of_every_use():
    """train_ops.ParamFanFn hands out aliases of parameters that several autograd nodes use (f and g share the flow blocks'
    parameters) and sums their gradients in its own backward: the result must be what autograd's per-use accumulation gives -
    including a parameter with three uses, an unused alias (no gradient) and in-place initialisation through `.data` after the
    aliases were taken (ActNorm's data-dependent init)."""
    import torch
    from puflow_amd.train_ops import ParamFanFn
    torch.manual_seed(0)
    a = torch.randn(3, 4, requires_grad=True)
    b = torch.randn(5, requires_grad=True)
    c = torch.randn(2, 2, requires_grad=True)
    x = torch.randn(4)

    def loss(a0, a1, a2, b0, b1, c0, c1):
        return (a0 @ x).sum() + (a1 * a1).sum() * 0.5 + (a2.sum() ** 2) + (b0 * 3).sum() + b1.pow(3).sum() + c0.trace()   # c1 unused

    al = ParamFanFn.apply((3, 2, 2), a, b, c)
    assert len(al) == 7 and all(t.shape == p.shape for t, p in zip(al, (a, a, a, b, b, c, c)))
    with torch.no_grad():
        b.data.mul_(2.0)                                    # aliases are views: they see the new values
    loss(*al).backward()
    got = [p.grad.clone() for p in (a, b, c)]
    for p in (a, b, c):
        p.grad = None
    loss(a, a, a, b, b, c, c).backward()
    for g, p in zip(got, (a, b, c)):
        assert torch.allclose(g, p.grad, rtol=1e-6, atol=1e-6)
