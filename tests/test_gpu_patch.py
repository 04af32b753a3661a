"""Patch pipeline operators + PatchHelper + the CLI on the GPU, against oracle/patch_ref.py."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import patch_ref as P
from oracle import ref_cpu as O
from puflow_amd.weights import synth_patches, synth_state_dict

DEV = "cuda:0"


@pytest.mark.parametrize("B,N,npoint", [(2, 2048, 32), (1, 5000, 78), (3, 700, 700), (1, 10240, 2054),
                                        (3, 40960, 600), (1, 99840, 300), (2, 200001, 64), (9, 8192, 100)])
def test_fps_bit_exact(B, N, npoint):
    """N >= 8192 takes the cooperative multi-workgroup kernel (1, 4 or 8 points per thread; several clouds at once)."""
    from puflow_amd import ops
    xyz = synth_patches(B, N, seed=N, surface=(N % 2 == 0))
    if N == 700 or N == 40960:
        xyz[0, 5] = xyz[0, 9]                          # duplicates: ties resolve to the first maximum
        xyz[0, N - 1] = xyz[0, 3]
    ref = P.fps(xyz, npoint)
    got = ops.furthest_point_sample(xyz.to(DEV), npoint)
    assert torch.equal(got.cpu().long(), ref)
    assert got[:, 0].eq(0).all()


@pytest.mark.parametrize("group,N,npoint", [(768, 30720, 3001), (1280, 99840, 1500), (1280, 20480, 2048), (1536, 36864, 777), (640, 20480, 500)])
def test_fps_group_hint_changes_no_index(group, N, npoint):
    """pf_fps_grouped: the layout hint of the merge (every `group` consecutive points are one patch) gives a wave of the
    cooperative kernel exactly one patch (12 / 20 / 24 points per thread) - other points per thread, other workgroups per cloud,
    the same samples bit for bit: against the oracle, on patch-ordered clouds with duplicates and on clouds in no order; a group
    the kernel has no shape for (640) is ignored."""
    from puflow_amd import ops
    g = torch.Generator().manual_seed(group + N)
    base = torch.nn.functional.normalize(torch.randn(2, 2000, 3, generator=g), dim=-1)
    npatch = N // group
    seeds = base[:, torch.randperm(2000, generator=g)[:npatch]]
    nn = torch.cdist(seeds, base).topk(128, largest=False).indices
    pts = torch.gather(base.unsqueeze(1).expand(2, npatch, 2000, 3), 2, nn.unsqueeze(-1).expand(2, npatch, 128, 3))
    rep = -(-group // 128)
    pts = (pts.repeat_interleave(rep, dim=2)[:, :, :group] + 0.01 * torch.randn(2, npatch, group, 3, generator=g))
    xyz = pts.reshape(2, N, 3).contiguous()
    xyz[0, 5] = xyz[0, 9]                              # duplicates: ties resolve to the first maximum
    xyz[1] = xyz[1, torch.randperm(N, generator=g)]    # second cloud: the same kind of points in no order (the hint is then wrong, not harmful)
    ref = P.fps(xyz, npoint)
    plain = ops.furthest_point_sample(xyz.to(DEV), npoint)
    hinted = ops.furthest_point_sample(xyz.to(DEV), npoint, group=group)
    assert torch.equal(plain.cpu().long(), ref)
    assert torch.equal(hinted, plain)


@pytest.mark.parametrize("sites,N,npoint", [(3, 8736, 617), (5, 11684, 2428), (9, 10000, 900), (2, 9000, 40)])
def test_fps_on_lattices_and_past_the_last_distinct_point(sites, N, npoint):
    """Cooperative FPS on lattice clouds: thousands of exact ties, and (first two cases) more samples than distinct points -
    once every min-distance is 0 the sequential algorithm keeps returning the smallest index.  The two-samples-per-exchange
    kernel once emitted (0, 1, 0, 1, ...) there: after c1 is added its own key stays the largest when everything is 0, so a
    second sample needs md > 0 (found by tools/stress_exact.py)."""
    from puflow_amd import ops
    g = torch.Generator().manual_seed(sites * 1000 + N)
    xyz = torch.round((torch.rand(2, N, 3, generator=g) * 2 - 1) * sites) / sites
    ref = P.fps(xyz, npoint)
    got = ops.furthest_point_sample(xyz.to(DEV), npoint)
    assert torch.equal(got.cpu().long(), ref)


def test_fps_matches_reference_torch_fps(golden_dir):
    """PINNED: `pf_fps` (single-workgroup and cooperative kernels) against the reference's own in-tree torch FPS
    (modules/utils/fps.py; tools/make_golden_patch.py), and PatchHelper.merge_pc (patch.py:162-165) built on it."""
    from puflow_amd import ops
    from puflow_amd.patch import PatchHelper
    g = np.load(os.path.join(golden_dir, "fps_ref.npz"))
    for tag in "abcde":
        xyz, ref = torch.from_numpy(g[f"{tag}_xyz"]), torch.from_numpy(g[f"{tag}_idx"])
        got = ops.furthest_point_sample(xyz.to(DEV), ref.shape[1])
        assert torch.equal(got.cpu().long(), ref), tag
    out = PatchHelper.merge_pc(torch.from_numpy(g["m_a"]).to(DEV), torch.from_numpy(g["m_b"]).to(DEV), 128)
    assert torch.equal(out.cpu(), torch.from_numpy(g["m_out"]))


@pytest.mark.parametrize("B,N,M,K", [(2, 2048, 32, 256), (1, 5000, 78, 256), (1, 300, 7, 300), (2, 1000, 5, 64)])
def test_knn_large_and_KNN_surface(B, N, M, K):
    from puflow_amd import ops
    ref_pts = synth_patches(B, N, seed=N + 1, surface=False)
    qry = ref_pts[:, :M].clone() + 0.01
    d_ref, i_ref = O.knn_canonical(qry, ref_pts, K)
    dist, idx = ops.KNN(k=K, transpose_mode=False)(ref_pts.transpose(1, 2).contiguous().to(DEV), qry.transpose(1, 2).contiguous().to(DEV))
    assert tuple(idx.shape) == (B, K, M) and idx.dtype == torch.int64
    assert torch.equal(idx.cpu().transpose(1, 2), i_ref) and torch.equal(dist.cpu().transpose(1, 2), d_ref)


def test_patch_helper_stages():
    from puflow_amd.patch import PatchHelper
    pc = synth_patches(2, 2048, seed=77)
    pc_n, c, fd = P.normalize_pc(pc)
    ph = PatchHelper(256, 4)
    got_n, gc, gfd = PatchHelper.normalize_pc(pc.to(DEV))
    np.testing.assert_allclose(got_n.cpu().numpy(), pc_n.numpy(), atol=1e-6)
    patches = PatchHelper.extract_knn_patch(pc_n.to(DEV), ph.knn, 256, 4)
    ref = P.extract_knn_patch(pc_n, 256, 4)
    assert tuple(patches.shape) == (2, 32, 256, 3) and torch.equal(patches.cpu(), ref)
    # merge: FPS over 32 x 1280 candidates down to 8216, then outlier removal
    cand = synth_patches(1, 32 * 1280, seed=78, surface=False).view(1, 32, 1280, 3)
    merged = PatchHelper.merge_patches(cand.to(DEV), 4120)                     # [B,3,4120]
    ref_m = P.merge_patches(cand, 4120)
    assert torch.equal(merged.transpose(1, 2).cpu(), ref_m)
    out = PatchHelper.remove_outliers(merged.transpose(1, 2).contiguous(), pc[:1].to(DEV), 24)
    ref_o = P.remove_outliers(ref_m, pc[:1], 24)
    assert tuple(out.shape) == (1, 4096, 3) and torch.equal(out.cpu(), ref_o)


def test_cli_end_to_end(tmp_path):
    """upsample CLI: .xyz in -> .xyz out (N*4 points, '%.6f'), network = HIP path; checked against the oracle pipeline."""
    from puflow_amd import upsample as U
    from puflow_amd.patch import PatchHelper
    src, dst = tmp_path / "in", tmp_path / "out"
    src.mkdir(); dst.mkdir()
    pc = synth_patches(1, 1024, seed=5)[0] * 3.0 + 1.5
    np.savetxt(src / "cloud.xyz", pc.numpy(), fmt="%.6f")
    sd = synth_state_dict(9)
    U.upsampling([str(src / "cloud.xyz")], str(dst), None, up_ratio=4, num_outlier=24, num_patch=256, seed=2021, state_dict=sd)
    out = np.loadtxt(dst / "cloud.xyz", dtype=np.float32)
    assert out.shape == (4096, 3) and np.isfinite(out).all()
    # same pipeline with the oracle network on CPU (same permutation: same seed)
    from puflow_amd.interpflow import PointInterpFlow
    np.random.seed(2021); torch.random.manual_seed(2021)
    PointInterpFlow(3)                                   # the CLI builds the network after seeding: same RNG consumption
    x = torch.from_numpy(np.loadtxt(src / "cloud.xyz", dtype=np.float32)).unsqueeze(0)
    x = x[:, torch.randperm(x.shape[1])].contiguous()
    xn, c, fd = P.normalize_pc(x)
    patches = P.extract_knn_patch(xn, 256, 4).reshape(16, 256, 3)
    pn, pc_, pfd = P.normalize_pc(patches)
    pred, _ = O.forward(sd, pn, 4)
    full = (torch.cat([pred, pn], 1) * pfd + pc_).reshape(1, 16, 1280, 3)
    cand = (full.reshape(1, -1, 3) * fd + c)[0].numpy()                          # 20 480 de-normalised candidates
    ref = P.remove_outliers(P.merge_patches(full, 4120) * fd + c, x, 24)[0].numpy()
    # The FPS merge is chaotic (one argmax flip re-seeds the whole selection), so the two 8192-point SELECTIONS
    # differ; what must hold: every point the CLI wrote is one of the oracle's candidates (network + de-normalise
    # parity, scale 3 -> 3e-5), and both selections cover the surface equally well.
    d = ((out[:, None, :] - cand[None, :, :]) ** 2).sum(-1).min(1)
    assert np.sqrt(d).max() < 3e-5
    xin = x[0].numpy()
    cd_out = float(O.chamfer_distance_mean(torch.from_numpy(out)[None], torch.from_numpy(xin)[None]))
    cd_ref = float(O.chamfer_distance_mean(torch.from_numpy(ref)[None], torch.from_numpy(xin)[None]))
    assert abs(cd_out - cd_ref) < 0.02 * cd_ref


def test_cli_batches_files_without_changing_any_cloud(tmp_path):
    """The CLI passes consecutive files of equal point count through the pipeline together (`cloud_batch`): every output
    file must be byte for byte what the one-file-at-a-time run (the reference's loop, upsample.py:42-57) writes, whatever
    the grouping - equal sizes batched, a different size in between, a batch limit that splits a run."""
    from puflow_amd import upsample as U
    src = tmp_path / "in"
    src.mkdir()
    sizes = {"a.xyz": 1024, "b.xyz": 1024, "c.xyz": 768, "d.xyz": 1024, "e.xyz": 1024, "f.xyz": 1024}
    for k, (name, n) in enumerate(sizes.items()):
        np.savetxt(src / name, (synth_patches(1, n, seed=40 + k)[0] * 2.0 - 0.5).numpy(), fmt="%.6f")
    sd = synth_state_dict(9)
    paths = [str(src / name) for name in sizes]
    outs = {}
    for cb in (1, 2, 8):
        dst = tmp_path / f"out{cb}"
        dst.mkdir()
        U.upsampling(paths, str(dst), None, up_ratio=4, num_outlier=24, num_patch=256, seed=2021, state_dict=sd, cloud_batch=cb)
        outs[cb] = {name: (dst / name).read_bytes() for name in sizes}
        assert all(len(v) > 0 for v in outs[cb].values())
    assert outs[2] == outs[1] and outs[8] == outs[1]
    assert np.loadtxt(tmp_path / "out8" / "c.xyz").shape == (768 * 4, 3)


def test_cli_sharding_over_ranks_does_not_change_any_cloud(tmp_path, monkeypatch):
    """Under torchrun the CLI shards the (sorted) file list over the ranks.  A file's shuffle comes from the process RNG in
    file order (upsample.py:43), so every rank replays the draws of the files it skips: each file must come out byte for byte
    as in the one-process run - here ranks 0 and 1 of 2 are run one after the other through the environment variables the
    CLI reads."""
    from puflow_amd import upsample as U
    src = tmp_path / "in"
    src.mkdir()
    names = ["b.xyz", "a.xyz", "d.xyz", "c.xyz", "e.xyz"]                     # created out of order: the list is sorted
    for k, name in enumerate(names):
        np.savetxt(src / name, (synth_patches(1, 1024 if k != 3 else 768, seed=60 + k)[0] * 1.5).numpy(), fmt="%.6f")
    sd = synth_state_dict(9)
    paths = [str(src / n) for n in names]

    def run(tag, rank, world):
        dst = tmp_path / tag
        dst.mkdir(exist_ok=True)
        monkeypatch.setenv("RANK", str(rank)); monkeypatch.setenv("WORLD_SIZE", str(world))
        U.upsampling(paths, str(dst), None, up_ratio=4, num_outlier=24, num_patch=256, seed=2021, state_dict=sd)
        return {p.name: p.read_bytes() for p in dst.iterdir()}

    one = run("one", 0, 1)
    r0, r1 = run("two", 0, 2), run("two", 1, 2)                                # both ranks write into the same directory
    assert set(one) == set(names) and set(r1) == set(names)
    assert r1 == one


def test_knn_large_streams_clouds_beyond_the_lds_limit():
    """knn_cuda.KNN takes any N (patch.py:33,107): clouds of more than 16 384 points stream through LDS in chunks; the
    result is still the exact (distance, index)-ordered top K."""
    from puflow_amd import ops
    N, M, K = 100000, 24, 256
    ref_pts = synth_patches(1, N, seed=4242, surface=True)
    ref_pts[0, 70000] = ref_pts[0, 11]                         # duplicate far apart in index: tie broken by index
    qry = ref_pts[:, [0, 11, 5000, 16383, 16384, 16385, 32768, 70000, 99999] + list(range(100, 115))].clone()
    d_ref, i_ref = O.knn_canonical(qry, ref_pts, K)
    dist, idx = ops.KNN(k=K, transpose_mode=True)(ref_pts.to(DEV), qry.to(DEV))
    assert torch.equal(idx.cpu(), i_ref) and torch.equal(dist.cpu(), d_ref)
    # K above the chunked path's limit is refused loudly, not truncated
    from puflow_amd._lib import PuflowHipError
    with pytest.raises(PuflowHipError):
        ops.KNN(k=9000, transpose_mode=True)(ref_pts.to(DEV), qry.to(DEV))


def test_fps_abort_word_is_reported():
    """A cooperative-FPS cloud whose workgroups timed out sets its abort word: the wrapper must raise, not hand out
    the (invalid) index row."""
    from puflow_amd import _lib, ops
    lib = _lib.load()
    B, N = 3, 10240
    mind = torch.zeros((B, N), dtype=torch.float32, device=DEV)
    ops._check_fps_abort(lib, mind, B, N)                        # clean: no exception
    import ctypes
    stride, word = ctypes.c_longlong(0), ctypes.c_longlong(0)
    assert lib.pf_fps_scratch_layout(N, ctypes.byref(stride), ctypes.byref(word)) == 1
    for status in (1, 2):                                        # 1 = gave up waiting, 2 = never finished (initial value)
        mind.zero_()
        mind.view(-1).view(torch.int64)[2 * stride.value + word.value] = status
        with pytest.raises(_lib.PuflowHipError):
            ops._check_fps_abort(lib, mind, B, N)
    # a real run leaves every cloud's status at 0 ("all steps completed")
    pts = synth_patches(B, N, seed=77).to(DEV)
    idx = torch.zeros((B, 64), dtype=torch.int32, device=DEV)
    _lib.check(lib.pf_fps(pts.data_ptr(), B, N, 64, mind.data_ptr(), idx.data_ptr(), None), "pf_fps")
    torch.cuda.synchronize()
    words = mind.view(-1).view(torch.int64)
    assert all(int(words[b * stride.value + word.value]) == 0 for b in range(B))
    assert lib.pf_fps_scratch_layout(2048, None, None) == 0      # single-workgroup kernel: no ring to check
    ops._check_fps_abort(lib, mind[:, :2048].contiguous(), B, 2048)


def test_pugan_5000_to_20000_pipeline():
    """BASELINE configs[3]: one 5000-point cloud -> int(5000 / 256 * 4) = 78 patches of 256 -> 78 x 1280 = 99 840
    candidates -> FPS 20 024 -> drop 24 outliers -> 20 000 points (patch.py:35-80,101; upsample.py:46-56).
    Stage by stage against the oracle: patch extraction bit-exact, the network within 1e-5 of the CPU oracle on every
    candidate, and - GIVEN the HIP candidates - the merge and the outlier removal bit-exact (the FPS merge is chaotic
    in its input, so the final selection is compared on identical candidates, and end to end as coverage)."""
    from puflow_amd.interpflow import PointInterpFlow
    from puflow_amd.patch import PatchHelper
    sd = synth_state_dict(31)
    net = PointInterpFlow(3); net.load_state_dict(sd); net.set_to_initialized_state(); net = net.to(DEV).eval()
    pc = synth_patches(1, 5000, seed=50) * 2.0 + 0.7
    ph = PatchHelper(256, 4)
    pcn, gc, gfd = PatchHelper.normalize_pc(pc.to(DEV))
    patches = PatchHelper.extract_knn_patch(pcn, ph.knn, 256, 4)
    assert tuple(patches.shape) == (1, 78, 256, 3)
    assert torch.equal(patches.cpu(), P.extract_knn_patch(pcn.cpu(), 256, 4))
    cand = PatchHelper.upsampling_patches(net, patches, 4)                          # [1, 78, 1280, 3]
    assert tuple(cand.shape) == (1, 78, 1280, 3)
    pn, pcen, pfd = P.normalize_pc(patches.cpu().reshape(78, 256, 3))
    pred_ref, _ = O.forward(sd, pn, 4)
    cand_ref = (torch.cat([pred_ref, pn], 1) * pfd + pcen).reshape(1, 78, 1280, 3)
    assert (cand.cpu() - cand_ref).abs().max() < 1e-5
    merged = PatchHelper.merge_patches(cand, 20024)                                 # [1, 3, 20024]
    ref_m = P.merge_patches(cand.cpu(), 20024)
    assert torch.equal(merged.transpose(1, 2).cpu(), ref_m)
    den = (merged * gfd + gc.transpose(1, 2)).transpose(1, 2).contiguous()
    out = PatchHelper.remove_outliers(den, pc.to(DEV), 24)
    ref_o = P.remove_outliers(den.cpu(), pc, 24)
    assert tuple(out.shape) == (1, 20000, 3) and torch.equal(out.cpu(), ref_o)
    # the one-call pipeline gives the same cloud
    e2e = PatchHelper.remove_outliers(ph.upsample(net, pc.to(DEV), npoint=20024, upratio=4), pc.to(DEV), 24)
    assert torch.equal(e2e, out)
    # coverage, end to end against the all-oracle pipeline (its own candidates, its own selection)
    cd_out = float(O.chamfer_distance_mean(out.cpu(), pc))
    ref_full = P.remove_outliers(P.merge_patches(cand_ref, 20024) * gfd.cpu() + gc.cpu(), pc, 24)
    cd_ref = float(O.chamfer_distance_mean(ref_full, pc))
    assert abs(cd_out - cd_ref) < 0.02 * cd_ref
