"""Build libpuflow_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libpuflow_hip.so")
SOURCES = ["api.hip", "knn.hip", "edgeconv.hip", "pointwise.hip", "flow.hip", "interp.hip", "chamfer.hip", "emd.hip", "train_ops.hip", "train_fused.hip", "train_mlp.hip", "train_flow.hip", "train_flowchain.hip", "train_glue.hip", "optim.hip", "patch_ops.hip", "cnf.hip", "xyz_io.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=on", "-Wno-unused-result"]
# The fused MFMA kernels never see NaNs; without the flag every fmaxf() is preceded by a canonicalising v_max x,x and
# the DPP row-max steps stay as v_mov_dpp + v_max instead of one v_max_f32_dpp (3x the instructions of a max-pool).
# No reassociation is enabled; the exact-order kernels (kNN, Chamfer, EMD, FPS, training ops) keep default semantics.
# -amdgpu-mfma-vgpr-form: keep MFMA accumulators in VGPRs; the AGPR form hipcc picks under pressure costs one
# v_accvgpr_read per accumulator register before any VALU use (every layer here) and halves the occupancy.
EXTRA_FLAGS = {s: ["-fno-honor-nans", "-mllvm", "-amdgpu-mfma-vgpr-form=" + os.environ.get("PF_MFMA_VGPR_FORM", "1")] for s in ("edgeconv.hip", "pointwise.hip", "flow.hip", "interp.hip", "cnf.hip")}
# interp_kernel<1,12> sits exactly on the 168-VGPR budget of three waves per SIMD; the scheduler's default register-pressure
# tracker overshoots it by four registers (20 bytes of scratch per lane, the only spilling kernel of the eval path), the GCN
# trackers do not (tools/check_resources.py: 168 VGPRs, no scratch)
# edgeconv.hip: three LDS weight fragments in flight instead of two (pf_mfma.h PfW2Lds::DEPTH; same VGPR count, same bits):
# +0.3 - 0.7 % on the headline step in two same-box A/Bs (depth 4 the same)
EXTRA_FLAGS["edgeconv.hip"] = EXTRA_FLAGS["edgeconv.hip"] + ["-DPF_W2LDS_DEPTH=" + os.environ.get("PF_EC_W2LDS_DEPTH", "3")]
if os.environ.get("PF_INTERP_TRACKERS", "1") != "0":
    EXTRA_FLAGS["interp.hip"] = EXTRA_FLAGS["interp.hip"] + ["-mllvm", "-amdgpu-use-amdgpu-trackers=1"]


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


# sources whose arithmetic the reduced-precision throughput build changes (they include pf_mfma.h's split-fp16 products)
F16_SOURCES = ("edgeconv.hip", "pointwise.hip", "flow.hip", "interp.hip")
LIB_F16 = LIB.replace(".so", "_f16.so")


def build(force: bool = False, verbose: bool = True, defines=(), tag: str = "", only=None) -> str:
    """defines/tag: build a variant `libpuflow_hip_<tag>.so` with extra -D flags (tools/tune_*.py, build_f16()).
    only: recompile just these sources with the flags and link them with the main build's objects for the rest."""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    lib = LIB if not tag else LIB.replace(".so", f"_{tag}.so")
    srcs = [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    hdrs = [os.path.join(CSRC, h) for h in os.listdir(CSRC) if h.endswith(".h")]
    hdrs.append(os.path.join(os.path.dirname(HERE), "include", "puflow_hip.h"))
    objdir = os.path.join(HERE, "build" + (f"_{tag}" if tag else ""))
    maindir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    objs = []
    procs = []
    for s in srcs:
        base = os.path.basename(s)
        if only is not None and base not in only:
            objs.append(os.path.join(maindir, base.replace(".hip", ".o")))       # shared with the main build
            continue
        o = os.path.join(objdir, base.replace(".hip", ".o"))
        objs.append(o)
        if force or _stale(o, [s] + hdrs):
            cmd = [hipcc] + FLAGS + EXTRA_FLAGS.get(os.path.basename(s), []) + [f"-D{d}" for d in defines] + ["-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((s, subprocess.Popen(cmd)))
    for s, p in procs:
        if p.wait() != 0:
            raise RuntimeError(f"hipcc failed on {s}")
    if force or procs or _stale(lib, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    global _REFRESHING
    if not tag and not defines and not _REFRESHING:
        _REFRESHING = True
        try:
            _refresh_variants(lib, verbose)
        finally:
            _REFRESHING = False
    return lib


_REFRESHING = False


def _refresh_variants(lib: str, verbose: bool) -> None:
    """The variant libraries the bench lines start children on link the main build's objects for every source they do not
    recompile: one that exists but is older than the library just built lacks whatever that build added (a new exported symbol
    makes _lib.load() refuse it - bench.py's grad_parity then reports `failed`) - keep them in step."""
    pkg = os.path.dirname(lib)
    if os.path.exists(os.path.join(pkg, "libpuflow_hip_f16.so")) and _stale(os.path.join(pkg, "libpuflow_hip_f16.so"), [lib]):
        build_f16(verbose=verbose)
    if any(os.path.exists(os.path.join(pkg, f"libpuflow_hip_{t}.so")) and _stale(os.path.join(pkg, f"libpuflow_hip_{t}.so"), [lib])
           for t in ("bwdf32", "gradf32")):
        build_gradf32(verbose=verbose)


def build_f16(force: bool = False, verbose: bool = True) -> str:
    """The reduced-precision throughput library: one fp16 product per 32-channel step instead of three (PF_MMN_TERMS=1)."""
    build(force=False, verbose=verbose)                                   # the objects it shares
    return build(force=force, verbose=verbose, defines=["PF_MMN_TERMS=1"], tag="f16", only=F16_SOURCES)


LIB_GRADF32 = LIB.replace(".so", "_gradf32.so")
GRADF32_DEFINES = ["PF_EC_BWDG_F32", "PF_EC_DW_F32", "PF_EC_FWD_F32"]
LIB_BWDF32 = LIB.replace(".so", "_bwdf32.so")
BWDF32_DEFINES = ["PF_EC_BWDG_F32", "PF_EC_DW_F32"]


def build_gradf32(force: bool = False, verbose: bool = True) -> str:
    """The A/B reference of the training step's arithmetic: the same kernels with plain f32 MFMA products where the default
    build multiplies split-bf16 (EdgeConv backward, weight gradients) / split-fp16 (conv_out forward) operands.  bench.py
    --mode train loads it in a child process and reports every parameter gradient's distance to it (`grad_parity`)."""
    build(force=False, verbose=verbose)
    # the same with the FORWARD left as in the default build: isolates the backward arithmetic (a forward that differs by 1e-6
    # already moves ill-conditioned gradients by 1e-3 through max-pool routes and the flow's conditioning)
    build(force=force, verbose=verbose, defines=BWDF32_DEFINES, tag="bwdf32", only=("train_fused.hip",))
    return build(force=force, verbose=verbose, defines=GRADF32_DEFINES, tag="gradf32", only=("train_fused.hip",))


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    build_f16(force="--force" in sys.argv)
    build_gradf32(force="--force" in sys.argv)
    print(LIB)
    print(LIB_F16)
    print(LIB_GRADF32)
