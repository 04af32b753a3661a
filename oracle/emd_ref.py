"""CPU oracle for the auction-algorithm EMD.  TEST INFRASTRUCTURE ONLY (see oracle/ref_cpu.py).

Restates the reference's CUDA module from its text - `metric/emd/emd_cuda.cu:23-282`
(kernels clear / calc_unass_* / Bid / GetMax / Assign / CalcDist), the Python wrapper
`metric/emd/emd_module.py:31-72` and `metric/loss.py:18-29` - in numpy.  The CUDA module has no CPU
path and cannot be built here (no nvcc), and the reference holds no expected values for it
(its `test_emd`, emd_module.py:81-98, only re-derives the distance from the returned
assignment): **parity unpinned** for this operator.  Checks that ARE available and are used
in tests/: self-consistency (dist == |x - y[assignment]|^2), bijection rate rising with
iterations, closeness to the optimal assignment (scipy linear_sum_assignment) for small eps.

Determinism rules chosen where the reference is racy:
  * Bid: best object = smallest k attaining the maximum value (strict `>` scan, cu:147-155);
    `better` = second largest value counting duplicates.
  * GetMax: the reference lets ANY bidder whose increment is within 1e-6 of the maximum win
    (last writer, cu:188-191); here the winner of object o is the bidder with the largest
    increment, largest index among exact ties.
  * non-finite predictions: when no object has a value above the -1e9 start (a NaN / inf prediction row) the point bids on
    the object with its own index at the minimum increment eps (the reference would index best_i = -1, cu:133,176-178);
    the NaN then shows in dist and the loss.
  * value arithmetic: `3.0 - sqrtf(.) - price` is evaluated in double (the literal 3.0 is a
    double in cu:146) and rounded to float once; squared distances are unfused fp32.
"""
from __future__ import annotations

import numpy as np

f32 = np.float32


def _sq_unfused(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """[U,1,3] - [1,n,3] style broadcast -> unfused fp32 ((dx*dx)+(dy*dy))+(dz*dz)."""
    d = (a - b).astype(f32)
    dx, dy, dz = d[..., 0], d[..., 1], d[..., 2]
    return ((dx * dx).astype(f32) + (dy * dy).astype(f32)).astype(f32) + (dz * dz).astype(f32)


def auction_one(x: np.ndarray, y: np.ndarray, eps: float, iters: int):
    """x (prediction), y (ground truth): [n,3] fp32.  -> (dist [n] fp32, assignment [n] int32)."""
    n = x.shape[0]
    x = x.astype(f32); y = y.astype(f32)
    eps = f32(eps)
    assignment = np.full(n, -1, np.int32)
    assignment_inv = np.full(n, -1, np.int32)
    price = np.zeros(n, f32)
    for it in range(iters):
        last = it == iters - 1
        U = np.nonzero(assignment == -1)[0]
        if U.size == 0:
            break                                                   # cu:105-106 (sample skipped from now on)
        s = _sq_unfused(y[None, :, :], x[U][:, None, :])           # y_k - x_i  (cu:141-143)
        v = ((3.0 - np.sqrt(s).astype(np.float64)) - price[None, :].astype(np.float64)).astype(f32)
        best_i = np.argmax(v, axis=1)                               # first maximum = smallest k
        best = v[np.arange(U.size), best_i]
        v2 = v.copy()
        v2[np.arange(U.size), best_i] = f32(-1e9)
        better = np.maximum(v2.max(axis=1), f32(-1e9)) if n > 1 else np.full(U.size, f32(-1e9))
        with np.errstate(invalid="ignore"):
            none = ~(v > f32(-1e9)).any(axis=1)                      # NaN / inf prediction row: own index, minimum increment
        best_i = np.where(none, U, best_i)
        best = np.where(none, f32(0), best); better = np.where(none, f32(0), better)
        inc = ((best - better).astype(f32) + eps).astype(f32)       # cu:175
        # winner per object: largest (inc, i)
        order = np.lexsort((U, inc))                                # ascending by inc then by i
        winner = {}
        for pos in order:
            winner[int(best_i[pos])] = pos                          # later (larger) overrides
        for pos in range(U.size):
            i, o = int(U[pos]), int(best_i[pos])
            if last or winner[o] == pos:
                if not last and assignment_inv[o] != -1:
                    assignment[assignment_inv[o]] = -1              # evict (cu:204-206)
                assignment_inv[o] = i
                assignment[i] = o
                price[o] = f32(price[o] + inc[pos])
    d = (x - y[assignment]).astype(f32)
    dist = ((d[:, 0] * d[:, 0]).astype(f32) + (d[:, 1] * d[:, 1]).astype(f32)).astype(f32) + (d[:, 2] * d[:, 2]).astype(f32)
    return dist.astype(f32), assignment


def emd_forward(xyz1: np.ndarray, xyz2: np.ndarray, eps: float, iters: int):
    """[B,n,3] x2 -> dist [B,n], assignment [B,n]  (emd_module.py:33-61)."""
    out = [auction_one(xyz1[b], xyz2[b], eps, iters) for b in range(xyz1.shape[0])]
    return np.stack([o[0] for o in out]), np.stack([o[1] for o in out])


def emd_backward(xyz1: np.ndarray, xyz2: np.ndarray, graddist: np.ndarray, assignment: np.ndarray) -> np.ndarray:
    """grad wrt xyz1 only: 2 g (x - y[assignment])  (cu:284-300; grad wrt xyz2 is zero, emd_module.py:68-72)."""
    B = xyz1.shape[0]
    ya = np.stack([xyz2[b][assignment[b]] for b in range(B)])
    g = (graddist.astype(f32) * f32(2))[..., None]
    return (g * (xyz1.astype(f32) - ya)).astype(f32)


def earth_mover_distance(preds: np.ndarray, gts: np.ndarray, eps: float = 0.005, iters: int = 50, radius=None) -> float:
    """EarthMoverDistance.forward (metric/loss.py:25-29): sum over batch and points of dist (/radius)."""
    dist, _ = emd_forward(preds, gts, eps, iters)
    if radius is not None:
        dist = dist / np.asarray(radius, f32).reshape(-1, 1)
    return float(dist.sum(dtype=np.float64))
