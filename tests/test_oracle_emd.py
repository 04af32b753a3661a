"""CPU checks of the auction-EMD oracle (oracle/emd_ref.py): the reference holds no expected values for
this operator, so the restatement is validated by the properties SURVEY.md Appendix B lists."""
import numpy as np
from scipy.optimize import linear_sum_assignment

from oracle import emd_ref as E


def _cube(n, seed):
    return np.random.default_rng(seed).random((n, 3), dtype=np.float32)


def test_self_consistency_and_bijection_rate():
    x, y = _cube(256, 0), _cube(256, 1)
    d50, a50 = E.auction_one(x, y, 0.005, 50)
    np.testing.assert_allclose(d50, ((x - y[a50]) ** 2).sum(-1), rtol=1e-5, atol=1e-8)
    _, a3 = E.auction_one(x, y, 0.005, 3)
    _, a500 = E.auction_one(x, y, 0.005, 500)
    u3, u50, u500 = (len(np.unique(a)) for a in (a3, a50, a500))
    assert u3 <= u50 <= u500 and u500 >= 250


def test_near_optimal_for_small_eps():
    n = 64
    x, y = _cube(n, 2), _cube(n, 3)
    cost = np.sqrt(((x[:, None] - y[None]) ** 2).sum(-1))
    r, c = linear_sum_assignment(cost)
    d, a = E.auction_one(x, y, 0.0005, 5000)
    assert len(np.unique(a)) == n
    assert np.sqrt(d).sum() <= cost[r, c].sum() + n * 0.0005 + 1e-4


def test_backward_formula():
    x, y = _cube(32, 4)[None], _cube(32, 5)[None]
    d, a = E.emd_forward(x, y, 0.005, 50)
    g = np.ones((1, 32), np.float32)
    gx = E.emd_backward(x, y, g, a)
    h = 1e-3
    xp = x.copy(); xp[0, 7, 1] += h
    num = (((xp[0] - y[0][a[0]]) ** 2).sum() - ((x[0] - y[0][a[0]]) ** 2).sum()) / h
    assert abs(num - gx[0, 7, 1]) < 5e-3
