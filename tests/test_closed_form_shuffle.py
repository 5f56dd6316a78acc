"""the data-parallel closed form of `partition_shuffle` (blas.rs:168-182) that
the HIP builder uses, checked against the oracle's literal loop on random predicates.

With p the predicate at each position, TL(x) = #true in [0,x), F = x - TL, T = Ttot - TL - p,
f_j / t_j the position of the j-th false from the left / true from the right (t_0 = n):
x is consumed from the left iff x < t_F; its fetch index is x + n - t_F (left) or
(n-1-x) + f_{T+1} + 1 (right); the element with fetch index n-1 is the never-examined `u`, which
lands on the pivot L = Ttot - p(u); examined trues keep x (left) or go to f_{T+1} (right);
examined falses go to t_F - 1 (left) or x - 1 (right)."""
import numpy as np

from voidin_amd import synth


def closed_form(p):
    p = np.asarray(p, bool)
    n = len(p)
    TL = np.concatenate([[0], np.cumsum(p)[:-1]])
    ttot = int(p.sum())
    x = np.arange(n)
    F = x - TL
    T = ttot - TL - p
    falsepos = np.full(n + 2, n)
    truepos = np.full(n + 2, -1)
    truepos[0] = n
    falsepos[(F + 1)[~p]] = x[~p]
    truepos[(T + 1)[p]] = x[p]
    tF = truepos[np.minimum(F, n + 1)]
    left = x < tF
    fj = falsepos[np.minimum(T + 1, n + 1)]
    fetch = np.where(left, x + n - tF, (n - 1 - x) + fj + 1)
    is_u = fetch == n - 1
    assert is_u.sum() == 1
    L = ttot - int(p[is_u][0])
    dest = np.where(left, np.where(p, x, tF - 1), np.where(p, fj, x - 1))
    dest = np.where(is_u, L, dest)
    arr = np.empty(n, np.int64)
    arr[dest] = x
    return L, arr


def test_closed_form_equals_literal_loop(oracle):
    rng = np.random.default_rng(7)
    for _ in range(3000):
        n = int(rng.integers(1, 90))
        keys = rng.random(n).astype(np.float32)
        pos = np.float32(rng.random())
        piv, ids = oracle.partition_shuffle(keys, np.arange(n), 0, n, pos)
        L, arr = closed_form(keys < pos)
        assert piv == L and np.array_equal(ids, arr)


def test_unexamined_element_lands_right_even_when_true(oracle):
    # ~half of random calls: the last-fetched element satisfies the predicate yet ends on the right
    rng = np.random.default_rng(11)
    hits = 0
    for _ in range(500):
        n = int(rng.integers(2, 40))
        keys = rng.random(n).astype(np.float32)
        piv, ids = oracle.partition_shuffle(keys, np.arange(n), 0, n, np.float32(0.5))
        hits += bool(keys[ids[piv]] < 0.5)
        assert (keys[ids[:piv]] < 0.5).all()
    assert 100 < hits < 400
