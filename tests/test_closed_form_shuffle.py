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


def closed_form_sparse(p, rng):
    """The form phase A runs (blas.hip a_ranks_kernel / a_apply_kernel): only the elements the shuffle swaps write the
    rank -> position tables, and an entry is believed only where it must be this round's.  The front pointer examines the
    originals at [0, pivot] in place and the back pointer the rest, pivot = Ttot - p(u): a position left of Ttot is consumed
    from the left, one right of it from the right, AT Ttot the t_F entry decides, and `u` is within one position of Ttot.
    The tables start as garbage (they are never cleared between rounds) and every position reads its entries, written or not."""
    p = np.asarray(p, bool)
    n = len(p)
    TL = np.concatenate([[0], np.cumsum(p)[:-1]]).astype(np.int64)
    ttot = int(p.sum())
    ftot = n - ttot
    truepos = rng.integers(0, n + 1, n + 2)
    falsepos = rng.integers(0, n + 1, n + 2)
    writes = 0
    for x in range(n):
        if p[x] and x + 1 >= ttot:
            truepos[ttot - TL[x] - 1] = x; writes += 1
        if not p[x] and x <= ttot + 1:
            falsepos[x - TL[x]] = x; writes += 1
    arr = np.full(n, -1, np.int64)
    L = None
    for x in range(n):
        F = x - TL[x]
        T = ttot - TL[x] - int(p[x])
        need_t = F != 0 and F <= ttot
        need_f = T + 1 <= ftot
        tp = truepos[F - 1] if need_t else truepos[0]
        fp = falsepos[T] if need_f else falsepos[0]
        tF = n if F == 0 else (tp if need_t else -1)
        fj = fp if need_f else n
        left = x < ttot or (x == ttot and x < tF)
        amb = ttot - 1 <= x <= ttot + 1
        fetch = x + n - tF if left else (n - 1 - x) + fj + 1
        if amb and fetch == n - 1:
            assert L is None
            dest = L = ttot - int(p[x])
        elif left:
            dest = x if p[x] else tF - 1
        else:
            dest = fj if p[x] else x - 1
        assert 0 <= dest < n and arr[dest] < 0
        arr[dest] = x
    assert L is not None
    return L, arr, writes


def test_sparse_tables_equal_literal_loop(oracle):
    rng = np.random.default_rng(23)
    moved = total = 0
    for it in range(2500):
        n = int(rng.integers(1, 120))
        keys = rng.random(n).astype(np.float32)
        pos = np.float32([0.0, 1.0, 0.125, 0.5, rng.random()][it % 5])       # no trues, all trues, an eighth, half, anything
        piv, ids = oracle.partition_shuffle(keys, np.arange(n), 0, n, pos)
        L, arr, w = closed_form_sparse(keys < pos, rng)
        assert piv == L and np.array_equal(ids, arr)
        moved += w; total += n
    assert moved < 0.6 * total                                # the dense form stores n entries


def test_sparse_tables_every_predicate_up_to_12(oracle):
    rng = np.random.default_rng(29)
    for n in range(1, 13):
        for bits in range(1 << n):
            p = np.array([(bits >> k) & 1 for k in range(n)], bool)
            keys = np.where(p, 0.25, 0.75).astype(np.float32)
            piv, ids = oracle.partition_shuffle(keys, np.arange(n), 0, n, np.float32(0.5))
            L, arr, _ = closed_form_sparse(p, rng)
            assert piv == L and np.array_equal(ids, arr)
