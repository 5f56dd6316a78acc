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


def fused_round_model(p_seg, act, item):
    """The one-launch round (blas.hip a_round_kernel), restated workgroup by workgroup: a segment of len(p_seg) positions in
    items of `item`, the shuffled window [act, n_seg); an item's workgroup sees only (1) its own predicates, (2) the scanned
    per-item true counts, (3) the predicates of the partner items it decides to read, and fills two small tables with the
    positions of the trues / falses whose ranks its own elements ask for.  Returns (pivot, arrangement) of the window."""
    p_seg = np.asarray(p_seg, bool)
    N = len(p_seg)
    n = N - act
    ni = (N + item - 1) // item
    inwin = np.arange(N) >= act
    cnt = np.array([int((p_seg[k * item:(k + 1) * item] & inwin[k * item:(k + 1) * item]).sum()) for k in range(ni)])
    incl = np.cumsum(cnt)
    ttot = int(incl[-1]) if ni else 0
    ftot = n - ttot
    arr = np.full(n, -1, np.int64)
    L = None
    for k in range(ni):
        p0, p1 = k * item, min(N, (k + 1) * item)
        if p1 <= act:
            continue                                                   # wholly frozen
        run0 = int(incl[k - 1]) if k else 0
        el = []
        tl = run0
        q_lo = T_lo = 1 << 60
        q_hi = T_hi = -1
        for pos in range(max(p0, act), p1):
            x, p = pos - act, bool(p_seg[pos])
            F, T = x - tl, ttot - tl - int(p)
            need_t = F != 0 and F <= ttot and x <= ttot and (not p or x + 1 >= ttot)
            need_f = T + 1 <= ftot and x >= ttot and (p or x <= ttot + 1)
            if need_t:
                q_lo, q_hi = min(q_lo, ttot - F), max(q_hi, ttot - F)
            if need_f:
                T_lo, T_hi = min(T_lo, T), max(T_hi, T)
            el.append((x, p, F, T, need_t, need_f))
            tl += int(p)
        # the items that hold the wanted ranks, by the interval test every lane makes on its four counts
        jt = [None, None]
        jf = [None, None]
        for i in range(ni):
            lo_t, hi_t = (int(incl[i - 1]) if i else 0), int(incl[i])
            a0, a1 = i * item, min(N, (i + 1) * item)
            w0, w1 = max(a0 - act, 0), max(a1 - act, 0)
            lo_f, hi_f = w0 - lo_t, w1 - hi_t
            if q_hi >= 0:
                if lo_t <= q_lo < hi_t: jt[0] = i
                if lo_t <= q_hi < hi_t: jt[1] = i
            if T_hi >= 0:
                if lo_f <= T_lo < hi_f: jf[0] = i
                if lo_f <= T_hi < hi_f: jf[1] = i
        tpos, fpos = {}, {}
        if q_hi >= 0:
            assert jt[0] is not None and jt[1] is not None and jt[0] <= jt[1]
            for j in range(jt[0], jt[1] + 1):
                rank = int(incl[j - 1]) if j else 0
                for pos in range(j * item, min(N, (j + 1) * item)):
                    if pos >= act and p_seg[pos]:
                        if q_lo <= rank <= q_hi: tpos[rank - q_lo] = pos - act
                        rank += 1
        if T_hi >= 0:
            assert jf[0] is not None and jf[1] is not None and jf[0] <= jf[1]
            for j in range(jf[0], jf[1] + 1):
                rank = max(j * item - act, 0) - (int(incl[j - 1]) if j else 0)
                for pos in range(j * item, min(N, (j + 1) * item)):
                    if pos >= act and not p_seg[pos]:
                        if T_lo <= rank <= T_hi: fpos[rank - T_lo] = pos - act
                        rank += 1
        for x, p, F, T, need_t, need_f in el:
            tF = n if F == 0 else (tpos[ttot - F - q_lo] if need_t else -1)
            fj = fpos[T - T_lo] if need_f else n
            left = x < ttot or (x == ttot and x < tF)
            amb = ttot - 1 <= x <= ttot + 1
            fetch = x + n - tF if left else (n - 1 - x) + fj + 1
            if amb and fetch == n - 1:
                assert L is None
                dest = L = ttot - int(p)
            elif left:
                dest = x if p else tF - 1
            else:
                dest = fj if p else x - 1
            assert 0 <= dest < n and arr[dest] < 0
            arr[dest] = x
    assert L is not None
    return L, arr


def test_fused_round_model_equals_literal_loop(oracle):
    """a_round_kernel's bookkeeping - rank ranges per item, partner items from the scanned counts, the two tables - against
    blas.rs:168-182 on windows that start inside an item, with item sizes small enough that partners span many items."""
    rng = np.random.default_rng(606)
    for it in range(4000):
        item = int(rng.choice([1, 2, 3, 4, 8, 16]))
        N = int(rng.integers(1, 150))
        act = int(rng.integers(0, N)) if it % 3 else 0
        dens = [0.0, 1.0, 0.05, 0.5, 0.95, float(rng.random())][it % 6]
        keys = np.where(rng.random(N) < dens, 0.25, 0.75).astype(np.float32)
        piv, ids = oracle.partition_shuffle(keys[act:], np.arange(N - act), 0, N - act, np.float32(0.5))
        L, arr = fused_round_model(keys < 0.5, act, item)
        assert piv == L and np.array_equal(ids, arr), (it, item, N, act)


def test_fused_round_model_every_predicate_up_to_11(oracle):
    for n in range(1, 12):
        for bits in range(1 << n):
            p = np.array([(bits >> k) & 1 for k in range(n)], bool)
            keys = np.where(p, 0.25, 0.75).astype(np.float32)
            piv, ids = oracle.partition_shuffle(keys, np.arange(n), 0, n, np.float32(0.5))
            for item in (2, 3):
                L, arr = fused_round_model(p, 0, item)
                assert piv == L and np.array_equal(ids, arr), (n, bits, item)
