// tlas_index_model.cpp — CPU model of the ALGORITHM of the indexed TLAS build (voidin_amd/csrc/tlas.hip, "build,
// indexed"): test infrastructure.  It restates, sequentially and without any GPU detail, the three claims the
// kernel rests on, so that they can be checked against the literal oracle (oracle/vd_oracle_tlas.c) on the CPU:
//
//  1. PRUNE.  For a group G of clusters with bounding box S (S.mn <= o.mn, o.mx <= S.mx for every o in G) and a target
//     box t, the box L = {min(t.mn, S.mx), max(t.mx, S.mn)} lies inside union(t, o) for every o in G, and the f32
//     evaluation of Aabb::area (intersection.rs:16-19: subtract, multiply, add - each monotone under round-to-nearest
//     on non-negative extents) is monotone under inclusion, so area(L) <= area(union(t, o)) in f32.  A group can
//     be skipped when area(L) > the best area found so far (strictly: an equal area may still win on the slot index).
//  2. CACHE.  The result of find_best_match(X) -> Y, found STRICT (no other candidate had the same area), stays the
//     answer for as long as X and Y are both alive and unchanged: every cluster that exists later is a cluster that
//     existed then, or a union of such clusters not containing X or Y, whose union area with X is >= the smallest of
//     theirs (monotone again) > area(X u Y).  The slot index plays no part, so slot relabelling cannot change it.
//  5. SPECULATION (claim 4 is further down, at the chain).  While c = best(b) is being answered, the answer of the query
//     that follows IF c == a - best(a u b), bounded by the chain element before a - can be computed on the state BEFORE
//     the merge: leave out the entries of a and b, count the entry that holds the last slot as slot b, same group
//     corners.  That is all a merge changes (tlas.rs:72-75), so the chain may take that answer without asking
//     (params[6]; the model also asks in the normal way and returns -2 if the two ever differ).
//  3. SLOTS.  The reference works on slot indices (tlas.rs:56-84): idx[a] = merged, idx[b] = idx[cnt-1].  When a is
//     the last slot, the merged cluster lands in slot b and the chain goes on with the stale index a (>= cnt): the
//     next find_best_match(a) sees the merged cluster itself as a candidate in slot b.  Modelled as in the kernel:
//     entries are clusters, `slot` is an attribute, exclusion is by slot.
//
// Build: g++ -O2 -ffp-contract=off -shared -fPIC (tests/test_tlas_index_model.py).
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

namespace {

constexpr uint32_t DEAD = 0xffffffffu;

inline int32_t key_of(float f) { int32_t i; std::memcpy(&i, &f, 4); return i ^ ((i >> 31) & 0x7fffffff); }
inline float min_to(float a, float b) { return key_of(b) < key_of(a) ? b : a; }
inline float max_to(float a, float b) { return key_of(b) > key_of(a) ? b : a; }
inline float area3(float dx, float dy, float dz) { return (dx * dy + dx * dz + dy * dz) * 2.0f; }
inline uint32_t bits(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }

struct Box { float mn[3], mx[3]; };

// the FAST arithmetic of the kernel: v_min/v_max for the union extents (no NaN anywhere: precondition)
inline float union_area(const Box& t, const Box& o) {
    const float dx = std::fmax(t.mx[0], o.mx[0]) - std::fmin(t.mn[0], o.mn[0]);
    const float dy = std::fmax(t.mx[1], o.mx[1]) - std::fmin(t.mn[1], o.mn[1]);
    const float dz = std::fmax(t.mx[2], o.mx[2]) - std::fmin(t.mn[2], o.mn[2]);
    return area3(dx, dy, dz) + 0.0f;
}
// claim 1: lower bound of union_area(t, o) over every o of a group whose INNER corner is I:
//   I.mn = max over the group of o.mn,  I.mx = min over the group of o.mx   (may be inverted)
// every member has o.mn <= I.mn and o.mx >= I.mx, so {min(t.mn, I.mn), max(t.mx, I.mx)} lies inside union(t, o).
inline float lower_bound(const Box& t, const Box& I) {
    const float dx = std::fmax(t.mx[0], I.mx[0]) - std::fmin(t.mn[0], I.mn[0]);
    const float dy = std::fmax(t.mx[1], I.mx[1]) - std::fmin(t.mn[1], I.mn[1]);
    const float dz = std::fmax(t.mx[2], I.mx[2]) - std::fmin(t.mn[2], I.mn[2]);
    return area3(dx, dy, dz) + 0.0f;
}
inline void grow(Box& S, const Box& b) {      // outer box (scene bounds)
    for (int k = 0; k < 3; ++k) { S.mn[k] = std::fmin(S.mn[k], b.mn[k]); S.mx[k] = std::fmax(S.mx[k], b.mx[k]); }
}
inline void shrink(Box& I, const Box& b) {    // inner corner of a group
    for (int k = 0; k < 3; ++k) { I.mn[k] = std::fmax(I.mn[k], b.mn[k]); I.mx[k] = std::fmin(I.mx[k], b.mx[k]); }
}
inline Box empty_box() { Box b; for (int k = 0; k < 3; ++k) { b.mn[k] = 1e30f; b.mx[k] = -1e30f; } return b; }
// inner corner of an EMPTY group: every lower bound through it is +inf (extent 2e30 squared overflows), i.e. pruned
inline Box empty_inner() { Box b; for (int k = 0; k < 3; ++k) { b.mn[k] = -1e30f; b.mx[k] = 1e30f; } return b; }

inline uint32_t spread6(uint32_t x) {     // 5 bits -> every sixth bit
    uint32_t r = 0;
    for (int b = 0; b < 5; ++b) r |= ((x >> b) & 1u) << (6 * b);
    return r;
}

struct Cache { uint32_t ver_x, e_y, ver_y, strict; };

struct Model {
    uint32_t n, E, slice, block, super_slices, phase2;
    std::vector<Box> ent_box; std::vector<uint32_t> ent_slot, ent_node, slot_ent;
    std::vector<Box> slice_box, super_box;
    std::vector<Cache> cache;
    uint64_t st_full = 0, st_cached = 0, st_cand = 0, st_slices = 0, st_phase2 = 0, st_nonstrict = 0, st_lb = 0, st_ownblock = 0;

    struct Hit { uint64_t key; uint32_t e; bool strict; };

    // the full query of the kernel.  `bound`: the union area with some cluster that is a candidate of this query (the
    // chain knows one most of the time: see tlas_index_model below), NaN when there is none - then the target's own
    // block of entries is scanned first and supplies the bound.  A group is skipped when its lower bound EXCEEDS the
    // bound; the bound is not tightened while the groups are visited (the kernel visits them in parallel).
    // skip1 / skip2 / from_slot -> to_slot: the speculative form (claim 5); DEAD = not used
    Hit query(uint32_t t_slot, const Box& tb, uint32_t e_t, float bound, uint32_t skip1 = DEAD, uint32_t skip2 = DEAD,
              uint32_t from_slot = 0xfffffffeu, uint32_t to_slot = 0) {
        uint64_t best = ~0ull; uint32_t best_e = 0; bool tie = false;
        auto eval = [&](uint32_t e) {
            uint32_t s = ent_slot[e];
            if (s == DEAD || s == t_slot || e == skip1 || e == skip2) return;
            if (s == from_slot) s = to_slot;
            const float a = union_area(tb, ent_box[e]);
            ++st_cand;
            if (!(a < 1e30f)) return;
            const uint64_t k = ((uint64_t)bits(a) << 32) | s;
            if ((k >> 32) == (best >> 32)) tie = true;
            if (k < best) { if ((k >> 32) != (best >> 32)) tie = false; best = k; best_e = e; }
        };
        const uint32_t n_slices = (uint32_t)slice_box.size(), per_block = block / slice;
        uint32_t blk = 0xffffffffu;
        bool have = bound == bound;
        if (!have) {
            blk = e_t / block;
            for (uint32_t e = blk * block; e < std::min(E, (blk + 1) * block); ++e) eval(e);
            const uint32_t bound_bits = (uint32_t)(best >> 32);
            std::memcpy(&bound, &bound_bits, 4);                          // 0xffffffff = NaN: still no bound
            have = best != ~0ull;
            ++st_ownblock;
        }
        for (uint32_t sp = 0; sp < super_box.size(); ++sp) {
            ++st_lb;
            if (have && !(lower_bound(tb, super_box[sp]) <= bound)) continue;
            for (uint32_t sl = sp * super_slices; sl < std::min(n_slices, (sp + 1) * super_slices); ++sl) {
                if (sl / per_block == blk) continue;                      // scanned already
                ++st_lb;
                if (have && !(lower_bound(tb, slice_box[sl]) <= bound)) continue;
                ++st_slices;
                for (uint32_t e = sl * slice; e < std::min(E, (sl + 1) * slice); ++e) eval(e);
            }
        }
        return Hit{best, best_e, !tie};
    }
};

}  // namespace

extern "C" {

// leaf_boxes: n x {mn[3], mx[3]}; out_box: (2n+1) x 6 floats; out_left/out_right/out_inst: 2n+1 each.
// params: {slice, block, super_slices, phase2_threshold, use_cache, refresh_every (merges; 0 = never), use_spec}; stats (may be null): 9 x u64.
// Returns 0, or -1 when the precondition (finite, |x| < 1e18, mn <= mx) fails (the kernel then runs the plain chain).
int tlas_index_model(const float* leaf_boxes, uint32_t n, const uint32_t* params, float* out_box, uint32_t* out_left,
                     uint32_t* out_right, uint32_t* out_inst, uint64_t* stats) {
    Model M;
    M.n = n; M.slice = params[0]; M.block = params[1]; M.super_slices = params[2]; M.phase2 = params[3];
    const bool use_cache = params[4] != 0;
    const bool use_spec = params[6] != 0 && !use_cache;
    uint64_t st_spec_used = 0;
    std::vector<Box> leaf(n);
    Box scene = empty_box();
    for (uint32_t i = 0; i < n; ++i) {
        std::memcpy(&leaf[i], leaf_boxes + 6 * (size_t)i, 24);
        for (int k = 0; k < 3; ++k) {
            const float a = leaf[i].mn[k], b = leaf[i].mx[k];
            if (!(std::fabs(a) < 1e18f) || !(std::fabs(b) < 1e18f) || !(a <= b)) return -1;
        }
        grow(scene, leaf[i]);
    }
    const size_t total = 2 * (size_t)n + 1;
    std::vector<Box> node_box(total);
    std::memset(node_box.data(), 0, total * sizeof(Box));
    std::fill(out_left, out_left + total, 0u); std::fill(out_right, out_right + total, 0u); std::fill(out_inst, out_inst + total, 0u);
    for (uint32_t i = 0; i < n; ++i) { node_box[i + 1] = leaf[i]; out_inst[i + 1] = i; }

    // ---- index: entries in Morton order of the 6-D points (mn, mx), 5 bits per coordinate over the scene box, stable.
    // (All that matters for exactness is that every live cluster is an entry; the order only decides how tight the
    // groups' inner corners are.  6-D because the reference's leaf boxes all reach back to the object-space mesh box
    // (tlas.rs:39 seeds the fold with it): two boxes are "near" when BOTH their corners are.) ----
    std::vector<uint32_t> order(n), code(n);
    for (uint32_t i = 0; i < n; ++i) {
        uint32_t c = 0;
        for (int k = 0; k < 3; ++k) {
            const float lo = scene.mn[k], ext = scene.mx[k] - scene.mn[k];
            float q0 = ext > 0.0f ? (leaf[i].mn[k] - lo) / ext * 32.0f : 0.0f;
            float q1 = ext > 0.0f ? (leaf[i].mx[k] - lo) / ext * 32.0f : 0.0f;
            q0 = std::fmin(std::fmax(q0, 0.0f), 31.0f); q1 = std::fmin(std::fmax(q1, 0.0f), 31.0f);
            c |= spread6((uint32_t)q0) << k;
            c |= spread6((uint32_t)q1) << (3 + k);
        }
        code[i] = c; order[i] = i;
    }
    std::stable_sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) { return code[x] < code[y]; });
    M.E = (n + M.block - 1) / M.block * M.block;
    M.ent_box.assign(M.E, empty_box()); M.ent_slot.assign(M.E, DEAD); M.ent_node.assign(M.E, 0); M.slot_ent.assign(n, 0);
    const uint32_t n_slices = M.E / M.slice;
    for (uint32_t e = 0; e < n; ++e) {
        const uint32_t i = order[e];
        M.ent_box[e] = leaf[i]; M.ent_slot[e] = i; M.ent_node[e] = i + 1; M.slot_ent[i] = e;
    }
    // inner corners of the groups.  They stay VALID without any update: a merged cluster replaces one of its two parts in
    // that part's entry and only grows (mn down, mx up), dead entries only loosen the corner.  refresh() tightens them.
    auto refresh = [&]() {
        M.slice_box.assign(n_slices, empty_inner());
        M.super_box.assign((n_slices + M.super_slices - 1) / M.super_slices, empty_inner());
        for (uint32_t e = 0; e < M.E; ++e)
            if (M.ent_slot[e] != DEAD) {
                shrink(M.slice_box[e / M.slice], M.ent_box[e]);
                shrink(M.super_box[e / M.slice / M.super_slices], M.ent_box[e]);
            }
    };
    refresh();
    const uint32_t refresh_every = params[5];
    uint32_t since_refresh = 0;
    // ---- claim 2: every leaf's nearest neighbour up front (the kernel does this with the whole GPU) ----
    M.cache.assign(M.E, Cache{0, 0, 0, 0});
    if (use_cache)
        for (uint32_t e = 0; e < n; ++e) {
            const Model::Hit h = M.query(M.ent_slot[e], M.ent_box[e], e, NAN);
            if (h.key != ~0ull && h.strict) M.cache[e] = Cache{M.ent_node[e], h.e, M.ent_node[h.e], 1u};
        }
    M.st_full = M.st_cand = M.st_slices = M.st_lb = M.st_ownblock = 0;   // count the sequential part only

    // ---- the chain (tlas.rs:56-84) on (slot, entry) pairs ----
    uint32_t cnt = n, used = n + 1;
    auto best = [&](uint32_t t_slot, uint32_t e_t, const Box& tb, float bound, uint32_t& out_e) -> uint32_t {
        // cached answer?  Only for a target that sits in its own slot (claim 3) and whose record is intact (claim 2)
        if (use_cache && M.ent_slot[e_t] == t_slot) {
            const Cache& c = M.cache[e_t];
            if (c.strict && c.ver_x == M.ent_node[e_t] && M.ent_slot[c.e_y] != DEAD && M.ent_node[c.e_y] == c.ver_y) {
                ++M.st_cached;
                out_e = c.e_y;
                return M.ent_slot[c.e_y];
            }
        }
        ++M.st_full;
        const Model::Hit h = M.query(t_slot, tb, e_t, bound);
        if (h.key == ~0ull) { out_e = e_t; return t_slot; }             // nothing: find_best_match returns the target
        if (!h.strict) ++M.st_nonstrict;
        if (use_cache && M.ent_slot[e_t] == t_slot)
            M.cache[e_t] = Cache{M.ent_node[e_t], h.e, M.ent_node[h.e], h.strict ? 1u : 0u};
        out_e = h.e;
        return (uint32_t)h.key;
    };
    uint32_t a = 0, ea = M.slot_ent[0], b, eb, c, ec;
    Box box_a = M.ent_box[ea], box_b;
    // claim 4 (bounds from the chain): the union area with ANY cluster that is a candidate of the query is an upper bound
    // of its answer.  For c = best(b) that is a itself (b = best(a) a moment ago) unless a is a stale index; for
    // b = best(a) right after a merge it is `prev`, the chain element before a - never merged while it is remembered:
    // merges take (a, b), and prev is neither (forgotten when it becomes b of a merge).
    bool have_prev = false; Box box_prev = empty_box(); uint32_t e_prev = 0;
    b = best(a, ea, box_a, NAN, eb); box_b = M.ent_box[eb];
    bool phase2 = false;
    // phase 2 state: plain slot arrays, as the reference keeps them
    std::vector<Box> sbox; std::vector<uint32_t> snode;
    auto best2 = [&](uint32_t t) -> uint32_t {
        ++M.st_phase2;
        float smallest = 1e30f; uint32_t bi = t;
        for (uint32_t i = 0; i < cnt; ++i) {
            if (i == t) continue;
            const float ar = union_area(sbox[t], sbox[i]);
            if (ar < smallest) { smallest = ar; bi = i; }
        }
        return bi;
    };
    while (cnt > 0) {
        if (!phase2 && cnt <= M.phase2) {
            // hand-over: slot arrays from the live entries (+ the stale slot a, which the chain may still name)
            phase2 = true;
            sbox.assign(n, empty_box()); snode.assign(n, 0);
            for (uint32_t e = 0; e < M.E; ++e)
                if (M.ent_slot[e] != DEAD) { sbox[M.ent_slot[e]] = M.ent_box[e]; snode[M.ent_slot[e]] = M.ent_node[e]; }
            if (a >= cnt) { sbox[a] = box_a; snode[a] = M.ent_node[ea]; }
        }
        if (phase2) {
            c = best2(b);
            if (a == c) {
                const uint32_t ia = snode[a], ib = snode[b];
                Box u;
                for (int k = 0; k < 3; ++k) { u.mn[k] = min_to(sbox[a].mn[k], sbox[b].mn[k]); u.mx[k] = max_to(sbox[a].mx[k], sbox[b].mx[k]); }
                node_box[used] = u; out_left[used] = ia; out_right[used] = ib; out_inst[used] = 0xffffffffu;
                sbox[a] = u; snode[a] = used;
                sbox[b] = sbox[cnt - 1]; snode[b] = snode[cnt - 1];
                used += 1; cnt -= 1;
                b = best2(a);
            } else { a = b; b = c; }
            continue;
        }
        // claim 5: the query a merge of (a, b) would ask next, answered on the state before the merge
        bool spec_valid = false; Model::Hit spec_hit{~0ull, 0, false};
        if (use_spec && M.ent_slot[ea] == a && cnt - 1 != a && have_prev && e_prev != eb && e_prev != ea) {
            Box u;
            for (int k = 0; k < 3; ++k) { u.mn[k] = min_to(box_a.mn[k], box_b.mn[k]); u.mx[k] = max_to(box_a.mx[k], box_b.mx[k]); }
            spec_hit = M.query(0xfffffffeu, u, ea, union_area(u, box_prev), ea, eb, cnt - 1, b);
            spec_valid = true;
        }
        {
            float bound = NAN;
            if (M.ent_slot[ea] == a) bound = union_area(box_b, box_a);        // a is a live candidate of best(b)
            else if (have_prev && e_prev != eb) bound = union_area(box_b, box_prev);
            c = best(b, eb, box_b, bound, ec);
        }
        if (a == c) {
            const uint32_t ia = M.ent_node[ea], ib = M.ent_node[eb];
            Box u;
            for (int k = 0; k < 3; ++k) { u.mn[k] = min_to(box_a.mn[k], box_b.mn[k]); u.mx[k] = max_to(box_a.mx[k], box_b.mx[k]); }
            node_box[used] = u; out_left[used] = ia; out_right[used] = ib; out_inst[used] = 0xffffffffu;
            const uint32_t last = cnt - 1;
            if (have_prev && (e_prev == eb || e_prev == ea)) have_prev = false;    // prev is consumed by this merge (or IS the
                                                                                   // merged entry: after a stale step a and prev name one cluster)
            M.ent_box[ea] = u; M.ent_node[ea] = used;
            M.ent_slot[eb] = DEAD;
            if (last == a) {                       // idx[b] = idx[a] = merged: the cluster moves to slot b, a goes stale
                M.ent_slot[ea] = b; M.slot_ent[b] = ea;
            } else if (last != b) {                // idx[b] = idx[last]
                const uint32_t el = M.slot_ent[last];
                M.ent_slot[el] = b; M.slot_ent[b] = el;
            }
            used += 1; cnt -= 1;
            box_a = u;
            if (refresh_every && ++since_refresh >= refresh_every) { refresh(); since_refresh = 0; }
            if (cnt == 0) break;                   // (never in phase 1 when phase2 >= 1: kept for phase2 == 0)
            if (spec_valid) {
                const Model::Hit real = M.query(a, box_a, ea, have_prev ? union_area(box_a, box_prev) : NAN);
                if (real.key != spec_hit.key || (real.key != ~0ull && real.e != spec_hit.e)) return -2;
                if (spec_hit.key == ~0ull) { b = a; eb = ea; } else { b = (uint32_t)spec_hit.key; eb = spec_hit.e; }
                ++st_spec_used;
            } else {
                b = best(a, ea, box_a, have_prev ? union_area(box_a, box_prev) : NAN, eb);
            }
            box_b = M.ent_box[eb];
        } else {
            have_prev = true; box_prev = box_a; e_prev = ea;
            a = b; ea = eb; box_a = box_b;
            b = c; eb = ec; box_b = M.ent_box[eb];
        }
    }
    uint32_t root;
    if (phase2) root = snode[a];
    else root = M.ent_node[ea];
    node_box[0] = node_box[root]; out_left[0] = out_left[root]; out_right[0] = out_right[root]; out_inst[0] = out_inst[root];
    std::memcpy(out_box, node_box.data(), total * sizeof(Box));
    if (stats) {
        stats[0] = M.st_full; stats[1] = M.st_cached; stats[2] = M.st_cand; stats[3] = M.st_slices; stats[4] = M.st_phase2;
        stats[5] = M.st_nonstrict; stats[6] = M.st_lb; stats[7] = M.st_ownblock; stats[8] = st_spec_used;
    }
    return 0;
}

}  // extern "C"
