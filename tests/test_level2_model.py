"""CPU model of level 2 of the sharded list ranking (csrc/step2_shard.hip: k_seg_mark / k_seg_walk1 / k_seg_jump / k_seg_walk2 and the
exchanges between them), in numpy, against a direct walk of every chain: what each rank computes from what it holds, what travels, and the
corner cases the kernels' comments name (a chain END's word carries its own length; a circle without a splitter is never visited, one with
splitters never reaches an end; a stretch on a circle hands out 'absent')."""
import numpy as np
import pytest

ABSENT = np.uint64(0xFFFFFFFFFFFFFFFF)


def sampled(s):
    """seg_sampled of step2_shard.hip: one segment in 64 by a hash of its number"""
    with np.errstate(over="ignore"):
        return ((np.asarray(s, np.uint64) * np.uint64(0x9E3779B97F4A7C15)) >> np.uint64(58)) == 0


def make_job(rng, n_chains, max_len, world, n_circles=0):
    """chains of segments in PAIRS (segment s and its reverse s ^ 1, linked the other way round) scattered over `world` ranks; -> per-rank
    (length, next-or-self) words in job-wide numbering, and the expected (end, distance to the end) of every segment"""
    lens = rng.integers(1, max_len + 1, n_chains + n_circles)
    n_seg = 2 * int(lens.sum())
    order = rng.permutation(n_seg // 2)                       # pair p = segments 2p, 2p + 1, in random places
    length = rng.integers(1, 64, n_seg // 2).astype(np.uint64)
    nxt = np.arange(n_seg, dtype=np.uint64)                   # a chain END points at itself
    seg_len = np.repeat(length, 2)
    exp_end = np.full(n_seg, ABSENT, np.uint64); exp_t = np.zeros(n_seg, np.uint64)
    at = 0
    for c, L in enumerate(lens):
        pairs = order[at:at + L]; at += L
        flip = rng.integers(0, 2, L)                          # which member of the pair lies on the forward chain
        fwd = 2 * pairs + flip; rev = fwd ^ 1
        circle = c >= n_chains
        for i in range(L - 1):
            nxt[fwd[i]] = fwd[i + 1]; nxt[rev[i + 1]] = rev[i]
        if circle:
            nxt[fwd[L - 1]] = fwd[0]; nxt[rev[0]] = rev[L - 1]
        else:
            for chain in (fwd, rev[::-1]):
                t = np.cumsum(seg_len[chain][::-1])[::-1]     # k-mers from a segment's head to the end of the chain
                exp_end[chain] = chain[-1]; exp_t[chain] = t
    # ownership: contiguous ranges of the job-wide numbers, as segbase[] gives them
    cuts = np.sort(rng.integers(0, n_seg // 2 + 1, world - 1)) * 2
    segbase = np.concatenate([[0], cuts, [n_seg]]).astype(np.int64)
    return seg_len, nxt, segbase, exp_end, exp_t


def level2(seg_len, nxt, segbase):
    """-> (Fend, T) per rank for ITS segments + the splitters (what the library holds after PH_L2_RESULTS), traffic counters"""
    world = len(segbase) - 1
    n = len(nxt)
    # --- exchange 1: every rank's words all-gathered: (seg_len, nxt) of the whole job on every rank
    ids = np.arange(n, dtype=np.uint64)
    is_end = nxt == ids
    head = is_end[ids ^ np.uint64(1)]                          # the reverse of s is a chain end: s is a chain head
    sp = head | sampled(ids)
    own = [np.arange(segbase[r], segbase[r + 1]) for r in range(world)]
    # --- walk 1, a rank from ITS splitters: (distance to the next splitter, it) or an end record; steps = segments between
    recs = {}
    steps_total = 0
    for r in range(world):
        for s in own[r][sp[own[r]]]:
            cur, acc, st = int(s), 0, 0
            while True:
                acc += int(seg_len[cur]); nx = int(nxt[cur])
                if nx == cur: recs[int(s)] = (0, int(s), acc, cur, st); break          # word (0, s), T, Fend
                if sp[nx]: recs[int(s)] = (acc, nx, None, None, st); break
                cur = nx; st += 1
            steps_total += st
    # --- exchange 2: the records all-gathered; every rank alike from here: pointer jumping over the splitters
    w_dist = {s: v[0] for s, v in recs.items()}; w_next = {s: v[1] for s, v in recs.items()}
    T = {s: v[2] for s, v in recs.items() if v[2] is not None}; F = {s: v[3] for s, v in recs.items() if v[3] is not None}
    for _ in range(64):
        changed = False
        nd, nn = dict(w_dist), dict(w_next)
        for v in recs:
            a = w_next[v]
            if a == v: continue
            b = w_next[a]
            if b == a: continue                               # arrived at the last splitter of the chain
            nd[v] = w_dist[v] + w_dist[a]; nn[v] = b; changed = True
        w_dist, w_next = nd, nn
        if not changed: break
    Fsp, Tsp = {}, {}
    for s in recs:                                            # k_seg_splitters_done
        e = w_next[s]
        if w_next[e] != e or e not in F: continue             # a circle of splitters
        Fsp[s] = F[e]; Tsp[s] = (0 if e == s else w_dist[s]) + T[e]
    # --- walk 2, a rank from ITS splitters again: results for the segments of the stretch, routed to their owners
    routed = [dict() for _ in range(world)]
    n_routed = 0
    for r in range(world):
        for s in own[r][sp[own[r]]]:
            s = int(s)
            ok = s in Fsp
            t = Tsp.get(s, 0)
            cur = s
            while True:
                nx = int(nxt[cur])
                if cur != s:
                    o = int(np.searchsorted(segbase, cur, side="right") - 1)
                    routed[o][cur] = (Fsp[s] if ok else None, t if ok else 0); n_routed += 1
                if ok: t -= int(seg_len[cur])
                if nx == cur or sp[nx]: break
                cur = nx
    assert n_routed == steps_total                            # the first walk's step counts reserve exactly the second walk's output
    out = []
    for r in range(world):
        Fend = np.full(n, ABSENT, np.uint64); Tt = np.zeros(n, np.uint64)
        for s, f in Fsp.items(): Fend[s] = f; Tt[s] = Tsp[s]  # the splitters' values are on every rank
        for s, (f, t) in routed[r].items():
            if f is not None: Fend[s] = f; Tt[s] = t
        out.append((Fend, Tt))
    return out, dict(splitters=len(recs), routed=n_routed)


@pytest.mark.parametrize("world,n_chains,max_len,n_circles", [(1, 40, 30, 0), (3, 60, 200, 0), (4, 25, 400, 3), (2, 5, 3, 2), (8, 300, 50, 5)])
def test_sharded_walks_rank_every_segment(world, n_chains, max_len, n_circles):
    rng = np.random.default_rng(1000 * world + n_chains)
    seg_len, nxt, segbase, exp_end, exp_t = make_job(rng, n_chains, max_len, world, n_circles)
    out, traffic = level2(seg_len, nxt, segbase)
    n = len(nxt)
    for r in range(world):
        Fend, T = out[r]
        mine = np.arange(segbase[r], segbase[r + 1])
        assert np.array_equal(Fend[mine], exp_end[mine])      # a segment on a circle stays ABSENT: flagged for the circle pass
        on_chain = exp_end[mine] != ABSENT
        assert np.array_equal(T[mine][on_chain], exp_t[mine][on_chain])
    # work: every non-splitter segment of a chain is visited by exactly one second walk; splitters ~ heads + 1/64
    assert traffic["routed"] <= n and traffic["splitters"] >= 2 * n_chains


def test_a_chain_end_word_carries_its_own_length():
    """one chain of three segments on three ranks, no sampled splitter in between: the head walks to the end and must count the end's length"""
    seg_len = np.array([5, 5, 7, 7, 11, 11], np.uint64)
    nxt = np.array([2, 1, 4, 1, 4, 3], np.uint64)             # 0 -> 2 -> 4 (end); reverse 5 -> 3 -> 1 (end)
    segbase = np.array([0, 2, 4, 6])
    if sampled(2) or sampled(3): pytest.skip("the hash samples the middle segment")
    out, _ = level2(seg_len, nxt, segbase)
    assert int(out[0][1][0]) == 23 and int(out[0][0][0]) == 4               # the head on rank 0: all three lengths, end = segment 4
    assert int(out[1][1][2]) == 18 and int(out[2][1][4]) == 11
    assert int(out[2][1][5]) == 23 and int(out[2][0][5]) == 1
