"""CPU models of two pieces of index arithmetic in k_superkmers_lane (w2rap_contigger_amd/csrc/step2_count.hip): the sliding-window minimum
over WIN = 46 m-mer keys by blocks of 16 (van Herk / Gil-Werman: suffix minima of the three blocks before the current one, running prefix
of the current one) and the cut of a read's k-mers into records (bucket change, or REC_MAXK k-mers).  The GPU parity tests compare the
kernel with the oracle; these pin the scheme itself, step for step as the kernel walks it."""
import numpy as np
import pytest

WIN, MMER, K, REC_MAXK = 46, 15, 60, 63
INF = 0xFFFFFFFF


def window_min_blocks(keys):
    """keys[j] = key of m-mer j of a read (len = gl - 14); -> window minimum of every k-mer p (len - 45 of them), block by block"""
    nj = len(keys)
    X3 = [INF] * 16; X2 = [INF] * 16; X1 = [INF] * 16
    out = {}
    jb = 0
    while jb < nj + 16:                                      # (the kernel runs whole blocks past the longest read of the wave)
        C = [INF] * 16
        pref = INF
        w21 = min(X2[0], X1[0])
        for i in range(16):
            j = jb + i
            key = int(keys[j]) if j < nj else INF
            C[i] = key; pref = min(pref, key)
            p = j - (WIN - 1)
            if p >= 0:
                mk = min(X3[i + 3], w21, pref) if i <= 12 else min(X2[i - 13], X1[0], pref)
                out[p] = mk
        for t in range(14, -1, -1):
            C[t] = min(C[t], C[t + 1])
        X3, X2, X1 = X2, X1, C
        jb += 16
    return [out[p] for p in range(nj - (WIN - 1))] if nj >= WIN else []


@pytest.mark.parametrize("gl", [60, 61, 75, 76, 77, 91, 92, 107, 108, 150, 151, 251, 400])
def test_window_minimum_by_blocks_of_16(gl):
    rng = np.random.default_rng(gl)
    keys = rng.permutation(1 << 20)[: gl - (MMER - 1)].astype(np.uint32)       # distinct keys, as the bijective m-mer hash gives
    got = window_min_blocks(keys)
    want = [int(keys[p:p + WIN].min()) for p in range(len(keys) - (WIN - 1))]
    assert got == want and len(got) == max(0, gl - (K - 1))


def cut_records(buckets):
    """buckets[p] = bucket of k-mer p; -> [(p0, n)] as the lane kernel cuts them (ALIGN64 = false)"""
    recs = []
    p0, cur = 0, None
    nk = len(buckets)
    for p in range(nk + 1):                                  # p == nk: the sentinel step that closes the last run
        isk = p < nk
        brk = isk and (p == 0 or buckets[p] != cur or p - p0 == REC_MAXK)
        if (brk and p) or (p == nk and nk):
            recs.append((p0, p - p0))
        if brk:
            p0, cur = p, buckets[p]
    return recs


@pytest.mark.parametrize("seed", range(6))
def test_records_cover_the_read_once_and_never_exceed_their_capacity(seed):
    rng = np.random.default_rng(seed)
    nk = int(rng.integers(1, 400))
    runs = rng.integers(1, 150, size=40)
    b = np.repeat(rng.integers(0, 5, size=40), runs)[:nk]
    recs = cut_records(list(b))
    assert sum(n for _, n in recs) == len(b) and recs[0][0] == 0
    for (p0, n), nxt in zip(recs, recs[1:] + [(len(b), 0)]):
        assert 1 <= n <= REC_MAXK and p0 + n == nxt[0]
        assert len(set(b[p0:p0 + n])) == 1                  # one bucket per record
        if nxt[0] < len(b) and b[nxt[0]] == b[p0]:
            assert n == REC_MAXK                             # the same bucket goes on only because the record was full
