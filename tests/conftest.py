import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
FIXTURES = ["random20k", "repeats_snps", "palindrome_circle"]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _guarded_rank(fn, rank, world, port, q, args):
    """child side of run_ranks: a rank that raises reports (rank, traceback) instead of leaving the parent to time out"""
    import traceback
    try:
        fn(rank, world, port, *args, q)
    except BaseException:
        q.put(("__rank_failed__", rank, traceback.format_exc()))
        raise


def run_ranks(fn, world, args=(), timeout=300):
    """spawns `world` processes running fn(rank, world, port, *args, q); -> their q.put() results (one each).
    A failing rank's traceback fails the test at once, and no rank is left behind (stuck in a collective, holding the GPU)."""
    import queue as _queue
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_guarded_rank, args=(fn, r, world, port, q, tuple(args))) for r in range(world)]
    for p in procs:
        p.start()
    outs = []
    try:
        for _ in procs:
            try:
                o = q.get(timeout=timeout)
            except _queue.Empty:
                raise AssertionError(f"no result from a rank within {timeout} s (exit codes so far: {[p.exitcode for p in procs]})")
            if isinstance(o, tuple) and len(o) == 3 and o[0] == "__rank_failed__":
                raise AssertionError(f"rank {o[1]} failed:\n{o[2]}")
            outs.append(o)
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0, f"a rank exited with {p.exitcode}"
    finally:
        for p in procs:
            if p.is_alive():
                p.terminate()
        for p in procs:
            p.join(timeout=10)
            if p.is_alive():
                p.kill()
    return outs


def load_fixture(name):
    """-> dict with packed/unpacked reads, PQVec blobs and raw quals of a golden fixture"""
    from w2rap_contigger_amd import formats as F
    pk, bo, ln = F.read_fastb(os.path.join(GOLDEN, name + ".fastb"))
    codes, off = F.unpack_bases(pk, bo, ln)
    pq, po = F.read_qualp(os.path.join(GOLDEN, name + ".qualp"))
    quals, qoff = F.qualp_to_raw(pq, po)
    assert np.array_equal(off, qoff)
    return dict(name=name, packed=pk, byte_off=bo, read_len=ln, codes=codes, off=off, pq=pq, pq_off=po, quals=quals)


@pytest.fixture(scope="session", params=FIXTURES)
def fixture_data(request):
    return load_fixture(request.param)


def golden_bytes(name, tag, ext):
    with open(os.path.join(GOLDEN, f"{name}.{tag}.{ext}"), "rb") as f:
        return f.read()


def relabel_compare(res_hbv, res_paths, ref_hbv, ref_paths, max_ties=0.002):
    """Compare (hbv, (offset, path_off, edges)) with a reference modulo edge relabelling
    (SURVEY.md 8c): same edge-sequence set, same vertices per edge, same adjacency, same paths
    except extension tie-breaks between parallel edges (Q14)."""
    def seqs(h):
        codes, off = h.edge_codes()
        off = off.astype(np.int64)
        return [codes[off[i]:off[i + 1]].tobytes() for i in range(h.n_edges)]
    ours, theirs = seqs(res_hbv), seqs(ref_hbv)
    assert sorted(ours) == sorted(theirs), "edge sequence sets differ"
    idx = {s: i for i, s in enumerate(theirs)}
    assert len(idx) == len(theirs)
    to_ref = np.array([idx[s] for s in ours], dtype=np.int64)

    def left_right(h):
        left = np.zeros(h.n_edges, np.int64); right = np.zeros(h.n_edges, np.int64)
        fo = h.from_off.astype(np.int64); to = h.to_off.astype(np.int64)
        for v in range(h.n_vertices):
            left[h.from_e[fo[v]:fo[v + 1]]] = v
            right[h.to_e[to[v]:to[v + 1]]] = v
        return left, right
    l1, r1 = left_right(res_hbv)
    l2, r2 = left_right(ref_hbv)
    assert res_hbv.n_vertices == ref_hbv.n_vertices
    assert np.array_equal(l1, l2[to_ref]) and np.array_equal(r1, r2[to_ref]), "vertex ids differ after relabelling"
    o1, p1, e1 = res_paths
    o2, p2, e2 = ref_paths
    assert len(o1) == len(o2)
    p1 = p1.astype(np.int64); p2 = p2.astype(np.int64)
    e1r = to_ref[e1] if len(e1) else e1
    ties = 0
    for i in range(len(o1)):
        a, b = e1r[p1[i]:p1[i + 1]], e2[p2[i]:p2[i + 1]]
        if o1[i] == o2[i] and len(a) == len(b) and np.array_equal(a, b):
            continue
        # allowed: same length, every differing position is a parallel edge (same vertices); offsets may
        # differ when the tie is on a left extension of different length
        assert len(a) == len(b), f"read {i}: path lengths differ {a} vs {b}"
        for x, y in zip(a, b):
            if x != y:
                assert l2[x] == l2[y] and r2[x] == r2[y], f"read {i}: {a} vs {b} is not a parallel-edge tie"
        ties += 1
    assert ties <= max(1, int(max_ties * len(o1))), f"{ties} tie-break differences"
    return ties


# ---------------------------------------------------------------------------------------------- oracle results, once per session
# The oracle is single-threaded C (about a minute per million reads); by round 5 the GPU suite spent two thirds of its 520 s waiting for it,
# several times for the same reads.  (1) oracle.run is memoised for the session by a hash of its inputs; (2) the big synthetic read sets
# come from ONE place (synth_reads / planted_reads below, memoised); (3) in a GPU session the oracle runs of those sets are started at
# once on a few host threads (the ctypes call releases the GIL, every run owns its state) and a test that needs one waits for that run
# only -- the fixture tests at the start of the session run meanwhile.
_ORACLE_MEMO = {}
_READS_MEMO = {}
_ORACLE_POOL = None


def _oracle_key(codes, quals, off, kw):
    import xxhash
    h = xxhash.xxh3_128()
    for a in (codes, quals, off):
        a = np.ascontiguousarray(a)
        h.update(str((a.dtype.str, a.shape)).encode()); h.update(a.view(np.uint8).reshape(-1).data)
    for k in sorted(kw):
        v = kw[k]
        h.update(k.encode())
        if isinstance(v, np.ndarray):
            v = np.ascontiguousarray(v); h.update(str((v.dtype.str, v.shape)).encode()); h.update(v.view(np.uint8).reshape(-1).data)
        else:
            h.update(repr(v).encode())
    return h.hexdigest()


def _install_oracle_memo():
    from oracle import oracle as O
    if getattr(O.run, "_memoised", False):
        return O
    plain = O.run

    def run(codes, quals, off, **kw):
        key = _oracle_key(codes, quals, off, kw)
        hit = _ORACLE_MEMO.get(key)
        if hit is None:
            hit = _ORACLE_MEMO[key] = plain(codes, quals, off, **kw)
        elif hasattr(hit, "result"):                      # a prefetch still running (or done): its result
            hit = _ORACLE_MEMO[key] = hit.result()
        return hit
    run._memoised = True
    run._plain = plain
    O.run = run
    return O


def oracle_prefetch(codes, quals, off, **kw):
    """starts oracle.run(codes, quals, off, **kw) on a host thread unless the session already has (or is computing) that result"""
    global _ORACLE_POOL
    O = _install_oracle_memo()
    key = _oracle_key(codes, quals, off, kw)
    if key in _ORACLE_MEMO:
        return
    if _ORACLE_POOL is None:
        from concurrent.futures import ThreadPoolExecutor
        _ORACLE_POOL = ThreadPoolExecutor(max_workers=max(1, min(8, (os.cpu_count() or 2) // 2)))
    _ORACLE_MEMO[key] = _ORACLE_POOL.submit(O.run._plain, codes, quals, off, **kw)


def _host_reads(d):
    from w2rap_contigger_amd import formats as F, synth
    codes = synth.unpack_fixed(d["packed"], synth.READ_LEN).cpu().numpy().reshape(-1)
    quals = d["quals"].cpu().numpy().reshape(-1)
    off = np.arange(d["n"] + 1, dtype=np.uint64) * synth.READ_LEN
    pk, bo, ln = F.pack_bases(codes, off)
    return dict(codes=codes, quals=quals, off=off, pk=pk, bo=bo, ln=ln, n=int(d["n"]))


def synth_reads(n, genome, seed):
    """host arrays (codes, quals, off, pk, bo, ln) of synth.generate_reads_device(n, genome, seed): the bench generator's reads"""
    key = ("synth", n, genome, seed)
    if key not in _READS_MEMO:
        import torch
        from w2rap_contigger_amd import synth
        d = synth.generate_reads_device(n, genome, seed, device="cuda")
        torch.cuda.synchronize()
        _READS_MEMO[key] = _host_reads(d)
        del d
        torch.cuda.empty_cache()
    return _READS_MEMO[key]


def planted_reads(n, seed):
    """the same for bench.planted_reads(n, seed): two haplotypes, planted repeat families, inverted copies"""
    key = ("planted", n, seed)
    if key not in _READS_MEMO:
        import torch
        import bench
        d = bench.planted_reads(n, seed, torch.device("cuda", 0))
        torch.cuda.synchronize()
        _READS_MEMO[key] = _host_reads(d)
        del d
        torch.cuda.empty_cache()
    return _READS_MEMO[key]


# the read sets whose full oracle run several -m gpu tests wait for (seconds of GPU work to make, a minute each for the oracle)
BENCH_LIKE = (1_200_000, 6_000_000, 91)
PREFETCH_SYNTH = [BENCH_LIKE, (1_100_000, 5_500_000, 78), (600_000, 3_000_000, 11), (400_000, 2_000_000, 5)]
PLANTED_LIKE = (1_200_000, 77)


@pytest.fixture(scope="session", autouse=True)
def _oracle_session(request):
    try:
        _install_oracle_memo()
    except Exception:
        yield
        return
    expr = request.config.getoption("-m") or ""
    if "gpu" in expr and "not gpu" not in expr and os.environ.get("W2RAP_TEST_NO_PREFETCH") != "1":
        try:
            import torch
            if torch.cuda.is_available():
                for spec in PREFETCH_SYNTH:
                    r = synth_reads(*spec)
                    oracle_prefetch(r["codes"], r["quals"], r["off"])
                r = planted_reads(*PLANTED_LIKE)
                oracle_prefetch(r["codes"], r["quals"], r["off"])
        except Exception as e:                            # the prefetch is an optimisation: a test that needs a result computes it itself
            print(f"[conftest] oracle prefetch not started: {e}", file=sys.stderr)
    yield
    if _ORACLE_POOL is not None:
        _ORACLE_POOL.shutdown(wait=False, cancel_futures=True)
