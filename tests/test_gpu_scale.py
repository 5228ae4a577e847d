"""The path every multi-GPU run takes -- dictionary, prune and unipaths sharded by bucket owner, read pathing through the minimizer-sampled
index + exact table (dist.distributed_count(gather=False) + dist.sharded_graph, what `bench.py --gpus N` drives) -- at the PER-GPU SHARES of
the BASELINE configs no single-GPU box can run whole:

  configs[2]  500 M reads / 8 GPUs   -> 62.5 M reads of a 312.5 Mbp genome on this GPU, byte-equal to the one-GPU (dictionary) path;
  configs[3]  Step 3 (K = 200) of it -> behind both, large-K graph and translated paths byte-equal;
  configs[4]  2 B reads, 17 Gbp / 8  -> 250 M reads of a 2.125 Gbp genome, three hash-range passes, 64-bit node ids: the size-independent
                                         properties, and the MEASURED peak of device memory per phase (w2rap_step2_device_peak_bytes) against
                                         the 288 GB of one MI355X -- the "fits 288 GB" of DESIGN.md section 5 as a number.

One rank owns everything here (world 1: one process group of one rank, no link is crossed); the two test hooks hand the cross-rank
machinery -- routed neighbour queries, segment chains, level-2 ranking -- the shares of an 8-rank job (W2RAP_TEST_SHARD_VIRTUAL=8: 7/8 of the
neighbour lookups go through the query path; W2RAP_TEST_SHARD_CUT=27: one chain link in 27 is treated as crossing ranks).  The oracle does
not run at these sizes (a minute per million reads): the checks are equality with the one-GPU path -- itself pinned to the oracle and the
reference at 1-8 M reads -- and the properties the domain offers."""
import json
import os
import socket

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

HBM_BYTES = 288e9


@pytest.fixture(scope="module")
def mods():
    import torch
    assert torch.cuda.is_available(), "the -m gpu tests need an MI355X"
    from w2rap_contigger_amd import formats as F, step2, step3, synth, dist as wd
    return F, step2, step3, synth, wd


@pytest.fixture(scope="module")
def world1():
    """a process group of ONE rank inside the pytest process (gloo: nothing travels at world 1, dist.py copies on the device)"""
    import torch.distributed as dist
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
    yield dist
    dist.destroy_process_group()


def _genome(n, seed):
    import torch
    gen = torch.Generator(device="cuda").manual_seed(seed)
    g = torch.empty(n, dtype=torch.uint8, device="cuda")
    for a in range(0, n, 1 << 30):                                      # (randint in pieces: its int64 scratch is 8 B per element)
        g[a:a + (1 << 30)] = torch.randint(0, 4, (min(1 << 30, n - a),), dtype=torch.uint8, device="cuda", generator=gen)
    return g


def _reads(synth, n, glen, seed):
    import torch
    g = _genome(glen, seed)
    d = synth.generate_reads_device(n, glen, seed, device="cuda", genome=g)
    del g
    d.pop("genome", None)
    torch.cuda.synchronize(); torch.cuda.empty_cache()
    return d


def _set(ctx, d):
    ctx.set_reads_device(d["n"], d["packed"].data_ptr(), d["byte_off"].data_ptr(), d["read_len"].data_ptr(), d["quals"].data_ptr(), d["qual_off"].data_ptr(), keepalive=d)


def _sharded_step(wd, ctx, n_passes=1, read_bytes=0):
    """what bench.py --gpus N runs on every rank; -> statistics, exchange log, and per phase the peak of the library's pool + the peak of
    what torch holds FOR THIS STEP meanwhile (the caller's reads, dist.py's exchange buffers) = an upper bound of the device memory the path
    needs.  (Tensors that earlier tests of the session left alive are not the path's: what torch holds beyond the reads before the step
    starts is subtracted, and reported.)"""
    import torch
    peaks = {}
    leftover = max(0, int(torch.cuda.memory_allocated()) - int(read_bytes)) if read_bytes else 0

    def phase(name):
        lib, tor = ctx.device_peak_bytes(reset=True), int(torch.cuda.max_memory_allocated()) - leftover
        torch.cuda.reset_peak_memory_stats()
        peaks[name] = dict(library=int(lib), torch=tor, total=int(lib) + tor)
    peaks["session_leftover"] = dict(library=0, torch=leftover, total=0)
    torch.cuda.reset_peak_memory_stats()
    ctx.device_peak_bytes(reset=True)
    be = wd.GpuBackend(ctx, "cuda:0")
    st = wd.distributed_count(be, 7, 4, gather=False, n_passes=n_passes)
    phase("count")
    info = wd.sharded_graph(be, st["S_local"], st, st["n_buckets"], n_passes=n_passes)
    phase("graph")
    ctx.path_reads()
    phase("path")
    return st, info, peaks


def _properties(F, res, st, n_reads, rng, min_pathed):
    """the size-independent properties of a Step-2 result (tests/test_gpu_parity.py test_properties_at_bench_size)"""
    h = res.hbv
    E = len(res.fwd_xlat)
    assert int(st["hist"].sum()) == st["D"] and int(st["hist"][4:].sum()) == st["S"]
    if st["hist"][100] == 0:
        assert int((np.arange(101, dtype=np.uint64) * st["hist"]).sum()) == st["M"]
    assert int((h.edge_len[res.fwd_xlat].astype(np.int64) - 59).sum()) == st["S"]          # every solid k-mer on exactly one unipath position
    assert np.array_equal(h.edge_len[res.fwd_xlat], h.edge_len[res.rev_xlat])
    ebo = h.edge_byte_off.astype(np.int64)

    def obj(o):
        a, b = int(ebo[o]), int(ebo[o + 1])
        return F.unpack_bases(h.edge_packed[a:b], np.array([0, b - a], np.uint64), np.array([h.edge_len[o]], np.uint32))[0]
    firsts = {}
    for x in rng.integers(0, E, 300):
        a, b = obj(int(res.fwd_xlat[x])), obj(int(res.rev_xlat[x]))
        assert np.array_equal(a, 3 - b[::-1])                                               # every object with its reverse complement
        firsts[int(x)] = a[:60].tobytes()
    xs = sorted(firsts)
    assert [firsts[x] for x in xs] == sorted(firsts[x] for x in xs)                         # unipaths in lexicographic order
    po = res.path_off.astype(np.int64)
    lens = np.diff(po)
    assert len(lens) == n_reads and res.n_reads_pathed > min_pathed * n_reads
    assert res.path_edges.min() >= 0 and res.path_edges.max() < h.n_edges
    multi = np.nonzero(lens > 1)[0]
    for i in (multi[rng.integers(0, len(multi), 5000)] if len(multi) else []):
        p = res.path_edges[po[i]:po[i + 1]]
        assert (res.vright[p[:-1]] == res.vleft[p[1:]]).all()                               # FixPaths adjacency


def _same_step2(F, a, b):
    assert np.array_equal(a.hist, b.hist)
    assert (a.n_kmer_instances, a.n_kmers_distinct, a.n_kmers_solid) == (b.n_kmer_instances, b.n_kmers_distinct, b.n_kmers_solid)
    assert F.hbv_to_bytes(a.hbv) == F.hbv_to_bytes(b.hbv)
    assert np.array_equal(a.path_offset, b.path_offset) and np.array_equal(a.path_off, b.path_off) and np.array_equal(a.path_edges, b.path_edges)
    assert (a.n_reads_pathed, a.n_reads_multipathed) == (b.n_reads_pathed, b.n_reads_multipathed)


def _note(name, obj):
    """what the run measured, for the log and -- when the test runs where it can write -- for profiles/ (gpurun_out/ on the GPU box)"""
    print(f"[scale] {name}: {json.dumps(obj)}")
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", f"scale_{name}.json"), "w") as f:
            json.dump(obj, f, indent=1)
    except OSError:
        pass


def test_configs2_and_3_share_sharded_equals_one_gpu(mods, world1, monkeypatch):
    """62.5 M reads / 312.5 Mbp: Step 2 through the sharded path + index (with the 8-rank shares of queries and segments) byte-equal to the
    one-GPU dictionary path; Step 3 at K = 200 behind either byte-equal too"""
    import torch
    F, step2, step3, synth, wd = mods
    n, glen = 62_500_000, 312_500_000
    d = _reads(synth, n, glen, 4201)
    read_bytes = sum(int(d[k].numel() * d[k].element_size()) for k in ("packed", "quals", "byte_off", "qual_off", "read_len"))
    with step2.Step2Context(0) as c1:
        _set(c1, d)
        st1 = c1.count_kmers(7, 4); c1.build_graph(None); c1.path_reads()
        one = c1.fetch()
        r3_one = step3.repath_after_step2(c1, 200)
    monkeypatch.setenv("W2RAP_TEST_SHARD_VIRTUAL", "8")
    monkeypatch.setenv("W2RAP_TEST_SHARD_CUT", "27")
    with step2.Step2Context(0) as c2:
        _set(c2, d)
        st, info, peaks = _sharded_step(wd, c2, read_bytes=read_bytes)
        sh = c2.fetch()
        r3_sh = step3.repath_after_step2(c2, 200)
        peaks["step3"] = dict(library=int(c2.device_peak_bytes()), torch=int(torch.cuda.max_memory_allocated()) - peaks["session_leftover"]["torch"])
        peaks["step3"]["total"] = peaks["step3"]["library"] + peaks["step3"]["torch"]
    _same_step2(F, sh, one)
    assert (st["M"], st["D"], st["S"]) == (st1["M"], st1["D"], st1["S"]) and np.array_equal(np.asarray(st["hist"]), np.asarray(st1["hist"]))
    _properties(F, sh, st, d["n"], np.random.default_rng(5), 0.95)
    # Step 3 (configs[3]'s share): the same large-K graph and the same translated paths behind both Step-2 paths
    assert F.hbv_to_bytes(r3_sh.hbv) == F.hbv_to_bytes(r3_one.hbv)
    assert np.array_equal(r3_sh.path_offset, r3_one.path_offset) and np.array_equal(r3_sh.path_off, r3_one.path_off) and np.array_equal(r3_sh.path_edges, r3_one.path_edges)
    assert np.array_equal(r3_sh.frag_count, r3_one.frag_count) and r3_sh.n_unique_places == r3_one.n_unique_places
    assert r3_sh.n_reads_pathed > 0.9 * d["n"] and r3_sh.hbv.n_edges > 0
    worst = max(p["total"] for p in peaks.values())
    _note("configs2_share", dict(reads=d["n"], genome=glen, kmers_solid=int(st["S"]), exchanges=info["exchanges"], read_bytes=read_bytes,
                                 peak_bytes=peaks, peak_incl_reads_GB=worst / 1e9, step3_unique_places=int(r3_sh.n_unique_places),
                                 step3_large_K_edge_objects=int(r3_sh.hbv.n_edges)))
    assert worst <= 0.9 * HBM_BYTES
    del d
    torch.cuda.empty_cache()


@pytest.mark.parametrize("ids", ["local32", "wide"])
def test_configs4_share_fits_288GB_measured(mods, world1, monkeypatch, ids):
    """250 M reads of a 2.125 Gbp genome (2 B reads and 17 Gbp over 8 GPUs), n_passes = 3, sharded path with the 8-rank shares of queries and
    segments: properties of the result, and the measured device-memory peak of every phase -- caller's reads and exchange buffers included.
    local32: what a rank of the 8-GPU job runs (job-wide node ids are 64-bit words in the sharded path anyway; the rank's 2.0 G k-mers are
    just below 2^31, so its LOCAL ids are 32-bit); wide: the same with 64-bit local ids forced, the shape beyond 2^31 k-mers per rank.
    Either way at most 0.9 x 288 GB (measured in round 6: 246 GB = 0.855, the counting phase; graph phase 220 / 236 GB)."""
    import torch
    F, step2, step3, synth, wd = mods
    free, total = torch.cuda.mem_get_info()
    if total < 250 * 2**30:
        pytest.skip("needs a 288 GB GPU")
    n, glen = 250_000_000, 2_125_000_000
    monkeypatch.setenv("W2RAP_TEST_SHARD_VIRTUAL", "8")
    monkeypatch.setenv("W2RAP_TEST_SHARD_CUT", "27")
    if ids == "wide":
        monkeypatch.setenv("W2RAP_WIDE_IDS", "1")
    d = _reads(synth, n, glen, 4404)
    read_bytes = sum(int(d[k].numel() * d[k].element_size()) for k in ("packed", "quals", "byte_off", "qual_off", "read_len"))
    with step2.Step2Context(0) as c:
        _set(c, d)
        st, info, peaks = _sharded_step(wd, c, n_passes=3, read_bytes=read_bytes)
        if ids == "local32":
            gl = c.good_len().astype(np.int64)
            assert st["M"] == int(np.where(gl > 60, gl - 59, 0).sum())
            del gl
        res = c.fetch()
    assert st["n_passes"] == 3 and st["S"] > 1_900_000_000
    _properties(F, res, st, d["n"], np.random.default_rng(7), 0.9)
    worst = max(p["total"] for p in peaks.values())
    per_solid = {k: (v["library"] / st["S"]) for k, v in peaks.items() if k != "session_leftover"}
    _note(f"configs4_share_{ids}", dict(reads=d["n"], genome=glen, n_passes=3, local_ids=ids, kmer_instances=int(st["M"]), kmers_solid=int(st["S"]), unipaths=int(len(res.fwd_xlat)),
                                        exchanges=info["exchanges"], read_bytes=read_bytes, peak_bytes=peaks, library_peak_bytes_per_solid_kmer=per_solid,
                                        peak_incl_reads_GB=worst / 1e9, hbm_GB=HBM_BYTES / 1e9, frac_of_hbm=worst / HBM_BYTES))
    limit = 0.9
    assert worst <= limit * HBM_BYTES, f"peak {worst / 1e9:.1f} GB of device memory (reads included) exceeds {limit} x 288 GB"
    del d, res
    torch.cuda.empty_cache()
