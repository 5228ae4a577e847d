"""The whole multi-GPU flow with TWO real ranks that share the one GPU of the test box: every rank drives its own
context of libw2rap_step2.so (its shard of the reads, the super-k-mer partition, owner-side counting of two source
segments, the replicated graph build, local pathing); only the transport differs from production -- gloo through host
memory instead of RCCL over xGMI (dist._host_staged).  Result: the same dictionary, graph and paths as one rank."""
import os

import numpy as np
import pytest

from conftest import load_fixture, run_ranks

pytestmark = pytest.mark.gpu


def _case(name):
    """a golden fixture, or bench-like synthetic reads big enough for the library to count in four bucket slices (so that the
    sliced record exchange and the incremental dictionary are exercised with more than one rank)"""
    if not name.startswith("synth"):
        return load_fixture(name)
    import torch
    from w2rap_contigger_amd import synth
    n = int(name[5:])
    d = synth.generate_reads_device(n, 5 * n, 79, device="cuda")
    codes = synth.unpack_fixed(d["packed"], synth.READ_LEN).cpu().numpy().reshape(-1)
    quals = d["quals"].cpu().numpy().reshape(-1)
    off = np.arange(d["n"] + 1, dtype=np.uint64) * synth.READ_LEN
    del d
    torch.cuda.empty_cache()
    return dict(codes=codes, quals=quals, off=off, read_len=np.full(len(off) - 1, synth.READ_LEN, np.uint32))


def _worker(rank, world, port, name, headroom, q):
    n_passes = 1
    sharded = False
    if isinstance(headroom, str) and headroom.startswith("sharded"):     # "sharded" or "sharded<passes>"
        sharded = True
        n_passes = int(headroom[7:] or 1)
        headroom = None
    if headroom == "wide":                   # 64-bit node ids and 33-bit rank words, as beyond 2^31 solid k-mers (BASELINE configs[2] replicated)
        os.environ["W2RAP_WIDE_IDS"] = "1"
        headroom = None
    if isinstance(headroom, str) and headroom.startswith("passes"):      # counting in hash-range passes (MapReduceEngine.h:286-299) with bucket owners
        n_passes = int(headroom[6:])
        headroom = None
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from w2rap_contigger_amd import dist as wd, formats as F, step2
        if headroom:
            wd.DICT_HEADROOM = headroom      # < 1: the incremental dictionary runs out of capacity -> dict_abort + classic gather
        fx = _case(name)
        n = len(fx["read_len"])
        cut = (n // world // 2) * 2
        lo_r, hi_r = rank * cut, (n if rank == world - 1 else (rank + 1) * cut)      # whole pairs per rank
        off = fx["off"].astype(np.int64)
        o = (off[lo_r:hi_r + 1] - off[lo_r]).astype(np.uint64)
        pk, bo, ln = F.pack_bases(fx["codes"][off[lo_r]:off[hi_r]], o)
        with step2.Step2Context(0) as ctx:
            ctx.set_reads_host(pk, bo, ln, quals=fx["quals"][off[lo_r]:off[hi_r]], qual_off=o)
            be = wd.GpuBackend(ctx, torch.device("cuda", 0))
            if sharded:                      # row e-3: every owner keeps its k-mers; dictionary, prune and unipaths sharded (the state machine of step2_shard.hip)
                st = wd.distributed_count(be, 7, 4, n_passes=n_passes, gather=False)
                info = wd.sharded_graph(be, st["S_local"], st, st["n_buckets"], n_passes=n_passes)
                assert info["solid_total"] == st["S"]
                si = ctx.shard_info()
                # this rank's share of the dictionary: its own solid k-mers only (the owners' bucket ranges are hash-uniform)
                assert si["solid_local"] <= 1.3 * si["solid_total"] / world + 64, si
            else:
                st = wd.distributed_count(be, 7, 4, n_passes=n_passes)
                ctx.build_graph(None)
            ctx.path_reads()
            res = ctx.fetch()
            r3 = wd.distributed_repath(ctx, 200)                         # Step 3 behind it: reads stay sharded, the large-K graph is replicated
        q.put((rank, lo_r, hi_r, st["M"], st["D"], st["S"], bool(st["fallback"]), st["hist"].tolist(), F.hbv_to_bytes(res.hbv),
               res.path_offset.copy(), res.path_off.copy(), res.path_edges.copy(),
               (F.hbv_to_bytes(r3.hbv), r3.path_offset.copy(), r3.path_off.copy(), r3.path_edges.copy(), r3.n_unique_places,
                np.asarray(r3.frag_count).copy(), r3.n_reads_pathed)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("name,world,headroom", [("repeats_snps", 2, None), ("repeats_snps", 3, None), ("synth1200000", 2, None),
                                                 ("synth1200000", 2, 0.5), ("synth1200000", 2, "wide"), ("repeats_snps", 2, "passes3"),
                                                 ("palindrome_circle", 3, "passes2"), ("synth1200000", 2, "passes3"),
                                                 ("repeats_snps", 2, "sharded"), ("palindrome_circle", 3, "sharded"), ("random20k", 4, "sharded2"),
                                                 ("synth1200000", 2, "sharded"), ("synth1200000", 3, "sharded3")])
def test_two_ranks_on_one_gpu_match_the_oracle(name, world, headroom):
    """headroom < 1: the capacity guessed from the first bucket slice is too small, so the sliced dictionary build is aborted
    on the GPU (dict_abort frees the half-built table and the gathered blocks) and the classic whole-set gather takes over"""
    from w2rap_contigger_amd import formats as F
    from oracle import oracle as O
    outs = run_ranks(_worker, world, (name, headroom), timeout=300)
    fx = _case(name)
    orc = O.run(fx["codes"], fx["quals"], fx["off"])
    ref_hbv = F.hbv_to_bytes(O.to_hbv(orc))
    po = orc.path_off.astype(np.int64)
    from oracle import oracle3 as O3
    o3 = O3.run(O.to_hbv(orc), (orc.path_offset, orc.path_off, orc.path_edges), 200)
    ref3 = F.hbv_to_bytes(O3.to_hbv(o3))
    po3 = o3.path_off.astype(np.int64)
    for rank, lo_r, hi_r, M, D, S, fallback, hist, hbv, p_offset, p_off, p_edges, s3 in outs:
        assert s3[0] == ref3 and s3[4] == len(o3.place_off) - 1        # the large-K graph of ALL reads on every rank
        # FragDist and the pathed counter are JOB-wide on every rank (all-reduced: the shards hold whole pairs)
        assert np.array_equal(s3[5].astype(np.int64), np.asarray(o3.frag).astype(np.int64))
        assert s3[6] == int((np.diff(po) > 0).sum())                 # "reads pathed" of Repath.cc:36-72: non-empty INPUT paths, job-wide
        assert np.array_equal(s3[1], o3.path_offset[lo_r:hi_r]) and np.array_equal(s3[2].astype(np.int64), po3[lo_r:hi_r + 1] - po3[lo_r])
        assert np.array_equal(s3[3], o3.path_edges[po3[lo_r]:po3[hi_r]])
        assert fallback == isinstance(headroom, float)
        assert (M, D, S) == (orc.n_instances, orc.n_distinct, len(orc.k_hi)) and hist == [int(x) for x in orc.hist]
        assert hbv == ref_hbv                                         # the replicated graph, canonical numbering
        assert np.array_equal(p_offset, orc.path_offset[lo_r:hi_r])  # this rank's reads
        assert np.array_equal(p_off.astype(np.int64), po[lo_r:hi_r + 1] - po[lo_r])
        assert np.array_equal(p_edges, orc.path_edges[po[lo_r]:po[hi_r]])
