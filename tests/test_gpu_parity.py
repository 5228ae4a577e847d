"""Parity of the HIP path (through the C ABI) with the reference goldens and the oracle.
Bit-exact: everything here is integer/byte/index work (the single fp64 decay in the extension
scoring is reproduced with unfused IEEE mul/sub)."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, FIXTURES, golden_bytes, load_fixture, relabel_compare

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mods():
    import torch
    assert torch.cuda.is_available(), "the -m gpu tests need an MI355X"
    from w2rap_contigger_amd import formats as F, step2, synth
    from oracle import oracle as O
    assert step2.lib().w2rap_step2_device_count() >= 1
    return F, step2, synth, O


@pytest.fixture(scope="module", params=FIXTURES)
def fx(request):
    return load_fixture(request.param)


@pytest.mark.parametrize("tag", ["ref", "ref8"])
def test_replay_is_byte_exact_vs_reference(mods, fx, tag):
    """one-shot C entry point, PQVec input, reference edge order replayed -> reference's own bytes"""
    F, step2, synth, O = mods
    name = fx["name"]
    hc, ho = O.edge_hint_from_hbv(F.read_hbv(os.path.join(GOLDEN, f"{name}.{tag}.hbv")))
    res = step2.build_read_qgraph(fx["packed"], fx["byte_off"], fx["read_len"], pq=fx["pq"], pq_off=fx["pq_off"],
                                  edge_order_hint=F.pack_bases(hc, ho))
    assert F.hbv_to_bytes(res.hbv) == golden_bytes(name, tag, "hbv")
    assert F.paths_to_bytes(res.path_offset, res.path_off, res.path_edges) == golden_bytes(name, tag, "paths")
    assert F.freqs_text(res.hist).encode() == golden_bytes(name, "ref", "freqs")


def test_stages_match_oracle(mods, fx):
    """a1 good_len, a2-a5 table + histogram, a6 pruned contexts, a7 (edge, offset) per k-mer"""
    F, step2, synth, O = mods
    orc = O.run(fx["codes"], fx["quals"], fx["off"])
    with step2.Step2Context(0) as ctx:
        ctx.set_reads_host(fx["packed"], fx["byte_off"], fx["read_len"], quals=fx["quals"], qual_off=fx["off"])   # raw quals
        st = ctx.count_kmers(7, 4)
        assert np.array_equal(ctx.good_len(), orc.good_len)
        assert (st["M"], st["D"], st["S"]) == (orc.n_instances, orc.n_distinct, len(orc.k_hi))
        assert np.array_equal(st["hist"], orc.hist)
        hi, lo, cnt, c, e, o = ctx.table(st["S"])
        order = np.lexsort((lo, hi))
        assert np.array_equal(hi[order], orc.k_hi) and np.array_equal(lo[order], orc.k_lo)
        assert np.array_equal(cnt[order], orc.k_count) and np.array_equal(c[order], orc.k_ctx)
        ctx.build_graph(None)
        hi, lo, cnt, c, e, o = ctx.table(st["S"])
        assert np.array_equal(e[order], orc.k_edge) and np.array_equal(o[order], orc.k_off)
        ctx.path_reads()
        res = ctx.fetch()
    assert F.hbv_to_bytes(res.hbv) == F.hbv_to_bytes(O.to_hbv(orc))
    assert np.array_equal(res.vleft, orc.left) and np.array_equal(res.vright, orc.right)
    assert np.array_equal(res.fwd_xlat, orc.fwdX) and np.array_equal(res.rev_xlat, orc.revX)
    assert np.array_equal(res.path_offset, orc.path_offset) and np.array_equal(res.path_off, orc.path_off)
    assert np.array_equal(res.path_edges, orc.path_edges)
    assert (res.n_reads_pathed, res.n_reads_multipathed) == (orc.pathed, orc.multipathed)


def test_canonical_mode_matches_reference_modulo_relabelling(mods, fx):
    F, step2, synth, O = mods
    res = step2.build_read_qgraph(fx["packed"], fx["byte_off"], fx["read_len"], pq=fx["pq"], pq_off=fx["pq_off"])
    name = fx["name"]
    relabel_compare(res.hbv, (res.path_offset, res.path_off, res.path_edges), F.read_hbv(os.path.join(GOLDEN, f"{name}.ref.hbv")),
                    F.read_paths(os.path.join(GOLDEN, f"{name}.ref.paths")))


@pytest.mark.parametrize("min_qual,min_freq", [(7, 4), (0, 1), (20, 2), (7, 50), (64, 4)])
def test_parameters_sweep_vs_oracle(mods, min_qual, min_freq):
    F, step2, synth, O = mods
    fx = load_fixture("random20k")
    orc = O.run(fx["codes"], fx["quals"], fx["off"], min_qual=min_qual, min_freq=min_freq)
    res = step2.build_read_qgraph(fx["packed"], fx["byte_off"], fx["read_len"], pq=fx["pq"], pq_off=fx["pq_off"],
                                  min_qual=min_qual, min_freq=min_freq)
    assert np.array_equal(res.hist, orc.hist)
    assert F.hbv_to_bytes(res.hbv) == F.hbv_to_bytes(O.to_hbv(orc))
    assert np.array_equal(res.path_offset, orc.path_offset) and np.array_equal(res.path_edges, orc.path_edges)


def _random_case(synth, seed, n_pairs, glen, extra_edge_cases):
    rng = np.random.default_rng(seed)
    contigs = [rng.integers(0, 4, glen, dtype=np.uint8)]
    if seed % 2:                                       # low-complexity + tandem repeat stretches
        contigs[0][100:400] = 0
        contigs[0][1000:1600] = np.tile(np.array([0, 1], np.uint8), 300)
        contigs[0][2000:2900] = np.tile(rng.integers(0, 4, 30, dtype=np.uint8), 30)
    codes, quals = synth.sample_reads(contigs, n_pairs, seed + 1)
    reads = [(codes[i].numpy(), quals[i].numpy()) for i in range(len(codes))]
    if extra_edge_cases:
        reads += synth.edge_case_reads(rng, contigs[0])
    order = rng.permutation(len(reads))
    reads = [reads[i] for i in order]
    lens = np.array([len(r[0]) for r in reads], np.uint64)
    off = np.zeros(len(reads) + 1, np.uint64); np.cumsum(lens, out=off[1:])
    return np.concatenate([r[0] for r in reads]), np.concatenate([r[1] for r in reads]), off


@pytest.mark.parametrize("seed", range(6))
def test_random_genomes_vs_oracle(mods, seed):
    """ragged, shuffled reads incl. the hand-made quirk reads, repeats and low-complexity sequence"""
    F, step2, synth, O = mods
    codes, quals, off = _random_case(synth, 100 + seed, 1500 + 500 * seed, 6000 + 3000 * seed, seed % 3 != 2)
    orc = O.run(codes, quals, off)
    pk, bo, ln = F.pack_bases(codes, off)
    res = step2.build_read_qgraph(pk, bo, ln, quals=quals, qual_off=off)
    assert np.array_equal(res.hist, orc.hist)
    assert F.hbv_to_bytes(res.hbv) == F.hbv_to_bytes(O.to_hbv(orc))
    assert np.array_equal(res.path_offset, orc.path_offset) and np.array_equal(res.path_off, orc.path_off)
    assert np.array_equal(res.path_edges, orc.path_edges)


def test_empty_and_degenerate_inputs(mods):
    F, step2, synth, O = mods
    z8, z64 = np.zeros(0, np.uint8), np.zeros(1, np.uint64)
    res = step2.build_read_qgraph(z8, z64, np.zeros(0, np.uint32), quals=z8, qual_off=z64)        # no reads at all
    assert res.hbv.n_edges == 0 and res.hbv.n_vertices == 0 and len(res.path_offset) == 0 and res.n_kmer_instances == 0
    # reads that are all too short / all low quality: no k-mers, every path empty
    rng = np.random.default_rng(3)
    lens = np.array([0, 10, 59, 60, 150, 150], np.uint64)
    off = np.zeros(7, np.uint64); np.cumsum(lens, out=off[1:])
    codes = rng.integers(0, 4, int(off[-1])).astype(np.uint8)
    quals = np.full(int(off[-1]), 2, np.uint8)
    quals[int(off[3]):int(off[4])] = 40                 # the 60-base read is all good: still nothing (len > K strict)
    pk, bo, ln = F.pack_bases(codes, off)
    res = step2.build_read_qgraph(pk, bo, ln, quals=quals, qual_off=off)
    orc = O.run(codes, quals, off)
    assert res.n_kmer_instances == 0 and orc.n_instances == 0 and res.hbv.n_edges == 0
    assert np.array_equal(res.path_off, np.zeros(7, np.uint64)) and np.array_equal(res.path_offset, orc.path_offset)


def test_wrong_hint_is_rejected(mods):
    F, step2, synth, O = mods
    fx = load_fixture("random20k")
    hc, ho = O.edge_hint_from_hbv(F.read_hbv(os.path.join(GOLDEN, "random20k.ref.hbv")))
    hc = hc.copy(); hc[3] ^= 2
    with pytest.raises(step2.Step2Error) as e:
        step2.build_read_qgraph(fx["packed"], fx["byte_off"], fx["read_len"], pq=fx["pq"], pq_off=fx["pq_off"],
                                edge_order_hint=F.pack_bases(hc, ho))
    assert e.value.code == 7


def test_lds_table_overflow_path(mods):
    """a bucket whose distinct set exceeds the LDS table is split by hash and recounted: force it with
    few reads (one bucket) that contain > 4096 distinct solid k-mers"""
    F, step2, synth, O = mods
    rng = np.random.default_rng(9)
    g = rng.integers(0, 4, 20_000, dtype=np.uint8)
    reads = []
    for s in range(0, 20_000 - 150, 30):               # tiling reads, 5 copies each
        reads += [g[s:s + 150]] * 5
    codes = np.concatenate(reads); n = len(reads)
    off = np.arange(n + 1, dtype=np.uint64) * 150
    quals = np.full(len(codes), 35, np.uint8)
    orc = O.run(codes, quals, off, stop_after=1)
    with step2.Step2Context(0) as ctx:
        ctx.set_reads_host(*F.pack_bases(codes, off), quals=quals, qual_off=off)
        m = ctx.quality_windows(7)
        recs, nrec, cnts, per = ctx.partition(1, 1)     # everything in ONE bucket
        assert ctx.kmers_per_part == [m]
        st = ctx.count_records(4, 1, 1, recs, cnts, m)
        assert st["D"] == orc.n_distinct and st["S"] == len(orc.k_hi) and np.array_equal(st["hist"], orc.hist)


def test_properties_at_scale(mods):
    """2 M reads (too slow for the oracle in a unit test): size-independent invariants"""
    import torch
    F, step2, synth, O = mods
    d = synth.generate_reads_device(2_000_000, 10_000_000, 77, device="cuda")
    torch.cuda.synchronize()
    with step2.Step2Context(0) as ctx:
        ctx.set_reads_device(d["n"], d["packed"].data_ptr(), d["byte_off"].data_ptr(), d["read_len"].data_ptr(),
                             d["quals"].data_ptr(), d["qual_off"].data_ptr(), keepalive=d)
        st = ctx.count_kmers(7, 4)
        gl = ctx.good_len().astype(np.int64)
        assert st["M"] == int(np.where(gl > 60, gl - 59, 0).sum())
        assert int(st["hist"].sum()) == st["D"] and int(st["hist"][4:].sum()) == st["S"]
        # no count saturates here, so sum i*hist[i] is the number of instances
        assert int((np.arange(101, dtype=np.uint64) * st["hist"]).sum()) == st["M"]
        hi, lo, cnt, c, e, o = ctx.table(st["S"])
        keys = np.stack([hi, lo], axis=1)
        assert len(np.unique(keys, axis=0)) == st["S"]                   # distinct
        ctx.build_graph(None); ctx.path_reads()
        hi, lo, cnt, c, e, o = ctx.table(st["S"])
        res = ctx.fetch()
    h = res.hbv
    E = len(res.fwd_xlat)
    # every k-mer lies on exactly one edge position: sum of edge k-mers == S
    assert int((h.edge_len[res.fwd_xlat].astype(np.int64) - 59).sum()) == st["S"]
    assert (e >= 0).all() and (e < E).all()
    assert len(np.unique(e.astype(np.int64) << 32 | o.astype(np.int64))) == st["S"]
    # every non-palindromic edge has its reverse complement as the next object
    codes, off = h.edge_codes(); off = off.astype(np.int64)
    for x in range(0, min(E, 200)):
        f, r = res.fwd_xlat[x], res.rev_xlat[x]
        a = codes[off[f]:off[f + 1]]; b = codes[off[r]:off[r + 1]]
        assert np.array_equal(a, 3 - b[::-1])
    # unipath order is lexicographic; vertices consistent with adjacency
    firsts = [codes[off[res.fwd_xlat[x]]:off[res.fwd_xlat[x]] + 60].tobytes() for x in range(E)]
    assert firsts == sorted(firsts)
    # FixPaths invariant: consecutive path edges are adjacent
    po = res.path_off.astype(np.int64)
    lens = np.diff(po)
    multi = np.nonzero(lens > 1)[0]
    for i in multi[:2000]:
        p = res.path_edges[po[i]:po[i + 1]]
        assert (res.vright[p[:-1]] == res.vleft[p[1:]]).all()
    assert res.n_reads_pathed == int((lens > 0).sum()) or res.n_reads_pathed >= int((lens > 0).sum())
    assert res.n_reads_pathed > 0.95 * d["n"]


def test_distributed_path_world1_equals_single(mods):
    """the multi-GPU building blocks (quality_windows -> partition -> count_records -> set_solid), chained on one
    GPU with 3 bucket owners emulated as 3 segments, give the same dictionary as count_kmers"""
    import torch
    F, step2, synth, O = mods
    from w2rap_contigger_amd import dist as wd
    fx = load_fixture("repeats_snps")
    orc = O.run(fx["codes"], fx["quals"], fx["off"])
    with step2.Step2Context(0) as ctx:
        ctx.set_reads_host(fx["packed"], fx["byte_off"], fx["read_len"], pq=fx["pq"], pq_off=fx["pq_off"])
        m = ctx.quality_windows(7)
        nb = ctx.default_buckets(m, 3)
        assert nb % 3 == 0
        recs, nrec, cnts, per = ctx.partition(nb, 3)
        assert sum(per) == nrec and sum(ctx.kmers_per_part) == m
        be = wd.GpuBackend(ctx, "cuda:0")
        rb = int(step2.lib().w2rap_step2_record_bytes())
        r = wd.dev_bytes(recs, nrec * rb, be.device).view(nrec, rb).clone()
        c = wd.dev_bytes(cnts, nb * 4, be.device).view(torch.int32).clone()
        # owner g counts its nb/3 buckets from one segment; gather the three solid sets
        his, los, ccs, hist, D = [], [], [], np.zeros(101, np.uint64), 0
        nbl = nb // 3
        start = 0
        for g in range(3):
            seg = r[start:start + per[g]].contiguous(); start += per[g]
            st = be.count_records(4, nbl, 1, seg, c[g * nbl:(g + 1) * nbl].contiguous(), ctx.kmers_per_part[g])
            hi, lo, cc = be.solid()
            his.append(hi.clone()); los.append(lo.clone()); ccs.append(cc.clone())
            hist += st["hist"]; D += st["D"]
        ghi, glo, gcc = torch.cat(his), torch.cat(los), torch.cat(ccs)
        be.set_solid(ghi, glo, gcc, m, D, hist)
        assert np.array_equal(hist, orc.hist) and D == orc.n_distinct and ghi.numel() == len(orc.k_hi)
        ctx.build_graph(None); ctx.path_reads()
        res = ctx.fetch()
    assert F.hbv_to_bytes(res.hbv) == F.hbv_to_bytes(O.to_hbv(orc))
    assert np.array_equal(res.path_edges, orc.path_edges) and np.array_equal(res.path_offset, orc.path_offset)


def test_count_records_from_three_source_segments(mods):
    """the owner-side counting of the multi-GPU path: every bucket's records arrive as THREE bucket-grouped
    segments (one per source rank) laid back to back; the kernel reads them as one logical record stream"""
    import torch
    F, step2, synth, O = mods
    from w2rap_contigger_amd import dist as wd
    fx = load_fixture("repeats_snps")
    orc = O.run(fx["codes"], fx["quals"], fx["off"], stop_after=1)
    n = len(fx["read_len"])
    cuts = [0, n // 5, n // 2 + 1, n]                     # three uneven shards of the reads
    off = fx["off"].astype(np.int64)
    segs, cnts, m_total, nb = [], [], 0, None
    with step2.Step2Context(0) as ctx:
        for g in range(3):
            a, b = cuts[g], cuts[g + 1]
            o = (off[a:b + 1] - off[a]).astype(np.uint64)
            pk, bo, ln = F.pack_bases(fx["codes"][off[a]:off[b]], o)
            ctx.set_reads_host(pk, bo, ln, quals=fx["quals"][off[a]:off[b]], qual_off=o)
            m_total += ctx.quality_windows(7)
            if nb is None:
                nb = 7                                     # few buckets: several hundred records each (more than one tile)
            recs, nrec, c, per = ctx.partition(nb, 1)
            segs.append(wd.dev_bytes(recs, nrec * int(step2.lib().w2rap_step2_record_bytes()), "cuda:0").clone())
            cnts.append(wd.dev_bytes(c, nb * 4, "cuda:0").clone())
        r = torch.cat(segs).contiguous(); c = torch.cat(cnts).contiguous()
        torch.cuda.synchronize()
        st = ctx.count_records(4, nb, 3, r.data_ptr(), c.data_ptr(), m_total)
    assert m_total == orc.n_instances
    assert st["D"] == orc.n_distinct and st["S"] == len(orc.k_hi) and np.array_equal(st["hist"], orc.hist)


@pytest.mark.parametrize("k1", ["lane", "wave"])
@pytest.mark.parametrize("spp", ["1", "2"])
def test_descriptor_overflow_list_and_retry(mods, monkeypatch, spp, k1):
    """K1 (either kernel: a lane per read, a wavefront per read) keeps `spp` record descriptors per read pass at fixed positions and
    spills the rest to an overflow list; with 1 or 2 slots nearly everything spills, the list overflows too and the pass is repeated
    with more slots"""
    F, step2, synth, O = mods
    monkeypatch.setenv("W2RAP_SPP", spp)
    monkeypatch.setenv("W2RAP_K1", k1)
    fx = load_fixture("random20k")
    orc = O.run(fx["codes"], fx["quals"], fx["off"], stop_after=1)
    with step2.Step2Context(0) as ctx:
        ctx.set_reads_host(fx["packed"], fx["byte_off"], fx["read_len"], pq=fx["pq"], pq_off=fx["pq_off"])
        st = ctx.count_kmers(7, 4)
    assert st["D"] == orc.n_distinct and st["S"] == len(orc.k_hi) and np.array_equal(st["hist"], orc.hist)


@pytest.mark.parametrize("case,nb", [("random20k", 7), ("random20k", 3001), ("len251", 11), ("ragged", 64)])
def test_both_partition_kernels_cut_the_same_records(mods, monkeypatch, case, nb):
    """K1 has two kernels -- a lane per read (default, reads up to ~315 good bases) and a wavefront per read (longer reads, W2RAP_K1=wave):
    with W2RAP_K1_ALIGN64 (runs end at bucket changes and at k-mer positions that are multiples of 64, the wavefront kernel's cuts; by
    default the lane kernel only cuts full records) the records they cut are the same multiset, bucket by bucket, whatever order the
    histogram atomics hand the ranks out in"""
    import torch
    F, step2, synth, O = mods
    from w2rap_contigger_amd import dist as wd
    if case == "random20k":
        fx = load_fixture(case)
        reads = dict(packed=fx["packed"], byte_off=fx["byte_off"], read_len=fx["read_len"], quals=fx["quals"], qual_off=fx["off"])
    else:
        rng = np.random.default_rng(77)
        contigs = [rng.integers(0, 4, 40_000, dtype=np.uint8)]
        if case == "len251":
            codes, quals = synth.sample_reads(contigs, 2000, 5, read_len=251, insert=400)
            n = codes.shape[0]
            codes = codes.numpy().reshape(-1); quals = quals.numpy().reshape(-1)
            off = np.arange(n + 1, dtype=np.uint64) * 251
        else:                                                 # every length from 0 to 200, in a shuffled order: odd byte offsets, reads of 59, 60, 61 bases
            lens = rng.permutation(np.repeat(np.arange(0, 201), 3))
            off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
            g = contigs[0]
            codes = np.concatenate([g[s:s + l] for s, l in zip(rng.integers(0, 39_000, lens.size), lens)]).astype(np.uint8)
            quals = np.full(codes.size, 30, np.uint8)
        pk, bo, ln = F.pack_bases(codes, off)
        reads = dict(packed=pk, byte_off=bo, read_len=ln, quals=quals, qual_off=off)
    rb = int(step2.lib().w2rap_step2_record_bytes())
    got = {}
    monkeypatch.setenv("W2RAP_K1_ALIGN64", "1")
    for k1 in ("lane", "wave"):
        monkeypatch.setenv("W2RAP_K1", k1)
        with step2.Step2Context(0) as ctx:
            ctx.set_reads_host(reads["packed"], reads["byte_off"], reads["read_len"], quals=reads["quals"], qual_off=reads["qual_off"])
            ctx.quality_windows(7)
            recs, nrec, c, per = ctx.partition(nb, 1)
            r = wd.dev_bytes(recs, nrec * rb, "cuda:0").clone().cpu().numpy().reshape(nrec, rb)
            cnt = wd.dev_bytes(c, nb * 4, "cuda:0").clone().cpu().numpy().view(np.uint32).copy()
        base = [int(x) for x in np.concatenate([[0], np.cumsum(cnt.astype(np.int64))])]
        assert base[-1] == nrec
        rows = [np.sort(np.ascontiguousarray(r[base[b]:base[b + 1]]).view([("v", "V%d" % rb)]).reshape(-1)) for b in range(nb)]
        got[k1] = (nrec, cnt, rows)
    assert got["lane"][0] == got["wave"][0] and got["lane"][0] > 0
    assert np.array_equal(got["lane"][1], got["wave"][1])
    for a, b in zip(got["lane"][2], got["wave"][2]):
        assert np.array_equal(a, b)


def test_overlapped_table_build_and_its_fallback(mods, monkeypatch):
    """count_kmers counts the buckets in four slices and builds the lookup table on a side stream meanwhile, sized from the
    first slice; with the test hook the extrapolated size is too small and the table is rebuilt the plain way.  Both must
    give the oracle's dictionary, graph and paths (the table feeds prune, unipaths and pathing)."""
    import torch
    F, step2, synth, O = mods
    from conftest import synth_reads
    r = synth_reads(400_000, 2_000_000, 5)                                          # ~6800 buckets: the sliced path
    codes, quals, off = r["codes"], r["quals"], r["off"]
    orc = O.run(codes, quals, off)
    ref = F.hbv_to_bytes(O.to_hbv(orc))
    for hook in (False, True):
        if hook:
            monkeypatch.setenv("W2RAP_TEST_SMALL_SCAP", "1")
        with step2.Step2Context(0) as ctx:
            ctx.set_reads_host(r["pk"], r["bo"], r["ln"], quals=quals, qual_off=off)
            st = ctx.count_kmers(7, 4)
            assert (st["M"], st["D"], st["S"]) == (orc.n_instances, orc.n_distinct, len(orc.k_hi))
            ctx.build_graph(None); ctx.path_reads()
            res = ctx.fetch()
        assert F.hbv_to_bytes(res.hbv) == ref
        assert np.array_equal(res.path_offset, orc.path_offset) and np.array_equal(res.path_edges, orc.path_edges)


def test_bench_distributed_code_path_on_one_gpu(mods):
    """bench.py through torch.distributed.run with the multi-GPU code path forced (RCCL process group,
    all_to_all_v of the super-k-mer records, all_gather_v of the solid k-mers) must report the same
    k-mer statistics as the single-GPU path."""
    import json, subprocess, sys
    from conftest import ROOT
    def run(extra_env, launcher):
        env = dict(os.environ, **extra_env)
        cmd = launcher + [os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "0", "--reads", "5e7",
                          "--no-cpu-baseline"]
        out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-2000:]
        return json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    single = run({}, [sys.executable])
    dist = run({"W2RAP_FORCE_DIST": "1"}, [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                                           "--master-addr", "127.0.0.1", "--master-port", "29577"])
    # the bench's own 50 M reads: large enough for the side-stream work of the sliced dictionary build to be still in flight when
    # the next phase starts (a missing stream dependency there showed only at this size -- W2RAP_TEST_NO_APPEND_WAIT=1 brings it
    # back and fails this test); graph and paths must have the same sizes
    for k in ("kmer_instances", "kmers_distinct", "kmers_solid", "unipaths", "edge_objects", "vertices", "reads_pathed", "path_elements"):
        assert single["config"][k] == dist["config"][k], k
    assert dist["n_gpus"] == 1 and dist["value"] > 0 and "roofline" in dist


def test_bench_with_two_ranks_sharing_the_gpu(mods):
    """bench.py --gpus 2 under torch.distributed.run, both ranks on the one GPU of the box (W2RAP_BENCH_SHARE_GPU: gloo instead of RCCL):
    the N > 1 flow of the bench -- one genome of N x 5 x reads bases, reads sharded by rank, replicated graphs compared across the ranks,
    job-wide totals -- gives the k-mer statistics of the same reads on one rank"""
    import json, subprocess, sys
    from conftest import ROOT
    def run(extra_env, launcher, gpus, reads, genome):
        env = dict(os.environ, **extra_env)
        cmd = launcher + [os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--steps", "1", "--warmup", "1", "--reads", str(reads), "--genome", str(genome),
                          "--no-cpu-baseline", "--no-extras"]
        out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
        assert out.returncode == 0, out.stderr[-3000:]
        return json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    two = run({"W2RAP_BENCH_SHARE_GPU": "1"}, [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                                               "--master-addr", "127.0.0.1", "--master-port", "29579"], 2, 1_000_000, 10_000_000)
    assert two["n_gpus"] == 2 and two["scaling"] == "weak" and "sharded over 2 GPUs" in two["config"]["workload"]
    c = two["config"]
    # 2 M reads of a 10 Mbp genome at 30x: nearly every 60-mer of both strands' canonical forms is solid, one unipath set for the job
    assert 9_900_000 < c["kmers_solid"] <= 10_000_000 and c["reads_pathed"] > 1_900_000
    assert two["value"] > 0 and two["roofline"]["kernel"].startswith("k_")


def test_heavy_bucket_rank_overflow_and_count_saturation(mods):
    """40 k identical poly-A reads put > 65535 records into ONE bucket of one batch (K1's 16-bit rank field overflows into the
    overflow list), saturate the count of A^60 at 255 and make it a one-k-mer circle (its successor is itself); a few thousand
    ordinary reads surround them."""
    F, step2, synth, O = mods
    rng = np.random.default_rng(21)
    g = rng.integers(0, 4, 30_000, dtype=np.uint8)
    reads = [np.zeros(150, np.uint8)] * 40_000
    for s in rng.integers(0, 30_000 - 150, 3_000):
        r = g[s:s + 150].copy()
        reads.append(r if rng.random() < 0.5 else (3 - r[::-1]).astype(np.uint8))
    order = rng.permutation(len(reads))
    codes = np.concatenate([reads[i] for i in order]); n = len(reads)
    off = np.arange(n + 1, dtype=np.uint64) * 150
    quals = np.full(len(codes), 35, np.uint8)
    orc = O.run(codes, quals, off)
    pk, bo, ln = F.pack_bases(codes, off)
    res = step2.build_read_qgraph(pk, bo, ln, quals=quals, qual_off=off)
    assert np.array_equal(res.hist, orc.hist) and res.hist[100] >= 1
    assert F.hbv_to_bytes(res.hbv) == F.hbv_to_bytes(O.to_hbv(orc))
    assert np.array_equal(res.path_offset, orc.path_offset) and np.array_equal(res.path_off, orc.path_off)
    assert np.array_equal(res.path_edges, orc.path_edges)


def test_full_parity_on_bench_like_reads(mods):
    """1.1 M reads from the bench generator (0.5 % errors, Q2 tails, both strands) -- enough for the batched partition, the sliced
    dictionary build and every gap shape of read pathing -- against the oracle: histogram, graph bytes and all paths."""
    import torch
    F, step2, synth, O = mods
    from conftest import synth_reads
    r = synth_reads(1_100_000, 5_500_000, 78)
    codes, quals, off, pk, bo, ln = r["codes"], r["quals"], r["off"], r["pk"], r["bo"], r["ln"]
    orc = O.run(codes, quals, off)
    res = step2.build_read_qgraph(pk, bo, ln, quals=quals, qual_off=off)
    assert np.array_equal(res.hist, orc.hist)
    assert F.hbv_to_bytes(res.hbv) == F.hbv_to_bytes(O.to_hbv(orc))
    assert np.array_equal(res.path_offset, orc.path_offset) and np.array_equal(res.path_off, orc.path_off)
    assert np.array_equal(res.path_edges, orc.path_edges)


def test_properties_at_bench_size(mods):
    """BASELINE configs[1] itself -- 50 M reads, 250 Mbp -- through the size-independent properties: k-mer instances from the
    quality windows, histogram sums, every solid k-mer on exactly one unipath position (sum of edge k-mers), every object with
    its reverse complement, lexicographic unipath order, FixPaths adjacency, pathing rate; and the multi-GPU code path (forced at
    world 1 elsewhere) is not needed for any of it."""
    import torch
    F, step2, synth, O = mods
    n = 50_000_000
    g = torch.randint(0, 4, (5 * n,), dtype=torch.uint8, device="cuda", generator=torch.Generator(device="cuda").manual_seed(42))
    d = synth.generate_reads_device(n, 5 * n, 42, device="cuda", genome=g)
    del g; d.pop("genome", None)
    torch.cuda.synchronize(); torch.cuda.empty_cache()
    with step2.Step2Context(0) as ctx:
        ctx.set_reads_device(d["n"], d["packed"].data_ptr(), d["byte_off"].data_ptr(), d["read_len"].data_ptr(),
                             d["quals"].data_ptr(), d["qual_off"].data_ptr(), keepalive=d)
        st = ctx.count_kmers(7, 4)
        gl = ctx.good_len().astype(np.int64)
        assert st["M"] == int(np.where(gl > 60, gl - 59, 0).sum())
        assert int(st["hist"].sum()) == st["D"] and int(st["hist"][4:].sum()) == st["S"]
        if st["hist"][100] == 0:                                       # nothing near saturation: sum i*hist[i] counts the instances
            assert int((np.arange(101, dtype=np.uint64) * st["hist"]).sum()) == st["M"]
        ctx.build_graph(None); ctx.path_reads()
        res = ctx.fetch()
    h = res.hbv
    E = len(res.fwd_xlat)
    assert int((h.edge_len[res.fwd_xlat].astype(np.int64) - 59).sum()) == st["S"]
    assert np.array_equal(h.edge_len[res.fwd_xlat], h.edge_len[res.rev_xlat])
    codes, off = h.edge_codes(); off = off.astype(np.int64)
    rng = np.random.default_rng(5)
    for x in rng.integers(0, E, 300):
        f, r = res.fwd_xlat[x], res.rev_xlat[x]
        a = codes[off[f]:off[f + 1]]; b = codes[off[r]:off[r + 1]]
        assert np.array_equal(a, 3 - b[::-1])
    firsts = [codes[off[res.fwd_xlat[x]]:off[res.fwd_xlat[x]] + 60].tobytes() for x in range(E)]
    assert firsts == sorted(firsts)
    po = res.path_off.astype(np.int64)
    lens = np.diff(po)
    assert res.n_reads_pathed > 0.95 * d["n"] and int((lens > 0).sum()) <= d["n"]
    multi = np.nonzero(lens > 1)[0]
    for i in multi[rng.integers(0, len(multi), 5000)] if len(multi) else []:
        p = res.path_edges[po[i]:po[i + 1]]
        assert (res.vright[p[:-1]] == res.vleft[p[1:]]).all()
    # every path element is an edge object; offsets stay inside the first edge's length window
    assert res.path_edges.min() >= 0 and res.path_edges.max() < h.n_edges


@pytest.mark.parametrize("read_len", [251, 1000])
def test_long_reads_vs_oracle(mods, read_len):
    """reads longer than one pass of 128 k-mer positions (K1 cuts them in several passes, records of up to 64 k-mers chain up,
    pathing walks many seeds and gaps per read): MiSeq-like 2x251 and 1 kb reads with errors and Q2 tails"""
    F, step2, synth, O = mods
    rng = np.random.default_rng(31 + read_len)
    contigs = [rng.integers(0, 4, 60_000, dtype=np.uint8)]
    contigs[0][5000:5600] = np.tile(rng.integers(0, 4, 40, dtype=np.uint8), 15)          # a tandem repeat inside the reads' reach
    codes, quals = synth.sample_reads(contigs, 3000, 17 + read_len, read_len=read_len, insert=max(read_len + 100, 400))
    n = codes.shape[0]
    codes = codes.numpy().reshape(-1); quals = quals.numpy().reshape(-1)
    off = np.arange(n + 1, dtype=np.uint64) * read_len
    orc = O.run(codes, quals, off)
    pk, bo, ln = F.pack_bases(codes, off)
    res = step2.build_read_qgraph(pk, bo, ln, quals=quals, qual_off=off)
    assert orc.n_instances > 0 and np.array_equal(res.hist, orc.hist)
    assert F.hbv_to_bytes(res.hbv) == F.hbv_to_bytes(O.to_hbv(orc))
    assert np.array_equal(res.path_offset, orc.path_offset) and np.array_equal(res.path_off, orc.path_off)
    assert np.array_equal(res.path_edges, orc.path_edges)


def test_inconsistent_host_arrays_are_rejected(mods):
    """w2rap_step2_set_reads checks host arrays before a kernel indexes by them"""
    F, step2, synth, O = mods
    fx = load_fixture("random20k")
    pk, bo, ln = fx["packed"], fx["byte_off"], fx["read_len"]
    bad_bo = bo.copy(); bad_bo[5] += 1
    with pytest.raises(step2.Step2Error, match="base_byte_off does not match"):
        step2.build_read_qgraph(pk, bad_bo, ln, pq=fx["pq"], pq_off=fx["pq_off"])
    bad_ln = ln.copy(); bad_ln[7] += 40
    with pytest.raises(step2.Step2Error, match="does not match read_len"):
        step2.build_read_qgraph(pk, bo, bad_ln, quals=fx["quals"], qual_off=fx["off"])
    bad_po = fx["pq_off"].copy(); bad_po[3] = bad_po[2]
    with pytest.raises(step2.Step2Error, match="pq_off is not ascending"):
        step2.build_read_qgraph(pk, bo, ln, pq=fx["pq"], pq_off=bad_po)
    # a PQVec cut short decodes to quality 0 for the missing values: a defined result, and the run completes
    pq = fx["pq"].copy(); po = fx["pq_off"]
    pq[int(po[10]):int(po[11])] = 0
    res = step2.build_read_qgraph(pk, bo, ln, pq=pq, pq_off=po)
    assert res.hbv.n_edges > 0


# ---------------------------------------------------------------------------------------------- node ids beyond 2^31
def _wide_case(mods, name):
    F, step2, synth, O = mods
    if name in FIXTURES:
        fx = load_fixture(name)
        return fx["codes"], fx["quals"], fx["off"]
    return _random_case(synth, 101, 2000, 9000, True)


@pytest.mark.parametrize("name", list(FIXTURES) + ["random_case"])
def test_wide_node_ids_equal_the_oracle(mods, name, monkeypatch):
    """The library numbers oriented nodes with 32-bit words below 2^31 solid k-mers and with 64-bit words (33-bit ids, 31-bit distances
    in the rank words) beyond; W2RAP_WIDE_IDS=1 forces the wide kernels on any input (BuildReadQGraph.cc:1092: the reference's
    dictionary has no ceiling).  Every stage must equal the oracle, with and without the reference's edge order replayed; the
    palindrome_circle fixture takes the wide min-jumping of the circle code."""
    F, step2, synth, O = mods
    monkeypatch.setenv("W2RAP_WIDE_IDS", "1")
    codes, quals, off = _wide_case(mods, name)
    orc = O.run(codes, quals, off)
    pk, bo, ln = F.pack_bases(codes, off)
    with step2.Step2Context(0) as ctx:
        ctx.set_reads_host(pk, bo, ln, quals=quals, qual_off=off)
        st = ctx.count_kmers(7, 4)
        hi, lo, cnt, c, e, o = ctx.table(st["S"])
        order = np.lexsort((lo, hi))
        assert np.array_equal(hi[order], orc.k_hi) and np.array_equal(c[order], orc.k_ctx)          # pruned contexts (k_prune<u64>)
        ctx.build_graph(None)
        hi, lo, cnt, c, e, o = ctx.table(st["S"])
        assert np.array_equal(e[order], orc.k_edge) and np.array_equal(o[order], orc.k_off)
        ctx.path_reads()
        res = ctx.fetch()
    assert F.hbv_to_bytes(res.hbv) == F.hbv_to_bytes(O.to_hbv(orc))
    assert np.array_equal(res.path_offset, orc.path_offset) and np.array_equal(res.path_off, orc.path_off)
    assert np.array_equal(res.path_edges, orc.path_edges)
    if name in FIXTURES:                                   # replay (k_edge_from_hint<u64>): the reference's own bytes
        fx = load_fixture(name)
        hc, ho = O.edge_hint_from_hbv(F.read_hbv(os.path.join(GOLDEN, f"{name}.ref.hbv")))
        r2 = step2.build_read_qgraph(fx["packed"], fx["byte_off"], fx["read_len"], pq=fx["pq"], pq_off=fx["pq_off"], edge_order_hint=F.pack_bases(hc, ho))
        assert F.hbv_to_bytes(r2.hbv) == golden_bytes(name, "ref", "hbv")
        assert F.paths_to_bytes(r2.path_offset, r2.path_off, r2.path_edges) == golden_bytes(name, "ref", "paths")


@pytest.mark.parametrize("wide", ["0", "1"])
@pytest.mark.parametrize("name", list(FIXTURES) + ["random_case"])
def test_two_level_splitter_ranking_forced_on_small_inputs(mods, name, wide, monkeypatch):
    """The splitter chains are ranked in two levels from 65536 listed splitters on (step2_graph.hip: one splitter in sixteen walks to the
    next such one, those alone jump, a second walk hands the stretch its final words; the plain jumping finishes what the walks leave out).
    W2RAP_RANK_HIER=1 forces it on the fixtures -- chains without a super-splitter, chains that are nothing else, the circles of
    palindrome_circle -- with 32- and 64-bit node ids, on one GPU and behind the sharded graph phase; =0 is the plain jumping alone."""
    F, step2, synth, O = mods
    codes, quals, off = _wide_case(mods, name)
    orc = O.run(codes, quals, off)
    pk, bo, ln = F.pack_bases(codes, off)
    monkeypatch.setenv("W2RAP_WIDE_IDS", wide)
    for hier, devices in (("1", None), ("0", None), ("1", [0, 0])):
        monkeypatch.setenv("W2RAP_RANK_HIER", hier)
        res = step2.build_read_qgraph(pk, bo, ln, quals=quals, qual_off=off, devices=devices)
        assert F.hbv_to_bytes(res.hbv) == F.hbv_to_bytes(O.to_hbv(orc)), (hier, devices)
        assert np.array_equal(res.path_offset, orc.path_offset) and np.array_equal(res.path_off, orc.path_off), (hier, devices)
        assert np.array_equal(res.path_edges, orc.path_edges), (hier, devices)


def test_more_than_2_31_solid_kmers_on_one_gpu(mods):
    """BASELINE configs[2] has ~2.5 G solid k-mers (2.5 Gbp genome); the reference has no ceiling there (new BRQ_Dict(kmers.size()),
    BuildReadQGraph.cc:1092).  One GPU, S > 2^31: 28 M reads of a 10 Gbp genome at min_freq 1 (every distinct k-mer is solid), and the
    size-independent properties of test_properties_at_bench_size.  Needs ~240 GB of HBM."""
    import torch
    F, step2, synth, O = mods
    free, total = torch.cuda.mem_get_info()
    if total < 250 * 2**30:
        pytest.skip("needs a 288 GB GPU")
    n, glen = 28_000_000, 10_000_000_000
    gen = torch.Generator(device="cuda").manual_seed(4242)
    g = torch.empty(glen, dtype=torch.uint8, device="cuda")
    for a in range(0, glen, 1 << 30):                                   # (randint in pieces: its int64 scratch is 8 B per element)
        g[a:a + (1 << 30)] = torch.randint(0, 4, (min(1 << 30, glen - a),), dtype=torch.uint8, device="cuda", generator=gen)
    d = synth.generate_reads_device(n, glen, 4242, device="cuda", genome=g)
    del g; d.pop("genome", None)
    torch.cuda.synchronize(); torch.cuda.empty_cache()
    with step2.Step2Context(0) as ctx:
        ctx.set_reads_device(d["n"], d["packed"].data_ptr(), d["byte_off"].data_ptr(), d["read_len"].data_ptr(),
                             d["quals"].data_ptr(), d["qual_off"].data_ptr(), keepalive=d)
        st = ctx.count_kmers(7, 1)
        assert st["S"] > 2**31, st
        gl = ctx.good_len().astype(np.int64)
        assert st["M"] == int(np.where(gl > 60, gl - 59, 0).sum())
        assert int(st["hist"].sum()) == st["D"] == st["S"]               # min_freq 1: every distinct k-mer is solid
        assert int((np.arange(101, dtype=np.uint64) * st["hist"]).sum()) == st["M"]
        ctx.build_graph(None); ctx.path_reads()
        cnts = ctx.counts()
        res = ctx.fetch()
    h = res.hbv
    E = len(res.fwd_xlat)
    assert cnts["unipaths"] == E and h.n_edges == cnts["edge_objects"]
    # every solid k-mer lies on exactly one unipath position
    assert int((h.edge_len[res.fwd_xlat].astype(np.int64) - 59).sum()) == st["S"]
    assert np.array_equal(h.edge_len[res.fwd_xlat], h.edge_len[res.rev_xlat])
    rng = np.random.default_rng(5)
    ebo = h.edge_byte_off.astype(np.int64)
    def obj(o):                                                         # bases of one edge object
        a, b = int(ebo[o]), int(ebo[o + 1])
        return F.unpack_bases(h.edge_packed[a:b], np.array([0, b - a], np.uint64), np.array([h.edge_len[o]], np.uint32))[0]
    sample = rng.integers(0, E, 300)
    firsts = {}
    for x in sample:
        f, r = int(res.fwd_xlat[x]), int(res.rev_xlat[x])
        a, b = obj(f), obj(r)
        assert np.array_equal(a, 3 - b[::-1])                            # every object with its reverse complement
        firsts[int(x)] = a[:60].tobytes()
    xs = sorted(firsts)
    assert [firsts[x] for x in xs] == sorted(firsts[x] for x in xs)     # unipaths in lexicographic order
    po = res.path_off.astype(np.int64)
    lens = np.diff(po)
    assert int((lens > 0).sum()) <= d["n"] and res.n_reads_pathed > 0.9 * d["n"]
    assert res.path_edges.min() >= 0 and res.path_edges.max() < h.n_edges
    multi = np.nonzero(lens > 1)[0]
    for i in (multi[rng.integers(0, len(multi), 5000)] if len(multi) else []):
        p = res.path_edges[po[i]:po[i + 1]]
        assert (res.vright[p[:-1]] == res.vleft[p[1:]]).all()            # FixPaths adjacency


@pytest.mark.parametrize("cfg,kpb", [("0", None), ("20", None), ("21", None), ("22", None), ("23", None), ("24", None), ("20", "30000"), ("22", "12000")])
def test_count_kernel_shapes_equal_the_oracle(mods, fx, monkeypatch, cfg, kpb):
    """the counting kernel in each of its shapes -- the round-1..3 kernel (full keys in LDS), the round-4 kernel (tag + reference slots over the
    bucket's resident records; two blocks per CU) with its tile / block variants -- and with buckets so large that the round-4 kernel DEFERS
    most of them to the list pass of the old one: table, contexts and histogram equal the oracle's every time"""
    F, step2, synth, O = mods
    monkeypatch.setenv("W2RAP_K3", cfg)
    if kpb:
        monkeypatch.setenv("W2RAP_KPB", kpb)
    orc = O.run(fx["codes"], fx["quals"], fx["off"], stop_after=1)
    with step2.Step2Context(0) as ctx:
        ctx.set_reads_host(fx["packed"], fx["byte_off"], fx["read_len"], quals=fx["quals"], qual_off=fx["off"])
        st = ctx.count_kmers(7, 4)
        assert (st["M"], st["D"], st["S"]) == (orc.n_instances, orc.n_distinct, len(orc.k_hi))
        assert np.array_equal(st["hist"], orc.hist)
        hi, lo, cnt, c, e, o = ctx.table(st["S"])
        order = np.lexsort((lo, hi))
        assert np.array_equal(hi[order], orc.k_hi) and np.array_equal(lo[order], orc.k_lo)
        assert np.array_equal(cnt[order], orc.k_count)


def _tail_into_interior_case():
    """a read that follows unipath U to its END and whose LOW-QUALITY tail goes on into the INTERIOR of another unipath W: the 59-mer S ends U (as
    a S) and lies inside W (as a' S b); inside quality windows a S is never followed by b, so the contexts hold no adjacency U -> W, W does not
    leave U's end vertex, and yet the k-mer S b behind U's end is solid.  The reference's pather looks every read k-mer up (BuildReadQGraph.cc:510-513)"""
    rng = np.random.default_rng(77)
    S = rng.integers(0, 4, 59, dtype=np.uint8)
    A = np.concatenate([rng.integers(0, 4, 400, dtype=np.uint8), [0], S])                          # ... a S   (the fragment ends with S)
    B = np.concatenate([rng.integers(0, 4, 300, dtype=np.uint8), [1], S, [2], rng.integers(0, 4, 300, dtype=np.uint8)])   # ... a' S b ...
    reads, quals = [], []
    for g in (A, B):
        for s in range(0, len(g) - 150 + 1, 7):
            for _ in range(3):
                reads.append(g[s:s + 150]); quals.append(np.full(150, 35, np.uint8))
        for _ in range(3):
            reads.append(g[len(g) - 150:]); quals.append(np.full(150, 35, np.uint8))
    # the chimeric reads: the last 100 bases of A, then b and W's continuation at quality 2
    tail = B[300 + 1 + 59:300 + 1 + 59 + 50]
    for _ in range(6):
        reads.append(np.concatenate([A[-100:], tail])); quals.append(np.concatenate([np.full(100, 35, np.uint8), np.full(50, 2, np.uint8)]))
    if len(reads) % 2:
        reads.append(reads[0]); quals.append(quals[0])
    codes = np.concatenate(reads).astype(np.uint8); q = np.concatenate(quals)
    off = np.arange(len(reads) + 1, dtype=np.uint64) * 150
    return codes, q, off


def test_low_quality_tail_into_the_interior_of_another_unipath(mods):
    F, step2, synth, O = mods
    codes, q, off = _tail_into_interior_case()
    orc = O.run(codes, q, off)
    pk, bo, ln = F.pack_bases(codes, off)
    res = step2.build_read_qgraph(pk, bo, ln, quals=q, qual_off=off)
    assert F.hbv_to_bytes(res.hbv) == F.hbv_to_bytes(O.to_hbv(orc))
    assert np.array_equal(res.path_offset, orc.path_offset) and np.array_equal(res.path_off, orc.path_off) and np.array_equal(res.path_edges, orc.path_edges)


def test_second_build_graph_on_one_count_is_a_state_error(mods, fx):
    """build_graph consumes the neighbour links of the count (include/w2rap_step2.h): a second call says so instead of reading freed memory"""
    F, step2, synth, O = mods
    with step2.Step2Context(0) as ctx:
        ctx.set_reads_host(fx["packed"], fx["byte_off"], fx["read_len"], quals=fx["quals"], qual_off=fx["off"])
        ctx.count_kmers(7, 4); ctx.build_graph(None)
        with pytest.raises(step2.Step2Error) as e:
            ctx.build_graph(None)
        assert e.value.code == 4
        ctx.count_kmers(7, 4); ctx.build_graph(None); ctx.path_reads()          # counting again makes it valid again


@pytest.mark.parametrize("env", [{"W2RAP_TEST_FP_LIMIT": "48"}, {"W2RAP_TEST_FP_SC": "6"}, {"W2RAP_TEST_FP_LIMIT": "200", "W2RAP_TEST_FP_SC": "20", "W2RAP_K3": "22"},
                                 {"W2RAP_TEST_FP_LIMIT": "16", "W2RAP_KPB": "20000"}])
def test_count_kernel_hash_classes(mods, fx, monkeypatch, env):
    """k_count_fp counts a bucket whose distinct set does not fit its table -- or whose solid set does not fit its staging area -- in hash classes
    (MapReduceEngine.h:288-291), refining a class again when it still does not fit; the hooks shrink both limits so that nearly every bucket of
    the fixtures takes that path, several levels deep: table, contexts and histogram equal the oracle's"""
    F, step2, synth, O = mods
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    orc = O.run(fx["codes"], fx["quals"], fx["off"])          # (the whole oracle: the table's contexts are the PRUNED ones)
    with step2.Step2Context(0) as ctx:
        ctx.set_reads_host(fx["packed"], fx["byte_off"], fx["read_len"], quals=fx["quals"], qual_off=fx["off"])
        st = ctx.count_kmers(7, 4)
        assert (st["M"], st["D"], st["S"]) == (orc.n_instances, orc.n_distinct, len(orc.k_hi))
        assert np.array_equal(st["hist"], orc.hist)
        hi, lo, cnt, c, e, o = ctx.table(st["S"])
        order = np.lexsort((lo, hi))
        assert np.array_equal(hi[order], orc.k_hi) and np.array_equal(lo[order], orc.k_lo)
        assert np.array_equal(cnt[order], orc.k_count) and np.array_equal(c[order], orc.k_ctx)


@pytest.mark.parametrize("env", [{}, {"W2RAP_TEST_FP_LIMIT": "64"}, {"W2RAP_KPB": "30000"}, {"W2RAP_K3": "22", "W2RAP_TEST_FP_SC": "12"}])
def test_fused_chunk_local_prune_equals_the_oracle(mods, fx, monkeypatch, env):
    """W2RAP_FUSED_PRUNE=1: k_count_fp does the chunk-local adjacency prune in its emit (the table still holds every distinct k-mer of the bucket);
    the list kernel's chunks keep k_prune_local, the open bits k_prune -- pruned contexts, (edge, offset) per k-mer, graph and paths as the oracle's,
    also when buckets are counted in hash classes or deferred"""
    F, step2, synth, O = mods
    monkeypatch.setenv("W2RAP_FUSED_PRUNE", "1")
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    orc = O.run(fx["codes"], fx["quals"], fx["off"])
    with step2.Step2Context(0) as ctx:
        ctx.set_reads_host(fx["packed"], fx["byte_off"], fx["read_len"], quals=fx["quals"], qual_off=fx["off"])
        st = ctx.count_kmers(7, 4)
        assert (st["M"], st["D"], st["S"]) == (orc.n_instances, orc.n_distinct, len(orc.k_hi))
        hi, lo, cnt, c, e, o = ctx.table(st["S"])
        order = np.lexsort((lo, hi))
        assert np.array_equal(hi[order], orc.k_hi) and np.array_equal(lo[order], orc.k_lo)
        assert np.array_equal(cnt[order], orc.k_count) and np.array_equal(c[order], orc.k_ctx)
        ctx.build_graph(None)
        hi, lo, cnt, c, e, o = ctx.table(st["S"])
        assert np.array_equal(e[order], orc.k_edge) and np.array_equal(o[order], orc.k_off)
        ctx.path_reads()
        res = ctx.fetch()
    assert F.hbv_to_bytes(res.hbv) == F.hbv_to_bytes(O.to_hbv(orc))
    assert np.array_equal(res.path_offset, orc.path_offset) and np.array_equal(res.path_edges, orc.path_edges)
