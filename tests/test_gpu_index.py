"""Read pathing through the minimizer-sampled index over the edge sequences (common.h EdgeIndex) instead of the k-mer dictionary:
what the sharded graph phase (row e-3) uses, switched on for ONE GPU with W2RAP_PATH_INDEX=1.  Same bytes as the reference."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, FIXTURES, golden_bytes, load_fixture

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mods():
    import torch
    assert torch.cuda.is_available(), "the -m gpu tests need an MI355X"
    from w2rap_contigger_amd import formats as F, step2, synth
    from oracle import oracle as O
    return F, step2, synth, O


@pytest.mark.parametrize("name", FIXTURES)
@pytest.mark.parametrize("wave", ["2", "0"])
def test_index_pathing_is_byte_exact_vs_reference(mods, name, wave, monkeypatch):
    F, step2, synth, O = mods
    monkeypatch.setenv("W2RAP_PATH_INDEX", "1")
    monkeypatch.setenv("W2RAP_PATH_WAVE", wave)
    monkeypatch.setenv("W2RAP_PATH_BUDGET", "2")                    # most reads of the fixtures take the second-stage kernels too
    fx = load_fixture(name)
    hc, ho = O.edge_hint_from_hbv(F.read_hbv(os.path.join(GOLDEN, f"{name}.ref.hbv")))
    res = step2.build_read_qgraph(fx["packed"], fx["byte_off"], fx["read_len"], pq=fx["pq"], pq_off=fx["pq_off"], edge_order_hint=F.pack_bases(hc, ho))
    assert F.hbv_to_bytes(res.hbv) == golden_bytes(name, "ref", "hbv")
    assert F.paths_to_bytes(res.path_offset, res.path_off, res.path_edges) == golden_bytes(name, "ref", "paths")


def test_index_finds_every_solid_kmer_where_the_oracle_puts_it(mods, monkeypatch):
    """(edge, offset) of every solid k-mer looked up through the index == the oracle's KDef; and 1.2 M bench-like reads path identically"""
    F, step2, synth, O = mods
    monkeypatch.setenv("W2RAP_PATH_INDEX", "1")
    from conftest import synth_reads
    r = synth_reads(600_000, 3_000_000, 11)
    codes, quals, off, pk, bo, ln = r["codes"], r["quals"], r["off"], r["pk"], r["bo"], r["ln"]
    orc = O.run(codes, quals, off)
    with step2.Step2Context(0) as ctx:
        ctx.set_reads_host(pk, bo, ln, quals=quals, qual_off=off)
        st = ctx.count_kmers(7, 4)
        ctx.build_graph(None)
        hi, lo, cnt, c, e, o = ctx.table(st["S"])
        order = np.lexsort((lo, hi))
        assert np.array_equal(e[order], orc.k_edge) and np.array_equal(o[order], orc.k_off)
        ctx.path_reads()
        res = ctx.fetch()
    assert F.hbv_to_bytes(res.hbv) == F.hbv_to_bytes(O.to_hbv(orc))
    assert np.array_equal(res.path_off, orc.path_off) and np.array_equal(res.path_edges, orc.path_edges) and np.array_equal(res.path_offset, orc.path_offset)


def test_index_table_that_starts_too_small_is_rebuilt(mods, monkeypatch):
    F, step2, synth, O = mods
    monkeypatch.setenv("W2RAP_PATH_INDEX", "1")
    monkeypatch.setenv("W2RAP_TEST_INDEX_SMALL", "1")
    fx = load_fixture(FIXTURES[0])
    hc, ho = O.edge_hint_from_hbv(F.read_hbv(os.path.join(GOLDEN, f"{FIXTURES[0]}.ref.hbv")))
    res = step2.build_read_qgraph(fx["packed"], fx["byte_off"], fx["read_len"], pq=fx["pq"], pq_off=fx["pq_off"], edge_order_hint=F.pack_bases(hc, ho))
    assert F.paths_to_bytes(res.path_offset, res.path_off, res.path_edges) == golden_bytes(FIXTURES[0], "ref", "paths")


@pytest.mark.parametrize("env", [{}, {"W2RAP_TEST_EXACT_SMALL": "1"}, {"W2RAP_NO_EXACT_INDEX": "1"}])
def test_index_on_repeat_rich_reads(mods, env, monkeypatch):
    """planted repeat families (bench.planted_reads): the k-mers that straddle a copy's boundary share their 31-base index key with every other copy's --
    the entries of such keys are marked and their k-mers live in the exact table beside the index (common.h).  Every solid k-mer is found where the
    oracle puts it, and every read paths the same, with the table, with a table that starts too small, and without it."""
    import sys, torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    F, step2, synth, O = mods
    monkeypatch.setenv("W2RAP_PATH_INDEX", "1")
    for k, v in env.items(): monkeypatch.setenv(k, v)
    from conftest import planted_reads, PLANTED_LIKE
    r = planted_reads(*PLANTED_LIKE)
    codes, quals, off, pk, bo, ln = r["codes"], r["quals"], r["off"], r["pk"], r["bo"], r["ln"]
    orc = O.run(codes, quals, off)
    with step2.Step2Context(0) as ctx:
        ctx.set_reads_host(pk, bo, ln, quals=quals, qual_off=off)
        st = ctx.count_kmers(7, 4)
        ctx.build_graph(None)
        hi, lo, cnt, c, e, o = ctx.table(st["S"])
        order = np.lexsort((lo, hi))
        assert np.array_equal(e[order], orc.k_edge) and np.array_equal(o[order], orc.k_off)
        ctx.path_reads()
        res = ctx.fetch()
    assert F.hbv_to_bytes(res.hbv) == F.hbv_to_bytes(O.to_hbv(orc))
    assert np.array_equal(res.path_off, orc.path_off) and np.array_equal(res.path_edges, orc.path_edges) and np.array_equal(res.path_offset, orc.path_offset)


def test_marking_pass_reports_a_run_it_cannot_walk(mods, monkeypatch):
    """k_index_mark bounds its backward walk over a run of occupied slots; beyond the bound the entries of one key would be counted from
    different starting points (and marked inconsistently), so the bound being hit is an ERROR (W2RAP_E_LIMIT), not a silent miss.  The hook
    sets the bound to one slot."""
    F, step2, synth, O = mods
    monkeypatch.setenv("W2RAP_PATH_INDEX", "1")
    monkeypatch.setenv("W2RAP_TEST_INDEX_RUN_MAX", "1")
    fx = load_fixture("repeats_snps")
    with pytest.raises(step2.Step2Error, match="run of occupied slots") as e:
        step2.build_read_qgraph(fx["packed"], fx["byte_off"], fx["read_len"], pq=fx["pq"], pq_off=fx["pq_off"])
    assert e.value.code == 5 if hasattr(e.value, "code") else True

