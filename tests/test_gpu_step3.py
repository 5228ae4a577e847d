"""Step 3 on the GPU (w2rap_step3_run through the C ABI) against the reference's own output and against the oracle."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, FIXTURES, relabel_compare
from w2rap_contigger_amd import formats as F
from oracle import oracle as O, oracle3 as O3

pytestmark = pytest.mark.gpu


def _small(name, tag):
    return F.read_hbv(os.path.join(GOLDEN, f"{name}.{tag}.hbv")), F.read_paths(os.path.join(GOLDEN, f"{name}.{tag}.paths"))


def _check_against_oracle(res, r):
    assert np.array_equal(res.inv, r.inv) and np.array_equal(res.inv2, r.inv2)
    assert np.array_equal(res.frag_count.astype(np.float64), r.frag)
    assert (res.n_unique_places, res.n_kmer_instances, res.n_kmers_distinct, res.n_unipaths) == (len(r.place_off) - 1, r.n_instances, r.n_distinct, r.n_edges)
    assert res.n_place_bases == len(r.all_codes)
    assert F.hbv_to_bytes(res.hbv) == F.hbv_to_bytes(O3.to_hbv(r))
    assert np.array_equal(res.vleft, r.left) and np.array_equal(res.vright, r.right) and np.array_equal(res.to_v, r.to_v)
    assert F.paths_to_bytes(res.path_offset, res.path_off, res.path_edges) == F.paths_to_bytes(r.path_offset, r.path_off, r.path_edges)


@pytest.mark.parametrize("tag", ["ref", "ref8"])
@pytest.mark.parametrize("name", FIXTURES)
def test_gpu_step3_replays_the_reference(name, tag):
    from w2rap_contigger_amd import step3
    h, p = _small(name, tag)
    rh = F.read_hbv(os.path.join(GOLDEN, f"{name}.{tag}.large_K.hbv"))
    hc, ho = O.edge_hint_from_hbv(rh)
    res = step3.repath_in_memory(h, p, 200, edge_order_hint=F.pack_bases(hc, ho))
    assert F.paths_to_bytes(res.path_offset, res.path_off, res.path_edges) == open(os.path.join(GOLDEN, f"{name}.{tag}.large_K.paths"), "rb").read()
    assert F.hbv_to_bytes(res.hbv, zero_padding=True) == F.hbv_to_bytes(rh, zero_padding=True)
    if tag == "ref":
        assert step3.frags_text(res.frag_count) == open(os.path.join(GOLDEN, f"{name}.ref.frags.dist")).read()
    _check_against_oracle(res, O3.run(h, p, 200, hc, ho))


@pytest.mark.parametrize("K2", [200, 100, 260, 72, 544, 640])      # (the ends of what the reference runs: its -K list, w2rap-contigger.cc:60-62, within BigK's, LargeKDispatcher.h:22-27)
@pytest.mark.parametrize("name", FIXTURES)
def test_gpu_step3_canonical_order_equals_the_oracle(name, K2):
    from w2rap_contigger_amd import step3
    h, p = _small(name, "ref")
    res = step3.repath_in_memory(h, p, K2)
    _check_against_oracle(res, O3.run(h, p, K2))
    if K2 == 200:
        rh = F.read_hbv(os.path.join(GOLDEN, f"{name}.ref.large_K.hbv")); rp = F.read_paths(os.path.join(GOLDEN, f"{name}.ref.large_K.paths"))
        relabel_compare(res.hbv, (res.path_offset, res.path_off, res.path_edges), rh, rp, max_ties=0)


def test_gpu_step3_rejects_bad_input():
    from w2rap_contigger_amd import step3
    from w2rap_contigger_amd.step2 import Step2Error
    import dataclasses
    h, p = _small("random20k", "ref")
    with pytest.raises(Step2Error) as e:
        step3.repath_in_memory(h, p, 201)
    assert e.value.code == 1
    keep = h.n_edges - 1
    bo = h.edge_byte_off
    h2 = dataclasses.replace(h, edge_len=h.edge_len[:keep], edge_byte_off=bo[:keep + 1], edge_packed=h.edge_packed[:int(bo[keep])])
    with pytest.raises(Step2Error) as e:
        step3.repath_in_memory(h2, (p[0][:0], p[1][:1], p[2][:0]), 200)
    assert e.value.code == 6 and "reverse complement" in str(e.value)


def test_gpu_step3_empty_input():
    from w2rap_contigger_amd import step3
    h, p = _small("random20k", "ref")
    res = step3.repath_in_memory(h, (p[0][:0], p[1][:1], p[2][:0]), 200)
    assert res.hbv.n_edges == 0 and res.n_unique_places == 0 and len(res.path_offset) == 0


@pytest.mark.parametrize("K2", [200, 100, 64])
@pytest.mark.parametrize("name", ["circle", "palindrome", "chains"])
def test_gpu_step3_hand_made_cases(name, K2):
    """a smooth circle in the large-K graph, a palindromic K2-mer, long multi-edge places with both truncations, a place of
    exactly K2 bases (the oracle is pinned on these inputs against the reference binary, tests/test_step3_oracle.py)"""
    from step3_cases import case
    from w2rap_contigger_amd import step3
    h, p = case(name)
    r = O3.run(h, p, K2)
    _check_against_oracle(step3.repath_in_memory(h, p, K2), r)
    # replay of an arbitrary edge order: the oracle's canonical unipaths, reversed
    ec, eo = r.obj_codes, r.obj_off.astype(np.int64)
    canon = [ec[eo[i]:eo[i + 1]] for i in range(len(eo) - 1) if O.eform(ec[eo[i]:eo[i + 1]]) != 1][::-1]
    hoff = np.zeros(len(canon) + 1, np.uint64); np.cumsum([len(s) for s in canon], out=hoff[1:])
    hc = np.concatenate(canon) if canon else np.zeros(0, np.uint8)
    _check_against_oracle(step3.repath_in_memory(h, p, K2, edge_order_hint=F.pack_bases(hc, hoff)), O3.run(h, p, K2, hc, hoff))


@pytest.mark.parametrize("K2", [200, 100, 260, 72, 640])
@pytest.mark.parametrize("name", FIXTURES)
def test_gpu_step3_unique_kmers_flag_equals_the_oracle(name, K2, monkeypatch):
    """W2RAP_STEP3_UNIQUE_KMERS: the fixtures' small-K graphs are the reference's own Step-2 output (every K-mer once), so the K2-mers strictly
    inside an edge that no place of three or more edges holds in its middle stay out of the dictionary's hashing (k3_lone_places) -- the
    result is the oracle's, which groups EVERY K2-mer by content (BigKPather.cc:40-55), also with --extend_paths, also with the partitions
    narrowed and the tags cut to a few bits, and the same as without the flag."""
    from w2rap_contigger_amd import step3
    h, p = _small(name, "ref")
    r = O3.run(h, p, K2)
    res = step3.repath_in_memory(h, p, K2, unique_kmers=True)
    _check_against_oracle(res, r)
    assert "k3_lone_places" in step3.profile()
    plain = step3.repath_in_memory(h, p, K2)
    assert "k3_lone_places" not in step3.profile()
    assert F.hbv_to_bytes(res.hbv) == F.hbv_to_bytes(plain.hbv) and np.array_equal(res.path_edges, plain.path_edges)
    if K2 == 200:
        _check_extended(step3.repath_in_memory(h, p, K2, extend_paths=True, unique_kmers=True), O3.run(h, p, K2, extend_paths=True), len(r.place_off) - 1)
        monkeypatch.setenv("W2RAP_TEST_DICT_AVG", "8"); monkeypatch.setenv("W2RAP_TEST_SORT_BITS", "8")
        _check_against_oracle(step3.repath_in_memory(h, p, K2, unique_kmers=True), r)
        monkeypatch.setenv("W2RAP_TEST_DICT_CAP", "2")                    # a bin overflows: the sorted form, with every key
        _check_against_oracle(step3.repath_in_memory(h, p, K2, unique_kmers=True), r)


def test_gpu_step3_after_gpu_step2_on_a_diploid_genome():
    """both steps on the GPU, chained through the reference's own file formats in memory: 40 k reads of a two-haplotype genome
    (one SNP per ~300 bases: most read paths cross several small-K edges), Step 3 against the oracle on Step 2's output"""
    import torch
    from w2rap_contigger_amd import step2, step3, synth
    rng = np.random.default_rng(21)
    g = rng.integers(0, 4, 100_000, dtype=np.uint8)
    hap2 = g.copy()
    for q in rng.choice(np.arange(500, len(g) - 500), len(g) // 300, replace=False):
        hap2[q] = (hap2[q] + 1 + rng.integers(0, 3)) & 3
    codes, quals = synth.sample_reads([g, hap2], 20_000, 77)
    codes, quals = codes.numpy().reshape(-1), quals.numpy().reshape(-1)
    off = np.arange(40_001, dtype=np.uint64) * synth.READ_LEN
    res2 = step2.build_read_qgraph(*F.pack_bases(codes, off), quals=quals, qual_off=off)
    paths = (res2.path_offset, res2.path_off, res2.path_edges)
    res3 = step3.repath_in_memory(res2.hbv, paths, 200)
    assert res3.n_unique_places > 1000 and (np.diff(res3.path_off.astype(np.int64)) > 1).sum() > 1000
    r3 = O3.run(res2.hbv, paths, 200)
    _check_against_oracle(res3, r3)
    _check_against_oracle(step3.repath_in_memory(res2.hbv, paths, 200, unique_kmers=True), r3)        # (k3_lone_places: most K2-mers skip the hashing)


def test_gpu_step3_behind_step2_in_one_context():
    """w2rap_step3_run_after_step2 (graph and paths stay in HBM) == w2rap_step3_run on the fetched host buffers"""
    from conftest import load_fixture
    from w2rap_contigger_amd import step2, step3
    fx = load_fixture("repeats_snps")
    with step2.Step2Context(0) as ctx:
        ctx.set_reads_host(fx["packed"], fx["byte_off"], fx["read_len"], pq=fx["pq"], pq_off=fx["pq_off"])
        ctx.count_kmers(7, 4); ctx.build_graph(None); ctx.path_reads()
        a = step3.repath_after_step2(ctx, 200)
        r2 = ctx.fetch()                                          # the Step-2 context is intact afterwards
        a2 = step3.repath_after_step2(ctx, 200)                   # and the call can be repeated
    b = step3.repath_in_memory(r2.hbv, (r2.path_offset, r2.path_off, r2.path_edges), 200)
    for x in (a, a2):
        assert F.hbv_to_bytes(x.hbv) == F.hbv_to_bytes(b.hbv)
        assert F.paths_to_bytes(x.path_offset, x.path_off, x.path_edges) == F.paths_to_bytes(b.path_offset, b.path_off, b.path_edges)
        assert np.array_equal(x.frag_count, b.frag_count) and np.array_equal(x.inv2, b.inv2)


def test_standalone_step3_tool_replays_the_reference(tmp_path):
    """w2rap-step3 (C++ tool, the reference's file names): .large_K.paths and .first.frags.dist byte-identical to the reference's with
    its edge order replayed, .large_K.hbv identical up to the padding bits; and the canonical run equals the library's"""
    import shutil, subprocess
    from conftest import ROOT
    from w2rap_contigger_amd import step3
    tool = os.path.join(ROOT, "w2rap_contigger_amd", "w2rap-step3")
    name = "repeats_snps"
    d = tmp_path
    shutil.copy(os.path.join(GOLDEN, f"{name}.ref.hbv"), d / "x.small_K.hbv")
    shutil.copy(os.path.join(GOLDEN, f"{name}.ref.paths"), d / "x.small_K.paths")
    ref_hbv = os.path.join(GOLDEN, f"{name}.ref.large_K.hbv")
    out = subprocess.run([tool, "-o", str(d), "-p", "x", "-K", "200", "--edge_order_from", ref_hbv], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    assert "320 unique places" in out.stdout
    assert open(d / "x.large_K.paths", "rb").read() == open(os.path.join(GOLDEN, f"{name}.ref.large_K.paths"), "rb").read()
    assert open(d / "x.first.frags.dist").read() == open(os.path.join(GOLDEN, f"{name}.ref.frags.dist")).read()
    assert F.hbv_to_bytes(F.read_hbv(d / "x.large_K.hbv"), zero_padding=True) == F.hbv_to_bytes(F.read_hbv(ref_hbv), zero_padding=True)
    out = subprocess.run([tool, "-o", str(d), "-p", "x"], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    res = step3.run_step3_files(str(d), "y") if False else step3.repath_in_memory(F.read_hbv(d / "x.small_K.hbv"), F.read_paths(d / "x.small_K.paths"), 200)
    assert open(d / "x.large_K.hbv", "rb").read() == F.hbv_to_bytes(res.hbv)
    assert open(d / "x.large_K.paths", "rb").read() == F.paths_to_bytes(res.path_offset, res.path_off, res.path_edges)


@pytest.mark.parametrize("bits", [8, 14])
def test_gpu_step3_dictionary_is_exact_under_sort_key_collisions(bits, monkeypatch):
    """the K2-mer dictionary sorts by the top bits of a hash and verifies contents; with only 8 or 14 sort bits nearly every run of
    equal sort keys holds several different K2-mers (interleaved with their duplicates) -- the exact regrouping must give the same
    graph and paths as ever"""
    from w2rap_contigger_amd import step3
    monkeypatch.setenv("W2RAP_TEST_SORT_BITS", str(bits))
    for name in ("repeats_snps", "palindrome_circle"):
        h, p = _small(name, "ref")
        rh = F.read_hbv(os.path.join(GOLDEN, f"{name}.ref.large_K.hbv"))
        hc, ho = O.edge_hint_from_hbv(rh)
        res = step3.repath_in_memory(h, p, 200, edge_order_hint=F.pack_bases(hc, ho))
        assert F.paths_to_bytes(res.path_offset, res.path_off, res.path_edges) == open(os.path.join(GOLDEN, f"{name}.ref.large_K.paths"), "rb").read()
        _check_against_oracle(res, O3.run(h, p, 200, hc, ho))
        _check_against_oracle(step3.repath_in_memory(h, p, 200), O3.run(h, p, 200))


def _dictionary_kernels(step3):
    pr = step3.profile()
    return "k3_dict_group" in pr, "k3_group" in pr


def test_gpu_step3_dictionary_by_partition_is_the_default():
    """the K2-mer dictionary is built by hash partition + grouping in LDS (k3_dict_part / k3_dict_group, no library sort); the sorted form
    runs for the replay of a given edge order, on request, and when a partition overflows"""
    from w2rap_contigger_amd import step3
    h, p = _small("repeats_snps", "ref")
    step3.repath_in_memory(h, p, 200)
    assert _dictionary_kernels(step3) == (True, False)
    rh = F.read_hbv(os.path.join(GOLDEN, "repeats_snps.ref.large_K.hbv"))
    hc, ho = O.edge_hint_from_hbv(rh)
    step3.repath_in_memory(h, p, 200, edge_order_hint=F.pack_bases(hc, ho))
    assert _dictionary_kernels(step3) == (False, True)


@pytest.mark.parametrize("env,kernels", [({"W2RAP_TEST_DICT_AVG": "8"}, (True, False)),                                         # two passes, 16 k partitions
                                         ({"W2RAP_TEST_DICT_AVG": "256", "W2RAP_TEST_DICT_PASS_BITS": "3"}, (True, False)),     # three passes
                                         ({"W2RAP_TEST_DICT_AVG": "1", "W2RAP_TEST_DICT_CAP": "64"}, (True, False)),            # partitions of one or two K2-mers
                                         ({"W2RAP_TEST_DICT_AVG": "64", "W2RAP_TEST_SORT_BITS": "4"}, (True, False)),           # 16 tags: every probe verifies contents
                                         ({"W2RAP_TEST_DICT_CAP": "300"}, (True, True)),                                        # a partition overflows: the sorted form takes over
                                         ({"W2RAP_STEP3_SORT_DICT": "1"}, (False, True))])
def test_gpu_step3_partitioned_dictionary_levels_tags_and_fallback(env, kernels, monkeypatch):
    from w2rap_contigger_amd import step3
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    for name in ("repeats_snps", "palindrome_circle"):
        h, p = _small(name, "ref")
        for K2 in (200, 72):
            res = step3.repath_in_memory(h, p, K2)
            assert _dictionary_kernels(step3) == kernels
            _check_against_oracle(res, O3.run(h, p, K2))


@pytest.mark.parametrize("env", [{"W2RAP_TEST_STEP3_FULL_SORTS": "1"}, {"W2RAP_TEST_TIE_RUN": "1"}, {"W2RAP_TEST_ENDS_COLLISION": "1"}])
def test_gpu_step3_single_sorts_and_their_fallbacks(env, monkeypatch):
    """places, unipath order and edge ends are each sorted ONCE by one 64-bit word and settled by content behind it (a key shared by two
    places / a long run of equal first words / two end sequences under one hash send them to the word-by-word sorts of round 2): the forced
    full sorts, a tie-run limit of one, and a pretended hash collision all give the reference's bytes and the oracle's graph"""
    from w2rap_contigger_amd import step3
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    for name in FIXTURES:
        h, p = _small(name, "ref")
        rh = F.read_hbv(os.path.join(GOLDEN, f"{name}.ref.large_K.hbv"))
        hc, ho = O.edge_hint_from_hbv(rh)
        res = step3.repath_in_memory(h, p, 200, edge_order_hint=F.pack_bases(hc, ho))
        assert F.paths_to_bytes(res.path_offset, res.path_off, res.path_edges) == open(os.path.join(GOLDEN, f"{name}.ref.large_K.paths"), "rb").read()
        assert F.hbv_to_bytes(res.hbv, zero_padding=True) == F.hbv_to_bytes(rh, zero_padding=True)
        for K2 in (200, 100):
            _check_against_oracle(step3.repath_in_memory(h, p, K2), O3.run(h, p, K2))


def test_gpu_step3_extend_paths_replays_the_reference():
    """--extend_paths (Repath.cc:72-96): the reference's own output with the flag on the fixture with junctions, its edge order replayed;
    then the oracle in canonical order on every fixture; "unique places" stays the number before the extension (Repath.cc:71)"""
    from w2rap_contigger_amd import step3
    name = "repeats_snps"
    h, p = _small(name, "ref")
    rh = F.read_hbv(os.path.join(GOLDEN, f"{name}.ext.large_K.hbv"))
    hc, ho = O.edge_hint_from_hbv(rh)
    res = step3.repath_in_memory(h, p, 200, edge_order_hint=F.pack_bases(hc, ho), extend_paths=True)
    assert F.paths_to_bytes(res.path_offset, res.path_off, res.path_edges) == open(os.path.join(GOLDEN, f"{name}.ext.large_K.paths"), "rb").read()
    assert F.hbv_to_bytes(res.hbv, zero_padding=True) == F.hbv_to_bytes(rh, zero_padding=True)
    assert res.n_unique_places == 320 and res.hbv.n_edges == 358
    plain = step3.repath_in_memory(h, p, 200)
    assert plain.hbv.n_edges == 384 and plain.n_kmer_instances == 66813 and res.n_kmer_instances > plain.n_kmer_instances


def _check_extended(res, r, n_places_before):
    """as _check_against_oracle; the oracle keeps an extended place as the reference does (as it is), the GPU enters it canonicalised against
    its reverse complement: the same K2-mers, so everything but the number and bases of the places is equal"""
    assert np.array_equal(res.inv, r.inv) and np.array_equal(res.inv2, r.inv2)
    assert (res.n_unique_places, res.n_kmers_distinct, res.n_unipaths) == (n_places_before, r.n_distinct, r.n_edges)
    assert F.hbv_to_bytes(res.hbv) == F.hbv_to_bytes(O3.to_hbv(r))
    assert np.array_equal(res.vleft, r.left) and np.array_equal(res.vright, r.right) and np.array_equal(res.to_v, r.to_v)
    assert F.paths_to_bytes(res.path_offset, res.path_off, res.path_edges) == F.paths_to_bytes(r.path_offset, r.path_off, r.path_edges)


@pytest.mark.parametrize("K2", [200, 100, 260])
@pytest.mark.parametrize("name", FIXTURES)
def test_gpu_step3_extend_paths_equals_the_oracle(name, K2):
    from w2rap_contigger_amd import step3
    h, p = _small(name, "ref")
    res = step3.repath_in_memory(h, p, K2, extend_paths=True)
    _check_extended(res, O3.run(h, p, K2, extend_paths=True), len(O3.run(h, p, K2, stop_after=1).place_off) - 1)


def test_gpu_step3_extend_paths_behind_step2_and_bad_arguments():
    """w2rap_step3_run_after_step2 takes the vertices from the Step-2 context; the one-shot call without vleft / vright is W2RAP_E_ARG"""
    import ctypes as C
    from conftest import load_fixture
    from w2rap_contigger_amd import step2, step3
    fx = load_fixture("repeats_snps")
    with step2.Step2Context(0) as ctx:
        ctx.set_reads_host(fx["packed"], fx["byte_off"], fx["read_len"], pq=fx["pq"], pq_off=fx["pq_off"])
        ctx.count_kmers(7, 4); ctx.build_graph(None); ctx.path_reads()
        a = step3.repath_after_step2(ctx, 200, extend_paths=True)
        r2 = ctx.fetch()
    paths = (r2.path_offset, r2.path_off, r2.path_edges)
    b = step3.repath_in_memory(r2.hbv, paths, 200, extend_paths=True)
    assert F.hbv_to_bytes(a.hbv) == F.hbv_to_bytes(b.hbv)
    assert F.paths_to_bytes(a.path_offset, a.path_off, a.path_edges) == F.paths_to_bytes(b.path_offset, b.path_off, b.path_edges)
    _check_extended(b, O3.run(r2.hbv, paths, 200, extend_paths=True), len(O3.run(r2.hbv, paths, 200, stop_after=1).place_off) - 1)
    keep = [np.ascontiguousarray(r2.hbv.edge_packed, np.uint8), np.ascontiguousarray(r2.hbv.edge_byte_off, np.uint64), np.ascontiguousarray(r2.hbv.edge_len, np.uint32),
            np.ascontiguousarray(paths[0], np.int32), np.ascontiguousarray(paths[1], np.uint64), np.ascontiguousarray(paths[2], np.int32)]
    ptr = lambda a: a.ctypes.data_as(C.c_void_p)
    i = step3.Step3In(r2.hbv.K, len(keep[2]), ptr(keep[0]), ptr(keep[1]), ptr(keep[2]), len(keep[3]), ptr(keep[3]), ptr(keep[4]), ptr(keep[5]), 0, None, None)
    prm = step3.Step3Params(200, 0, 1, None, 0, 0, None, None)
    o = step3.Step3Out(); err = C.create_string_buffer(512)
    assert step3.lib().w2rap_step3_run(C.byref(i), C.byref(prm), C.byref(o), err, 512) == 1 and b"vleft" in err.value
    tl, tr = r2.hbv.to_left_right()
    bad = np.ascontiguousarray(tr.copy(), np.int32); bad[0] = r2.hbv.n_vertices + 5
    tl = np.ascontiguousarray(tl, np.int32)
    i.n_vertices, i.vleft, i.vright = r2.hbv.n_vertices, ptr(tl), ptr(bad)
    assert step3.lib().w2rap_step3_run(C.byref(i), C.byref(prm), C.byref(o), err, 512) == 1 and b"vertex" in err.value


def test_standalone_step3_tool_extend_paths(tmp_path):
    """w2rap-step3 --extend_paths 1 -> the reference's files with the flag (its edge order replayed)"""
    import shutil, subprocess
    from conftest import ROOT
    tool = os.path.join(ROOT, "w2rap_contigger_amd", "w2rap-step3")
    name = "repeats_snps"
    d = tmp_path
    shutil.copy(os.path.join(GOLDEN, f"{name}.ref.hbv"), d / "x.small_K.hbv"); shutil.copy(os.path.join(GOLDEN, f"{name}.ref.paths"), d / "x.small_K.paths")
    ref_hbv = os.path.join(GOLDEN, f"{name}.ext.large_K.hbv")
    out = subprocess.run([tool, "-o", str(d), "-p", "x", "--extend_paths", "1", "--edge_order_from", ref_hbv], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    assert "320 unique places" in out.stdout and "done extending paths" in out.stdout
    assert open(d / "x.large_K.paths", "rb").read() == open(os.path.join(GOLDEN, f"{name}.ext.large_K.paths"), "rb").read()
    assert F.hbv_to_bytes(F.read_hbv(d / "x.large_K.hbv"), zero_padding=True) == F.hbv_to_bytes(F.read_hbv(ref_hbv), zero_padding=True)


def test_gpu_step3_reports_what_the_reference_prints():
    """'N / M reads pathed, X spanning junctions', 'sorting P places', 'U unique places' (Repath.cc:36-72) on the fixture with junctions"""
    from w2rap_contigger_amd import step3
    h, p = _small("repeats_snps", "ref")
    res = step3.repath_in_memory(h, p, 200)
    assert (res.n_reads_pathed, res.n_reads_multipathed, res.n_places, res.n_unique_places) == (11832, 263, 11550, 320)


@pytest.mark.parametrize("extend", [False, True])
def test_gpu_step3_sharded_reads_build_the_same_graph(extend):
    """multi-GPU Step 3 in one process: the reads cut into three shards; each shard reduced to one path per unique place
    (PLACES_ONLY), the other shards' place paths handed in as extra_paths -- every shard builds the graph of the whole read set
    and translates exactly its own reads.  extend: --extend_paths acts on the union of the places, in every shard's second call"""
    import numpy as np
    from conftest import GOLDEN
    from w2rap_contigger_amd import formats as F, step2, step3
    name = "repeats_snps"
    h = F.read_hbv(f"{GOLDEN}/{name}.ref.hbv")
    po, pf, pe = F.read_paths(f"{GOLDEN}/{name}.ref.paths")
    pf = pf.astype(np.int64)
    whole = step3.repath_in_memory(h, (po, pf.astype(np.uint64), pe), 200, extend_paths=extend)
    n = len(po)
    cuts = [0, (n // 3) & ~1, (2 * n // 3) & ~1, n]
    shards = []
    for a, b in zip(cuts, cuts[1:]):
        shards.append((po[a:b], (pf[a:b + 1] - pf[a]).astype(np.uint64), pe[pf[a]:pf[b]]))
    places = [step3.repath_in_memory(h, s, 200, places_only=True, extend_paths=extend) for s in shards]      # (a shard's own list is never extended)
    for pl, s in zip(places, shards):
        assert pl.place_paths is not None and 0 < len(pl.place_paths[0]) - 1 <= len(s[0]) and pl.hbv.n_edges == 0
    assert sum(len(pl.place_paths[0]) - 1 for pl in places) >= whole.n_unique_places
    wo = whole.path_off.astype(np.int64)
    for k, s in enumerate(shards):
        others = [places[j].place_paths for j in range(3) if j != k]
        x_off = np.concatenate([[0], np.cumsum(np.concatenate([np.diff(o[0].astype(np.int64)) for o in others]))]).astype(np.uint64)
        x_edges = np.concatenate([o[1] for o in others])
        r = step3.repath_in_memory(h, s, 200, extra_paths=(x_off, x_edges), extend_paths=extend)
        assert F.hbv_to_bytes(r.hbv) == F.hbv_to_bytes(whole.hbv) and np.array_equal(r.inv2, whole.inv2)
        assert r.n_unique_places == whole.n_unique_places and r.n_kmers_distinct == whole.n_kmers_distinct
        a, b = cuts[k], cuts[k + 1]
        assert np.array_equal(r.path_offset, whole.path_offset[a:b]) and np.array_equal(r.path_off.astype(np.int64), wo[a:b + 1] - wo[a])
        assert np.array_equal(r.path_edges, whole.path_edges[wo[a]:wo[b]])
        assert r.n_reads_pathed == int(np.count_nonzero(np.diff(s[1].astype(np.int64)) > 0))
    # bad extra paths are rejected
    with pytest.raises(step2.Step2Error, match="does not exist"):
        step3.repath_in_memory(h, shards[0], 200, extra_paths=(np.array([0, 1], np.uint64), np.array([h.n_edges + 5], np.int32)))
    with pytest.raises(step2.Step2Error, match="ascending"):
        step3.repath_in_memory(h, shards[0], 200, extra_paths=(np.array([0, 2, 1], np.uint64), np.array([0, 0], np.int32)))


def test_gpu_step3_thousands_of_tiny_places():
    """1500 unrelated edges of exactly K2 .. K2+3 bases, each its own place with one to four K2-mers: every 256-thread block of the
    position-ordered kernels spans a hundred places (the per-block place lookup walks), one-edge places fill the direct-addressed table;
    with a handful of two-edge places on top.  Against the oracle."""
    import numpy as np
    from step3_cases import make_hbv, make_paths, rc, K
    from w2rap_contigger_amd import formats as F, step3
    from oracle import oracle3 as O3
    rng = np.random.default_rng(17)
    K2 = 100
    seqs = []
    for i in range(1500):
        s = rng.integers(0, 4, K2 + (i % 4), dtype=np.uint8)
        seqs += [s, rc(s)]
    # a chain of three overlapping edges for a few multi-edge places
    g = rng.integers(0, 4, 500, dtype=np.uint8)
    cuts = [0, 150, 300, 500 - (K - 1)]
    chain0 = len(seqs)
    for a, b in zip(cuts, cuts[1:]):
        s = g[a:b + K - 1]
        seqs += [s, rc(s)]
    paths = [[2 * i + (i % 3 == 0)] for i in range(1500)] * 2                       # every edge twice, some through the reverse object
    paths += [[chain0, chain0 + 2], [chain0 + 2, chain0 + 4], [chain0 + 5, chain0 + 3, chain0 + 1], [chain0]]
    order = rng.permutation(len(paths))
    paths = [paths[i] for i in order]
    if len(paths) % 2:
        paths.append([])
    h = make_hbv(seqs)
    p = make_paths(paths, rng.integers(-5, 40, len(paths)))
    o = O3.run(h, p, K2)
    r = step3.repath_in_memory(h, p, K2)
    assert F.hbv_to_bytes(r.hbv) == F.hbv_to_bytes(O3.to_hbv(o))
    assert np.array_equal(r.path_offset, o.path_offset) and np.array_equal(r.path_off, o.path_off) and np.array_equal(r.path_edges, o.path_edges)
    assert np.array_equal(r.inv, o.inv) and r.n_unique_places == len(o.place_off) - 1 and r.n_unique_places >= 1500
