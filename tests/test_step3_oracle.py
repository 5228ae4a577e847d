"""The Step-3 oracle (oracle/step3_oracle.cc: Involution, FragDist, RepathInMemory) against the reference's OWN Step-3 output
(tests/golden/*.large_K.*, written by oracle/_ref/ref_step3 = the unmodified reference code, 1 and 8 threads): with the
reference's edge order replayed the large-K paths are byte-identical and the large-K graph is identical up to the
padding bits of each edge's last byte; in canonical order everything matches modulo edge relabelling."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, FIXTURES, relabel_compare
from w2rap_contigger_amd import formats as F
from oracle import oracle as O, oracle3 as O3


def _small(name, tag):
    return F.read_hbv(os.path.join(GOLDEN, f"{name}.{tag}.hbv")), F.read_paths(os.path.join(GOLDEN, f"{name}.{tag}.paths"))


def _large(name, tag):
    return F.read_hbv(os.path.join(GOLDEN, f"{name}.{tag}.large_K.hbv")), F.read_paths(os.path.join(GOLDEN, f"{name}.{tag}.large_K.paths"))


@pytest.mark.parametrize("tag", ["ref", "ref8"])
@pytest.mark.parametrize("name", FIXTURES)
def test_oracle3_replays_the_reference_byte_for_byte(name, tag):
    h, p = _small(name, tag)
    rh, rp = _large(name, tag)
    hc, ho = O.edge_hint_from_hbv(rh)                                  # the reference's large-K unipaths in ITS order
    r = O3.run(h, p, 200, hc, ho)
    assert F.paths_to_bytes(r.path_offset, r.path_off, r.path_edges) == open(os.path.join(GOLDEN, f"{name}.{tag}.large_K.paths"), "rb").read()
    assert F.hbv_to_bytes(O3.to_hbv(r), zero_padding=True) == F.hbv_to_bytes(rh, zero_padding=True)
    if tag == "ref":
        assert O3.frags_text(r.frag) == open(os.path.join(GOLDEN, f"{name}.ref.frags.dist")).read()
    # Involution: the partner's sequence is the reverse complement
    codes, off = h.edge_codes(); off = off.astype(np.int64)
    for e in range(h.n_edges):
        a, b = codes[off[e]:off[e + 1]], codes[off[r.inv[e]]:off[r.inv[e] + 1]]
        assert np.array_equal(a, 3 - b[::-1])
    assert np.array_equal(r.inv[r.inv], np.arange(h.n_edges))
    assert np.array_equal(r.inv2[r.inv2], np.arange(len(r.inv2)))


@pytest.mark.parametrize("name", FIXTURES)
def test_oracle3_canonical_order_matches_modulo_relabelling(name):
    h, p = _small(name, "ref")
    rh, rp = _large(name, "ref")
    r = O3.run(h, p, 200)
    assert r.n_distinct <= r.n_instances and r.n_edges * 2 >= len(r.inv2)
    relabel_compare(O3.to_hbv(r), (r.path_offset, r.path_off, r.path_edges), rh, rp, max_ties=0)
    # 1-thread and 8-thread reference runs give the same graph, numbered differently
    rh8, rp8 = _large(name, "ref8")
    h8, p8 = _small(name, "ref8")
    r8 = O3.run(h8, p8, 200)
    relabel_compare(O3.to_hbv(r8), (r8.path_offset, r8.path_off, r8.path_edges), rh8, rp8, max_ties=0)


def test_oracle3_rejects_a_graph_without_reverse_complements():
    h, p = _small("random20k", "ref")
    import dataclasses
    keep = h.n_edges - 1                                               # drop the last edge object: its partner loses its RC
    bo = h.edge_byte_off
    h2 = dataclasses.replace(h, edge_len=h.edge_len[:keep], edge_byte_off=bo[:keep + 1], edge_packed=h.edge_packed[:int(bo[keep])])
    with pytest.raises(RuntimeError, match="reverse complement"):
        O3.run(h2, (p[0][:0], p[1][:1], p[2][:0]), 200)


@pytest.mark.skipif(not os.path.exists(O3.REF3_BIN), reason="oracle/_ref/ref_step3 not built (needs /root/reference once)")
@pytest.mark.parametrize("K2", [100, 260, 72, 640])
def test_oracle3_other_large_k_against_the_reference_run_here(K2, tmp_path):
    """the goldens are K2 = 200; other members of the reference's K list are checked against the reference binary itself"""
    import shutil
    name = "repeats_snps"
    d = tmp_path
    shutil.copy(os.path.join(GOLDEN, f"{name}.ref.hbv"), d / "t.small_K.hbv")
    shutil.copy(os.path.join(GOLDEN, f"{name}.ref.paths"), d / "t.small_K.paths")
    O3.run_reference3(str(d), "t", K2, 1)
    rh = F.read_hbv(d / "t.large_K.hbv")
    hc, ho = O.edge_hint_from_hbv(rh)
    h, p = _small(name, "ref")
    r = O3.run(h, p, K2, hc, ho)
    assert F.paths_to_bytes(r.path_offset, r.path_off, r.path_edges) == open(d / "t.large_K.paths", "rb").read()
    assert F.hbv_to_bytes(O3.to_hbv(r), zero_padding=True) == F.hbv_to_bytes(rh, zero_padding=True)


def test_oracle3_extend_paths_replays_the_reference_byte_for_byte():
    """--extend_paths (Repath.cc:72-96): the golden is the reference's own output with EXTEND_PATHS on the fixture with junctions
    (97 extended places beside the 320; a different large-K graph: 358 edge objects against 384)"""
    name = "repeats_snps"
    h, p = _small(name, "ref")
    rh = F.read_hbv(os.path.join(GOLDEN, f"{name}.ext.large_K.hbv"))
    hc, ho = O.edge_hint_from_hbv(rh)
    r = O3.run(h, p, 200, hc, ho, extend_paths=True)
    assert len(r.place_off) - 1 == 417
    assert F.paths_to_bytes(r.path_offset, r.path_off, r.path_edges) == open(os.path.join(GOLDEN, f"{name}.ext.large_K.paths"), "rb").read()
    assert F.hbv_to_bytes(O3.to_hbv(r), zero_padding=True) == F.hbv_to_bytes(rh, zero_padding=True)
    r0 = O3.run(h, p, 200)
    assert (len(r0.place_off) - 1, r0.n_instances) == (320, 66813) and r.n_instances == 123670


@pytest.mark.skipif(not os.path.exists(O3.REF3_BIN), reason="oracle/_ref/ref_step3 not built (needs /root/reference once)")
@pytest.mark.parametrize("name,K2", [("repeats_snps", 100), ("palindrome_circle", 200), ("random20k", 260)])
def test_oracle3_extend_paths_against_the_reference_run_here(name, K2, tmp_path):
    import shutil
    d = tmp_path
    shutil.copy(os.path.join(GOLDEN, f"{name}.ref.hbv"), d / "t.small_K.hbv")
    shutil.copy(os.path.join(GOLDEN, f"{name}.ref.paths"), d / "t.small_K.paths")
    O3.run_reference3(str(d), "t", K2, 1, extend_paths=True)
    rh = F.read_hbv(d / "t.large_K.hbv")
    hc, ho = O.edge_hint_from_hbv(rh)
    h, p = _small(name, "ref")
    r = O3.run(h, p, K2, hc, ho, extend_paths=True)
    assert F.paths_to_bytes(r.path_offset, r.path_off, r.path_edges) == open(d / "t.large_K.paths", "rb").read()
    assert F.hbv_to_bytes(O3.to_hbv(r), zero_padding=True) == F.hbv_to_bytes(rh, zero_padding=True)


@pytest.mark.skipif(not os.path.exists(O3.REF3_BIN), reason="oracle/_ref/ref_step3 not built (needs /root/reference once)")
@pytest.mark.parametrize("K2", [200, 100])
@pytest.mark.parametrize("name", ["circle", "palindrome", "chains"])
def test_oracle3_hand_made_cases_against_the_reference_run_here(name, K2, tmp_path):
    """a smooth circle, a palindromic K2-mer, long multi-edge places: the oracle against the reference binary itself"""
    from step3_cases import case
    h, p = case(name)
    d = tmp_path
    F.write_hbv(d / "t.small_K.hbv", h)
    F.write_paths(d / "t.small_K.paths", *p)
    O3.run_reference3(str(d), "t", K2, 1)
    rh = F.read_hbv(d / "t.large_K.hbv")
    hc, ho = O.edge_hint_from_hbv(rh)
    r = O3.run(h, p, K2, hc, ho)
    assert F.paths_to_bytes(r.path_offset, r.path_off, r.path_edges) == open(d / "t.large_K.paths", "rb").read()
    assert F.hbv_to_bytes(O3.to_hbv(r), zero_padding=True) == F.hbv_to_bytes(rh, zero_padding=True)
    if name == "circle":                                         # the large-K graph holds an edge that starts and ends in one vertex
        assert any(r.left[i] == r.right[i] for i in range(len(r.left)))
