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


# ---- W2RAP_STEP3_UNIQUE_KMERS (include/w2rap_step3.h, csrc/step3_repath.hip k3_mid_edges / k3_lone_places): the claim the GPU's shortcut rests
# on, checked on the CPU against the oracle's places -- which are what BigKPather's dictionary is built from (Repath.cc:101-132, BigKPather.cc:40-55)
def _lone_rule(place_off, place_edges, inv):
    """-> lone[u]: place u is ONE edge that no place of three or more edges holds in its middle (nor its inverse), and is not its own inverse"""
    po = place_off.astype(np.int64)
    shared = inv == np.arange(len(inv))
    for u in range(len(po) - 1):
        mid = place_edges[po[u] + 1:po[u + 1] - 1]
        shared[mid] = True; shared[inv[mid]] = True
    return np.array([po[u + 1] - po[u] == 1 and not shared[place_edges[po[u]]] for u in range(len(po) - 1)])


def _canon(kmer_codes):
    rc = (3 - kmer_codes[::-1])
    a, b = kmer_codes.tobytes(), rc.tobytes()
    return a if a <= b else b


def _occurrences(r, K2):
    """{canonical K2-mer: number of occurrences in the places' sequences}, and the list of (place, t, canonical K2-mer)"""
    ao = r.all_off.astype(np.int64)
    count, occ = {}, []
    for u in range(len(ao) - 1):
        seq = r.all_codes[ao[u]:ao[u + 1]]
        for t in range(len(seq) - K2 + 1):
            c = _canon(seq[t:t + K2])
            count[c] = count.get(c, 0) + 1
            occ.append((u, t, c))
    return count, occ


@pytest.mark.parametrize("K2", [200, 100])
@pytest.mark.parametrize("name", FIXTURES)
def test_k2mers_strictly_inside_an_unshared_edge_occur_once(name, K2):
    """In the reference's own Step-2 graphs (every K-mer once) a K2-mer strictly inside a one-edge place whose edge no longer place holds in its
    middle has NO second occurrence -- neither forward nor reverse-complemented, neither in its own place nor in any other: the K2-mers the GPU
    leaves out of the dictionary's hashing are exactly such ones.  Not vacuous: most occurrences of every fixture are of this kind."""
    h, p = _small(name, "ref")
    r = O3.run(h, p, K2, stop_after=1)
    lone = _lone_rule(r.place_off, r.place_edges, r.inv)
    count, occ = _occurrences(r, K2)
    ao = r.all_off.astype(np.int64)
    n_lone = 0
    for u, t, c in occ:
        L = ao[u + 1] - ao[u]
        if lone[u] and 0 < t < L - K2:
            n_lone += 1
            assert count[c] == 1, (u, t)
    assert len(occ) == O3.run(h, p, K2).n_instances
    assert n_lone > len(occ) // 2


def test_the_lone_rule_needs_a_graph_of_unique_kmers():
    """the same rule on a graph that is NOT a unipath graph of distinct K-mers -- two edge objects (and their inverses) with one sequence --
    marks K2-mers as single that occur twice: why the shortcut sits behind a flag only a caller with Step 2's graph may set"""
    rng = np.random.default_rng(9)
    s = rng.integers(0, 4, 400, dtype=np.uint8)
    seqs = [s, s.copy(), 3 - s[::-1], 3 - s[::-1]]                       # objects 0 and 1 spell the same bases; 2 and 3 their reverse complement
    codes = np.concatenate(seqs); off = np.arange(5, dtype=np.uint64) * 400
    inv = np.array([2, 3, 0, 1])
    place_off = np.array([0, 1, 2], np.uint64); place_edges = np.array([0, 1], np.int32)       # two one-edge places
    lone = _lone_rule(place_off, place_edges, inv)
    assert lone.all()
    K2 = 200
    count = {}
    for e in place_edges:
        seq = codes[int(off[e]):int(off[e + 1])]
        for t in range(len(seq) - K2 + 1):
            c = _canon(seq[t:t + K2]); count[c] = count.get(c, 0) + 1
    assert set(count.values()) == {2}
