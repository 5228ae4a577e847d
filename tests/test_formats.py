"""Host logic: the Step-1/2/3 on-disk formats (no GPU)."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, FIXTURES, golden_bytes
from w2rap_contigger_amd import formats as F


def test_pack_unpack_roundtrip_ragged():
    rng = np.random.default_rng(5)
    lens = np.array([0, 1, 3, 4, 5, 59, 60, 61, 150, 151, 0, 7], dtype=np.uint64)
    off = np.zeros(len(lens) + 1, np.uint64)
    np.cumsum(lens, out=off[1:])
    codes = rng.integers(0, 4, int(off[-1])).astype(np.uint8)
    pk, bo, ln = F.pack_bases(codes, off)
    assert list(np.diff(bo)) == [int((l + 3) // 4) for l in lens]
    c2, o2 = F.unpack_bases(pk, bo, ln)
    assert np.array_equal(c2, codes) and np.array_equal(o2, off)


def test_fastb_qualp_roundtrip(tmp_path):
    rng = np.random.default_rng(6)
    lens = np.array([150] * 20 + [0, 10, 300, 255, 256], dtype=np.uint64)
    off = np.zeros(len(lens) + 1, np.uint64)
    np.cumsum(lens, out=off[1:])
    codes = rng.integers(0, 4, int(off[-1])).astype(np.uint8)
    quals = rng.integers(0, 64, int(off[-1])).astype(np.uint8)
    quals[:150] = 33                                   # a constant read -> 0-bit block
    F.write_fastb(tmp_path / "a.fastb", *F.pack_bases(codes, off))
    F.write_qualp(tmp_path / "a.qualp", quals, off)
    c2, o2 = F.unpack_bases(*F.read_fastb(tmp_path / "a.fastb"))
    q2, qo2 = F.qualp_to_raw(*F.read_qualp(tmp_path / "a.qualp"))
    assert np.array_equal(c2, codes) and np.array_equal(o2, off)
    assert np.array_equal(q2, quals) and np.array_equal(qo2, off)


def test_pq_rejects_q64():
    with pytest.raises(ValueError):
        F.pq_encode(np.array([64], np.uint8))          # feudal/PQVec.cc:30-35 is fatal on q > 63


@pytest.mark.parametrize("name", FIXTURES)
@pytest.mark.parametrize("tag", ["ref", "ref8"])
def test_reference_files_parse_and_reserialise(name, tag):
    """the reference's own .hbv/.paths parse to the last byte and re-serialise identically"""
    h = F.read_hbv(os.path.join(GOLDEN, f"{name}.{tag}.hbv"))
    assert h.K == 60
    assert F.hbv_to_bytes(h) == golden_bytes(name, tag, "hbv")
    o, po, e = F.read_paths(os.path.join(GOLDEN, f"{name}.{tag}.paths"))
    assert F.paths_to_bytes(o, po, e) == golden_bytes(name, tag, "paths")
    # every edge >= K bases; edges at a vertex share their (K-1)-mer (HyperBasevector::TestValid)
    codes, off = h.edge_codes()
    off = off.astype(np.int64)
    assert (h.edge_len >= 60).all()
    fo = h.from_off.astype(np.int64)
    for v in range(h.n_vertices):
        es = h.from_e[fo[v]:fo[v + 1]]
        firsts = {codes[off[x]:off[x] + 59].tobytes() for x in es}
        assert len(firsts) <= 1


def test_freqs_text():
    hist = np.arange(101)
    t = F.freqs_text(hist)
    assert t.splitlines()[0] == "1, 1" and t.splitlines()[-1] == "100, 100" and len(t.splitlines()) == 100
