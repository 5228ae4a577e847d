"""The Step-1 oracle (oracle/step1_oracle.cc: paired fastq -> packed bases + PQVec qualities) against the reference's OWN Step-1 output
(tests/golden/step1.ref.fastb/.qualp, written by oracle/_ref/ref_step1 = the unmodified ExtractReads + WriteAll): byte-identical files."""
import os

import numpy as np
import pytest

from conftest import GOLDEN
from w2rap_contigger_amd import formats as F
from oracle import oracle1 as O1


def _fq():
    return open(os.path.join(GOLDEN, "step1_r1.fastq"), "rb").read(), open(os.path.join(GOLDEN, "step1_r2.fastq"), "rb").read()


def test_oracle1_reproduces_the_reference_files(tmp_path):
    r = O1.run(*_fq())
    F.write_fastb(tmp_path / "a.fastb", r["packed"], r["byte_off"], r["read_len"])
    F.write_qualp_blobs(tmp_path / "a.qualp", r["pq"], r["pq_off"])
    assert open(tmp_path / "a.fastb", "rb").read() == open(os.path.join(GOLDEN, "step1.ref.fastb"), "rb").read()
    assert open(tmp_path / "a.qualp", "rb").read() == open(os.path.join(GOLDEN, "step1.ref.qualp"), "rb").read()
    assert len(r["read_len"]) == 1000 and r["read_len"].min() == 0 and r["read_len"].max() == 251
    q, qo = F.qualp_to_raw(r["pq"], r["pq_off"])                    # and the blobs decode to the qualities
    assert np.array_equal(q, r["quals"])


def test_oracle1_pq_encoder_is_the_references_run_length_code():
    """quirk Q17: PowerOf2::ceilLg2lkp returns 64 - ceil(log2 v), so the optimal-partition search never joins two different
    qualities: one 3-byte block per run of equal values, runs cut at 255"""
    q = np.array([30] * 5 + [2] * 3 + [40] + [35] * 300, np.uint8)
    b = O1.pq_encode(q)
    runs = [(5, 30), (3, 2), (1, 40), (255, 35), (45, 35)]           # a run longer than 255 is cut after 255
    exp = b"".join(bytes([n, (v << 3) & 0xFF, v >> 5]) for n, v in runs) + b"\0"
    assert b == exp
    assert np.array_equal(F.pq_decode(b), q)
    with pytest.raises(ValueError):
        O1.pq_encode(np.array([64], np.uint8))


@pytest.mark.parametrize("f1,f2,msg", [
    (b"@a\nACGT\n+\nIIII\n", b"", "different numbers of records"),
    (b"@a\nACGT\n+\nIIII\n", b"@a\nACGT\n+\n", "incomplete record"),
    (b"@a\nACGT\n+\nIII\n", b"@a\nACGT\n+\nIIII\n", "inconsistent base/quality lengths"),
    (b"@a\nACXT\n+\nIIII\n", b"@a\nACGT\n+\nIIII\n", "illegal base character"),
    (b"@a\nACGT\n+\nIII\x7f\n", b"@a\nACGT\n+\nIIII\n", "> 63"),
    (b"@a\nACGT\n+\nIIII\n\n", b"@a\nACGT\n+\nIIII\n\n", "incomplete record"),
])
def test_oracle1_fatal_inputs(f1, f2, msg):
    with pytest.raises(RuntimeError, match=msg):
        O1.run(f1, f2)


def test_oracle1_last_line_without_newline_and_empty_input():
    r = O1.run(b"@a\nACGN\n+\nII#I", b"@b\nttga\n+\n!!!!")
    assert list(r["read_len"]) == [4, 4] and list(r["quals"]) == [40, 40, 2, 40, 0, 0, 0, 0]
    codes, _ = F.unpack_bases(r["packed"], r["byte_off"], r["read_len"])
    assert list(codes) == [0, 1, 2, 0, 3, 3, 2, 0]
    assert len(O1.run(b"", b"")["read_len"]) == 0


def _interleave(f1, f2):
    a, b = f1.split(b"\n"), f2.split(b"\n")
    n = len(a) // 4
    out = []
    for i in range(n):
        out += a[4 * i:4 * i + 4] + b[4 * i:4 * i + 4]
    return b"\n".join(out) + b"\n"


def test_oracle1_one_interleaved_file_gives_the_pairs_files(tmp_path):
    """ExtractReads.cc:481-568: a single fastq file with alternating mates -> the same frag_reads_orig files as the pair (golden)"""
    f1, f2 = _fq()
    r = O1.run_files([_interleave(f1, f2)])
    F.write_fastb(tmp_path / "a.fastb", r["packed"], r["byte_off"], r["read_len"])
    F.write_qualp_blobs(tmp_path / "a.qualp", r["pq"], r["pq_off"])
    assert open(tmp_path / "a.fastb", "rb").read() == open(os.path.join(GOLDEN, "step1.ref.fastb"), "rb").read()
    assert open(tmp_path / "a.qualp", "rb").read() == open(os.path.join(GOLDEN, "step1.ref.qualp"), "rb").read()
    with pytest.raises(RuntimeError, match="even number of entries"):
        O1.run_files([b"@r1\nACGT\n+\nIIII\n" * 3])
    for bad in (b">r1\nACGT\n+\nIIII\n", b"@\nACGT\n+\nIIII\n", b"@ r\nACGT\n+\nIIII\n", b"@/1\nACGT\n+\nIIII\n", b""):
        with pytest.raises(RuntimeError, match="first line"):
            O1.run_files([bad])
    with pytest.raises(RuntimeError, match="more than two"):
        O1.run_files([b"@a/1\nA\n+\nI\n", b"@a/2\nA\n+\nI\n", b"@a 3\nA\n+\nI\n"])


@pytest.mark.skipif(not os.path.exists(O1.REF1_BIN), reason="needs the reference build (oracle/_ref)")
def test_oracle1_file_grouping_against_the_reference_binary(tmp_path):
    """which files pair up is decided by their first read names (ExtractReads.cc:218-258): same name -> a pair in the order given; a file
    whose name nobody shares is read on its own; groups come out sorted by name -- the real reference on the same files"""
    f1, f2 = _fq()
    a = f1.split(b"\n"); b = f2.split(b"\n")
    b[0] = b"@zzz/2 renamed"                                        # the second file no longer shares the first file's read name
    n = (len(a) // 4) & ~1
    cases = {
        "inter": [("i.fastq", _interleave(f1, f2))],
        "unpaired_two": [("a.fastq", b"\n".join(a[:4 * n]) + b"\n"), ("b.fastq", b"\n".join(b[:4 * n]) + b"\n")],
        "unpaired_two_swapped": [("b.fastq", b"\n".join(b[:4 * n]) + b"\n"), ("a.fastq", b"\n".join(a[:4 * n]) + b"\n")],
        "pair_swapped": [("r2.fastq", f2), ("r1.fastq", f1)],
        "pair_and_single": [("r1.fastq", f1), ("s.fastq", b"@aaa x\nACGTN\n+\nIIII#\n@aaa y\nTTGCA\n+\n#IIII\n"), ("r2.fastq", f2)],
    }
    for name, files in cases.items():
        d = tmp_path / name
        d.mkdir()
        for fn, text in files:
            open(d / fn, "wb").write(text)
        O1.run_reference1(str(d), ",".join(str(d / fn) for fn, _ in files))
        r = O1.run_files([t for _, t in files])
        F.write_fastb(d / "o.fastb", r["packed"], r["byte_off"], r["read_len"])
        F.write_qualp_blobs(d / "o.qualp", r["pq"], r["pq_off"])
        assert open(d / "o.fastb", "rb").read() == open(d / "frag_reads_orig.fastb", "rb").read(), name
        assert open(d / "o.qualp", "rb").read() == open(d / "frag_reads_orig.qualp", "rb").read(), name
