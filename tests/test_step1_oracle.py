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
