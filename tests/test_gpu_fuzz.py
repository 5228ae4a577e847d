"""Mutated inputs: the GPU path and the oracle accept and reject the same inputs, with the same message class and, when accepted,
the same bytes.  (Step 1 text is the one place where arbitrary user bytes reach a kernel.)"""
import os

import numpy as np
import pytest

from conftest import GOLDEN
from w2rap_contigger_amd import formats as F, step1, step2
from oracle import oracle1 as O1

pytestmark = pytest.mark.gpu

CLASSES = ("different numbers of records", "incomplete record", "inconsistent base/quality lengths", "illegal base character", "> 63")


@pytest.fixture(scope="module", autouse=True)
def _gpu():
    import torch
    assert torch.cuda.is_available(), "the -m gpu tests need an MI355X"


def _both(f1, f2):
    try:
        o = O1.run(f1, f2); oe = None
    except RuntimeError as e:
        o = None; oe = str(e)
    try:
        g = step1.extract_reads(f1, f2); ge = None
    except step2.Step2Error as e:
        g = None; ge = str(e)
    assert (o is None) == (g is None), f"oracle: {oe!r}  gpu: {ge!r}"
    if o is None:
        oc = [c for c in CLASSES if c in oe]; gc = [c for c in CLASSES if c in ge]
        assert oc and gc, (oe, ge)
        # (an input with several defects: the reference names the one its record-by-record loop meets first -- quality values even
        #  later, they are converted in batches -- the GPU names them by kind: line structure, lengths, bases, qualities)
        return None
    assert np.array_equal(g.packed, o["packed"]) and np.array_equal(g.read_len, o["read_len"]) and np.array_equal(g.quals, o["quals"])
    assert np.array_equal(g.pq, o["pq"]) and np.array_equal(g.pq_off, o["pq_off"])
    return g


def test_fuzz_step1_mutated_fastq():
    lines1 = open(os.path.join(GOLDEN, "step1_r1.fastq"), "rb").read().split(b"\n")[:160]
    lines2 = open(os.path.join(GOLDEN, "step1_r2.fastq"), "rb").read().split(b"\n")[:160]
    base1, base2 = b"\n".join(lines1) + b"\n", b"\n".join(lines2) + b"\n"
    assert _both(base1, base2) is not None
    rng = np.random.default_rng(99)
    accepted = rejected = 0
    for it in range(150):
        bufs = [bytearray(base1), bytearray(base2)]
        for _ in range(int(rng.integers(1, 3))):
            b = bufs[int(rng.integers(0, 2))]
            kind = int(rng.integers(0, 12))                            # 5 and above: harmless
            p = int(rng.integers(0, len(b)))
            if kind == 0:
                b[p] = int(rng.integers(0, 256))                       # any byte
            elif kind == 1:
                del b[p:p + int(rng.integers(1, 40))]                  # a hole
            elif kind == 2:
                b[p:p] = bytes(rng.integers(0, 256, int(rng.integers(1, 20)), dtype=np.uint8))   # noise
            elif kind == 3:
                b[p:p] = b"\n"                                         # a stray newline
            elif kind == 4:
                del b[p:]                                              # truncation
            else:
                b[p] = ord("N") if b[p] in b"ACGT" else b[p]           # harmless
        r = _both(bytes(bufs[0]), bytes(bufs[1]))
        accepted += r is not None; rejected += r is None
    assert accepted >= 10 and rejected >= 50, (accepted, rejected)


def test_fuzz_step1_random_bytes_and_degenerate_texts():
    rng = np.random.default_rng(7)
    for n in (1, 2, 3, 4, 5, 15, 16, 17, 63, 64, 65, 4095, 4096, 4097, 70000):
        a = bytes(rng.integers(0, 256, n, dtype=np.uint8)); b = bytes(rng.integers(0, 256, n, dtype=np.uint8))
        _both(a, b)
        _both(b"\n" * n, b"\n" * n)                                    # only newlines: empty records, n % 4 decides
        _both(b"@\n\n+\n\n" * (n % 50 + 1), b"@\n\n+\n\n" * (n % 50 + 1))
    # one very long record (rounds of 256 characters, runs over 255) next to tiny ones
    q = bytes(33 + (i // 300) % 40 for i in range(5000))
    s = bytes(b"ACGT"[i % 4] for i in range(5000))
    rec = b"@long\n" + s + b"\n+\n" + q + b"\n" + b"@a\nA\n+\n!\n" + b"@b\nAC\n+\n!~\n"[:0] + b"@c\nACG\n+\n#I#\n"
    assert _both(rec, rec).n_reads == 6


def test_fuzz_hbv_and_paths_files_truncated(tmp_path):
    """the standalone tools on truncated / foreign files: an error message and exit code 1, never a crash"""
    import subprocess
    from conftest import ROOT
    pre = os.path.join(GOLDEN, "random20k.ref")
    hb = open(pre + ".hbv", "rb").read()
    exe = os.path.join(ROOT, "w2rap_contigger_amd", "w2rap-hbv2gfa")
    rng = np.random.default_rng(3)
    for cut in [0, 7, 11, 12, 20, 100] + [int(x) for x in rng.integers(0, len(hb), 12)]:
        open(tmp_path / "t.hbv", "wb").write(hb[:cut])
        r = subprocess.run([exe, "-i", str(tmp_path / "t"), "-o", str(tmp_path / "o")], capture_output=True)
        assert r.returncode == 1 and b"cannot read" in r.stderr, (cut, r.returncode, r.stderr[-200:])
    for it in range(10):                                               # flipped bytes: either a clean run or a clean error
        m = bytearray(hb); m[int(rng.integers(12, len(hb)))] ^= 1 << int(rng.integers(0, 8))
        open(tmp_path / "t.hbv", "wb").write(bytes(m))
        r = subprocess.run([exe, "-i", str(tmp_path / "t"), "-o", str(tmp_path / "o")], capture_output=True)
        assert r.returncode in (0, 1), (it, r.returncode, r.stderr[-300:])
