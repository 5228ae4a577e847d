"""hbvtool (canonicaliser + diff for .hbv/.paths pairs): the reference's 1-thread and 8-thread outputs, which differ as files, canonicalise
to the same graph bytes; paths agree up to the parallel-edge extension ties of SURVEY.md Q14."""
import io
import os

import numpy as np
import pytest

from conftest import GOLDEN, FIXTURES
from w2rap_contigger_amd import formats as F, hbvtool


@pytest.mark.parametrize("name", FIXTURES)
@pytest.mark.parametrize("kind", ("", ".large_K"))
def test_canonical_form_of_two_reference_runs_is_one_file(name, kind, tmp_path):
    a, b = os.path.join(GOLDEN, f"{name}.ref{kind}"), os.path.join(GOLDEN, f"{name}.ref8{kind}")
    ha, hb = F.read_hbv(a + ".hbv"), F.read_hbv(b + ".hbv")
    ca, pa, ma = hbvtool.canonicalise(ha, F.read_paths(a + ".paths"))
    cb, pb, mb = hbvtool.canonicalise(hb, F.read_paths(b + ".paths"))
    assert F.hbv_to_bytes(ca, zero_padding=True) == F.hbv_to_bytes(cb, zero_padding=True)
    # canonical = unipaths in lexicographic order, each followed by its reverse complement
    codes, off = ca.edge_codes(); off = off.astype(np.int64)
    seqs = [codes[off[i]:off[i + 1]].tobytes() for i in range(ca.n_edges)]
    fw = [s for s in seqs if hbvtool._form(np.frombuffer(s, np.uint8)) != 1]
    assert fw == sorted(fw) and sorted(np.asarray(ma)) == list(range(ha.n_edges))
    # idempotent, and the files round-trip through the CLI
    c2, _, m2 = hbvtool.canonicalise(ca, pa)
    assert F.hbv_to_bytes(c2) == F.hbv_to_bytes(ca) and np.array_equal(m2, np.arange(ca.n_edges))
    assert hbvtool.main(["canon", a, str(tmp_path / "x")]) == 0
    assert open(tmp_path / "x.hbv", "rb").read() == F.hbv_to_bytes(ca)
    out = io.StringIO()
    assert hbvtool.diff(a, b, out) == 0, out.getvalue()
    assert "identical after canonicalisation" in out.getvalue() and " 0 differ otherwise" in out.getvalue()


def test_diff_reports_real_differences(tmp_path):
    a = os.path.join(GOLDEN, "random20k.ref")
    out = io.StringIO()
    assert hbvtool.diff(a, os.path.join(GOLDEN, "repeats_snps.ref"), out) == 1 and "edge sequence sets differ" in out.getvalue()
    o, p, e = F.read_paths(a + ".paths")
    e = e.copy(); e[5] ^= 1                                   # one read sent down the other strand
    F.write_hbv(tmp_path / "m.hbv", F.read_hbv(a + ".hbv")); F.write_paths(tmp_path / "m.paths", o, p, e)
    out = io.StringIO()
    assert hbvtool.diff(a, str(tmp_path / "m"), out) == 1 and " 1 differ otherwise" in out.getvalue()
    assert hbvtool.main(["frobnicate"]) == 2
