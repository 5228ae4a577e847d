"""The GFA-dump oracle (oracle/oracle_gfa.py: hbv2gfa's statistics and <prefix>_raw.gfa) against the reference tool's OWN output
(tests/golden/*.ref_raw.gfa / *.ref_gfa_stats.txt, written by oracle/_ref/ref_hbv2gfa = the reference's hbv2gfa main, -g 20)."""
import os
import shutil

import numpy as np
import pytest

from conftest import GOLDEN
from w2rap_contigger_amd import formats as F
from oracle import oracle_gfa as OG

GRAPHS = ("palindrome_circle.ref", "repeats_snps.ref", "repeats_snps.ref.large_K")


@pytest.mark.parametrize("g", GRAPHS)
def test_gfa_oracle_reproduces_the_reference_tool(g):
    h = F.read_hbv(os.path.join(GOLDEN, g + ".hbv"))
    assert OG.raw_gfa(h) == open(os.path.join(GOLDEN, g + ".ref_raw.gfa"), "rb").read()
    assert OG.stats_text(h, 20000) == open(os.path.join(GOLDEN, g + ".ref_gfa_stats.txt")).read()


def test_gfa_oracle_involution_is_an_involution():
    h = F.read_hbv(os.path.join(GOLDEN, "repeats_snps.ref8.hbv"))
    inv = OG.involution(h)
    assert np.array_equal(inv[inv], np.arange(h.n_edges)) and np.array_equal(h.edge_len[inv], h.edge_len)


@pytest.mark.skipif(not os.path.exists(OG.REF_GFA_BIN), reason="needs the reference build (oracle/_ref)")
@pytest.mark.parametrize("g", ("random20k.ref8", "random20k.ref.large_K", "palindrome_circle.ref8.large_K"))
def test_gfa_oracle_against_the_reference_binary(g, tmp_path):
    shutil.copy(os.path.join(GOLDEN, g + ".hbv"), tmp_path / "g.hbv")
    shutil.copy(os.path.join(GOLDEN, g + ".paths"), tmp_path / "g.paths")
    txt, gfa = OG.run_reference_gfa(str(tmp_path), "g", "o", 5)
    h = F.read_hbv(os.path.join(GOLDEN, g + ".hbv"))
    assert OG.raw_gfa(h) == gfa
    assert OG.stats_text(h, 5000) == txt.split("=== Graph stats === \n")[1].split("Dumping gfa")[0]
