"""The GFA dump (hbv2gfa without line finding) on the GPU through the C ABI (include/w2rap_gfa.h) against the reference tool's own output
(tests/golden/*.ref_raw.gfa, *.ref_gfa_stats.txt) and the GFA oracle."""
import os
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN, ROOT
from w2rap_contigger_amd import formats as F, gfa, step2
from oracle import oracle_gfa as OG

pytestmark = pytest.mark.gpu

GOLD = ("palindrome_circle.ref", "repeats_snps.ref", "repeats_snps.ref.large_K")
MORE = ("random20k.ref", "random20k.ref8", "repeats_snps.ref8", "palindrome_circle.ref8", "random20k.ref.large_K", "palindrome_circle.ref.large_K",
        "repeats_snps.ref8.large_K")


@pytest.fixture(scope="module", autouse=True)
def _gpu():
    import torch
    assert torch.cuda.is_available(), "the -m gpu tests need an MI355X"


@pytest.mark.parametrize("g", GOLD)
def test_gpu_gfa_is_the_reference_tools_output(g):
    h = F.read_hbv(os.path.join(GOLDEN, g + ".hbv"))
    r = gfa.gfa_dump(h, 20000)
    assert r.gfa == open(os.path.join(GOLDEN, g + ".ref_raw.gfa"), "rb").read()
    assert r.stats_text() == open(os.path.join(GOLDEN, g + ".ref_gfa_stats.txt")).read()
    assert np.array_equal(r.inv, OG.involution(h))
    assert r.n_segments == r.gfa.count(b"S\t") and r.n_links == r.gfa.count(b"L\t") and r.gfa_len == len(r.gfa)


@pytest.mark.parametrize("g", MORE)
def test_gpu_gfa_equals_the_oracle(g):
    h = F.read_hbv(os.path.join(GOLDEN, g + ".hbv"))
    for gs in (0, 3000, 10 ** 7):
        r = gfa.gfa_dump(h, gs)
        assert r.gfa == OG.raw_gfa(h)
        assert r.stats_text() == OG.stats_text(h, gs)


def test_gpu_gfa_tool_and_flags(tmp_path):
    exe = os.path.join(ROOT, "w2rap_contigger_amd", "w2rap-hbv2gfa")
    g = "repeats_snps.ref"
    pre = os.path.join(GOLDEN, g)
    out = subprocess.run([exe, "-i", pre, "-o", str(tmp_path / "o"), "-g", "20"], check=True, capture_output=True, text=True).stdout
    assert open(tmp_path / "o_raw.gfa", "rb").read() == open(pre + ".ref_raw.gfa", "rb").read()
    assert out.split("=== Graph stats === \n")[1].split("Dumping gfa")[0] == open(pre + ".ref_gfa_stats.txt").read()
    assert out.startswith("hbv2gfa from w2rap-contigger\nReading graph and paths...\n   DONE!\n") and "Graph has 438 edges" in out
    out = subprocess.run([exe, "-i", pre, "-o", str(tmp_path / "s"), "--stats_only", "1"], check=True, capture_output=True, text=True).stdout
    assert not os.path.exists(tmp_path / "s_raw.gfa") and "N50: " in out and "Dumping gfa" not in out
    assert subprocess.run([exe, "-i", pre, "-o", str(tmp_path / "l"), "-l", "1"], capture_output=True).returncode == 1
    assert subprocess.run([exe, "-i", str(tmp_path / "nothing"), "-o", str(tmp_path / "l")], capture_output=True).returncode == 1
    # python mirror of the tool
    r = gfa.run_hbv2gfa(pre, str(tmp_path / "p"), 20)
    assert open(tmp_path / "p_raw.gfa", "rb").read() == open(pre + ".ref_raw.gfa", "rb").read()
    r = gfa.gfa_dump(F.read_hbv(pre + ".hbv"), flags=gfa.NO_FETCH)
    assert r.gfa == b"" and r.gfa_len == os.path.getsize(pre + ".ref_raw.gfa")


def test_gpu_gfa_rejects_a_graph_without_reverse_complements():
    """TestInvolution (hbv2gfa.cc:53) aborts on such a graph in the reference; here W2RAP_E_GRAPH"""
    h = F.read_hbv(os.path.join(GOLDEN, "palindrome_circle.ref.hbv"))
    pk = h.edge_packed.copy()
    pk[int(h.edge_byte_off[3]) + 5] ^= 0x10                       # one base of one object changed
    bad = F.HBV(h.K, h.from_off, h.from_v, h.from_e, h.to_off, h.to_e, pk, h.edge_byte_off, h.edge_len)
    with pytest.raises(step2.Step2Error) as e:
        gfa.gfa_dump(bad)
    assert e.value.code == 6
    fo = h.from_off.copy(); fo[-1] -= 1                             # adjacency lists that do not hold every object
    with pytest.raises(step2.Step2Error, match="every edge object"):
        gfa.gfa_dump(F.HBV(h.K, fo, h.from_v, h.from_e, h.to_off, h.to_e, h.edge_packed, h.edge_byte_off, h.edge_len))
    fe = h.from_e.copy(); fe[0] = h.n_edges + 7                     # ... or name one that does not exist
    with pytest.raises(step2.Step2Error, match="does not exist"):
        gfa.gfa_dump(F.HBV(h.K, h.from_off, h.from_v, fe, h.to_off, h.to_e, h.edge_packed, h.edge_byte_off, h.edge_len))


def test_gpu_gfa_of_a_fresh_step2_and_step3_graph():
    """graphs straight from Step 2 and Step 3 on the GPU (canonical edge order), with short and long edges, against the oracle"""
    from w2rap_contigger_amd import step3
    pk, bo, ln = F.read_fastb(f"{GOLDEN}/repeats_snps.fastb")
    pq, po = F.read_qualp(f"{GOLDEN}/repeats_snps.qualp")
    res = step2.build_read_qgraph(pk, bo, ln, pq=pq, pq_off=po)
    r = gfa.gfa_dump(res.hbv, 120000)
    assert r.gfa == OG.raw_gfa(res.hbv) and r.stats_text() == OG.stats_text(res.hbv, 120000)
    r3 = step3.repath_in_memory(res.hbv, (res.path_offset, res.path_off, res.path_edges), 200)
    g3 = gfa.gfa_dump(r3.hbv, 50000)
    assert g3.gfa == OG.raw_gfa(r3.hbv) and g3.stats_text() == OG.stats_text(r3.hbv, 50000)
    assert np.array_equal(g3.inv, r3.inv2)                          # Step 3's own involution of its graph


@pytest.mark.parametrize("name", ("random20k", "repeats_snps", "palindrome_circle"))
def test_gpu_canonical_mode_writes_the_canonicalised_reference_graph(name):
    """hbvtool.canonicalise(the reference's file) == what Step 2 / Step 3 write without an edge-order hint, byte for byte (graphs)"""
    from w2rap_contigger_amd import hbvtool, step3
    pk, bo, ln = F.read_fastb(f"{GOLDEN}/{name}.fastb")
    pq, po = F.read_qualp(f"{GOLDEN}/{name}.qualp")
    res = step2.build_read_qgraph(pk, bo, ln, pq=pq, pq_off=po)
    ref, _, _ = hbvtool.canonicalise(F.read_hbv(f"{GOLDEN}/{name}.ref8.hbv"))
    assert F.hbv_to_bytes(res.hbv) == F.hbv_to_bytes(ref)
    r3 = step3.repath_in_memory(res.hbv, (res.path_offset, res.path_off, res.path_edges), 200)
    ref3, _, _ = hbvtool.canonicalise(F.read_hbv(f"{GOLDEN}/{name}.ref.large_K.hbv"))
    assert F.hbv_to_bytes(r3.hbv, zero_padding=True) == F.hbv_to_bytes(ref3, zero_padding=True)
