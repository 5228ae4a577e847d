"""Drop-in check against the reference's own consumer: its Step 3 (Involution, FragDist,
RepathInMemory -- oracle/_ref/ref_step3, the unmodified reference code) must build the same
large-K graph and paths from OUR Step-2 output as from its own (modulo edge numbering)."""
import os
import shutil
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN, FIXTURES, load_fixture, relabel_compare
from w2rap_contigger_amd import formats as F
from oracle import oracle as O

REF3 = os.path.join(os.path.dirname(O.REF_BIN), "ref_step3")
needs_ref3 = pytest.mark.skipif(not os.path.exists(REF3), reason="oracle/_ref/ref_step3 not built (needs /root/reference once)")


def run_step3(d, prefix="t", K=200):
    subprocess.run([REF3, str(d), prefix, str(K), "1"], check=True, capture_output=True)
    return F.read_hbv(os.path.join(d, f"{prefix}.large_K.hbv")), F.read_paths(os.path.join(d, f"{prefix}.large_K.paths"))


def reference_step3(name, tmp):
    d = tmp / "ref"; d.mkdir()
    shutil.copy(os.path.join(GOLDEN, f"{name}.ref.hbv"), d / "t.small_K.hbv")
    shutil.copy(os.path.join(GOLDEN, f"{name}.ref.paths"), d / "t.small_K.paths")
    return run_step3(d)


@needs_ref3
@pytest.mark.parametrize("name", FIXTURES)
def test_reference_step3_accepts_oracle_output(name, tmp_path):
    fx = load_fixture(name)
    r = O.run(fx["codes"], fx["quals"], fx["off"])                    # canonical edge order (differs from the reference's)
    d = tmp_path / "ours"; d.mkdir()
    F.write_hbv(d / "t.small_K.hbv", O.to_hbv(r))
    F.write_paths(d / "t.small_K.paths", r.path_offset, r.path_off, r.path_edges)
    h_ours, p_ours = run_step3(d)
    h_ref, p_ref = reference_step3(name, tmp_path)
    assert h_ours.K == 200 and h_ours.n_edges == h_ref.n_edges
    relabel_compare(h_ours, p_ours, h_ref, p_ref, max_ties=0.01)


@needs_ref3
@pytest.mark.gpu
@pytest.mark.parametrize("name", FIXTURES)
def test_reference_step3_accepts_gpu_output(name, tmp_path):
    from w2rap_contigger_amd import step2
    d = tmp_path / "ours"; d.mkdir()
    shutil.copy(os.path.join(GOLDEN, f"{name}.fastb"), d / "frag_reads_orig.fastb")
    shutil.copy(os.path.join(GOLDEN, f"{name}.qualp"), d / "frag_reads_orig.qualp")
    step2.run_step2_files(str(d), "t")                                # our Step 2, reference file names
    assert open(d / "small_K.freqs").read() == open(os.path.join(GOLDEN, f"{name}.ref.freqs")).read()
    h_ours, p_ours = run_step3(d)
    h_ref, p_ref = reference_step3(name, tmp_path)
    relabel_compare(h_ours, p_ours, h_ref, p_ref, max_ties=0.01)


@pytest.mark.gpu
def test_standalone_tool_matches_library(tmp_path):
    """w2rap-step2 (C++ tool, reference file names) == the Python path, and replays a reference order byte-exactly"""
    from conftest import ROOT, golden_bytes
    tool = os.path.join(ROOT, "w2rap_contigger_amd", "w2rap-step2")
    name = "palindrome_circle"
    d = tmp_path
    shutil.copy(os.path.join(GOLDEN, f"{name}.fastb"), d / "frag_reads_orig.fastb")
    shutil.copy(os.path.join(GOLDEN, f"{name}.qualp"), d / "frag_reads_orig.qualp")
    out = subprocess.run([tool, "-o", str(d), "-p", "x", "--edge_order_from", os.path.join(GOLDEN, f"{name}.ref8.hbv")],
                         capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    assert open(d / "x.small_K.hbv", "rb").read() == golden_bytes(name, "ref8", "hbv")
    assert open(d / "x.small_K.paths", "rb").read() == golden_bytes(name, "ref8", "paths")
    assert open(d / "small_K.freqs", "rb").read() == golden_bytes(name, "ref", "freqs")
