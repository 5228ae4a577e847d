"""INTEGRATION.md section B made real: the reference's UNMODIFIED main (src/modules/w2rap-contigger.cc) and objects, linked with
integration/BuildReadQGraph_gpu.cc in BuildReadQGraph.o's place and with libw2rap_step2.so (oracle/Makefile `gpu_contigger` ->
oracle/_ref/w2rap-contigger-gpu), run `--from_step 2 --to_step 3` on the golden fixtures: its Step 2 is the GPU library behind the
reference's own call (w2rap-contigger.cc:338), its Step 3 the reference's own code consuming that graph in memory."""
import io
import os
import shutil
import subprocess

import pytest

from conftest import GOLDEN, FIXTURES, ROOT

pytestmark = pytest.mark.gpu
BIN = os.path.join(ROOT, "oracle", "_ref", "w2rap-contigger-gpu")


def _run(tmp_path, name, extra=(), env=None):
    out = tmp_path / name
    out.mkdir()
    shutil.copy(f"{GOLDEN}/{name}.fastb", out / "frag_reads_orig.fastb")
    shutil.copy(f"{GOLDEN}/{name}.qualp", out / "frag_reads_orig.qualp")
    e = dict(os.environ); e.update(env or {})
    r = subprocess.run([BIN, "-r", "unused.fastq", "-o", str(out), "-p", "asm", "-t", "8", "-m", "32", "-K", "200", "--from_step", "2", "--to_step", "3",
                        "--dump_all", "1", *extra], capture_output=True, text=True, timeout=600, env=e)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    return out, r.stdout


@pytest.mark.skipif(not os.path.exists(BIN), reason="oracle/_ref/w2rap-contigger-gpu not built (needs /root/reference at build time)")
@pytest.mark.parametrize("name", FIXTURES)
def test_reference_main_with_the_gpu_shim(tmp_path, name):
    from w2rap_contigger_amd import formats as F, hbvtool
    out, log = _run(tmp_path, name)
    assert "reads pathed" in log
    assert open(out / "small_K.freqs").read() == open(f"{GOLDEN}/{name}.ref.freqs").read()
    # the small-K graph the reference's main dumped = the canonicalised golden (the reference's own numbering is arbitrary)
    ref2, _, _ = hbvtool.canonicalise(F.read_hbv(f"{GOLDEN}/{name}.ref.hbv"))
    assert open(out / "asm.small_K.hbv", "rb").read() == F.hbv_to_bytes(ref2)
    buf = io.StringIO()
    assert hbvtool.diff(str(out / "asm.small_K"), f"{GOLDEN}/{name}.ref", buf) == 0, buf.getvalue()
    # Step 3 = the reference's own RepathInMemory on that graph: equal to its golden large-K graph and paths modulo relabelling
    buf = io.StringIO()
    assert hbvtool.diff(str(out / "asm.large_K"), f"{GOLDEN}/{name}.ref.large_K", buf) == 0, buf.getvalue()
    assert open(out / "asm.first.frags.dist").read() == open(f"{GOLDEN}/{name}.ref.frags.dist").read()


@pytest.mark.skipif(not os.path.exists(BIN), reason="oracle/_ref/w2rap-contigger-gpu not built")
def test_reference_main_with_the_gpu_shim_on_two_ranks(tmp_path):
    """W2RAP_GPUS=2 would take devices 0 and 1; on a one-GPU box the library reports the missing device through the shim's FatalErr"""
    import torch
    name = "repeats_snps"
    if torch.cuda.device_count() >= 2:
        out, log = _run(tmp_path, name, env={"W2RAP_GPUS": "2"})
        assert open(out / "small_K.freqs").read() == open(f"{GOLDEN}/{name}.ref.freqs").read()
    else:
        out = tmp_path / name
        out.mkdir()
        shutil.copy(f"{GOLDEN}/{name}.fastb", out / "frag_reads_orig.fastb")
        shutil.copy(f"{GOLDEN}/{name}.qualp", out / "frag_reads_orig.qualp")
        e = dict(os.environ); e["W2RAP_GPUS"] = "2"
        r = subprocess.run([BIN, "-r", "x", "-o", str(out), "-p", "asm", "--from_step", "2", "--to_step", "2"], capture_output=True, text=True, timeout=600, env=e)
        assert r.returncode != 0 and "device" in (r.stdout + r.stderr)
