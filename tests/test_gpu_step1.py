"""Step 1 (paired fastq -> frag_reads_orig.fastb/.qualp) on the GPU through the C ABI (include/w2rap_step1.h) against the reference's own
files (tests/golden/step1.ref.*, written by the unmodified ExtractReads + WriteAll) and against the Step-1 oracle."""
import gzip
import os
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN, ROOT
from w2rap_contigger_amd import formats as F, step1, step2
from oracle import oracle1 as O1

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _gpu():
    import torch
    assert torch.cuda.is_available(), "the -m gpu tests need an MI355X"


def _fq():
    return open(os.path.join(GOLDEN, "step1_r1.fastq"), "rb").read(), open(os.path.join(GOLDEN, "step1_r2.fastq"), "rb").read()


def _same(res, orc):
    assert np.array_equal(res.packed, orc["packed"]) and np.array_equal(res.byte_off, orc["byte_off"]) and np.array_equal(res.read_len, orc["read_len"])
    assert np.array_equal(res.quals, orc["quals"])
    assert np.array_equal(res.pq, orc["pq"]) and np.array_equal(res.pq_off, orc["pq_off"])


def test_gpu_step1_writes_the_references_files(tmp_path):
    res = step1.run_step1_files(os.path.join(GOLDEN, "step1_r1.fastq") + "," + os.path.join(GOLDEN, "step1_r2.fastq"), str(tmp_path))
    assert open(tmp_path / "frag_reads_orig.fastb", "rb").read() == open(os.path.join(GOLDEN, "step1.ref.fastb"), "rb").read()
    assert open(tmp_path / "frag_reads_orig.qualp", "rb").read() == open(os.path.join(GOLDEN, "step1.ref.qualp"), "rb").read()
    _same(res, O1.run(*_fq()))
    assert res.n_reads == 1000 and res.n_bases == int(res.read_len.sum())


def test_gpu_step1_tool_plain_and_gz(tmp_path):
    """w2rap-step1 with the reference's flags; .gz inputs inflate to the same result"""
    exe = os.path.join(ROOT, "w2rap_contigger_amd", "w2rap-step1")
    for sub, gz in (("plain", False), ("gz", True)):
        d = tmp_path / sub
        d.mkdir()
        names = []
        for k in (1, 2):
            src = os.path.join(GOLDEN, f"step1_r{k}.fastq")
            if gz:
                dst = str(d / f"r{k}.fastq.gz")
                with gzip.open(dst, "wb") as f:
                    f.write(open(src, "rb").read())
                names.append(dst)
            else:
                names.append(src)
        subprocess.run([exe, "-r", ",".join(names), "-o", str(d)], check=True, capture_output=True)
        assert open(d / "frag_reads_orig.fastb", "rb").read() == open(os.path.join(GOLDEN, "step1.ref.fastb"), "rb").read()
        assert open(d / "frag_reads_orig.qualp", "rb").read() == open(os.path.join(GOLDEN, "step1.ref.qualp"), "rb").read()
    r = subprocess.run([exe, "-r", "", "-o", str(tmp_path)], capture_output=True)
    assert r.returncode == 2
    r = subprocess.run([exe, "-r", "/nonexistent/a.fastq,/nonexistent/b.fastq", "-o", str(tmp_path)], capture_output=True)
    assert r.returncode == 1 and b"cannot read" in r.stderr


def _random_fastq(rng, n, maxlen, qual_runs, lower=False, n_frac=0.01):
    """n records; run-structured qualities (long runs, runs > 255, all values 0..63)"""
    out = []
    alpha = np.frombuffer(b"ACGTacgt" if lower else b"ACGT", np.uint8)
    for i in range(n):
        L = int(rng.integers(0, maxlen + 1))
        b = alpha[rng.integers(0, len(alpha), L)].copy()
        b[rng.random(L) < n_frac] = ord("N")
        q = np.zeros(L, np.uint8)
        p = 0
        while p < L:
            r = int(rng.integers(1, qual_runs + 1))
            q[p:p + r] = rng.integers(0, 64)
            p += r
        out.append(b"@r%d some text\n" % i + b.tobytes() + b"\n+\n" + (q + 33).tobytes() + b"\n")
    return b"".join(out)


@pytest.mark.parametrize("n,maxlen,runs,lower", [(1, 7, 3, False), (257, 150, 40, False), (3000, 251, 8, True), (500, 1500, 700, False), (64, 5, 1, True)])
def test_gpu_step1_random_equals_the_oracle(n, maxlen, runs, lower):
    rng = np.random.default_rng(n * 31 + maxlen)
    f1, f2 = _random_fastq(rng, n, maxlen, runs, lower), _random_fastq(rng, n, maxlen, runs, lower)
    res = step1.extract_reads(f1, f2)
    _same(res, O1.run(f1, f2))
    q, qo = F.qualp_to_raw(res.pq, res.pq_off)                       # the blobs decode to the raw qualities
    assert np.array_equal(q, res.quals) and np.array_equal(qo, res.qual_off)


def test_gpu_step1_flags_and_edge_inputs():
    f1, f2 = _fq()
    a = step1.extract_reads(f1, f2, flags=step1.NO_PQ)
    assert a.pq is None and np.array_equal(a.quals, O1.run(f1, f2)["quals"])
    b = step1.extract_reads(f1, f2, flags=step1.NO_FETCH)
    assert b.n_reads == 1000 and len(b.packed) == 0 and b.n_bases == a.n_bases
    # last line without a newline; empty reads; empty files
    r = step1.extract_reads(b"@a\nACGN\n+\nII#I", b"@b\nttga\n+\n!!!!")
    _same(r, O1.run(b"@a\nACGN\n+\nII#I", b"@b\nttga\n+\n!!!!"))
    assert list(r.read_len) == [4, 4] and list(r.quals) == [40, 40, 2, 40, 0, 0, 0, 0]
    e = b"@a\n\n+\n\n"
    _same(step1.extract_reads(e, e), O1.run(e, e))
    z = step1.extract_reads(b"", b"")
    assert z.n_reads == 0 and len(z.read_len) == 0 and list(z.byte_off) == [0]
    # text with vertical tabs in the header lines (the word-wise newline count has to stay exact)
    v = b"@a\x0b\x0b\x0a" + b"ACGT" * 20 + b"\n+\x0b\n" + b"I" * 80 + b"\n"
    _same(step1.extract_reads(v, v), O1.run(v, v))


@pytest.mark.parametrize("f1,f2,msg", [
    (b"@a\nACGT\n+\nIIII\n", b"", "different numbers of records"),
    (b"@a\nACGT\n+\nIIII\n", b"@a\nACGT\n+\n", "incomplete record"),
    (b"@a\nACGT\n+\nIII\n", b"@a\nACGT\n+\nIIII\n", "inconsistent base/quality lengths"),
    (b"@a\nACXT\n+\nIIII\n", b"@a\nACGT\n+\nIIII\n", "illegal base character"),
    (b"@a\nACGT\n+\nIII\x7f\n", b"@a\nACGT\n+\nIIII\n", "> 63"),
    (b"@a\nACGT\n+\nIIII\n\n", b"@a\nACGT\n+\nIIII\n\n", "incomplete record"),
    (b"@a\nACGT\n+\nIIII\n@b\nAC\n", b"@a\nACGT\n+\nIIII\n", "different numbers of records"),      # the missing header is seen first
    (b"@a\nACGT\n+\nIIII\n@b\nAC\n", b"@a\nACGT\n+\nIIII\n@b\nAC\n+\nII\n", "incomplete record"),
    (b"@a\nACGT\n+\nIIII\n@b\nAC\n+\nII\n", b"@a\nACGT\n+\nIIII\n", "different numbers of records"),
    (b"@a\nACGT\n+\nII I\n", b"@a\nACGT\n+\nIIII\n", "> 63"),
])
def test_gpu_step1_fatal_inputs_as_the_reference(f1, f2, msg):
    """the reference's fatal conditions (ExtractReads.cc:399-452, PQVec.cc:30-35): same condition in the oracle and on the GPU"""
    with pytest.raises(RuntimeError, match=msg):
        O1.run(f1, f2)
    with pytest.raises(step2.Step2Error, match=msg):
        step1.extract_reads(f1, f2)


def test_gpu_step1_feeds_step2():
    """fastq -> Step 1 -> Step 2 on the GPU ends at the reference's own small-K graph: the golden fixture's reads written out as a
    pair of fastq files, ingested, and the reference's .hbv/.paths reproduced byte for byte (edge order replayed)"""
    from oracle import oracle as O
    name = "repeats_snps"
    pk, bo, ln = F.read_fastb(f"{GOLDEN}/{name}.fastb")
    pq, po = F.read_qualp(f"{GOLDEN}/{name}.qualp")
    codes, off = F.unpack_bases(pk, bo, ln)
    quals, _ = F.qualp_to_raw(pq, po)
    fq = [[], []]
    for r in range(len(ln)):
        a, b = int(off[r]), int(off[r + 1])
        fq[r & 1].append(b"@r%d\n" % r + np.frombuffer(b"ACGT", np.uint8)[codes[a:b]].tobytes() + b"\n+\n" + (quals[a:b] + 33).astype(np.uint8).tobytes() + b"\n")
    s1 = step1.extract_reads(b"".join(fq[0]), b"".join(fq[1]))
    assert np.array_equal(s1.packed, pk) and np.array_equal(s1.read_len, ln) and np.array_equal(s1.quals, quals)
    ref_hbv = F.read_hbv(f"{GOLDEN}/{name}.ref.hbv")
    hc, ho = O.edge_hint_from_hbv(ref_hbv)
    res = step2.build_read_qgraph(s1.packed, s1.byte_off, s1.read_len, pq=s1.pq, pq_off=s1.pq_off, edge_order_hint=F.pack_bases(hc, ho))
    assert F.hbv_to_bytes(res.hbv) == open(f"{GOLDEN}/{name}.ref.hbv", "rb").read()
    assert F.paths_to_bytes(res.path_offset, res.path_off, res.path_edges) == open(f"{GOLDEN}/{name}.ref.paths", "rb").read()


def test_gpu_step1_device_text_into_step2_context():
    """the text already in HBM (device pointers), the reads left in HBM as a Step-2 context's reads (w2rap_step1_run_into_step2):
    counting, graph and paths straight behind -- equal to Step 2 run on the host copies"""
    import torch
    rng = np.random.default_rng(5)
    genome = rng.integers(0, 4, 3000)
    acgt = np.frombuffer(b"ACGT", np.uint8)
    recs = [[], []]
    for i in range(1200):
        s = int(rng.integers(0, 3000 - 400))
        for k, (a, rc) in enumerate(((s, False), (s + 250, True))):
            c = genome[a:a + 150]
            if rc:
                c = 3 - c[::-1]
            q = np.full(150, 37, np.uint8); q[140:] = 2
            recs[k].append(b"@p%d/%d\n" % (i, k + 1) + acgt[c].tobytes() + b"\n+\n" + (q + 33).tobytes() + b"\n")
    f1, f2 = b"".join(recs[0]), b"".join(recs[1])
    host = step1.extract_reads(f1, f2)
    # odd device addresses too: the library copies an unaligned text to an aligned buffer
    for shift in (0, 3):
        d1 = torch.frombuffer(bytearray(b"\0" * shift + f1), dtype=torch.uint8).cuda()
        d2 = torch.frombuffer(bytearray(b"\0" * shift + f2), dtype=torch.uint8).cuda()
        with step2.Step2Context(0) as ctx:
            dev = step1.extract_reads((d1.data_ptr() + shift, len(f1)), (d2.data_ptr() + shift, len(f2)), ctx=ctx)
            assert np.array_equal(dev.packed, host.packed) and np.array_equal(dev.pq, host.pq) and np.array_equal(dev.quals, host.quals)
            assert {"k1_count_nl", "k1_list_nl", "k1_lens", "k1_unpack", "k1_pq_write"} <= set(step1.profile())
            ctx.count_kmers(7, 4); ctx.build_graph(None); ctx.path_reads()
            res = ctx.fetch()
            # a second ingest into the same context replaces the reads and the results
            step1.extract_reads(f1, f2, flags=step1.NO_FETCH | step1.NO_PQ, ctx=ctx)
            ctx.count_kmers(7, 4); ctx.build_graph(None); ctx.path_reads()
            res2 = ctx.fetch()
        alone = step1.extract_reads((d1.data_ptr() + shift, len(f1)), (d2.data_ptr() + shift, len(f2)))      # device text, no Step-2 context
        assert np.array_equal(alone.packed, host.packed) and np.array_equal(alone.pq, host.pq) and np.array_equal(alone.pq_off, host.pq_off)
        ref = step2.build_read_qgraph(host.packed, host.byte_off, host.read_len, quals=host.quals, qual_off=host.qual_off)
        for r in (res, res2):
            assert F.hbv_to_bytes(r.hbv) == F.hbv_to_bytes(ref.hbv) and np.array_equal(r.path_edges, ref.path_edges) and np.array_equal(r.hist, ref.hist)
        assert res.hbv.n_edges > 0 and res.n_reads_pathed > 2000
    # a failed ingest leaves the context without reads
    with step2.Step2Context(0) as ctx:
        with pytest.raises(step2.Step2Error, match="incomplete record"):
            step1.extract_reads(f1[:-10], f2[:-200], ctx=ctx)
        ctx.count_kmers(7, 4)
        assert ctx.counts()["kmer_instances"] == 0


def test_gpu_steps_1_2_3_chained_in_hbm_end_at_the_references_large_k_graph():
    """fastq text -> Step 1 -> Step 2 -> Step 3, each stage taking its input where the previous one left it in HBM (no host copies in
    between): the large-K graph is the canonicalised reference graph byte for byte, the paths are the oracle chain's"""
    from w2rap_contigger_amd import hbvtool, step3
    from oracle import oracle as O, oracle3 as O3
    name = "repeats_snps"
    pk, bo, ln = F.read_fastb(f"{GOLDEN}/{name}.fastb")
    codes, off = F.unpack_bases(pk, bo, ln)
    quals, _ = F.qualp_to_raw(*F.read_qualp(f"{GOLDEN}/{name}.qualp"))
    fq = [[], []]
    for r in range(len(ln)):
        a, b = int(off[r]), int(off[r + 1])
        fq[r & 1].append(b"@r%d\n" % r + np.frombuffer(b"ACGT", np.uint8)[codes[a:b]].tobytes() + b"\n+\n" + (quals[a:b] + 33).astype(np.uint8).tobytes() + b"\n")
    f1, f2 = b"".join(fq[0]), b"".join(fq[1])
    with step2.Step2Context(0) as ctx:
        step1.extract_reads(f1, f2, flags=step1.NO_PQ | step1.NO_FETCH, ctx=ctx)
        ctx.count_kmers(7, 4); ctx.build_graph(None); ctx.path_reads()
        r3 = step3.repath_after_step2(ctx, 200)
        r2 = ctx.fetch()
    ref3, _, _ = hbvtool.canonicalise(F.read_hbv(f"{GOLDEN}/{name}.ref.large_K.hbv"))
    assert F.hbv_to_bytes(r3.hbv, zero_padding=True) == F.hbv_to_bytes(ref3, zero_padding=True)
    o1 = O1.run(f1, f2)
    c1, _ = F.unpack_bases(o1["packed"], o1["byte_off"], o1["read_len"])
    o2 = O.run(c1, o1["quals"], np.concatenate([[0], np.cumsum(o1["read_len"])]).astype(np.uint64))
    assert F.hbv_to_bytes(r2.hbv) == F.hbv_to_bytes(O.to_hbv(o2)) and np.array_equal(r2.path_edges, o2.path_edges)
    o3 = O3.run(O.to_hbv(o2), (o2.path_offset, o2.path_off, o2.path_edges), 200)
    assert F.hbv_to_bytes(r3.hbv) == F.hbv_to_bytes(O3.to_hbv(o3))
    assert np.array_equal(r3.path_offset, o3.path_offset) and np.array_equal(r3.path_off, o3.path_off) and np.array_equal(r3.path_edges, o3.path_edges)


def test_gpu_pipeline_steps_1_to_3_with_the_references_file_names(tmp_path):
    """python -m w2rap_contigger_amd.pipeline: --from_step 1 --to_step 3 in one process against runs split at every step boundary
    (files written by one run, read by the next): the same files; the graphs are the canonicalised reference graphs"""
    from w2rap_contigger_amd import hbvtool, pipeline
    name = "repeats_snps"
    pk, bo, ln = F.read_fastb(f"{GOLDEN}/{name}.fastb")
    codes, off = F.unpack_bases(pk, bo, ln)
    quals, _ = F.qualp_to_raw(*F.read_qualp(f"{GOLDEN}/{name}.qualp"))
    fq = [[], []]
    for r in range(len(ln)):
        a, b = int(off[r]), int(off[r + 1])
        fq[r & 1].append(b"@p%d/%d\n" % (r >> 1, 1 + (r & 1)) + np.frombuffer(b"ACGT", np.uint8)[codes[a:b]].tobytes() + b"\n+\n" + (quals[a:b] + 33).astype(np.uint8).tobytes() + b"\n")
    for k in (0, 1):
        open(tmp_path / f"r{k + 1}.fastq", "wb").write(b"".join(fq[k]))
    reads = f"{tmp_path}/r1.fastq,{tmp_path}/r2.fastq"
    one = tmp_path / "one"; split = tmp_path / "split"
    assert pipeline.main(["-r", reads, "-o", str(one), "-p", "asm", "--dump_all", "1", "-t", "8", "-m", "100"]) == 0
    for a, b in ((1, 1), (2, 2), (3, 3)):
        assert pipeline.main(["-r", reads, "-o", str(split), "-p", "asm", "--from_step", str(a), "--to_step", str(b)]) == 0
    files = ["frag_reads_orig.fastb", "frag_reads_orig.qualp", "small_K.freqs", "asm.small_K.hbv", "asm.small_K.paths", "asm.large_K.hbv", "asm.large_K.paths",
             "asm.first.frags.dist"]
    for f in files:
        assert open(one / f, "rb").read() == open(split / f, "rb").read(), f
    assert open(one / "small_K.freqs").read() == open(f"{GOLDEN}/{name}.ref.freqs").read()
    assert open(one / "asm.first.frags.dist").read() == open(f"{GOLDEN}/{name}.ref.frags.dist").read()
    ref2, _, _ = hbvtool.canonicalise(F.read_hbv(f"{GOLDEN}/{name}.ref.hbv"))
    assert open(one / "asm.small_K.hbv", "rb").read() == F.hbv_to_bytes(ref2)
    ref3, _, _ = hbvtool.canonicalise(F.read_hbv(f"{GOLDEN}/{name}.ref.large_K.hbv"))
    assert F.hbv_to_bytes(F.read_hbv(str(one / "asm.large_K.hbv")), zero_padding=True) == F.hbv_to_bytes(ref3, zero_padding=True)
    import io
    out = io.StringIO()
    assert hbvtool.diff(str(one / "asm.large_K"), f"{GOLDEN}/{name}.ref.large_K", out) == 0, out.getvalue()
    assert pipeline.main(["-r", reads, "-o", str(one), "-p", "asm", "--from_step", "2", "--to_step", "5"]) == 1      # steps 4-7 are the reference's
    # without --dump_all the read files are written all the same (the reference writes them when `dump_all || to_step < 6`,
    # w2rap-contigger.cc:312-318, and its --from_step 4 loads them again, :322-328), the small-K files are not (:343-347)
    hand = tmp_path / "handover"
    assert pipeline.main(["-r", reads, "-o", str(hand), "-p", "asm", "--from_step", "1", "--to_step", "3"]) == 0
    for f in ("frag_reads_orig.fastb", "frag_reads_orig.qualp", "small_K.freqs", "asm.large_K.hbv", "asm.large_K.paths", "asm.first.frags.dist"):
        assert open(hand / f, "rb").read() == open(one / f, "rb").read(), f
    assert not (hand / "asm.small_K.hbv").exists()


def test_gpu_step1_interleaved_file_and_file_grouping(tmp_path):
    """one fastq file with alternating mates (W2RAP_STEP1_INTERLEAVED) gives the pair's golden files; several files are grouped by their
    first read names as the reference groups them (extract_read_files / the tool) -- against the oracle's restatement, which is pinned
    to the reference binary in tests/test_step1_oracle.py"""
    from test_step1_oracle import _interleave
    f1, f2 = _fq()
    inter = _interleave(f1, f2)
    r = step1.extract_reads(inter, b"", flags=step1.INTERLEAVED)
    _same(r, O1.run(f1, f2))
    F.write_fastb(tmp_path / "a.fastb", r.packed, r.byte_off, r.read_len); F.write_qualp_blobs(tmp_path / "a.qualp", r.pq, r.pq_off)
    assert open(tmp_path / "a.fastb", "rb").read() == open(os.path.join(GOLDEN, "step1.ref.fastb"), "rb").read()
    assert open(tmp_path / "a.qualp", "rb").read() == open(os.path.join(GOLDEN, "step1.ref.qualp"), "rb").read()
    # fatal as in the reference
    for text, msg in ((b"@r1\nACGT\n+\nIIII\n" * 3, "even number of entries"), (b"@r1\nACGT\n+\nIIII\n@r2\nAC\n", "incomplete record"),
                      (b"@r1\nACGT\n+\nIII\n@r2\nAC\n+\nII\n", "inconsistent base/quality lengths"), (b"@r1\nACXT\n+\nIIII\n" * 3, "illegal base character")):
        with pytest.raises(RuntimeError, match=msg):
            O1.run_single(text)
        with pytest.raises(step2.Step2Error, match=msg):
            step1.extract_reads(text, b"", flags=step1.INTERLEAVED)
    for bad in (b">r1\nACGT\n+\nIIII\n", b"@ r\nACGT\n+\nIIII\n", b""):
        with pytest.raises(step2.Step2Error, match="first line"):
            step1.extract_read_files([bad])
    with pytest.raises(step2.Step2Error, match="more than two"):
        step1.extract_read_files([b"@a/1\nA\n+\nI\n", b"@a/2\nA\n+\nI\n", b"@a 3\nA\n+\nI\n"])
    # grouping: a pair (given as r2, r1: kept in that order), a file of its own sorted in front of it, another behind
    single_a = b"@aaa x\nACGTN\n+\nIIII#\n@aaa y\nTTGCA\n+\n#IIII\n"
    b2 = f2.split(b"\n"); b2[0] = b"@zzz/2 renamed"; n = (len(b2) // 4) & ~1
    lone = b"\n".join(b2[:4 * n]) + b"\n"
    texts = [f2, single_a, lone, f1]
    assert step1.plan_files(texts) == O1.plan_files(texts) == [(1,), (0, 3), (2,)]
    g = step1.extract_read_files(texts)
    o = O1.run_files(texts)
    assert np.array_equal(g.packed, o["packed"]) and np.array_equal(g.byte_off, o["byte_off"]) and np.array_equal(g.read_len, o["read_len"])
    assert np.array_equal(g.pq, o["pq"]) and np.array_equal(g.pq_off, o["pq_off"]) and np.array_equal(g.quals, o["quals"])
    # the tool on the same files
    names = []
    for k, t in enumerate(texts):
        names.append(str(tmp_path / f"f{k}.fastq")); open(names[-1], "wb").write(t)
    exe = os.path.join(ROOT, "w2rap_contigger_amd", "w2rap-step1")
    d = tmp_path / "tool"; d.mkdir()
    subprocess.run([exe, "-r", ",".join(names), "-o", str(d)], check=True, capture_output=True)
    F.write_fastb(tmp_path / "o.fastb", o["packed"], o["byte_off"], o["read_len"]); F.write_qualp_blobs(tmp_path / "o.qualp", o["pq"], o["pq_off"])
    assert open(d / "frag_reads_orig.fastb", "rb").read() == open(tmp_path / "o.fastb", "rb").read()
    assert open(d / "frag_reads_orig.qualp", "rb").read() == open(tmp_path / "o.qualp", "rb").read()
    r = subprocess.run([exe, "-r", str(tmp_path / "f1.fastq") + "," + names[0], "-o", str(d)], capture_output=True)      # single_a + f2: both unpaired; fine
    assert r.returncode == 0
    open(tmp_path / "bad.fastq", "wb").write(b">x\nA\n+\nI\n")
    r = subprocess.run([exe, "-r", str(tmp_path / "bad.fastq"), "-o", str(d)], capture_output=True)
    assert r.returncode == 1 and b"first line" in r.stderr
    # the pipeline module with one interleaved file: Step 1 straight into Step 2's context
    from w2rap_contigger_amd import pipeline
    open(tmp_path / "inter.fastq", "wb").write(inter)
    assert pipeline.main(["-r", str(tmp_path / "inter.fastq"), "-o", str(tmp_path / "p"), "-p", "x", "--to_step", "2", "--dump_all", "1"]) == 0
    assert open(tmp_path / "p" / "frag_reads_orig.fastb", "rb").read() == open(os.path.join(GOLDEN, "step1.ref.fastb"), "rb").read()
