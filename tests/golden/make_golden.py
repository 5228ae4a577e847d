#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by running the REAL reference
Step 2 (oracle/_ref/ref_step2, built from /root/reference by oracle/Makefile).

Run in the build container only (the reference does not travel):
    python tests/golden/make_golden.py

For every fixture <name> this writes
    <name>.fastb / <name>.qualp      inputs in the reference's own Step-1 output format
    <name>.ref.hbv / <name>.ref.paths / <name>.ref.freqs
                                      the reference's Step-2 outputs (-t 1, deterministic)
    <name>.ref8.hbv / <name>.ref8.paths   the same with 8 threads (different edge order)
and, from the REAL reference Step 3 (oracle/_ref/ref_step3, K2 = 200) run on those Step-2 outputs,
    <name>.ref.large_K.hbv / .paths / <name>.ref.frags.dist      (1 thread, from <name>.ref.*)
    <name>.ref8.large_K.hbv / .paths                              (8 threads, from <name>.ref8.*)
and, for Step 1 (fastq ingest), step1_r1.fastq / step1_r2.fastq with the REAL reference's frag_reads_orig.fastb/.qualp for them
(step1.ref.fastb / step1.ref.qualp, from oracle/_ref/ref_step1; `make_golden.py step1`).
and, from the REAL reference hbv2gfa tool (oracle/_ref/ref_hbv2gfa -g 20; `make_golden.py gfa`), for the graphs named in GFA_GOLDENS
    <graph>.ref_raw.gfa and <graph>.ref_gfa_stats.txt (its stdout between "=== Graph stats === " and "Dumping gfa").
All of these are data (inputs and expected outputs); no reference source is stored.
"""
import os
import shutil
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from w2rap_contigger_amd import formats as F, synth  # noqa: E402
from oracle import oracle as O  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))

FIXTURES = {
    # name: (genome kind, genome seed, read pairs, read seed, with hand-made edge-case reads)
    "random20k": ("random", 11, 2000, 101, True),
    "repeats_snps": ("repeats_snps", 12, 6000, 102, False),
    "palindrome_circle": ("palindrome_circle", 13, 2300, 103, False),
}


def make_inputs(kind, gseed, n_pairs, rseed, edge_cases):
    contigs = synth.fixture_genome(kind, gseed)
    codes, quals = synth.sample_reads(contigs, n_pairs, rseed)
    codes, quals = codes.numpy(), quals.numpy()
    reads = [(codes[i], quals[i]) for i in range(len(codes))]
    if edge_cases:
        reads += synth.edge_case_reads(np.random.default_rng(rseed + 1), contigs[0])
    lens = np.array([len(r[0]) for r in reads], dtype=np.uint64)
    off = np.zeros(len(reads) + 1, dtype=np.uint64)
    np.cumsum(lens, out=off[1:])
    return np.concatenate([r[0] for r in reads]), np.concatenate([r[1] for r in reads]), off


def step3_goldens(name):
    from oracle import oracle3 as O3
    for threads, tag in ((1, "ref"), (8, "ref8")):
        with tempfile.TemporaryDirectory() as d:
            shutil.copy(os.path.join(HERE, f"{name}.{tag}.hbv"), os.path.join(d, "t.small_K.hbv"))
            shutil.copy(os.path.join(HERE, f"{name}.{tag}.paths"), os.path.join(d, "t.small_K.paths"))
            O3.run_reference3(d, "t", 200, threads)
            shutil.copy(os.path.join(d, "t.large_K.hbv"), os.path.join(HERE, f"{name}.{tag}.large_K.hbv"))
            shutil.copy(os.path.join(d, "t.large_K.paths"), os.path.join(HERE, f"{name}.{tag}.large_K.paths"))
            if threads == 1:
                shutil.copy(os.path.join(d, "t.first.frags.dist"), os.path.join(HERE, f"{name}.ref.frags.dist"))
    if name == "repeats_snps":                 # the reference's --extend_paths (Repath.cc:72-96) on the fixture with junctions
        with tempfile.TemporaryDirectory() as d:
            shutil.copy(os.path.join(HERE, f"{name}.ref.hbv"), os.path.join(d, "t.small_K.hbv"))
            shutil.copy(os.path.join(HERE, f"{name}.ref.paths"), os.path.join(d, "t.small_K.paths"))
            O3.run_reference3(d, "t", 200, 1, extend_paths=True)
            shutil.copy(os.path.join(d, "t.large_K.hbv"), os.path.join(HERE, f"{name}.ext.large_K.hbv"))
            shutil.copy(os.path.join(d, "t.large_K.paths"), os.path.join(HERE, f"{name}.ext.large_K.paths"))


def step1_goldens():
    """step1_r1.fastq / step1_r2.fastq (1000 reads of random20k, its ragged edge-case reads included, half of the Q2 bases written as N)
    and the reference's own Step-1 output for them: step1.ref.fastb / step1.ref.qualp (oracle/_ref/ref_step1)"""
    from oracle import oracle1 as O1
    pk, bo, ln = F.read_fastb(os.path.join(HERE, "random20k.fastb"))
    codes, off = F.unpack_bases(pk, bo, ln)
    quals, _ = F.qualp_to_raw(*F.read_qualp(os.path.join(HERE, "random20k.qualp")))
    off = off.astype(np.int64)
    n = len(ln)
    pick = list(range(0, 598)) + list(range(n - 402, n))
    rng = np.random.default_rng(3)
    L = "ACGT"
    f1, f2 = open(os.path.join(HERE, "step1_r1.fastq"), "w"), open(os.path.join(HERE, "step1_r2.fastq"), "w")
    for j, r in enumerate(pick):
        q = quals[off[r]:off[r + 1]]
        s = "".join(("N" if (qq == 2 and rng.random() < 0.5) else L[c]) for c, qq in zip(codes[off[r]:off[r + 1]], q))
        if j % 7 == 3:
            s = s.lower().replace("n", "N")                      # lower-case bases are legal (Base::char2Val), a lower-case n is not
        (f1 if j % 2 == 0 else f2).write(f"@read{j // 2}/{1 + j % 2} some comment\n{s}\n+\n{''.join(chr(33 + int(x)) for x in q)}\n")
    f1.close(); f2.close()
    with tempfile.TemporaryDirectory() as d:
        for f in ("step1_r1.fastq", "step1_r2.fastq"):
            shutil.copy(os.path.join(HERE, f), os.path.join(d, f))
        O1.run_reference1(d, "step1_r1.fastq,step1_r2.fastq", 1)
        shutil.copy(os.path.join(d, "frag_reads_orig.fastb"), os.path.join(HERE, "step1.ref.fastb"))
        shutil.copy(os.path.join(d, "frag_reads_orig.qualp"), os.path.join(HERE, "step1.ref.qualp"))


GFA_GOLDENS = ("palindrome_circle.ref", "repeats_snps.ref", "repeats_snps.ref.large_K")


def gfa_goldens():
    from oracle import oracle_gfa as OG
    for g in GFA_GOLDENS:
        with tempfile.TemporaryDirectory() as d:
            shutil.copy(os.path.join(HERE, g + ".hbv"), os.path.join(d, "g.hbv"))
            shutil.copy(os.path.join(HERE, g + ".paths"), os.path.join(d, "g.paths"))
            txt, gfa = OG.run_reference_gfa(d, "g", "o", 20)
        open(os.path.join(HERE, g + ".ref_raw.gfa"), "wb").write(gfa)
        open(os.path.join(HERE, g + ".ref_gfa_stats.txt"), "w").write(txt.split("=== Graph stats === \n")[1].split("Dumping gfa")[0])


def main():
    O.build(ref=True)
    if len(sys.argv) > 1 and sys.argv[1] == "gfa":
        return gfa_goldens()
    if len(sys.argv) > 1 and sys.argv[1] == "step1":
        return step1_goldens()
    if len(sys.argv) > 1 and sys.argv[1] == "step3":          # only the Step-3 goldens, from the committed Step-2 ones
        for name in FIXTURES:
            step3_goldens(name)
        return
    for name, spec in FIXTURES.items():
        codes, quals, off = make_inputs(*spec)
        fastb, qualp = os.path.join(HERE, name + ".fastb"), os.path.join(HERE, name + ".qualp")
        F.write_fastb(fastb, *F.pack_bases(codes, off))
        F.write_qualp(qualp, quals, off)
        for threads, tag in ((1, "ref"), (8, "ref8")):
            with tempfile.TemporaryDirectory() as d:
                shutil.copy(fastb, os.path.join(d, "frag_reads_orig.fastb"))
                shutil.copy(qualp, os.path.join(d, "frag_reads_orig.qualp"))
                secs = O.run_reference(d, "t", threads=threads)
                shutil.copy(os.path.join(d, "t.small_K.hbv"), os.path.join(HERE, f"{name}.{tag}.hbv"))
                shutil.copy(os.path.join(d, "t.small_K.paths"), os.path.join(HERE, f"{name}.{tag}.paths"))
                if threads == 1:
                    shutil.copy(os.path.join(d, "small_K.freqs"), os.path.join(HERE, f"{name}.ref.freqs"))
        print(f"{name}: {len(off) - 1} reads, reference Step 2 took {secs:.2f}s")
        step3_goldens(name)


if __name__ == "__main__":
    main()
