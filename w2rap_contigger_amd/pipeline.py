"""Steps 1 to 3 of w2rap-contigger with the reference's flags and file names, on the GPU.

Mirrors ``w2rap-contigger -r r1.fastq,r2.fastq -o OUT -p PREFIX [-K 200] [--min_freq 4] [--min_qual 7] --from_step A --to_step B`` for
1 <= A <= B <= 3 (src/modules/w2rap-contigger.cc:300-383): consecutive steps run in one process hand their data over in HBM
(w2rap_step1_run_into_step2, the staged Step-2 entry points, w2rap_step3_run_after_step2), a run that starts at step 2 or 3 loads the
files the previous step wrote, and every step writes what the reference writes:
    step 1: OUT/frag_reads_orig.fastb, .qualp  -- ALWAYS (the reference writes them when `dump_all || to_step < 6`,
            w2rap-contigger.cc:312-318, and its steps 2..6 load them again, :322-328: a hand-over to `--from_step 4` needs them)
    step 2: OUT/PREFIX.small_K.hbv, .paths (last step or dump_all, :343-347), OUT/small_K.freqs (always, BuildReadQGraph.cc:1108)
    step 3: OUT/PREFIX.large_K.hbv, .paths, OUT/PREFIX.first.frags.dist
Steps 4-7 are the reference's (``w2rap-contigger ... --from_step 4``).  The HIP library is the only implementation (no CPU fallback).

    python -m w2rap_contigger_amd.pipeline -r r1.fastq.gz,r2.fastq.gz -o OUT -p asm --from_step 1 --to_step 3
"""
from __future__ import annotations

import argparse
import os
import sys

from . import formats as F, step1, step2, step3


def run(read_files, out_dir, prefix, large_k=200, min_freq=4, min_qual=7, from_step=1, to_step=3, dump_all=False, device=0, log=print, extend_paths=False):
    if not (1 <= from_step <= to_step <= 3):
        raise ValueError("steps 1..3 only (from_step <= to_step); steps 4-7 are the reference's")
    os.makedirs(out_dir, exist_ok=True)
    pre = os.path.join(out_dir, prefix)
    out = {}
    with step2.Step2Context(device) as ctx:
        if from_step == 1:
            log("--== Step 1: Reading input files ==--")
            names = [x for x in read_files.split(",") if x]
            texts = [step1._slurp(x) for x in names]
            plan = step1.plan_files(texts, names)
            last = to_step == 1
            if len(plan) == 1:                                      # one pair or one interleaved file: straight into Step 2's context
                g = plan[0]
                s1 = step1.extract_reads(texts[g[0]], texts[g[1]] if len(g) == 2 else b"", device,
                                         flags=(step1.INTERLEAVED if len(g) == 1 else 0),
                                         ctx=None if last else ctx)
            else:                                                   # several groups: concatenated on the host, handed over as host arrays
                s1 = step1.extract_read_files(texts, device, names)
                if not last:
                    ctx.set_reads_host(s1.packed, s1.byte_off, s1.read_len, quals=s1.quals, qual_off=s1.qual_off)
            out["step1"] = s1
            # to_step <= 3 < 6: the reference always writes the read files here (w2rap-contigger.cc:312-318)
            if True:
                F.write_fastb(os.path.join(out_dir, "frag_reads_orig.fastb"), s1.packed, s1.byte_off, s1.read_len)
                F.write_qualp_blobs(os.path.join(out_dir, "frag_reads_orig.qualp"), s1.pq, s1.pq_off)
            log(f"Reading input files DONE: {s1.n_reads} reads, {s1.n_bases} bases")
        if from_step <= 2 <= to_step:
            log("--== Step 2: Building first (small K) graph ==--")
            if from_step == 2:
                pk, bo, ln = F.read_fastb(os.path.join(out_dir, "frag_reads_orig.fastb"))
                pq, po = F.read_qualp(os.path.join(out_dir, "frag_reads_orig.qualp"))
                ctx.set_reads_host(pk, bo, ln, pq=pq, pq_off=po)
            st = ctx.count_kmers(min_qual, min_freq)
            ctx.build_graph(None)
            ctx.path_reads()
            with open(os.path.join(out_dir, "small_K.freqs"), "w") as f:
                f.write(F.freqs_text(st["hist"]))
            if to_step == 2 or dump_all:
                r2 = ctx.fetch()
                out["step2"] = r2
                F.write_hbv(pre + ".small_K.hbv", r2.hbv)
                F.write_paths(pre + ".small_K.paths", r2.path_offset, r2.path_off, r2.path_edges)
            log(f"Building first graph DONE: {st['M']} k-mer instances, {st['S']} solid, {ctx.counts()['edge_objects']} edge objects")
        if to_step == 3:
            log("--== Step 3: Repathing to second (large K) graph ==--")
            if from_step == 3:
                r3 = step3.run_step3_files(out_dir, prefix, large_k, device, extend_paths=extend_paths)
            else:
                r3 = step3.repath_after_step2(ctx, large_k, extend_paths=extend_paths)
                F.write_hbv(pre + ".large_K.hbv", r3.hbv)
                F.write_paths(pre + ".large_K.paths", r3.path_offset, r3.path_off, r3.path_edges)
                with open(pre + ".first.frags.dist", "w") as f:
                    f.write(step3.frags_text(r3.frag_count))
            out["step3"] = r3
            log(f"Repathing to second graph DONE: {r3.n_unique_places} unique places, {r3.hbv.n_edges} large-K edge objects")
    return out


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(prog="python -m w2rap_contigger_amd.pipeline", description="w2rap-contigger steps 1-3 on the GPU")
    ap.add_argument("-r", "--read_files", default="")
    ap.add_argument("-o", "--out_dir", required=True)
    ap.add_argument("-p", "--prefix", required=True)
    ap.add_argument("-K", "--large_k", type=int, default=200)
    ap.add_argument("--min_freq", type=int, default=4)
    ap.add_argument("--min_qual", type=int, default=7)
    ap.add_argument("--from_step", type=int, default=1)
    ap.add_argument("--to_step", type=int, default=3)
    ap.add_argument("--dump_all", type=int, default=0)
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--extend_paths", default="0", help="the reference's --extend_paths (TCLAP bool: 0/1/true/false), Repath.cc:72-96")
    for ignored in ("-t", "-m", "-d", "--tmp_dir", "-s", "--pair_sample"):       # the reference's resource flags: accepted, not needed here
        ap.add_argument(ignored, default=None, help=argparse.SUPPRESS)
    a = ap.parse_args(argv)
    if a.from_step == 1 and not a.read_files:
        ap.error("-r is required from step 1")
    try:
        run(a.read_files, a.out_dir, a.prefix, a.large_k, a.min_freq, a.min_qual, a.from_step, a.to_step, bool(a.dump_all), a.device,
            extend_paths=str(a.extend_paths).lower() in ("1", "true"))
    except (step2.Step2Error, ValueError, OSError) as e:
        print(f"w2rap pipeline: {e}", file=sys.stderr)
        return 1
    return 0


if __name__ == "__main__":
    sys.exit(main())
