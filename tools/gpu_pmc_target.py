#!/usr/bin/env python3
"""One (or k) complete pass of a step on synthetic reads generated in HBM: the target of
`rocprofv3 --pmc ... -- python3 tools/gpu_pmc_target.py 5e7 1 [step2|step1|step3]`."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from w2rap_contigger_amd import step2, synth

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 50_000_000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
mode = sys.argv[3] if len(sys.argv) > 3 else "step2"
dev = torch.device("cuda", 0)
if mode == "step3":
    import bench
    from w2rap_contigger_amd import step3
    d = bench.diploid_reads(n, 2000, 42, dev)
else:
    genome = torch.randint(0, 4, (n * 5,), dtype=torch.uint8, device=dev, generator=torch.Generator(device=dev).manual_seed(42))
    d = synth.generate_reads_device(n, n * 5, 42, device=dev, genome=genome)
    del genome
d.pop("genome", None)
torch.cuda.synchronize(); torch.cuda.empty_cache()
if mode == "step1":
    import bench
    from w2rap_contigger_amd import step1
    t1, _ = bench.fastq_text_device(d, 0, dev); t2, _ = bench.fastq_text_device(d, 1, dev)
    del d; torch.cuda.synchronize(); torch.cuda.empty_cache()
    with step2.Step2Context(0) as ctx:
        for _ in range(reps):
            r = step1.extract_reads((t1.data_ptr(), t1.numel()), (t2.data_ptr(), t2.numel()), flags=step1.NO_FETCH, ctx=ctx)
    print("reads", r.n_reads, "bases", r.n_bases, "pq", r.n_pq_bytes)
    sys.exit(0)
with step2.Step2Context(0) as ctx:
    ctx.set_reads_device(d["n"], d["packed"].data_ptr(), d["byte_off"].data_ptr(), d["read_len"].data_ptr(),
                         d["quals"].data_ptr(), d["qual_off"].data_ptr(), keepalive=d)
    for _ in range(reps if mode == "step2" else 1):
        st = ctx.count_kmers(7, 4); ctx.build_graph(None); ctx.path_reads()
    print("M", st["M"], "D", st["D"], "S", st["S"])
    if mode == "step3":
        for _ in range(reps):
            r3 = step3.repath_after_step2(ctx, 200, fetch=False)
        print("N2", r3.n_kmer_instances)
