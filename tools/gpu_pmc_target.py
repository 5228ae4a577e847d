#!/usr/bin/env python3
"""Profiling target (run directly under rocprofv3): one warm-up + one timed whole Step 2 at N reads."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from w2rap_contigger_amd import step2, synth
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 50_000_000
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 1
d = synth.generate_reads_device(n, n * 5, 42, device="cuda")
torch.cuda.synchronize(); torch.cuda.empty_cache()
with step2.Step2Context(0) as ctx:
    ctx.set_reads_device(d["n"], d["packed"].data_ptr(), d["byte_off"].data_ptr(), d["read_len"].data_ptr(), d["quals"].data_ptr(), d["qual_off"].data_ptr(), keepalive=d)
    for it in range(iters):
        st = ctx.count_kmers(7, 4); ctx.build_graph(None); ctx.path_reads()
    print("M", st["M"], "S", st["S"], {k: round(v[0] / iters, 2) for k, v in ctx.profile().items() if v[0] / iters > 1})
