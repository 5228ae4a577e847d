#!/usr/bin/env python3
"""Per-kernel milliseconds of one Step 2 on the planted workload of bench.py (two haplotypes + repeat families):
python3 tools/gpu_planted_kernels.py [reads]   (W2RAP_PATH_INDEX=1: read pathing through the minimizer-sampled index)"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from w2rap_contigger_amd import step2
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 50_000_000
dev = torch.device("cuda", 0)
d = bench.planted_reads(n, 4343, dev)
torch.cuda.synchronize(); torch.cuda.empty_cache()
with step2.Step2Context(0) as c:
    c.set_reads_device(d["n"], d["packed"].data_ptr(), d["byte_off"].data_ptr(), d["read_len"].data_ptr(), d["quals"].data_ptr(), d["qual_off"].data_ptr(), keepalive=d)
    for it in range(3):
        c.profile(reset=True)
        t0 = time.perf_counter(); c.count_kmers(7, 4); t1 = time.perf_counter(); c.build_graph(None); t2 = time.perf_counter(); c.path_reads(); torch.cuda.synchronize(); t3 = time.perf_counter()
    p = c.profile(reset=True)
    print(f"count {1e3 * (t1 - t0):.1f} graph {1e3 * (t2 - t1):.1f} path {1e3 * (t3 - t2):.1f} ms; index={os.environ.get('W2RAP_PATH_INDEX', '0')}")
    print({k: round(v[0], 2) for k, v in sorted(p.items(), key=lambda kv: -kv[1][0]) if v[0] > 0.2})
    print(c.counts())
