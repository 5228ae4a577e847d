#!/usr/bin/env python3
"""Sizes of the K3 emit chunks (solid k-mers per minimizer bucket) at bench scale: what the bucket-local prune and the ranking tiles work on.
usage: gpu_chunk_hist.py [reads=50e6]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from w2rap_contigger_amd import step2, synth
from w2rap_contigger_amd.dist import dev_bytes
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 50_000_000
dev = torch.device("cuda", 0)
d = synth.generate_reads_device(n, n * 5, 42, device=dev)
with step2.Step2Context(0) as ctx:
    ctx.set_reads_device(d["n"], d["packed"].data_ptr(), d["byte_off"].data_ptr(), d["read_len"].data_ptr(), d["quals"].data_ptr(), d["qual_off"].data_ptr(), keepalive=d)
    st = ctx.count_kmers(7, 4)
    stp, cnp, nc = ctx.chunk_buffers()
    cn = dev_bytes(cnp, nc * 4, dev).view(torch.int32).to(torch.int64)
    tot = int(cn.sum())
    print(f"S {st['S']}  chunks {nc}  k-mers in chunks {tot}  mean {tot / max(nc, 1):.1f}  max {int(cn.max())}")
    for lo, hi in ((0, 64), (64, 128), (128, 256), (256, 512), (512, 1024), (1024, 1 << 30)):
        m = (cn > lo) & (cn <= hi)
        print(f"  {lo:5d} < n <= {hi:<10d} chunks {int(m.sum()):9d} ({float(m.sum()) / nc:6.3f})   k-mers {int(cn[m].sum()):11d} ({float(cn[m].sum()) / tot:6.3f})")
