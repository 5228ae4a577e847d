#!/usr/bin/env python3
"""Per-kernel hipEvent times of one count phase (K0-K5) at bench scale: python3 tools/gpu_count_profile.py [reads]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from w2rap_contigger_amd import step2, synth
dev = torch.device("cuda", 0)
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 50_000_000
g = torch.randint(0, 4, (n * 5,), dtype=torch.uint8, device=dev, generator=torch.Generator(device=dev).manual_seed(42))
d = synth.generate_reads_device(n, n * 5, 42, device=dev, genome=g); del g; d.pop("genome", None)
torch.cuda.synchronize(); torch.cuda.empty_cache()
with step2.Step2Context(0) as ctx:
    ctx.set_reads_device(d["n"], d["packed"].data_ptr(), d["byte_off"].data_ptr(), d["read_len"].data_ptr(), d["quals"].data_ptr(),
                         d["qual_off"].data_ptr(), keepalive=d)
    for it in range(2):
        ctx.count_kmers(7, 4); p = ctx.profile()
    for k, v in sorted(p.items(), key=lambda kv: -kv[1][0]):
        print(f"{k:24s} {v[0]:8.2f} ms  {v[1]} launches")
