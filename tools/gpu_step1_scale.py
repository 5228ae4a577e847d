#!/usr/bin/env python3
"""Step 1 at BASELINE size on one MI355X: the round trip reads -> fastq text -> Step 1 gives back the generator's arrays bit for bit
(checked on a host-sized part), the reads handed to Step 2 in place give the same graph as set_reads on the generator's arrays, and what the host-text
path costs over PCIe.   usage: gpu_step1_scale.py [reads=50e6] [host_reads=4e6]"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from w2rap_contigger_amd import step1, step2, synth  # noqa: E402

n_reads = int(float(sys.argv[1])) if len(sys.argv) > 1 else 50_000_000
n_host = int(float(sys.argv[2])) if len(sys.argv) > 2 else 4_000_000
dev = torch.device("cuda", 0)
d = synth.generate_reads_device(n_reads, n_reads * 5, 42, device=dev)
d.pop("genome", None)
t1, W = bench.fastq_text_device(d, 0, dev)
t2, _ = bench.fastq_text_device(d, 1, dev)
torch.cuda.synchronize()
out = {"reads": d["n"], "fastq_bytes": t1.numel() + t2.numel()}

# 1. direct Step 2 on the generator's arrays
with step2.Step2Context(0) as ctx:
    ctx.set_reads_device(d["n"], d["packed"].data_ptr(), d["byte_off"].data_ptr(), d["read_len"].data_ptr(), d["quals"].data_ptr(), d["qual_off"].data_ptr(), keepalive=d)
    ctx.count_kmers(7, 4); ctx.build_graph(None); ctx.path_reads()
    direct = ctx.counts()
# 2. fastq text -> Step 1 -> the context's reads -> Step 2
with step2.Step2Context(0) as ctx:
    t0 = time.perf_counter()
    r = step1.extract_reads((t1.data_ptr(), t1.numel()), (t2.data_ptr(), t2.numel()), flags=step1.NO_PQ | step1.NO_FETCH, ctx=ctx)
    torch.cuda.synchronize()
    out["into_step2_ms_first_call"] = (time.perf_counter() - t0) * 1e3
    t0 = time.perf_counter()
    r = step1.extract_reads((t1.data_ptr(), t1.numel()), (t2.data_ptr(), t2.numel()), flags=step1.NO_PQ | step1.NO_FETCH, ctx=ctx)
    torch.cuda.synchronize()
    out["into_step2_ms"] = (time.perf_counter() - t0) * 1e3
    out["into_step2_kernel_ms"] = {k: v[0] for k, v in step1.profile().items()}
    ctx.count_kmers(7, 4); ctx.build_graph(None); ctx.path_reads()
    chained = ctx.counts()
out["step2_direct"] = direct
out["step2_behind_step1"] = chained
out["same_graph"] = direct == chained
assert direct == chained, (direct, chained)

# 3. host text (what w2rap_step1_run is given by the tool): upload + ingest + download of everything
nb = n_host // 2 * W
h1, h2 = t1[:nb].cpu().numpy().tobytes(), t2[:nb].cpu().numpy().tobytes()
step1.extract_reads(h1[:W * 1000], h2[:W * 1000])
t0 = time.perf_counter()
rh = step1.extract_reads(h1, h2)
wall = time.perf_counter() - t0
out["host_path"] = {"reads": rh.n_reads, "fastq_bytes": 2 * nb, "wall_s": wall, "ms_upload": rh.ms_upload, "ms_index": rh.ms_index, "ms_encode": rh.ms_encode,
                    "reads_per_s": rh.n_reads / wall, "upload_GB_per_s": 2 * nb / (rh.ms_upload * 1e-3) / 1e9}
# the round trip at this size, on the host: Step 1's arrays == the generator's
k = rh.n_reads
assert np.array_equal(rh.packed.reshape(k, -1), d["packed"][:k].cpu().numpy()) and np.array_equal(rh.quals.reshape(k, -1), d["quals"][:k].cpu().numpy())
out["host_round_trip_equal"] = True
print(json.dumps(out))
