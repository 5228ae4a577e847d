#!/usr/bin/env python3
"""One-off: where the device memory of the sharded path goes.  W2RAP_TRACE_MEM=1 makes the library print its live blocks at every new peak;
this runs the per-GPU share of a config at world 1 (gloo group of one rank) and prints the phases' peaks.
    python3 tools/gpu_mem_trace.py reads genome [n_passes] [sharded=1]"""
import os, sys, socket, json
os.environ.setdefault("W2RAP_TRACE_MEM", "1")
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from w2rap_contigger_amd import step2, synth, dist as wd
import torch.distributed as dist
import test_gpu_scale as T
n = int(float(sys.argv[1])); g = int(float(sys.argv[2])); P = int(sys.argv[3]) if len(sys.argv) > 3 else 1
sharded = (sys.argv[4] != "0") if len(sys.argv) > 4 else True
s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
d = T._reads(synth, n, g, 4404)
torch.cuda.reset_peak_memory_stats()
with step2.Step2Context(0) as c:
    T._set(c, d)
    c.device_peak_bytes(reset=True)
    if sharded:
        st, info, peaks = T._sharded_step(wd, c, n_passes=P)
    else:
        st = c.count_kmers(7, 4); pc = c.device_peak_bytes(reset=True); c.build_graph(None); pg = c.device_peak_bytes(reset=True); c.path_reads(); pp = c.device_peak_bytes(reset=True)
        peaks = dict(count=pc, graph=pg, path=pp)
    if not sharded:
        peaks = {k: dict(library=v, torch=0, total=v) for k, v in peaks.items()}
    print(json.dumps(dict(S=int(st["S"]), M=int(st["M"]), peaks_GB={k: {a: b / 1e9 for a, b in v.items()} for k, v in peaks.items()},
                          library_per_solid={k: v["library"] / st["S"] for k, v in peaks.items()})))
dist.destroy_process_group()
