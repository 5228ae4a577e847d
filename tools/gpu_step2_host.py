#!/usr/bin/env python3
"""PCIe-inclusive Step 2: the one-shot entry point w2rap_step2_run with HOST buffers (upload of bases and qualities, compute, download of
graph and paths), against the device-resident step.   usage: gpu_step2_host.py [reads=50e6] [devices, e.g. 0,0]"""
import json, os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from w2rap_contigger_amd import step2, synth

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 50_000_000
devices = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else None      # e.g. 0,0: two ranks of the C-ABI multi-GPU path on one GPU
d = synth.generate_reads_device(n, n * 5, 42, device="cuda")
pk = d["packed"].cpu().numpy().reshape(-1); qs = d["quals"].cpu().numpy().reshape(-1)
bo = d["byte_off"].cpu().numpy().astype(np.uint64); qo = d["qual_off"].cpu().numpy().astype(np.uint64); ln = d["read_len"].cpu().numpy().astype(np.uint32)
del d; torch.cuda.empty_cache()
out = {"reads": n, "devices": devices, "host_input_bytes": int(pk.nbytes + qs.nbytes + bo.nbytes + qo.nbytes + ln.nbytes)}
for it in range(3):
    t0 = time.perf_counter()
    tm = {}
    res = step2.build_read_qgraph(pk, bo, ln, quals=qs, qual_off=qo, timing=tm, devices=devices)
    out[f"wall_s_{it}"] = time.perf_counter() - t0
    out[f"run_s_{it}"] = tm["run_s"]                       # w2rap_step2_run alone, without this wrapper's numpy copies of the result
out["device_ms"] = {"count": res.ms_count, "graph": res.ms_graph, "path": res.ms_path}
out["kmers_per_s_pcie_inclusive"] = res.n_kmer_instances / out["run_s_1"]
out["output_bytes"] = int(res.path_edges.nbytes + res.path_off.nbytes + res.path_offset.nbytes + res.hbv.edge_packed.nbytes)
print(json.dumps(out))
