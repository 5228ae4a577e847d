#!/usr/bin/env python3
"""One-off: the path every multi-GPU run takes -- sharded graph + minimizer index + exact table, behind the one in-process call
with `world` ranks on this GPU and the hooks that give them the query / segment shares of an 8-rank job -- against the REAL reference
(oracle/_ref/ref_step2, all host cores) on the SAME reads, at a size the unit tests cannot afford: the reference's edge numbering replayed, the
three output files compared byte for byte (bench.py's verdict functions).  Also the one-GPU dictionary path and the one-GPU index path.
    python3 tools/gpu_parity_ref_scale.py reads [planted=1] [world=4]"""
import json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from w2rap_contigger_amd import step2
n = int(float(sys.argv[1])); planted = (sys.argv[2] != "0") if len(sys.argv) > 2 else True; world = int(sys.argv[3]) if len(sys.argv) > 3 else 4
dev = torch.device("cuda", 0)
mem_kb = {l.split(":")[0]: int(l.split()[1]) for l in open("/proc/meminfo") if l.split(":")[0] in ("MemTotal", "MemAvailable")}
print(f"host: {os.cpu_count()} cores, MemTotal {mem_kb['MemTotal'] / 1e6:.0f} GB, MemAvailable {mem_kb['MemAvailable'] / 1e6:.0f} GB", flush=True)
# the reference holds 24 B per k-mer instance (~85 per read) in its leaves and as much again in the merges, plus the reads
need_gb = n * 85 * 24 * 2.5 / 1e9 + n * 600 / 1e9
if need_gb > 0.7 * mem_kb["MemAvailable"] / 1e6:
    n2 = int(0.7 * mem_kb["MemAvailable"] / 1e6 / (85 * 24 * 2.5 / 1e9 + 600 / 1e9)) // 2 * 2
    print(f"{n} reads would need ~{need_gb:.0f} GB of host memory for the reference: {n2} reads instead", flush=True)
    n = n2
t0 = time.time()
secs, cores, kind, sample, d, leaves, ref = bench.cpu_baseline(n, n * 5, 6161, dev, planted=planted)
print(f"reference ({kind}, {cores} threads): {secs:.1f} s of Step 2 on {d['n']} {'planted' if planted else 'uniform'} reads ({time.time() - t0:.1f} s with the file I/O)", flush=True)
assert ref is not None, "oracle/_ref/ref_step2 is not on this box"
out = {"reads": int(d["n"]), "planted": planted, "reference_seconds": secs, "threads": cores}
with step2.Step2Context(0) as c:
    c.set_reads_device(d["n"], d["packed"].data_ptr(), d["byte_off"].data_ptr(), d["read_len"].data_ptr(), d["quals"].data_ptr(), d["qual_off"].data_ptr(), keepalive=d)
    out["one_gpu_dictionary"] = bench.same_as_reference(c, ref)
os.environ["W2RAP_PATH_INDEX"] = "1"
with step2.Step2Context(0) as c:
    c.set_reads_device(d["n"], d["packed"].data_ptr(), d["byte_off"].data_ptr(), d["read_len"].data_ptr(), d["quals"].data_ptr(), d["qual_off"].data_ptr(), keepalive=d)
    out["one_gpu_index"] = bench.same_as_reference(c, ref)
os.environ.pop("W2RAP_PATH_INDEX")
out["sharded"] = bench.sharded_as_reference(d, ref, 0, world=world, cut=27, virtual=8)
for k in ("one_gpu_dictionary", "one_gpu_index", "sharded"):
    v = out[k]
    print(f"{k}: same as the reference {v['same_graph_as_gpu']} (freqs {v['freqs_bytes_equal']}, hbv {v['hbv_bytes_equal']}, paths {v['paths_bytes_equal']}; "
          f"reads differing by parallel-edge ties {v['path_reads_differing_by_parallel_edge_ties']}, otherwise {v['path_reads_differing_otherwise']}); "
          f"{v['edge_objects']} edge objects, {v['kmers_solid']} solid k-mers" + (f"; {v['route']}" if "route" in v else ""), flush=True)
print(json.dumps(out))
sys.exit(0 if all(out[k]["same_graph_as_gpu"] for k in ("one_gpu_dictionary", "one_gpu_index", "sharded")) else 1)
