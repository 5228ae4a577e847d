#!/usr/bin/env python3
"""The planted workload of bench.py (repeats + second haplotype) at a few sizes: path-length histogram, unipath sizes; at the small size
against the oracle.   usage: gpu_planted_check.py [reads ...]"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from w2rap_contigger_amd import step2, synth, formats as F
from oracle import oracle as O
dev = torch.device("cuda", 0)
for n in [int(float(x)) for x in (sys.argv[1:] or ["1e6", "5e7"])]:
    d = bench.planted_reads(n, 4343, dev)
    with step2.Step2Context(0) as ctx:
        ctx.set_reads_device(d["n"], d["packed"].data_ptr(), d["byte_off"].data_ptr(), d["read_len"].data_ptr(), d["quals"].data_ptr(), d["qual_off"].data_ptr(), keepalive=d)
        st = ctx.count_kmers(7, 4); ctx.build_graph(None)
        torch.cuda.synchronize(); t0 = time.perf_counter(); ctx.path_reads(); torch.cuda.synchronize(); t1 = time.perf_counter()
        res = ctx.fetch()
        prof = ctx.profile(reset=True)
    pl = np.diff(res.path_off.astype(np.int64))
    nk = res.hbv.edge_len[res.fwd_xlat].astype(np.int64) - 59
    print(f"n={n} S={st['S']} unipaths={len(res.fwd_xlat)} path ms {(t1-t0)*1e3:.1f} mean path len {pl.mean():.3f}")
    print("  kernels:", {k: round(v[0], 2) for k, v in prof.items() if "path" in k})
    print("  path len hist", np.bincount(pl)[:16].tolist(), "max", int(pl.max()))
    print("  unipath k-mers: median", int(np.median(nk)), "mean", float(nk.mean()), "hist(log2)", np.bincount(np.log2(nk).astype(int)).tolist())
    if n <= 2_000_000:
        codes = synth.unpack_fixed(d["packed"], synth.READ_LEN).cpu().numpy().reshape(-1); quals = d["quals"].cpu().numpy().reshape(-1)
        off = np.arange(d["n"] + 1, dtype=np.uint64) * synth.READ_LEN
        orc = O.run(codes, quals, off)
        print("  oracle: paths equal", np.array_equal(res.path_edges, orc.path_edges) and np.array_equal(res.path_offset, orc.path_offset), "graph equal",
              F.hbv_to_bytes(res.hbv) == F.hbv_to_bytes(O.to_hbv(orc)))
    del d; torch.cuda.empty_cache()
