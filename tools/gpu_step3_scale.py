#!/usr/bin/env python3
"""Step 2 -> Step 3 on the GPU at scale: python3 tools/gpu_step3_scale.py [reads] [snp_every] [check]
reads: synthetic PE150 reads (30x of a genome of reads*5 bases); snp_every > 0: a second haplotype with one SNP per that many
bases (half the reads from each) so that read paths cross small-K edges; check=1: compare Step 3 with the oracle (CPU, slow).
Prints the phase times, the per-kernel profile and the sizes."""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from w2rap_contigger_amd import formats as F, step2, step3, synth

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000
snp = int(sys.argv[2]) if len(sys.argv) > 2 else 0
check = len(sys.argv) > 3 and sys.argv[3] == "1"
dev = torch.device("cuda", 0)
G = n * 5
if snp:
    rng = np.random.default_rng(5)
    g = rng.integers(0, 4, G // 2, dtype=np.uint8)
    h2 = g.copy()
    pos = rng.choice(np.arange(500, len(g) - 500), len(g) // snp, replace=False)
    h2[pos] = (h2[pos] + 1 + rng.integers(0, 3, len(pos))) & 3
    genome = torch.from_numpy(np.concatenate([g, h2])).to(dev)      # (reads that straddle the seam are a negligible oddity)
    d = synth.generate_reads_device(n, len(genome), 42, device=dev, genome=genome)
else:
    d = synth.generate_reads_device(n, G, 42, device=dev)
d.pop("genome", None)
torch.cuda.synchronize()
with step2.Step2Context(0) as ctx:
    ctx.set_reads_device(d["n"], d["packed"].data_ptr(), d["byte_off"].data_ptr(), d["read_len"].data_ptr(), d["quals"].data_ptr(), d["qual_off"].data_ptr(), keepalive=d)
    ctx.count_kmers(7, 4); ctx.build_graph(None); ctx.path_reads()
    for it in range(2):                                   # steps 2 and 3 in one process: graph and paths stay in HBM
        t0 = time.perf_counter()
        r3c = step3.repath_after_step2(ctx, 200)
        wallc = time.perf_counter() - t0
    print(f"step 3 behind step 2 (device-resident): wall {wallc * 1e3:.1f} ms; device ms: places {r3c.ms_places:.1f} dict {r3c.ms_dict:.1f} graph {r3c.ms_graph:.1f} paths {r3c.ms_paths:.1f}")
    for k, v in sorted(step3.profile().items(), key=lambda kv: -kv[1][0])[:14]:
        print(f"  {k:24s} {v[0]:9.3f} ms  {v[1]} launches")
    r2 = ctx.fetch()
del d; torch.cuda.empty_cache()
paths = (r2.path_offset, r2.path_off, r2.path_edges)
print(f"step 2: {r2.hbv.n_edges} edge objects, {len(r2.path_offset)} reads, {int((np.diff(r2.path_off.astype(np.int64)) > 1).sum())} multi-edge paths", flush=True)
for it in range(2):
    t0 = time.perf_counter()
    r3 = step3.repath_in_memory(r2.hbv, paths, 200)
    wall = time.perf_counter() - t0
print(f"step 3: wall {wall * 1e3:.1f} ms (with upload/download); device ms: places {r3.ms_places:.1f} dict {r3.ms_dict:.1f} graph {r3.ms_graph:.1f} paths {r3.ms_paths:.1f}")
print(f"  places {r3.n_places} unique {r3.n_unique_places} bases {r3.n_place_bases} K2-mers {r3.n_kmer_instances} distinct {r3.n_kmers_distinct} unipaths {r3.n_unipaths} "
      f"objects {r3.hbv.n_edges} vertices {r3.hbv.n_vertices} multi-edge large-K paths {int((np.diff(r3.path_off.astype(np.int64)) > 1).sum())}")
same = (F.hbv_to_bytes(r3.hbv) == F.hbv_to_bytes(r3c.hbv) and np.array_equal(r3.path_edges, r3c.path_edges) and np.array_equal(r3.path_offset, r3c.path_offset))
print("host-buffer entry == device-resident entry:", same)
if not same:
    sys.exit(1)
if check:
    from oracle import oracle3 as O3
    t0 = time.perf_counter()
    r = O3.run(r2.hbv, paths, 200)
    print(f"oracle: {time.perf_counter() - t0:.1f} s")
    ok = (F.hbv_to_bytes(r3.hbv) == F.hbv_to_bytes(O3.to_hbv(r)) and
          F.paths_to_bytes(r3.path_offset, r3.path_off, r3.path_edges) == F.paths_to_bytes(r.path_offset, r.path_off, r.path_edges) and
          np.array_equal(r3.inv, r.inv) and np.array_equal(r3.frag_count.astype(np.float64), r.frag))
    print("PARITY", "ok" if ok else "MISMATCH")
    sys.exit(0 if ok else 1)
