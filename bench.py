#!/usr/bin/env python3
"""bench.py -- Step-2 (k=60) graph build + read pathing on N MI355X GPUs.

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one complete pass of the hot path (quality windows -> canonical 60-mer
counting -> solid dictionary -> adjacency prune -> unipaths -> vertices -> read pathing +
extension + FixPaths) over the synthetic reads, which are already resident in HBM when the
timed region starts.  Workload at N=1 = BASELINE.json configs[1]: 50 M synthetic PE150 reads
(250 Mbp genome, 30x), generated in HBM with the distributions of SURVEY.md 8d.  At N>1 the
workload is BASELINE configs[2] scaled to N GPUs: 62.5 M reads per GPU (500 M / 8) of ONE genome
of N x 312.5 Mbp (30x; at N = 8 exactly configs[2]: 500 M reads, 2.5 Gbp), reads sharded by
rank, the k-mer shuffle an RCCL all_to_all_v, dictionary / prune / unipaths sharded by bucket owner (row e-3), pathing local -- weak scaling.

Prints ONE JSON line (rank 0).  `value` = job-wide canonical k-mer instances per second over
the whole step; the phase rates, the roofline object of the dominant kernel, the count-phase
fraction of BASELINE.md section 3, the host-resident (PCIe-inclusive) rate of the one-shot C
entry point, a second workload with planted repeats and a second haplotype, and the CPU
baseline (the real reference's Step 2, oracle/_ref, timed on this box's host cores on a
bounded sample) are carried alongside.
"""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
from w2rap_contigger_amd import formats as F, step2, synth  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec, /opt/skills/guides/MI355X_MICROARCH.md
GUIDE_COPY_GBS = 6290.0        # ... and what a float4 copy reaches there ("6.29 TB/s measured, 79 %"), MI355X_MICROARCH.md:35
B_K = 41.0                     # algorithmic bytes per k-mer instance, SURVEY.md 8(d): 2*17 + 188/91 + 18*D/M
B_R = 300.0                    # algorithmic bytes per read for pathing, SURVEY.md 8(d)
# Counters of the kernels from rocprofv3 PMC passes over this very workload (separate --pmc runs of the default 50 M-read step, recipe of
# MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE in passes of their own; traffic = 2 * FETCH + WRITE, the gfx950 correction for wide reads).
# They are constants of a COMMITTED profile -- profiles/pmc_step2.json, written by tools/pmc_summary.py --json from the passes of
# tools/r04_pmc.sh, never edited by hand -- NOT counters of the run that prints the line: the line says so (`traffic_from_profile`).
# Totals per STEP; a kernel that runs as several launches per step gets its share per launch.
PMC_JSON = os.path.join(ROOT, "profiles", "pmc_step2.json")


def pmc_profile():
    try:
        with open(PMC_JSON) as f:
            return json.load(f)
    except Exception:
        return {"kernels": {}}


def pmc_of(prof, kname):
    """the profile's entry of a kernel (names in the profile carry template arguments, the library's own names do not)"""
    for k, v in prof.get("kernels", {}).items():
        if k.split("<")[0] == kname.split("<")[0]:
            return v
    return None


def _env_flag(name):
    """an environment switch the way the library reads it (step2_run.hip): set to a non-zero number"""
    v = os.environ.get(name, "").strip()
    try:
        return bool(v) and int(v) != 0
    except ValueError:
        return False


SIMDS, CLOCK_HZ = 256 * 4, 2.4e9      # MI355X: 256 CUs x 4 SIMDs, 2.4 GHz peak engine clock (MI355X_MICROARCH.md)


# The one JSON line goes to the process's ORIGINAL stdout; everything else that writes to file descriptor 1 while the bench runs (RCCL prints
# a version banner there when a process group comes up) is sent to stderr, so that stdout carries exactly one line.
_REAL_STDOUT = None


def guard_stdout():
    global _REAL_STDOUT
    if _REAL_STDOUT is None:
        sys.stdout.flush()
        _REAL_STDOUT = os.fdopen(os.dup(1), "w")
        os.dup2(2, 1)


def emit(line):
    out = _REAL_STDOUT or sys.stdout
    out.write(line + "\n")
    out.flush()


def with_copy_rate(roofline, dev):
    """adds the rate of a plain device copy on this box (the library's own 16-B-per-lane copy kernel, bytes read + written; SURVEY.md 8d: the
    roofline is quoted against the 8 TB/s spec AND against what a copy reaches) and the fraction of IT to a roofline object"""
    try:
        with step2.Step2Context(dev.index or 0) as c:
            bw = c.copy_bandwidth(4 << 30, 5)
        roofline["measured_copy_GBs"] = bw
        form = ["grid-stride", "non-temporal loads and stores", "a contiguous stretch per block"][int(step2.lib().w2rap_step2_copy_bench_form())]
        roofline["measured_copy_kernel"] = f"k_copy16 (libw2rap_step2: uint4 per lane, four loads in flight; best of three forms -- here: {form} -- and of 8, 16, 32 blocks per CU), 4 GiB, 5 repetitions"
        roofline["frac_of_measured_copy"] = roofline["achieved"] / bw
        # MI355X_MICROARCH.md quotes 6.29 TB/s for a float4 copy; where this box's copy stays below it, the guide's figure is the denominator to trust
        roofline["guide_copy_GBs"] = GUIDE_COPY_GBS
        roofline["frac_of_guide_copy"] = roofline["achieved"] / GUIDE_COPY_GBS
    except Exception as e:                                      # never let the side measurement break the bench line
        roofline["measured_copy_GBs"] = None
        roofline["measured_copy_error"] = str(e)[:200]
    return roofline


def cpu_baseline(n_reads, genome_len, seed, dev, planted=False):
    """Reference Step 2 (oracle/_ref/ref_step2, the unmodified reference code) on the host cores, on a bounded sample of the same workload (8 M
    reads by default: eight 1 M-read leaves of its task tree, BuildReadQGraph.cc:1018,1266, so that the counting phase runs on eight threads and
    the serial merges and the serial dictionary fill show); falls back to our single-threaded port if the binary is absent.  The reference's
    OWN output files of that run (.small_K.hbv, .small_K.paths, small_K.freqs) are kept as bytes: the caller runs the GPU path on the same
    sample and compares (same_graph_as_gpu)."""
    from oracle import oracle as O
    d = planted_reads(n_reads, seed, dev) if planted else synth.generate_reads_device(n_reads, genome_len, seed, device=dev)
    codes = synth.unpack_fixed(d["packed"], synth.READ_LEN).cpu().numpy().reshape(-1)
    quals = d["quals"].cpu().numpy().reshape(-1)
    off = np.arange(d["n"] + 1, dtype=np.uint64) * synth.READ_LEN
    cores = os.cpu_count() or 1
    leaves = 1
    while (d["n"] + leaves - 1) // leaves > 1_000_000:        # createDictOMPRecursive halves until a part has <= 1 M reads
        leaves *= 2
    sample = f"{d['n']} synthetic PE150 reads, {genome_len} bp genome (same generator, 30x)"
    ref_files = None
    if os.path.exists(O.REF_BIN):
        with tempfile.TemporaryDirectory() as tmp:
            F.write_fastb(os.path.join(tmp, "frag_reads_orig.fastb"), *F.pack_bases(codes, off))
            F.write_qualp(os.path.join(tmp, "frag_reads_orig.qualp"), quals, off)
            secs = O.run_reference(tmp, "b", threads=cores)
            ref_files = {"hbv": open(os.path.join(tmp, "b.small_K.hbv"), "rb").read(), "paths": open(os.path.join(tmp, "b.small_K.paths"), "rb").read(),
                         "freqs": open(os.path.join(tmp, "small_K.freqs")).read(), "hbv_obj": F.read_hbv(os.path.join(tmp, "b.small_K.hbv"))}
        kind = "reference"
    else:
        t0 = time.perf_counter()
        O.run(codes, quals, off)
        secs = time.perf_counter() - t0
        cores, kind, leaves = 1, "port", 1
    return secs, cores, kind, sample, d, leaves, ref_files


def same_as_reference(ctx, ref_files, res=None, solid=None):
    """the GPU path on the reads already set in ctx, the reference's own edge numbering replayed (it is arbitrary: SURVEY.md 8c) -> the three
    output files must be the reference's, byte for byte; path differences are counted and split into parallel-edge extension ties (Q14) and others.
    res: a result fetched by another route (the in-process multi-rank call) to be judged the same way"""
    from oracle import oracle as O
    if res is None:
        hc, ho = O.edge_hint_from_hbv(ref_files["hbv_obj"])
        st = ctx.count_kmers(7, 4); ctx.build_graph(F.pack_bases(hc, ho)); ctx.path_reads()
        res = ctx.fetch()
        solid = int(st["S"])
    same_freqs = F.freqs_text(res.hist) == ref_files["freqs"]
    same_hbv = F.hbv_to_bytes(res.hbv) == ref_files["hbv"]
    same_paths = same_hbv and F.paths_to_bytes(res.path_offset, res.path_off, res.path_edges) == ref_files["paths"]
    ties = other = 0
    if same_hbv and not same_paths:
        # not byte-equal: tell extension ties between PARALLEL edges (same end vertices, equal score: quirk Q14, the reference's own 1- and
        # 8-thread runs differ by them) from real differences -- only the latter fail the verdict
        import io, struct
        from w2rap_contigger_amd import hbvtool
        buf = ref_files["paths"]
        (n,) = struct.unpack_from("<Q", buf, 0)
        p = 8
        left, right = hbvtool._left_right(res.hbv)
        po = res.path_off.astype(np.int64)
        for r in range(n):
            o, l = struct.unpack_from("<iH", buf, p); p += 6
            e = np.frombuffer(buf, dtype="<i4", count=l, offset=p); p += 4 * l
            g = res.path_edges[po[r]:po[r + 1]]
            if o == int(res.path_offset[r]) and l == len(g) and np.array_equal(e, g):
                continue
            if l == len(g) and all(u == v or (left[u] == left[v] and right[u] == right[v]) for u, v in zip(e, g)):
                ties += 1
            else:
                other += 1
    ok_paths = same_paths or (same_hbv and other == 0)
    return {"same_graph_as_gpu": bool(same_freqs and same_hbv and ok_paths), "freqs_bytes_equal": bool(same_freqs), "hbv_bytes_equal": bool(same_hbv),
            "paths_bytes_equal": bool(same_paths), "path_reads_differing_by_parallel_edge_ties": ties, "path_reads_differing_otherwise": other, "compared": "small_K.freqs, .small_K.hbv and .small_K.paths of the reference's run on this sample against the GPU path "
            "on the same reads with the reference's edge numbering replayed (edge_order_hint)", "edge_objects": int(res.hbv.n_edges), "kmers_solid": solid}


def sharded_as_reference(dp, ref_files, device, world=2, cut=27, virtual=8):
    """The path every multi-GPU run takes -- dictionary, prune and unipaths sharded by bucket owner, read pathing through the minimizer-sampled
    index + exact table -- on the reference's sample: `world` ranks of the one in-process call share this GPU, the two test hooks hand the
    cross-rank machinery the query and segment shares of an 8-rank job (results do not depend on them), the reference's edge numbering is
    replayed.  -> the verdict of same_as_reference for that route."""
    from oracle import oracle as O
    hc, ho = O.edge_hint_from_hbv(ref_files["hbv_obj"])
    hp = dp["packed"].cpu().numpy().reshape(-1); hq = dp["quals"].cpu().numpy().reshape(-1)
    hbo = dp["byte_off"].cpu().numpy().astype(np.uint64); hqo = dp["qual_off"].cpu().numpy().astype(np.uint64); hln = dp["read_len"].cpu().numpy().astype(np.uint32)
    keep = {k: os.environ.get(k) for k in ("W2RAP_TEST_SHARD_CUT", "W2RAP_TEST_SHARD_VIRTUAL")}
    os.environ["W2RAP_TEST_SHARD_CUT"] = str(cut); os.environ["W2RAP_TEST_SHARD_VIRTUAL"] = str(virtual)
    try:
        res = step2.build_read_qgraph(hp, hbo, hln, quals=hq, qual_off=hqo, devices=[device] * world, edge_order_hint=F.pack_bases(hc, ho))
    finally:
        for k, v in keep.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    out = same_as_reference(None, ref_files, res=res, solid=int(res.n_kmers_solid))
    out["route"] = (f"w2rap_step2_run, n_gpus = {world} on this one GPU: graph sharded by bucket owner, read pathing through the index + exact table; "
                    f"W2RAP_TEST_SHARD_CUT={cut}, W2RAP_TEST_SHARD_VIRTUAL={virtual}")
    return out


B_K2 = 2 * 12.0 + 0.25 + 4 + 4   # Step 3, algorithmic bytes per K2-mer occurrence: a (hash, position) record written once and read back once,
                                 # its 2-bit base, the id it learns, the successor link it writes (NOTES.md section 9)


def diploid_reads(n_reads, snp_every, seed, dev):
    """SURVEY 8d diploid variant: two haplotypes of n_reads*5/2 bases, one SNP per snp_every bases, reads 50/50"""
    rng = np.random.default_rng(seed)
    g = rng.integers(0, 4, n_reads * 5 // 2, dtype=np.uint8)
    h2 = g.copy()
    pos = rng.choice(np.arange(500, len(g) - 500), max(1, len(g) // snp_every), replace=False)
    h2[pos] = (h2[pos] + 1 + rng.integers(0, 3, len(pos))) & 3
    genome = torch.from_numpy(np.concatenate([g, h2])).to(dev)
    d = synth.generate_reads_device(n_reads, len(genome), seed, device=dev, genome=genome)
    d.pop("genome", None)
    return d


def planted_reads(n_reads, seed, dev):
    """SURVEY.md 8d "planted features" at bench scale: two haplotypes of n_reads*5/2 bases (one SNP per 2 kb on the second), and on both
    2000 copies of 40 repeat families (500 .. 5000 bp, a third of the copies inverted) -- exact repeats longer than k, inverted repeats
    and SNP bubbles, which a uniform random genome does not have.  Reads 50/50 from the haplotypes, 30x in total."""
    rng = np.random.default_rng(seed)
    G = n_reads * 5 // 2
    g = rng.integers(0, 4, G, dtype=np.uint8)
    fams = [rng.integers(0, 4, int(L), dtype=np.uint8) for L in rng.integers(500, 5001, 40)]
    for k in range(2000):
        f = fams[k % len(fams)]
        at = int(rng.integers(1000, G - 6000))
        g[at:at + len(f)] = (3 - f[::-1]) if k % 3 == 0 else f
    h2 = g.copy()
    pos = rng.choice(np.arange(500, G - 500), max(1, G // 2000), replace=False)
    h2[pos] = (h2[pos] + 1 + rng.integers(0, 3, len(pos))) & 3
    genome = torch.from_numpy(np.concatenate([g, h2])).to(dev)
    d = synth.generate_reads_device(n_reads, len(genome), seed, device=dev, genome=genome)
    d.pop("genome", None)
    return d


def main_step3(a, extra=False):
    """Step 3 (Involution, FragDist, RepathInMemory at K2) straight behind Step 2 on one GPU: a step = one whole Step 3 on the graph and
    the read paths that Step 2 left in HBM (w2rap_step3_run_after_step2, results kept on the device).  Prints ONE JSON line
    (extra: returns the result instead, no CPU baseline -- the N = 1 Step-2 line carries it under `extras`)."""
    a.reads = a.reads or 50e6
    from w2rap_contigger_amd import step3
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU (libw2rap_step2 has no CPU fallback)")
    dev = torch.device("cuda", 0)
    n_reads = int(a.reads)
    d = diploid_reads(n_reads, a.snp_every, 42, dev)
    torch.cuda.synchronize(dev); torch.cuda.empty_cache()
    ctx = step2.Step2Context(0)
    ctx.set_reads_device(d["n"], d["packed"].data_ptr(), d["byte_off"].data_ptr(), d["read_len"].data_ptr(), d["quals"].data_ptr(), d["qual_off"].data_ptr(), keepalive=d)
    ctx.count_kmers(7, 4); ctx.build_graph(None); ctx.path_reads()
    sizes2 = ctx.counts()
    for _ in range(a.warmup):
        step3.repath_after_step2(ctx, a.K2, fetch=False)
    torch.cuda.synchronize(dev)
    prof = {}
    t0 = time.perf_counter()
    for _ in range(a.steps):
        r3 = step3.repath_after_step2(ctx, a.K2, fetch=False)
        for k, v in step3.profile().items():
            o = prof.get(k, (0.0, 0)); prof[k] = (o[0] + v[0], o[1] + v[1])
    torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t0
    ms_per_step = elapsed / a.steps * 1e3
    N2 = r3.n_kmer_instances
    kname, (kms, klaunches) = max(prof.items(), key=lambda kv: kv[1][0])
    # the sort's launches differ in size (one over all K2-mer occurrences, small ones over edges and ends): price the entry as a whole per step
    per_step_ms = kms / a.steps
    achieved = N2 * (24.0 if "sort" in kname else B_K2) / (per_step_ms * 1e-3) / 1e9
    result = {
        "metric": f"step3_K{a.K2}_kmer_occurrences_per_s", "value": N2 / (ms_per_step * 1e-3), "unit": "K2-mers/s", "n_gpus": 1, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
        "config": {"workload": f"Step 3 (large-K repath, K2={a.K2}) behind Step 2 on {d['n']} synthetic PE150 reads, two haplotypes of {n_reads * 5 // 2} bp with one SNP per "
                               f"{a.snp_every} bp (SURVEY 8d diploid variant; BASELINE configs[3] scaled to one GPU)",
                   "reads_total": d["n"], "small_k_edge_objects": sizes2["edge_objects"], "places": r3.n_places, "unique_places": r3.n_unique_places,
                   "place_bases": r3.n_place_bases, "k2mer_occurrences": N2, "k2mers_distinct": r3.n_kmers_distinct, "unipaths": r3.n_unipaths,
                   "large_k_edge_objects": r3.hbv.n_edges if r3.hbv.n_edges else None},
        "phase_ms": {"places": r3.ms_places, "dictionary": r3.ms_dict, "graph": r3.ms_graph, "paths": r3.ms_paths},
        "device_ms_per_step": r3.ms_places + r3.ms_dict + r3.ms_graph + r3.ms_paths,
        "reads_repathed_per_s": d["n"] / (ms_per_step * 1e-3),
        "roofline": {"bound": "hbm", "kernel": kname, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                     "algorithmic_bytes_per_unit": 24.0 if "sort" in kname else B_K2, "unit_kind": "K2-mer occurrences", "units_per_step": N2,
                     "ms_per_step": per_step_ms, "launches_per_step": klaunches / a.steps,
                     "note": "rocPRIM radix sort of the (hash, position) pairs over 40 key bits: 5 passes x 24 B of traffic for 24 algorithmic bytes"
                             if "sort" in kname else None},
        "kernel_ms_per_step": {k: v[0] / a.steps for k, v in sorted(prof.items(), key=lambda kv: -kv[1][0])[:20]},
    }
    if extra:
        ctx.close(); del d; torch.cuda.empty_cache()
        return result
    if not a.no_cpu_baseline:
        # the REAL reference's Step 3 (oracle/_ref/ref_step3) on the host cores, on the Step-2 output of a bounded sample of the same workload
        from oracle import oracle3 as O3
        ctx.close(); del d; torch.cuda.empty_cache()
        n_cpu = int(a.cpu_reads)
        dc = diploid_reads(n_cpu, a.snp_every, 4242, dev)
        with step2.Step2Context(0) as c2:
            c2.set_reads_device(dc["n"], dc["packed"].data_ptr(), dc["byte_off"].data_ptr(), dc["read_len"].data_ptr(), dc["quals"].data_ptr(), dc["qual_off"].data_ptr(), keepalive=dc)
            c2.count_kmers(7, 4); c2.build_graph(None); c2.path_reads()
            s3 = step3.repath_after_step2(c2, a.K2, fetch=False)
            r2 = c2.fetch()
        cores = os.cpu_count() or 1
        if os.path.exists(O3.REF3_BIN):
            with tempfile.TemporaryDirectory() as tmp:
                F.write_hbv(os.path.join(tmp, "b.small_K.hbv"), r2.hbv)
                F.write_paths(os.path.join(tmp, "b.small_K.paths"), r2.path_offset, r2.path_off, r2.path_edges)
                tc = time.perf_counter()
                O3.run_reference3(tmp, "b", a.K2, cores)
                secs = time.perf_counter() - tc
            kind = "reference"
        else:
            tc = time.perf_counter()
            O3.run(r2.hbv, (r2.path_offset, r2.path_off, r2.path_edges), a.K2)
            secs = time.perf_counter() - tc
            cores, kind = 1, "port"
        result["cpu_baseline"] = {"value": s3.n_kmer_instances / secs, "unit": "K2-mers/s", "cores": cores, "kind": kind, "seconds": secs,
                                  "sample": f"Step-2 output of {dc['n']} reads of the same workload ({s3.n_kmer_instances} K2-mer occurrences); includes reading "
                                            f".small_K.hbv/.paths and writing .large_K.hbv/.paths", "reads_per_s": dc["n"] / secs}
    with_copy_rate(result["roofline"], dev)
    emit(json.dumps(result))


# Step 1, same recipe (profiles/r02_pmc.md): (FETCH_SIZE bytes, WRITE_SIZE bytes) per launch at the default 50 M reads
PMC_R02_STEP1 = {"k1_pq_write": (4.90e9, 18.12e9), "k1_unpack": (11.50e9, 9.60e9), "k1_list_nl": (1.18e9 / 2, 1.63e9 / 2), "k1_count_nl": (8.90e9 / 2, 2.26e9 / 2)}


def fastq_text_device(d, mate, dev, chunk=1 << 20):
    """the text of one fastq file of the pair for reads d (mates interleaved: this file holds reads mate, mate+2, ...), built in HBM:
    Illumina-style header with the record number, 150 bases, '+', 150 quality characters (q + 33).  -> (u8 tensor, bytes per record)"""
    n_rec = d["n"] // 2
    pre = b"@A00123:45:HXXXXXXXX:1:1101:"
    suf = (" %d:N:0:ATCACG\n" % (mate + 1)).encode()
    L = synth.READ_LEN
    H = len(pre) + 10 + len(suf)
    W = H + L + 3 + L + 1
    text = torch.empty((n_rec, W), dtype=torch.uint8, device=dev)
    text[:, :len(pre)] = torch.tensor(list(pre), dtype=torch.uint8, device=dev)
    text[:, len(pre) + 10:H] = torch.tensor(list(suf), dtype=torch.uint8, device=dev)
    text[:, H + L:H + L + 3] = torch.tensor(list(b"\n+\n"), dtype=torch.uint8, device=dev)
    text[:, W - 1] = 10
    acgt = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
    for a in range(0, n_rec, chunk):
        b = min(n_rec, a + chunk)
        idx = torch.arange(a, b, device=dev, dtype=torch.int64)
        for k in range(10):
            text[a:b, len(pre) + 9 - k] = ((idx // 10 ** k) % 10 + 48).to(torch.uint8)
        rows = idx * 2 + mate
        codes = synth.unpack_fixed(d["packed"][rows], L)
        text[a:b, H:H + L] = acgt[codes.long()]
        text[a:b, H + L + 3:W - 1] = d["quals"][rows] + 33
    return text.reshape(-1), W


def main_step1(a, extra=False):
    """Step 1 (paired fastq -> bases + PQVec qualities) on one GPU: a step = one whole ingest of the two texts, which lie in
    HBM when the timed region starts; results stay on the device.  Prints ONE JSON line (extra: returns the result instead)."""
    a.reads = a.reads or 50e6
    from w2rap_contigger_amd import step1
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU (libw2rap_step2 has no CPU fallback)")
    # N > 1 (torch.distributed.run, one rank per GPU): the records shard by rank with no exchange at all -- every rank ingests its own
    # pair of texts (weak scaling); the process group only carries the barrier and the max over ranks of the timed region
    world, rank, local_rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))
    if extra:
        world, rank = 1, 0
    # W2RAP_BENCH_SHARE_GPU=1 (tests on a 1-GPU box): every rank drives cuda:0 with its own library context and the ranks talk through gloo
    # (host-staged exchange, dist._host_staged) -- the whole N > 1 flow of this file except RCCL itself
    share_gpu = os.environ.get("W2RAP_BENCH_SHARE_GPU") == "1"
    if share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0"); os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
        if share_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    n_reads = int(a.reads) & ~1
    genome_len = int(a.genome) if a.genome else n_reads * 5
    d = synth.generate_reads_device(n_reads, genome_len, 42 + 7919 * rank, device=dev)
    d.pop("genome", None)
    t1, W = fastq_text_device(d, 0, dev)
    t2, _ = fastq_text_device(d, 1, dev)
    n = d["n"]
    del d
    torch.cuda.synchronize(dev); torch.cuda.empty_cache()
    ctx = step2.Step2Context(local_rank)
    args = ((t1.data_ptr(), t1.numel()), (t2.data_ptr(), t2.numel()))
    for _ in range(a.warmup):
        step1.extract_reads(*args, device=local_rank, flags=step1.NO_FETCH, ctx=ctx)
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    prof = {}
    t0 = time.perf_counter()
    for _ in range(a.steps):
        r = step1.extract_reads(*args, device=local_rank, flags=step1.NO_FETCH, ctx=ctx)
        for k, v in step1.profile().items():
            o = prof.get(k, (0.0, 0)); prof[k] = (o[0] + v[0], o[1] + v[1])
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
        dist.destroy_process_group()
        if rank:
            return
    ms_per_step = elapsed / a.steps * 1e3
    text_bytes = t1.numel() + t2.numel()
    lines_bytes = 2.0 * r.n_bases + 2 * n                   # the sequence and quality lines with their newlines
    # algorithmic bytes per step of each kernel (NOTES.md section 10): what it must read and write once
    alg = {"k1_count_nl": text_bytes + text_bytes / 8 + text_bytes / 16384 * 4, "k1_list_nl": text_bytes / 8 + text_bytes / 16384 * 8 + 4 * n * 8,
           "k1_unpack": lines_bytes + 4 * 8 * n + 16 * n + r.n_packed_bytes + r.n_bases + 4 * n,
           "k1_pq_write": r.n_bases + 16 * n + r.n_pq_bytes}
    step_alg = text_bytes + r.n_packed_bytes + r.n_bases + r.n_pq_bytes + 28 * n
    kname, (kms, klaunches) = max(((k, v) for k, v in prof.items() if k in alg), key=lambda kv: kv[1][0])
    per_launch_ms = kms / klaunches
    launches_per_step = klaunches / a.steps
    achieved = alg[kname] / launches_per_step / (per_launch_ms * 1e-3) / 1e9
    result = {
        "metric": "step1_fastq_reads_per_s", "value": world * n / (ms_per_step * 1e-3), "unit": "reads/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
        "config": {"workload": f"Step 1 (paired fastq ingest) of {n} synthetic PE150 reads per GPU: two fastq texts of {t1.numel()} B ({W} B per record), "
                               f"SURVEY 8d base/quality distributions (BASELINE configs[1] read set as fastq)",
                   "reads_total": world * n, "parallelism": f"records sharded x{world}, no exchange", "fastq_bytes": text_bytes, "packed_base_bytes": r.n_packed_bytes, "quality_bytes": r.n_bases, "pqvec_bytes": r.n_pq_bytes},
        "phase_ms": {"line_index": r.ms_index, "encode": r.ms_encode},
        "fastq_GB_per_s": text_bytes / (ms_per_step * 1e-3) / 1e9,
        "step_algorithmic_GB_per_s": step_alg / (ms_per_step * 1e-3) / 1e9,
        "roofline": {"bound": "hbm", "kernel": kname, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                     "traffic": (2 * PMC_R02_STEP1[kname][0] + PMC_R02_STEP1[kname][1]) if (n == 50_000_000 and kname in PMC_R02_STEP1) else None,
                     "algorithmic_bytes_per_unit": alg[kname] / n, "unit_kind": "reads", "units_per_launch": n / launches_per_step,
                     "ms_per_launch": per_launch_ms, "launches_per_step": launches_per_step},
        "kernel_ms_per_step": {k: v[0] / a.steps for k, v in sorted(prof.items(), key=lambda kv: -kv[1][0])[:12]},
    }
    if extra:
        ctx.close(); del t1, t2; torch.cuda.empty_cache()
        return result
    if not a.no_cpu_baseline and world == 1:
        # the REAL reference's Step 1 (oracle/_ref/ref_step1: ExtractReads + WriteAll) on the host cores, on the first records of the same texts
        from oracle import oracle1 as O1
        n_cpu = min(int(a.cpu_reads) & ~1, n)
        nb = n_cpu // 2 * W
        h1, h2 = t1[:nb].cpu().numpy().tobytes(), t2[:nb].cpu().numpy().tobytes()
        cores = os.cpu_count() or 1
        if os.path.exists(O1.REF1_BIN):
            with tempfile.TemporaryDirectory() as tmp:
                open(os.path.join(tmp, "r1.fastq"), "wb").write(h1); open(os.path.join(tmp, "r2.fastq"), "wb").write(h2)
                secs = O1.run_reference1(tmp, os.path.join(tmp, "r1.fastq") + "," + os.path.join(tmp, "r2.fastq"), cores)
            kind = "reference"
        else:
            tc = time.perf_counter()
            O1.run(h1, h2)
            secs = time.perf_counter() - tc
            cores, kind = 1, "port"
        result["cpu_baseline"] = {"value": n_cpu / secs, "unit": "reads/s", "cores": cores, "kind": kind, "seconds": secs,
                                  "sample": f"the first {n_cpu} reads of the same two fastq texts, read from files in a temporary directory (time inside ExtractReads)"}
    with_copy_rate(result["roofline"], dev)
    emit(json.dumps(result))


def main_pipeline(a, extra=False):
    """Steps 1 -> 2 -> 3 in one process without leaving the GPU: a step = two fastq texts in HBM -> reads (Step 1, raw qualities, no PQVec) ->
    small-K graph + paths (Step 2) -> large-K graph + paths (Step 3), each stage taking its input where the previous one left it.
    Diploid workload of --step3.  Prints ONE JSON line."""
    a.reads = a.reads or 50e6
    from w2rap_contigger_amd import step1, step3
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU (libw2rap_step2 has no CPU fallback)")
    dev = torch.device("cuda", 0)
    n_reads = int(a.reads) & ~1
    d = diploid_reads(n_reads, a.snp_every, 42, dev)
    t1, W = fastq_text_device(d, 0, dev)
    t2, _ = fastq_text_device(d, 1, dev)
    n = d["n"]
    del d
    torch.cuda.synchronize(dev); torch.cuda.empty_cache()
    ctx = step2.Step2Context(0)
    args = ((t1.data_ptr(), t1.numel()), (t2.data_ptr(), t2.numel()))

    def one():
        ts = [time.perf_counter()]
        step1.extract_reads(*args, flags=step1.NO_PQ | step1.NO_FETCH, ctx=ctx); torch.cuda.synchronize(dev); ts.append(time.perf_counter())
        st = ctx.count_kmers(7, 4); ctx.build_graph(None); ctx.path_reads(); torch.cuda.synchronize(dev); ts.append(time.perf_counter())
        r3 = step3.repath_after_step2(ctx, a.K2, fetch=False); torch.cuda.synchronize(dev); ts.append(time.perf_counter())
        return st, r3, [(y - x) * 1e3 for x, y in zip(ts, ts[1:])]
    for _ in range(a.warmup):
        one()
    torch.cuda.synchronize(dev)
    stage = [0.0, 0.0, 0.0]
    t0 = time.perf_counter()
    for _ in range(a.steps):
        st, r3, ms = one()
        stage = [x + y for x, y in zip(stage, ms)]
    torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t0
    ms_per_step = elapsed / a.steps * 1e3
    sizes2 = ctx.counts()
    result = {
        "metric": "steps123_reads_per_s", "value": n / (ms_per_step * 1e-3), "unit": "reads/s", "n_gpus": 1, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
        "config": {"workload": f"fastq text -> Step 1 -> Step 2 (k=60) -> Step 3 (K2={a.K2}) on {n} synthetic PE150 reads, two haplotypes of {n_reads * 5 // 2} bp with one "
                               f"SNP per {a.snp_every} bp; everything stays in HBM between the steps", "reads_total": n, "fastq_bytes": t1.numel() + t2.numel(),
                   "kmer_instances": st["M"], "kmers_solid": st["S"], "small_k_edge_objects": sizes2["edge_objects"], "k2mer_occurrences": r3.n_kmer_instances,
                   "large_k_unipaths": r3.n_unipaths},
        "stage_ms": {"step1": stage[0] / a.steps, "step2": stage[1] / a.steps, "step3": stage[2] / a.steps},
        "kmers_per_s_whole_pipeline": st["M"] / (ms_per_step * 1e-3),
    }
    if extra:
        ctx.close(); del t1, t2; torch.cuda.empty_cache()
        return result
    if not a.no_cpu_baseline:
        # the REAL reference's Steps 1, 2 and 3 (oracle/_ref) on the host cores, on the first records of the same texts, through its files
        from oracle import oracle as O, oracle1 as O1, oracle3 as O3
        n_cpu = min(int(a.cpu_reads) & ~1, n)
        if all(os.path.exists(b) for b in (O1.REF1_BIN, O.REF_BIN, O3.REF3_BIN)):
            cores = os.cpu_count() or 1
            ctx.close(); del t1, t2; torch.cuda.empty_cache()
            dc = diploid_reads(n_cpu, a.snp_every, 4242, dev)                  # the same workload at the sample's size (30x coverage of ITS genome)
            c1, _ = fastq_text_device(dc, 0, dev); c2, _ = fastq_text_device(dc, 1, dev)
            with tempfile.TemporaryDirectory() as tmp:
                open(os.path.join(tmp, "r1.fastq"), "wb").write(c1.cpu().numpy().tobytes()); open(os.path.join(tmp, "r2.fastq"), "wb").write(c2.cpu().numpy().tobytes())
                tc = time.perf_counter()
                O1.run_reference1(tmp, os.path.join(tmp, "r1.fastq") + "," + os.path.join(tmp, "r2.fastq"), cores); s1 = time.perf_counter()
                O.run_reference(tmp, "b", threads=cores); s2 = time.perf_counter()
                O3.run_reference3(tmp, "b", a.K2, cores); s3 = time.perf_counter()
            secs = s3 - tc
            result["cpu_baseline"] = {"value": n_cpu / secs, "unit": "reads/s", "cores": cores, "kind": "reference", "seconds": secs,
                                      "stage_seconds": {"step1": s1 - tc, "step2": s2 - s1, "step3": s3 - s2},
                                      "sample": f"{n_cpu} reads of the same workload (two haplotypes of {n_cpu * 5 // 2} bp) as fastq files; the three reference steps as "
                                                f"separate runs through their files"}
    emit(json.dumps(result))


def main_gfa(a, extra=False):
    """GFA dump (hbv2gfa without line finding) of the Step-2 graph of the bench workload: a step = one w2rap_gfa_dump (involution,
    canonical forms, statistics, all S and L lines built in HBM, text not fetched).  The graph comes over PCIe (62 MB of packed bases) inside
    the step; the roofline entry is the segment writer's own device time.  Prints ONE JSON line."""
    a.reads = a.reads or 50e6
    from w2rap_contigger_amd import gfa
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU (libw2rap_step2 has no CPU fallback)")
    dev = torch.device("cuda", 0)
    n_reads = int(a.reads)
    d = synth.generate_reads_device(n_reads, int(a.genome) if a.genome else n_reads * 5, 42, device=dev)
    d.pop("genome", None)
    with step2.Step2Context(0) as ctx:
        ctx.set_reads_device(d["n"], d["packed"].data_ptr(), d["byte_off"].data_ptr(), d["read_len"].data_ptr(), d["quals"].data_ptr(), d["qual_off"].data_ptr(), keepalive=d)
        ctx.count_kmers(7, 4); ctx.build_graph(None); ctx.path_reads()
        res = ctx.fetch()
    del d; torch.cuda.empty_cache()
    h = res.hbv
    for _ in range(a.warmup):
        gfa.gfa_dump(h, flags=gfa.NO_FETCH)
    prof = {}
    t0 = time.perf_counter()
    for _ in range(a.steps):
        r = gfa.gfa_dump(h, flags=gfa.NO_FETCH)
        for k, v in gfa.profile().items():
            o = prof.get(k, (0.0, 0)); prof[k] = (o[0] + v[0], o[1] + v[1])
    elapsed = time.perf_counter() - t0
    ms_per_step = elapsed / a.steps * 1e3
    seg_ms = prof["kg_seg_write"][0] / prof["kg_seg_write"][1]
    seg_bytes = r.canonical_size * 0.25 + r.segment_bytes           # bases read at 2 bits, the S lines written once
    achieved = seg_bytes / (seg_ms * 1e-3) / 1e9
    result = {
        "metric": "gfa_dump_bases_per_s", "value": r.canonical_size / (ms_per_step * 1e-3), "unit": "bases/s", "n_gpus": 1, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
        "config": {"workload": f"GFA dump of the small-K graph of {n_reads} synthetic PE150 reads (BASELINE configs[1]): {h.n_edges} edge objects, {r.n_segments} segments, "
                               f"{r.n_links} links, {r.gfa_len} bytes of text", "edge_objects": h.n_edges, "gfa_bytes": r.gfa_len, "canonical_bases": r.canonical_size},
        "device_ms": {"involution": r.ms_involution, "dump": r.ms_dump},
        "roofline": {"bound": "hbm", "kernel": "kg_seg_write", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                     "algorithmic_bytes_per_unit": seg_bytes / max(1, r.canonical_size), "unit_kind": "bases", "units_per_launch": r.canonical_size, "ms_per_launch": seg_ms},
        "kernel_ms_per_step": {k: v[0] / a.steps for k, v in sorted(prof.items(), key=lambda kv: -kv[1][0])[:12]},
    }
    if extra:
        return result
    if not a.no_cpu_baseline:
        # the REAL reference's hbv2gfa (oracle/_ref/ref_hbv2gfa) on the same graph, files in a temporary directory
        from oracle import oracle_gfa as OG
        if os.path.exists(OG.REF_GFA_BIN):
            with tempfile.TemporaryDirectory() as tmp:
                F.write_hbv(os.path.join(tmp, "g.hbv"), h)
                F.write_paths(os.path.join(tmp, "g.paths"), res.path_offset[:2], res.path_off[:3], res.path_edges[:int(res.path_off[2])])
                tc = time.perf_counter()
                _, ref_text = OG.run_reference_gfa(tmp, "g", "o")
                secs = time.perf_counter() - tc
            same = gfa.gfa_dump(h).gfa == ref_text
            result["cpu_baseline"] = {"value": r.canonical_size / secs, "unit": "bases/s", "cores": 1, "kind": "reference", "seconds": secs,
                                      "sample": "the same graph, whole tool run (reads .hbv, writes _raw.gfa)", "same_text_as_gpu": bool(same)}
    with_copy_rate(result["roofline"], dev)
    emit(json.dumps(result))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--reads", type=float, default=0, help="reads per GPU (default: 50e6 on one GPU = configs[1]; 62.5e6 = 500e6 / 8 on several = configs[2])")
    ap.add_argument("--genome", type=float, default=0, help="genome length (default: 30x coverage of all the reads of the job)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-reads", type=float, default=8e6, help="reads of the CPU-baseline sample (8 M = eight 1 M-read leaves of the reference's task tree)")
    ap.add_argument("--no-planted-parity", action="store_true", help="skip the real-reference comparison on the planted generator")
    ap.add_argument("--planted-cpu-reads", type=float, default=4e6, help="reads of the planted-generator sample the reference is run on")
    ap.add_argument("--no-extras", action="store_true", help="skip the untimed extras of the N=1 line: host-resident one-shot call, planted workload")
    ap.add_argument("--step3", action="store_true", help="measure Step 3 (large-K repath, SURVEY 8f N1) behind Step 2 instead: its own JSON line")
    ap.add_argument("--step1", action="store_true", help="measure Step 1 (paired fastq ingest, SURVEY 8f N3) instead: its own JSON line")
    ap.add_argument("--pipeline", action="store_true", help="measure Steps 1 -> 2 -> 3 chained on the GPU (fastq text in HBM -> large-K graph): its own JSON line")
    ap.add_argument("--gfa", action="store_true", help="measure the GFA dump of the Step-2 graph (SURVEY 8f N4) instead: its own JSON line")
    ap.add_argument("--K2", type=int, default=200)
    ap.add_argument("--snp-every", type=int, default=2000, help="--step3: second haplotype with one SNP per this many bases (SURVEY 8d diploid variant)")
    a = ap.parse_args()
    guard_stdout()
    if a.step3:
        return main_step3(a)
    if a.step1:
        return main_step1(a)
    if a.gfa:
        return main_gfa(a)
    if a.pipeline:
        return main_pipeline(a)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    parity_failed = False
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
        a.gpus = world
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU (libw2rap_step2 has no CPU fallback)")
    share_gpu = os.environ.get("W2RAP_BENCH_SHARE_GPU") == "1"      # (tests on a 1-GPU box: see main_step1)
    if share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # W2RAP_FORCE_DIST=1 drives the multi-GPU code path (process group, all_to_all_v, all_gather_v) even
    # with one rank, so that it can be exercised on a 1-GPU box
    use_dist = world > 1 or os.environ.get("W2RAP_FORCE_DIST") == "1"
    if use_dist:
        import torch.distributed as dist
        from w2rap_contigger_amd import dist as wd
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if share_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    # N = 1: BASELINE configs[1] (50 M reads, 250 Mbp).  N > 1: configs[2] scaled to N GPUs -- 62.5 M reads per GPU (500 M / 8) of one genome of
    # N x 312.5 Mbp, i.e. at N = 8 exactly configs[2] (500 M reads, 2.5 Gbp): the reads are sharded, the genome (and with it the
    # replicated dictionary and graph: S ~ genome length) is the job's, not the rank's.
    n_reads = int(a.reads) if a.reads else (50_000_000 if world == 1 else 62_500_000)
    genome_len = int(a.genome) if a.genome else n_reads * 5 * world
    if world == 1:
        workload = f"{n_reads} synthetic PE150 reads, {genome_len} bp genome, k=60 Step-2 graph + read pathing (BASELINE configs[1]); min_qual 7, min_freq 4"
    else:
        workload = (f"{n_reads * world} synthetic PE150 reads sharded over {world} GPUs ({n_reads} per GPU), ONE {genome_len} bp genome, k=60 Step-2 graph + read pathing "
                    f"(BASELINE configs[2]{'' if (world == 8 and n_reads == 62_500_000) else f' scaled to {world} GPUs: 500 M reads / 2.5 Gbp at 8'}); min_qual 7, min_freq 4")
    # the same genome on every rank (generated in pieces: randint's int64 scratch is 8 B per element); rank-specific reads
    gen = torch.Generator(device=dev).manual_seed(42)
    genome = torch.empty(genome_len, dtype=torch.uint8, device=dev)
    for g0 in range(0, genome_len, 1 << 30):
        genome[g0:g0 + (1 << 30)] = torch.randint(0, 4, (min(1 << 30, genome_len - g0),), dtype=torch.uint8, device=dev, generator=gen)
    d = synth.generate_reads_device(n_reads, genome_len, 42 + 7919 * rank, device=dev, genome=genome)
    del genome
    d.pop("genome", None)
    torch.cuda.synchronize(dev)
    torch.cuda.empty_cache()                 # hand the generator's scratch back before the library allocates
    ctx = step2.Step2Context(local_rank)
    ctx.set_reads_device(d["n"], d["packed"].data_ptr(), d["byte_off"].data_ptr(), d["read_len"].data_ptr(),
                         d["quals"].data_ptr(), d["qual_off"].data_ptr(), keepalive=d)
    backend = wd.GpuBackend(ctx, dev) if use_dist else None

    def barrier():
        torch.cuda.synchronize(dev)
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize(dev)

    # several GPUs: dictionary, prune and unipaths stay SHARDED by bucket owner (row e-3; W2RAP_REPLICATED_GRAPH=1: gathered and rebuilt
    # on every rank as in rounds 1-4)
    sharded = use_dist and not _env_flag("W2RAP_REPLICATED_GRAPH")

    xinfo = {}                                   # the last step's sharded-graph exchanges (dist.sharded_graph)

    def one_step():
        t0 = time.perf_counter()
        if sharded:
            st = wd.distributed_count(backend, 7, 4, gather=False)
            t1 = time.perf_counter()
            xinfo.clear(); xinfo.update(wd.sharded_graph(backend, st["S_local"], st, st["n_buckets"]))
            t2 = time.perf_counter()
        else:
            if use_dist:
                st = wd.distributed_count(backend, 7, 4)
            else:
                st = ctx.count_kmers(7, 4)
            t1 = time.perf_counter()
            ctx.build_graph(None)
            t2 = time.perf_counter()
        ctx.path_reads()
        torch.cuda.synchronize(dev)
        t3 = time.perf_counter()
        return st, (t1 - t0, t2 - t1, t3 - t2)

    for _ in range(a.warmup):
        one_step()
    ctx.profile(reset=True)
    barrier()
    t_begin = time.perf_counter()
    phases = np.zeros(3)
    st = None
    for _ in range(a.steps):
        st, ph = one_step()
        phases += ph
    barrier()
    elapsed = time.perf_counter() - t_begin
    prof = ctx.profile(reset=True)
    sizes = ctx.counts()                      # of the last step: graph and paths as held on this rank
    if use_dist and world > 1:                # every rank must hold the same graph
        # (sharded graph: a context holds the solid k-mers of ITS buckets only; the job's count is st["S"])
        g = torch.tensor([int(st["S"]) if sharded else sizes["kmers_solid"], sizes["unipaths"], sizes["edge_objects"], sizes["vertices"]], dtype=torch.int64, device=dev)
        lo, hi = g.clone(), g.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN); dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        if not torch.equal(lo, hi):
            sys.exit(f"rank {rank}: the graphs differ between ranks: min {lo.tolist()} max {hi.tolist()}")
        p = torch.tensor([sizes["reads_pathed"], sizes["path_elements"]], dtype=torch.int64, device=dev)
        dist.all_reduce(p)
        sizes["reads_pathed"], sizes["path_elements"] = int(p[0].item()), int(p[1].item())
    # ---- the first run on several GPUs validates itself: who took part, do all ranks hold the same graph, is it the
    #      graph ONE rank builds from the same reads (a 2 M-read subsample: every rank's first reads, gathered on rank 0 and run there alone)
    selfcheck = None
    if use_dist:
        import hashlib
        res_g = ctx.fetch()
        hb = F.hbv_to_bytes(res_g.hbv)
        digest = torch.tensor(list(hashlib.sha256(hb).digest()[:8]), dtype=torch.int64, device=dev)
        if share_gpu:
            dcpu = digest.cpu(); parts = [torch.zeros_like(dcpu) for _ in range(world)]
            dist.all_gather(parts, dcpu)
            alld = [tuple(x.tolist()) for x in parts]
        else:
            out = torch.empty(world * 8, dtype=torch.int64, device=dev)
            dist.all_gather_into_tensor(out, digest)
            alld = [tuple(out[8 * r:8 * r + 8].tolist()) for r in range(world)]
        try:
            uuid = str(torch.cuda.get_device_properties(dev).uuid)
        except Exception:
            uuid = f"cuda:{local_rank}"
        uu = [None] * world
        dist.all_gather_object(uu, uuid)
        selfcheck = {"rccl_ranks": dist.get_world_size(), "backend": dist.get_backend(), "device_uuids": uu,
                     "graph_equal_across_ranks": all(x == alld[0] for x in alld), "graph_sha256_16": hashlib.sha256(hb).hexdigest()[:16]}
        del res_g
        # subsample: the first n_sub / world reads of every rank; distributed (sharded) on all ranks, then alone on rank 0
        n_sub = (min(2_000_000, d["n"] * world) // world // 2) * 2
        sub = step2.Step2Context(local_rank)
        nb_sub = int(d["byte_off"][n_sub].item()) if n_sub < d["n"] else int(d["packed"].numel())
        sub.set_reads_device(n_sub, d["packed"].data_ptr(), d["byte_off"].data_ptr(), d["read_len"].data_ptr(), d["quals"].data_ptr(), d["qual_off"].data_ptr(), keepalive=d)
        bsub = wd.GpuBackend(sub, dev)
        if sharded:
            sst = wd.distributed_count(bsub, 7, 4, gather=False)
            wd.sharded_graph(bsub, sst["S_local"], sst, sst["n_buckets"])
        else:
            sst = wd.distributed_count(bsub, 7, 4); sub.build_graph(None)
        sub.path_reads()
        rs = sub.fetch()
        sub.close()
        pk_all = [None] * world
        rl = synth.READ_LEN
        mine = (d["packed"].view(-1)[:nb_sub].cpu().numpy().copy(), d["quals"].view(-1)[:n_sub * rl].cpu().numpy().copy())
        dist.gather_object(mine, pk_all if rank == 0 else None, dst=0)
        paths_mine = (rs.path_offset.copy(), np.diff(rs.path_off.astype(np.int64)), rs.path_edges.copy())
        pa_all = [None] * world
        dist.gather_object(paths_mine, pa_all if rank == 0 else None, dst=0)
        if rank == 0:
            pk = np.concatenate([x[0] for x in pk_all]); qq = np.concatenate([x[1] for x in pk_all])
            nn = n_sub * world
            bo = (np.arange(nn + 1, dtype=np.uint64) * np.uint64((rl + 3) // 4)); qo = np.arange(nn + 1, dtype=np.uint64) * np.uint64(rl)
            with step2.Step2Context(local_rank) as one:
                one.set_reads_host(pk, bo, np.full(nn, rl, np.uint32), quals=qq, qual_off=qo)
                s1 = one.count_kmers(7, 4); one.build_graph(None); one.path_reads()
                r1 = one.fetch()
            same_stats = (int(s1["M"]), int(s1["D"]), int(s1["S"])) == (int(sst["M"]), int(sst["D"]), int(sst["S"])) and np.array_equal(np.asarray(s1["hist"]), np.asarray(sst["hist"]))
            same_graph = F.hbv_to_bytes(r1.hbv) == F.hbv_to_bytes(rs.hbv)
            same_paths = (np.array_equal(np.concatenate([x[0] for x in pa_all]), r1.path_offset) and
                          np.array_equal(np.concatenate([x[1] for x in pa_all]), np.diff(r1.path_off.astype(np.int64))) and
                          np.array_equal(np.concatenate([x[2] for x in pa_all]), r1.path_edges))
            selfcheck.update({"stats_equal_single_rank": bool(same_stats), "graph_equal_single_rank": bool(same_graph), "paths_equal_single_rank": bool(same_paths),
                              "single_rank_sample": f"{nn} reads (the first {n_sub} of every rank)"})
            try:
                from w2rap_contigger_amd import scale_model
                selfcheck["model_ms_per_step"] = scale_model.predict(world)["ms_per_step"]
            except Exception as ex:
                selfcheck["model_ms_per_step"] = None; selfcheck["model_error"] = str(ex)[:200]
    m_total = int(st["M"])
    if use_dist:
        t = torch.tensor([elapsed] + list(phases), dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t[0].item())
        phases = t[1:].cpu().numpy()
    ms_per_step = elapsed / a.steps * 1e3
    phases = phases / a.steps
    result = None
    if rank == 0:
        # dominant kernel (hipEvents on the library's stream, summed over the timed steps)
        kname, (kms, klaunches) = max(prof.items(), key=lambda kv: kv[1][0])
        avg_ms = kms / klaunches
        per_step_launches = klaunches / a.steps
        # units one launch processes: the counting kernels see this rank's share of the k-mers
        if kname.startswith("k_path"):
            units, per_unit, what = d["n"] / per_step_launches, B_R, "reads"
        else:
            units, per_unit, what = (m_total / world) / max(per_step_launches, 1), B_K, "k-mers"
        achieved = units * per_unit / (avg_ms * 1e-3) / 1e9
        # what the committed PMC profile of this command says about the kernel: HBM bytes it really moves, its VALU share, what limits it
        traffic = hbm_actual = valu_frac = valu_wave = prof_tie = None
        limiter = "unprofiled"
        prof_json = pmc_profile()
        pk = pmc_of(prof_json, kname) if (world == 1 and d["n"] == 50_000_000) else None
        if pk:
            step_traffic = 2 * pk["fetch_bytes"] + pk["write_bytes"]
            traffic = step_traffic / max(per_step_launches, 1)
            hbm_actual = traffic / (avg_ms * 1e-3) / 1e9                                 # GB/s the kernel really draws from HBM
            valu_frac = pk["insts_valu"] / max(per_step_launches, 1) * 2.0 / (SIMDS * CLOCK_HZ * avg_ms * 1e-3)    # wave-VALU x 2 clocks (SIMD-32) / SIMD-clocks
            valu_wave = pk.get("valu_active_frac")                                       # SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES: per WAVE; x resident waves per SIMD = how busy the VALU is
            limiter = "hbm" if hbm_actual > 0.5 * HBM_PEAK_GBS else ("valu-issue" if valu_frac > 0.25 else "latency")
            # the tie between that profile and THIS run: the same kernel, the same number of launches per step, and its time
            # there (kernels serialised under the counter passes) against its time here -- a profile of another build or workload shows up as a
            # mismatch in the line itself
            prof_tie = {"profile_launches_per_step": pk.get("launches"), "profile_ms_per_step": pk.get("ms"), "run_ms_per_step": avg_ms * per_step_launches,
                        "same_launch_count": pk.get("launches") == per_step_launches,
                        "ms_ratio_run_over_profile": (avg_ms * per_step_launches / pk["ms"]) if pk.get("ms") else None}
        # In the single-GPU step the counting kernel shares the GPU with the dictionary build (k_table_insert runs on a
        # side stream while the next bucket slice is counted): its launches are longer than on their own.  One extra,
        # untimed step without that overlap gives the kernel's own duration next to the live one.
        alone = None
        # (skipped under rocprofv3, whose per-kernel averages should be those of the timed launches)
        profiled = "rocprof" in os.environ.get("LD_PRELOAD", "") or any(k.startswith("ROCPROF") for k in os.environ)
        if not use_dist and kname.startswith("k_count") and not profiled:
            os.environ["W2RAP_NO_OVERLAP"] = "1"
            try:
                ctx.profile(reset=True)
                one_step()
                pa = ctx.profile(reset=True)
            finally:
                os.environ.pop("W2RAP_NO_OVERLAP", None)
            if kname in pa and pa[kname][1]:
                ms_alone = pa[kname][0] / pa[kname][1]
                ach = (m_total / world) * per_unit / (ms_alone * 1e-3) / 1e9
                alone = {"avg_launch_ms": ms_alone, "achieved": ach, "frac": ach / HBM_PEAK_GBS,
                         "note": "same kernel, one launch over all buckets, nothing else on the GPU"}
        result = {
            "metric": "step2_k60_canonical_kmers_per_s", "value": m_total / (ms_per_step * 1e-3), "unit": "k-mers/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "config": {"workload": workload,
                       "reads_total": d["n"] * world, "reads_per_gpu": d["n"], "genome_bp": genome_len, "kmer_instances": m_total, "kmers_distinct": int(st["D"]),
                       "kmers_solid": int(st["S"]), "unipaths": sizes["unipaths"], "edge_objects": sizes["edge_objects"],
                       "vertices": sizes["vertices"], "reads_pathed": sizes["reads_pathed"], "path_elements": sizes["path_elements"],
                       "parallelism": ("one GPU (no shuffle; dictionary pathing)" if not use_dist else
                                       f"reads sharded x{world}, k-mer shuffle all_to_all_v, " + ("graph sharded (dictionary, prune, unipaths by bucket owner; the E-sized rest on every rank)"
                                                                                                 if sharded else "graph replicated"))},
            "phase_ms": {"count": phases[0] * 1e3, "graph": phases[1] * 1e3, "path": phases[2] * 1e3},
            "kmers_per_s_count_phase": m_total / phases[0],
            "reads_pathed_per_s": d["n"] * world / phases[2],
            # `achieved` is ALGORITHMIC bytes over time (the contract's pricing, against the HBM roof); `bound` is what the PMC profile says limits the
            # kernel -- it keeps its tables in LDS and moves a small fraction of those bytes: `hbm_actual_GBs`, `valu_frac`
            "roofline": {"bound": limiter if limiter != "unprofiled" else "hbm", "priced_against": "hbm", "kernel": kname, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "hbm_actual_GBs": hbm_actual, "hbm_actual_frac": (hbm_actual / HBM_PEAK_GBS) if hbm_actual else None,
                         "valu_frac": valu_frac, "valu_active_frac_per_wave": valu_wave, "valu_frac_note": "wave-VALU instructions of the profile x 2 clocks (SIMD-32) / (1024 SIMDs x 2.4 GHz x launch time)",
                         "traffic_from_profile": os.path.relpath(PMC_JSON, ROOT) if traffic is not None else None,     # a constant of the committed profile of this command, not a counter of this run
                         "traffic_profile_tie": prof_tie,
                         # BASELINE.md section 3's own formula for the WHOLE counting phase (K0-K5, wall clock): (M x 41 B) / t_count / peak, per GPU
                         "count_phase_frac": (m_total / world) * B_K / phases[0] / 1e9 / HBM_PEAK_GBS,
                         "path_phase_frac": d["n"] * B_R / phases[2] / 1e9 / HBM_PEAK_GBS,
                         "algorithmic_bytes_per_unit": per_unit, "unit_kind": what, "units_per_launch": units,
                         "avg_launch_ms": avg_ms, "launches_per_step": per_step_launches,
                         "overlapped_with": "k_table_insert (side stream)" if (not use_dist and per_step_launches > 1) else None,
                         "not_overlapped": alone},
            "kernel_ms_per_step": {k: v[0] / a.steps for k, v in sorted(prof.items(), key=lambda kv: -kv[1][0])[:60]},
        }
        if xinfo:
            # per-exchange wall times of the LAST timed step on rank 0 (what W2RAP_TRACE_SHARD prints): one multi-GPU run calibrates the
            # link efficiency scale_model.py assumes (LINK_EFF); `exchanges` says what each one moved
            result["exchange_ms"] = xinfo.get("exchange_ms")
            result["exchanges"] = xinfo.get("exchange_log")
        if selfcheck:
            result.update(selfcheck)                      # top-level keys: rccl_ranks, device_uuids, graph_equal_across_ranks, *_equal_single_rank, model_ms_per_step
    if rank == 0 and world == 1 and not a.no_extras:
        # ---- SURVEY 8d metric (1): reads resident in HOST memory -> the one-shot C entry point (upload through the pinned staging pump, compute,
        # download of graph and paths).  Untimed extra; the SECOND call is quoted (the first pays the context's device pool once per process).
        ctx.close(); ctx = None
        try:
            hp = d["packed"].cpu().numpy().reshape(-1); hq = d["quals"].cpu().numpy().reshape(-1)
            hbo = d["byte_off"].cpu().numpy().astype(np.uint64); hqo = d["qual_off"].cpu().numpy().astype(np.uint64); hln = d["read_len"].cpu().numpy().astype(np.uint32)
            runs = []
            for _ in range(3):
                tm = {}
                rr = step2.build_read_qgraph(hp, hbo, hln, quals=hq, qual_off=hqo, device=local_rank, timing=tm)
                runs.append(tm["run_s"])
            result["kmers_per_s_host_resident"] = rr.n_kmer_instances / runs[1]
            result["host_resident_second_call_s"] = runs[1]
            result["host_resident"] = {"first_call_s": runs[0], "second_call_s": runs[1], "third_call_s": runs[2], "input_bytes": int(hp.nbytes + hq.nbytes + hbo.nbytes + hqo.nbytes + hln.nbytes),
                                       "output_bytes": int(rr.path_edges.nbytes + rr.path_off.nbytes + rr.path_offset.nbytes + rr.hbv.edge_packed.nbytes),
                                       "note": "w2rap_step2_run on pageable host arrays (raw qualities), PCIe both ways included; never `value`.  The bases and a one-bit quality mask go up first, the raw qualities travel under the counting and graph phases (W2RAP_NO_UPLOAD_OVERLAP=1: the plain order)"}
            result["host_resident_second_call_s"] = runs[1]
            del hp, hq, hbo, hqo, hln, rr
        except Exception as e:
            result["host_resident"] = {"error": str(e)[:300]}
        # the same call on what the reference's caller HOLDS (w2rap-contigger.cc:326-338): the bases and the PQVec byte strings of its VecPQVec --
        # produced here by Step 1 from fastq text, as its pipeline does; the qualities are decoded on the device (k_decode_pq)
        try:
            from w2rap_contigger_amd import step1
            t1, _ = fastq_text_device(d, 0, dev); t2, _ = fastq_text_device(d, 1, dev)
            with step2.Step2Context(local_rank) as c1:
                r1 = step1.extract_reads((t1.data_ptr(), t1.numel()), (t2.data_ptr(), t2.numel()), ctx=c1)
            del t1, t2
            torch.cuda.empty_cache()
            runs = []
            for _ in range(2):
                tm = {}
                rr = step2.build_read_qgraph(r1.packed, r1.byte_off, r1.read_len, pq=r1.pq, pq_off=r1.pq_off, device=local_rank, timing=tm)
                runs.append(tm["run_s"])
            result["host_resident_pq"] = {"first_call_s": runs[0], "second_call_s": runs[1],
                                          "input_bytes": int(r1.packed.nbytes + r1.byte_off.nbytes + r1.read_len.nbytes + r1.pq.nbytes + r1.pq_off.nbytes),
                                          "pq_bytes": int(r1.pq.nbytes), "kmers_per_s": rr.n_kmer_instances / runs[1],
                                          "note": "w2rap_step2_run on pageable host arrays with the qualities as PQVec byte strings (what VecPQVec holds), PCIe both ways included; "
                                                  "the synthetic qualities (five values at random per base) make run-length PQVecs 2.4x LARGER than raw bytes -- binned real data is the opposite"}
            del r1, rr
        except Exception as e:
            result["host_resident_pq"] = {"error": str(e)[:300]}
        # ... and once on BINNED qualities -- eight distinct values in runs (what current Illumina instruments write, and the case the
        # reference's run-length PQVec encoding was made for): the PQVec bytes shrink below the raw bytes and the call with them
        try:
            from w2rap_contigger_amd import step1
            gq = torch.Generator(device=dev).manual_seed(99)
            bins = torch.tensor([2, 11, 18, 23, 27, 32, 36, 40], dtype=torch.uint8, device=dev)
            nq = d["quals"].numel()
            db = dict(d)
            qb = torch.empty(nq, dtype=torch.uint8, device=dev)
            for a0 in range(0, nq, 1 << 28):                              # runs of ~25 equal values (geometric), the bin of a run drawn at random; Q2 tails stay
                b0 = min(nq, a0 + (1 << 28))
                starts = torch.rand(b0 - a0, generator=gq, device=dev) < 0.04
                run = torch.cumsum(starts.to(torch.int32), 0)
                val = bins[((run.to(torch.int64) * 2654435761) >> 7) % 8]
                src = d["quals"].view(-1)[a0:b0]
                qb[a0:b0] = torch.where(src == 2, src, val)
                del starts, run, val
            db["quals"] = qb.view(d["quals"].shape)
            t1, _ = fastq_text_device(db, 0, dev); t2, _ = fastq_text_device(db, 1, dev)
            del db, qb
            with step2.Step2Context(local_rank) as c1:
                r1 = step1.extract_reads((t1.data_ptr(), t1.numel()), (t2.data_ptr(), t2.numel()), ctx=c1)
            del t1, t2
            torch.cuda.empty_cache()
            runs = []
            for _ in range(2):
                tm = {}
                rr = step2.build_read_qgraph(r1.packed, r1.byte_off, r1.read_len, pq=r1.pq, pq_off=r1.pq_off, device=local_rank, timing=tm)
                runs.append(tm["run_s"])
            result["host_resident_pq_binned"] = {"first_call_s": runs[0], "second_call_s": runs[1],
                                                 "input_bytes": int(r1.packed.nbytes + r1.byte_off.nbytes + r1.read_len.nbytes + r1.pq.nbytes + r1.pq_off.nbytes),
                                                 "pq_bytes": int(r1.pq.nbytes), "raw_quality_bytes": int(nq), "kmers_per_s": rr.n_kmer_instances / runs[1],
                                                 "note": "the same call on PQVec byte strings of BINNED qualities: eight distinct values in runs of ~25 (Q2 tails kept)"}
            del r1, rr
        except Exception as e:
            result["host_resident_pq_binned"] = {"error": str(e)[:300]}
        del d
        torch.cuda.empty_cache()
        # ---- a second workload beside configs[1] (never instead of it): planted repeats, inverted repeats and a second haplotype
        try:
            dp = planted_reads(n_reads, 4343, dev)
            torch.cuda.synchronize(dev); torch.cuda.empty_cache()
            with step2.Step2Context(local_rank) as cp:
                cp.set_reads_device(dp["n"], dp["packed"].data_ptr(), dp["byte_off"].data_ptr(), dp["read_len"].data_ptr(), dp["quals"].data_ptr(), dp["qual_off"].data_ptr(), keepalive=dp)
                for _ in range(1 + 3):
                    torch.cuda.synchronize(dev)
                    tp0 = time.perf_counter()
                    stp = cp.count_kmers(7, 4); tp1 = time.perf_counter()
                    cp.build_graph(None); tp2 = time.perf_counter()
                    cp.path_reads(); torch.cuda.synchronize(dev); tp3 = time.perf_counter()
                szp = cp.counts()
            result["planted_workload"] = {"workload": f"{dp['n']} PE150 reads of two haplotypes of {n_reads * 5 // 2} bp (1 SNP / 2 kb) with 2000 planted copies of 40 repeat families "
                                                      f"(500-5000 bp, a third inverted); last of 3 timed steps", "ms_per_step": (tp3 - tp0) * 1e3,
                                          "value": int(stp["M"]) / (tp3 - tp0), "unit": "k-mers/s",
                                          "phase_ms": {"count": (tp1 - tp0) * 1e3, "graph": (tp2 - tp1) * 1e3, "path": (tp3 - tp2) * 1e3},
                                          "kmer_instances": int(stp["M"]), "kmers_solid": int(stp["S"]), "unipaths": szp["unipaths"], "edge_objects": szp["edge_objects"],
                                          "vertices": szp["vertices"], "reads_pathed": szp["reads_pathed"], "path_elements": szp["path_elements"]}
            result["planted_ms_per_step"] = (tp3 - tp0) * 1e3
            result["planted_count_ms"], result["planted_graph_ms"], result["planted_path_ms"] = (tp1 - tp0) * 1e3, (tp2 - tp1) * 1e3, (tp3 - tp2) * 1e3
            del dp
        except Exception as e:
            result["planted_workload"] = {"error": str(e)[:300]}
        torch.cuda.empty_cache()
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        if ctx is not None:
            ctx.close()
        d = None
        torch.cuda.empty_cache()
        n_cpu = int(a.cpu_reads)
        secs, cores, kind, sample, dc, leaves, ref_files = cpu_baseline(n_cpu, n_cpu * 5, 4242, dev)
        parity = None
        with step2.Step2Context(local_rank) as c2:       # M of the sample from our own K0 (exact)
            c2.set_reads_device(dc["n"], dc["packed"].data_ptr(), dc["byte_off"].data_ptr(), dc["read_len"].data_ptr(),
                                dc["quals"].data_ptr(), dc["qual_off"].data_ptr(), keepalive=dc)
            m_cpu = c2.quality_windows(7)
            if ref_files is not None:
                # the GPU path on the SAME sample against the files the reference has just written: the line carries the verdict,
                # and the process exits non-zero on a difference (below, after the line is out)
                try:
                    parity = same_as_reference(c2, ref_files)
                except Exception as e:
                    parity = {"same_graph_as_gpu": False, "error": str(e)[:300]}
        result["cpu_baseline"] = {"value": m_cpu / secs, "unit": "k-mers/s", "cores": cores, "kind": kind, "sample": sample,
                                  "seconds": secs, "reads_per_s": dc["n"] / secs, "threads": cores, "task_tree_leaves": leaves,
                                  "note": "the reference's counting runs one thread per 1 M-read leaf of its task tree (BuildReadQGraph.cc:1018,1266): "
                                          f"{leaves} of the {cores} threads work in that phase, the merges near the root and the dictionary fill are serial"}
        if parity is not None:
            result["cpu_baseline"].update(parity)
            parity_failed = not parity["same_graph_as_gpu"]
        del dc
        torch.cuda.empty_cache()
        # the same comparison on the PLANTED generator (repeats, inverted repeats, SNP bubbles: SURVEY 8d "plus planted features") -- the
        # uniform genome exercises none of the extension ties, bubbles or many-part reads at scale
        if ref_files is not None and not a.no_planted_parity:
            try:
                n_pl = int(a.planted_cpu_reads)
                secs2, _, _, _, dp2, _, ref2 = cpu_baseline(n_pl, n_pl * 5, 5151, dev, planted=True)
                with step2.Step2Context(local_rank) as c3:
                    c3.set_reads_device(dp2["n"], dp2["packed"].data_ptr(), dp2["byte_off"].data_ptr(), dp2["read_len"].data_ptr(), dp2["quals"].data_ptr(), dp2["qual_off"].data_ptr(), keepalive=dp2)
                    pp = same_as_reference(c3, ref2)
                result["planted_same_graph_as_gpu"] = bool(pp["same_graph_as_gpu"])
                result["planted_parity"] = dict(pp, reads=dp2["n"], reference_seconds=secs2)
                parity_failed = parity_failed or not pp["same_graph_as_gpu"]
                # ... and the same sample through the routes every MULTI-GPU run takes: read pathing through the
                # minimizer-sampled index + exact table on one GPU (W2RAP_PATH_INDEX=1), and the sharded graph phase behind the one
                # in-process call with two ranks on this GPU
                try:
                    os.environ["W2RAP_PATH_INDEX"] = "1"
                    with step2.Step2Context(local_rank) as c4:
                        c4.set_reads_device(dp2["n"], dp2["packed"].data_ptr(), dp2["byte_off"].data_ptr(), dp2["read_len"].data_ptr(), dp2["quals"].data_ptr(), dp2["qual_off"].data_ptr(), keepalive=dp2)
                        pi = same_as_reference(c4, ref2)
                finally:
                    os.environ.pop("W2RAP_PATH_INDEX", None)
                result["planted_same_graph_as_gpu_index"] = bool(pi["same_graph_as_gpu"])
                result["planted_parity_index"] = pi
                ps = sharded_as_reference(dp2, ref2, local_rank)
                result["planted_same_graph_as_gpu_sharded"] = bool(ps["same_graph_as_gpu"])
                result["planted_parity_sharded"] = ps
                parity_failed = parity_failed or not pi["same_graph_as_gpu"] or not ps["same_graph_as_gpu"]
                del dp2
            except Exception as e:
                result.setdefault("planted_same_graph_as_gpu", False)
                result.setdefault("planted_same_graph_as_gpu_index", False)
                result.setdefault("planted_same_graph_as_gpu_sharded", False)
                result["planted_parity_error"] = str(e)[:300]
                parity_failed = True
            torch.cuda.empty_cache()
    if rank == 0 and world == 1 and not a.no_extras:
        # ---- the "next" rows of SURVEY.md 8f on the same box in the same run (untimed extras like planted_workload: each is its own
        # workload and its own steps; the full lines come from bench.py --step3 / --step1 / --gfa / --pipeline)
        import copy
        extras = {}
        for name, fn in (("step3", main_step3), ("step1", main_step1), ("gfa", main_gfa), ("pipeline", main_pipeline)):
            try:
                a2 = copy.copy(a); a2.reads = 0; a2.steps = 2; a2.warmup = 1; a2.no_cpu_baseline = True
                torch.cuda.empty_cache()
                r = fn(a2, extra=True)
                e = {"metric": r["metric"], "value": r["value"], "unit": r["unit"], "ms_per_step": r["ms_per_step"], "steps": a2.steps, "workload": r["config"]["workload"]}
                if "roofline" in r:
                    e["roofline"] = {k: r["roofline"].get(k) for k in ("kernel", "achieved", "peak", "unit", "frac", "bound")}
                for k in ("phase_ms", "stage_ms", "device_ms"):
                    if k in r:
                        e[k] = r[k]
                extras[name] = e
            except Exception as ex:
                extras[name] = {"error": str(ex)[:300]}
        result["extras"] = extras
    if rank == 0:
        with_copy_rate(result["roofline"], dev)
    emit(json.dumps(result))
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    if parity_failed:
        sys.exit("bench.py: the GPU path's output differs from the reference's on a cpu_baseline sample (cpu_baseline.same_graph_as_gpu or one of the "
                 "planted_same_graph_as_gpu* keys is false)")


if __name__ == "__main__":
    main()
