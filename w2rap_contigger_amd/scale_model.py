"""Per-phase cost model of `bench.py --gpus N` (BASELINE configs[2] scaled to N GPUs: fixed reads per GPU, ONE genome of N x 312.5 Mbp)
for the SHARDED graph phase (row e-3: dictionary, prune and unipaths stay with the bucket owners).

No multi-GPU node was available to the builder in rounds 1-6.  What one GPU CAN measure, it measured -- three runs of the sharded code path
forced to run at world 1 on the per-GPU share of the workload (`W2RAP_FORCE_DIST=1 python bench.py --reads 62.5e6 --genome 312.5e6`),
committed under profiles/ and loaded below:
  * r06_dist_world1.json: the plain run (every k-mer local: no queries, ~2 segments per unipath);
  * r06_dist_world1_cut27_v8.json: with the two test hooks that give ONE rank the list sizes of an 8-rank job -- W2RAP_TEST_SHARD_VIRTUAL=8
    (the neighbour k-mers whose bucket would belong to another of 8 owners are asked for through the routed query path: 81 M queries, 0.26
    per solid k-mer) and W2RAP_TEST_SHARD_CUT=27 (one local chain link in 27 is handed to the cross-rank machinery: 23 M segments, what
    2/47 x 7/8 of the links crossing ranks gives): the KERNEL cost per GPU of everything that crosses ranks at N = 8;
  * r05_dist_world1_cut3_v0.json: one link in 3 cut -- 208 M segments, the segment count of the whole 8-rank JOB: the cost of what level 2
    still does on every rank alike (streaming over the job's words, the splitter jumping).
  * r06_one_gpu_62M.json: the ONE-GPU path (no shuffle, dictionary pathing, every side-stream overlap) on the same 62.5 M reads: what
    `efficiency_vs_one_gpu` is quoted against -- the sharded path at world 1 is itself 24 % slower than that (equal kernel time, lost overlap:
    NOTES.md round 6), so "efficiency against the sharded world-1 run" flatters the scaling.
What is NOT measured is every byte on a link: link times are priced at the rates of MI355X_MICROARCH.md with LINK_EFF (an assumption).
`tests/test_scale_model.py` pins the arithmetic on CPU; `bench.py --gpus N` prints `model_ms_per_step` beside the measured time.

What scales how:
  * per GPU, constant in N (weak scaling): quality windows, partition (K1/K2), owner-side counting (K3) with the owner's own dictionary built
    under it, the sharded part of the graph phase (prune, links, level-1 ranking, segments, the level-2 WALKS from a rank's own splitters,
    middle bases, edge deposit, index listing, the rank's range of the filter), read pathing;
  * per GPU, growing with (N-1)/N: the kernels of the three query rounds and of the segment level (owner computation, routing, answers);
  * exchanges: the k-mer shuffle ((N-1)/N of the records, slice by slice under the counting); A: 24 B per query + 8 B per answer; B, C:
    8 + 8 B; level 2: 8 B per job segment gathered, 32 B per splitter / head, 16 B per segment routed; edge stream all-reduce (2 x 0.25 B per
    job base), index entries (1.4 B per job base) and filter words (0.5 B per job base) gathered;
  * REPLICATED on every rank, proportional to the JOB: the streaming part of level 2 (splitter marks, jumping over 1/64 of the segments),
    the insertion of the gathered index entries and the exact table beside the index, the byte codes, the E-sized unipath bookkeeping.
"""
import json
import os
from dataclasses import dataclass, asdict

XGMI_LINK_GBS = 153.0       # per link and direction, 7 links per GPU (MI355X_MICROARCH.md)
LINK_EFF = 0.7              # fraction of the link rate a large point-to-point copy reaches (assumption until measured)
_PROFILES = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
PROFILE = os.path.join(_PROFILES, "r06_dist_world1.json")
PROFILE_8 = os.path.join(_PROFILES, "r06_dist_world1_cut27_v8.json")
PROFILE_ONE_GPU = os.path.join(_PROFILES, "r06_one_gpu_62M.json")
PROFILE_JOB = os.path.join(_PROFILES, "r05_dist_world1_cut3_v0.json")

_REPLICATED_GENOME = ("k_index_insert", "k_exact_insert", "k_unpack_codes", "k_ends", "k_radix_sort_pairs", "k_heads_shard", "k_edges_sorted", "k_edges_hint")
_REPLICATED_SEGMENTS = ("k_seg_mark", "k_seg_finish", "k_seg_jump", "k_seg_splitters_done", "k_seg_splitters_store", "k_l2_apply", "k_stripes_compact")


@dataclass
class World1:
    """milliseconds of the SHARDED code path at world 1 on the per-GPU share of the workload (62.5 M reads, 312.5 Mbp); defaults = the
    committed profiles (from_profile() reads the files)"""
    quality: float = 1.8
    partition: float = 27.4          # K1 + K2 over the rank's reads
    count: float = 65.0              # exchange + owner-side counting of the four slices, the owner's own dictionary built under it
    graph_sharded: float = 59.0      # prune, links, level-1 ranking, segments, middle bases, edge deposit, index listing, filter range: ~ S_1
    graph_replicated_per_gbase: float = 11.3   # per 10^9 genome bases of the JOB: index insert 7.7, unpack codes 1.5, unipath sort + ends ~2
    path: float = 22.5               # this rank's reads against the index
    # what crosses ranks, per GPU at the list sizes of an 8-rank job (cut27_v8 minus the plain run): kernels only
    cross_rank_at_8: float = 25.0
    queries_per_kmer_at_8: float = 0.26        # A queries per solid k-mer at (N-1)/N = 7/8
    segments_per_kmer_at_8: float = 0.0744     # chain segments (both orientations) per solid k-mer at 7/8
    # what level 2 does on every rank alike, per segment of the JOB (cut3: 208 M segments): marks, finish, jumping, records applied; + ~40 B of memsets
    level2_replicated_ns_per_segment: float = 0.03
    record_bytes_per_gpu: float = 9.1e9      # super-k-mer records a rank produces (284 M of 32 B each)
    solid_per_gpu: float = 312e6             # S_1
    genome_bases_per_gpu: float = 312.5e6

    @staticmethod
    def from_profile(path=PROFILE, path_8=PROFILE_8, path_job=PROFILE_JOB):
        w = World1()
        try:
            d = json.load(open(path))
            k = d["kernel_ms_per_step"]; ph = d["phase_ms"]
            w.path = ph["path"]
            w.quality = k.get("k_good_len", w.quality)
            w.partition = k.get("k_superkmers_lane", 0.0) + k.get("k_scatter_records", 0.0)
            repl = sum(k.get(n, 0.0) for n in _REPLICATED_GENOME)
            w.graph_sharded = ph["graph"] - repl                                      # host-side exchange overhead stays with the sharded part
            w.graph_replicated_per_gbase = repl / (d["config"]["genome_bp"] / 1e9)
            w.count = ph["count"] - w.quality - w.partition
            w.solid_per_gpu = float(d["config"]["kmers_solid"]); w.genome_bases_per_gpu = float(d["config"]["genome_bp"])
            d8 = json.load(open(path_8))
            w.cross_rank_at_8 = d8["phase_ms"]["graph"] - ph["graph"]
            dj = json.load(open(path_job))
            kj = dj["kernel_ms_per_step"]
            seg_job = 2.0 / 3.0 * float(dj["config"]["kmers_solid"])                  # one link in three cut: two segments per three k-mers
            w.level2_replicated_ns_per_segment = sum(kj.get(n, 0.0) for n in _REPLICATED_SEGMENTS) * 1e6 / seg_job + 40.0 / 3.0e3       # + 40 B of memsets at ~3 TB/s
        except Exception:
            pass
        return w


def predict(n_gpus: int, w: World1 = None) -> dict:
    """-> per-phase ms at n_gpus and the step time; weak-scaling efficiency = t(1) / t(N)"""
    w = w or World1.from_profile()
    n = max(1, int(n_gpus))
    links = max(1, min(n - 1, 7))
    bw = links * XGMI_LINK_GBS * 1e9 * LINK_EFF                     # bytes per second into / out of one GPU
    far = (n - 1) / n
    rel = far / 0.875                                               # list sizes relative to the 8-rank emulation
    # the shuffle: (n-1)/n of the records leave the rank; it runs under the counting except for slice 0
    shuffle = w.record_bytes_per_gpu * far / bw * 1e3 if n > 1 else 0.0
    exposed_shuffle = max(0.0, shuffle - w.count)
    # kernels of the query rounds and of the segment level, per GPU
    cross = w.cross_rank_at_8 * rel if n > 1 else 0.0
    queries = w.solid_per_gpu * w.queries_per_kmer_at_8 * rel
    seg_own = w.solid_per_gpu * w.segments_per_kmer_at_8 * rel       # segments among a rank's k-mers
    seg_job = seg_own * n
    # bytes on the links, per GPU: A (24 + 8 B), B and C (16 B per segment end), level 2 (8 B per job segment gathered from the others,
    # 32 B per splitter and head ~ 1/32 of them, 16 B per own segment routed)
    xchg_bytes = queries * 32 + seg_own * 2 * 16 + seg_job * far * (8 + 32 / 32) + seg_own * far * 16
    xchg = xchg_bytes / bw * 1e3 if n > 1 else 0.0
    level2 = seg_job * w.level2_replicated_ns_per_segment * 1e-6 if n > 1 else 0.0
    # the edge stream (0.25 B per base of the JOB) summed over the ranks, the index entries (2 x 16 B per ~23 bases) and the filter words
    # (4 B per 8 bases) gathered
    stream = (w.genome_bases_per_gpu * n * (2 * 0.25 + 1.4 + 0.5)) * far / bw * 1e3 if n > 1 else 0.0
    repl = w.graph_replicated_per_gbase * w.genome_bases_per_gpu * n / 1e9
    phases = {"quality": w.quality, "partition": w.partition, "count": w.count, "shuffle_exposed": exposed_shuffle, "graph_sharded": w.graph_sharded,
              "cross_rank_kernels": cross, "graph_exchanges": xchg, "level2_replicated": level2, "edge_stream_allreduce": stream, "graph_replicated": repl,
              "path": w.path}
    total = sum(phases.values())
    replicated = level2 + repl
    return {"n_gpus": n, "phase_ms": phases, "ms_per_step": total, "replicated_ms": replicated}


def one_gpu_ms(path=PROFILE_ONE_GPU):
    """ms per step of the ONE-GPU path on the per-GPU share of the workload (the committed profile), or None"""
    try:
        return float(json.load(open(path))["ms_per_step"])
    except Exception:
        return None


def table(w: World1 = None):
    """per N: the model's step, its weak-scaling efficiency against the sharded path's own world-1 run, and -- the honest figure -- against
    the ONE-GPU path on the same per-GPU share (efficiency_vs_one_gpu)"""
    w = w or World1.from_profile()
    t1 = predict(1, w)["ms_per_step"]
    t_one = one_gpu_ms()
    rows = []
    for n in (1, 2, 4, 8):
        p = predict(n, w)
        rows.append({"n_gpus": n, "ms_per_step": round(p["ms_per_step"], 1), "weak_scaling_efficiency": round(t1 / p["ms_per_step"], 3),
                     "efficiency_vs_one_gpu": round(t_one / p["ms_per_step"], 3) if t_one else None,
                     "replicated_share": round(p["replicated_ms"] / p["ms_per_step"], 3)})
    return rows


if __name__ == "__main__":
    w = World1.from_profile()
    print(json.dumps({"assumptions": asdict(w), "link_GBs": XGMI_LINK_GBS, "link_eff": LINK_EFF,
                      "validated": "world 1 (kernel costs of the cross-rank machinery measured at the list sizes of 8 ranks; links priced, not measured)",
                      "prediction": table(w)}, indent=1))
