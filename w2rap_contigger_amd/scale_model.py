"""Per-phase cost model of `bench.py --gpus N` (BASELINE configs[2] scaled to N GPUs: fixed reads per GPU, ONE genome of N x 312.5 Mbp),
re-derived in round 5 for the SHARDED graph phase (row e-3: dictionary, prune and unipaths stay with the bucket owners).

No multi-GPU node was available to the builder in rounds 1-5: this model says what the first hardware run should show, phase by phase, from
numbers that WERE measured -- the per-kernel times of the sharded code path forced to run at world 1 on the per-GPU share of the workload
(`W2RAP_FORCE_DIST=1 W2RAP_TRACE=1 python bench.py --reads 62.5e6 --genome 312.5e6`: profiles/r05_dist_world1.json, loaded below when it is
there), the sizes of what travels, and the link rates of MI355X_MICROARCH.md.  EVERYTHING beyond world 1 is an unvalidated model (LINK_EFF
is an assumption).  `tests/test_scale_model.py` pins the arithmetic on CPU; `bench.py --gpus N` prints `model_ms_per_step` beside the
measured time.

What scales how:
  * per GPU, constant in N (weak scaling): quality windows, partition (K1/K2), owner-side counting (K3) with the owner's own dictionary built
    under it, the sharded part of the graph phase (prune, links, level-1 ranking, segments, middle bases, edge deposit), read pathing;
  * exchanges: the k-mer shuffle ((N-1)/N of the records, slice by slice under the counting); the prune's neighbour queries (24 B per
    query and answer, ~0.7 (N-1)/N per solid k-mer), context and segment queries (16 B, ~2/47 (N-1)/N per node);
  * REPLICATED on every rank, proportional to the JOB: level 2 of the list ranking (segments ~ 2 x 2/47 x (N-1)/N x S_total: 32 B gathered
    each, ~3 jump launches over them), the packed edge stream (all-reduce of 0.25 B per genome base) and what is built from it: byte codes,
    pathing index, absence filter (genome-sized), the E-sized unipath bookkeeping.
"""
import json
import os
from dataclasses import dataclass, asdict

XGMI_LINK_GBS = 153.0       # per link and direction, 7 links per GPU (MI355X_MICROARCH.md)
LINK_EFF = 0.7              # fraction of the link rate a large point-to-point copy reaches (assumption until measured)
PROFILE = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r05_dist_world1.json")


@dataclass
class World1:
    """milliseconds of the SHARDED code path at world 1 on the per-GPU share of the workload (62.5 M reads, 312.5 Mbp); defaults =
    profiles/r05_dist_world1.json as committed (from_profile() reads the file)"""
    quality: float = 2.1
    partition: float = 27.4          # K1 + K2 over the rank's reads
    count: float = 65.0              # exchange + owner-side counting of the four slices, the owner's dictionary built under it
    first_slice_exposed: float = 8.1  # counts, offsets, exchange + launch of slice 0, which nothing hides (inside `count` at world 1)
    graph_sharded: float = 48.0      # prune (local, shard, final), links, level-1 ranking, segments, middle bases, edge deposit: ~ S_1
    graph_replicated_per_gbase: float = 31.0   # per 10^9 genome bases: pack + unpack codes 2.3, pathing index 8.5, absence filter 18.3 (its own stream), a8 ~2
    level2_ns_per_segment: float = 0.25        # ns per gathered segment: 3 jump launches + unpack + finish + heads over random 8-B words (k_split_jump: 0.17 ns per splitter and launch)
    path: float = 22.1               # this rank's reads against the index
    record_bytes_per_gpu: float = 9.1e9      # super-k-mer records a rank produces (284 M of 32 B each)
    solid_per_gpu: float = 312e6             # S_1
    genome_bases_per_gpu: float = 312.5e6

    @staticmethod
    def from_profile(path=PROFILE):
        w = World1()
        try:
            d = json.load(open(path))
            k = d["kernel_ms_per_step"]; ph = d["phase_ms"]
            w.path = ph["path"]
            sharded = sum(k.get(n, 0.0) for n in ("k_prune_local", "k_prune_shard", "k_prune_final", "k_links_shard", "k_rank_tiles", "k_split_jump", "k_rank_finish",
                                                 "k_seg_number", "k_mid_shard", "k_assign_shard"))
            repl = sum(k.get(n, 0.0) for n in ("k_pack_words", "k_unpack_codes", "k_pack_codes", "k_index_fill", "k_filter32", "k_ends", "rocprim_radix_sort_pairs"))
            w.graph_sharded = sharded + max(0.0, ph["graph"] - sharded - repl)          # host-side exchange overhead stays with the sharded part
            w.graph_replicated_per_gbase = repl / (d["config"]["genome_bp"] / 1e9)
            w.count = ph["count"] - w.quality - w.partition
            w.solid_per_gpu = float(d["config"]["kmers_solid"]); w.genome_bases_per_gpu = float(d["config"]["genome_bp"])
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
    # the shuffle: (n-1)/n of the records leave the rank; it runs under the counting except for slice 0
    shuffle = w.record_bytes_per_gpu * far / bw * 1e3 if n > 1 else 0.0
    exposed_shuffle = max(0.0, shuffle - w.count)
    # the sharded graph phase's query / answer rounds: A (24 B x 0.7 per solid k-mer), B and C (16 B x 2 x 2/47 per k-mer), both directions
    xchg = (w.solid_per_gpu * far * (0.7 * 24 + 2 * (2 / 47) * 2 * 16)) / bw * 1e3 if n > 1 else 0.0
    # level 2: segments of the JOB, gathered (32 B) and ranked on every rank
    segments = 2 * (2 / 47) * far * w.solid_per_gpu * n
    level2 = segments * 32 * far / bw * 1e3 + segments * w.level2_ns_per_segment * 1e-6 if n > 1 else 0.0
    # the edge stream (0.25 B per base of the JOB) summed over the ranks, then everything genome-sized on every rank
    stream = (w.genome_bases_per_gpu * n / 4) * 2 * far / bw * 1e3 if n > 1 else 0.0
    repl = w.graph_replicated_per_gbase * w.genome_bases_per_gpu * n / 1e9
    phases = {"quality": w.quality, "partition": w.partition, "count": w.count, "shuffle_exposed": exposed_shuffle, "graph_sharded": w.graph_sharded,
              "graph_exchanges": xchg, "level2_replicated": level2, "edge_stream_allreduce": stream, "graph_replicated": repl, "path": w.path}
    total = sum(phases.values())
    replicated = level2 + repl
    return {"n_gpus": n, "phase_ms": phases, "ms_per_step": total, "replicated_ms": replicated}


def table(w: World1 = None):
    w = w or World1.from_profile()
    t1 = predict(1, w)["ms_per_step"]
    rows = []
    for n in (1, 2, 4, 8):
        p = predict(n, w)
        rows.append({"n_gpus": n, "ms_per_step": round(p["ms_per_step"], 1), "weak_scaling_efficiency": round(t1 / p["ms_per_step"], 3),
                     "replicated_share": round(p["replicated_ms"] / p["ms_per_step"], 3)})
    return rows


if __name__ == "__main__":
    w = World1.from_profile()
    print(json.dumps({"assumptions": asdict(w), "link_GBs": XGMI_LINK_GBS, "link_eff": LINK_EFF, "validated": "world 1 only", "prediction": table(w)}, indent=1))
