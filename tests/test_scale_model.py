"""The per-phase cost model of `bench.py --gpus N` (w2rap_contigger_amd/scale_model.py): its arithmetic, pinned on CPU, and its tie to the
committed world-1 profile of the sharded path (profiles/r06_dist_world1.json)."""
import json
import os

from w2rap_contigger_amd import scale_model as SM


def test_world_1_is_the_sum_of_its_measured_phases():
    w = SM.World1()
    p = SM.predict(1, w)
    assert abs(p["ms_per_step"] - (w.quality + w.partition + w.count + w.graph_sharded + w.graph_replicated_per_gbase * w.genome_bases_per_gpu / 1e9 + w.path)) < 1e-9
    assert p["phase_ms"]["graph_exchanges"] == 0.0 and p["phase_ms"]["level2_replicated"] == 0.0 and p["phase_ms"]["cross_rank_kernels"] == 0.0


def test_the_defaults_are_the_committed_profile():
    """the model's inputs are read from the committed profile of the forced world-1 run, not copied by hand"""
    assert os.path.exists(SM.PROFILE)
    d = json.load(open(SM.PROFILE))
    w = SM.World1.from_profile()
    assert abs(w.path - d["phase_ms"]["path"]) < 1e-9
    p = SM.predict(1, w)
    assert abs(p["ms_per_step"] - d["ms_per_step"]) / d["ms_per_step"] < 0.05      # the model's world-1 step is the measured one (phases partition it)
    assert "graph sharded" in d["config"]["parallelism"]
    # the cross-rank machinery's kernel cost at the list sizes of 8 ranks and level 2's replicated part are MEASURED (the two test hooks)
    d8 = json.load(open(SM.PROFILE_8))
    assert abs(w.cross_rank_at_8 - (d8["phase_ms"]["graph"] - d["phase_ms"]["graph"])) < 1e-9 and 5 < w.cross_rank_at_8 < 60
    assert os.path.exists(SM.PROFILE_JOB) and 0.01 < w.level2_replicated_ns_per_segment < 0.1
    p8 = SM.predict(8, w)
    assert abs(p8["phase_ms"]["cross_rank_kernels"] - w.cross_rank_at_8) < 1e-9


def test_sharded_phases_stay_and_replicated_ones_grow_with_the_job():
    w = SM.World1()
    p1, p8 = SM.predict(1, w), SM.predict(8, w)
    for k in ("quality", "partition", "count", "graph_sharded", "path"):
        assert p8["phase_ms"][k] == p1["phase_ms"][k]
    assert abs(p8["phase_ms"]["graph_replicated"] - 8 * p1["phase_ms"]["graph_replicated"]) < 1e-9
    rows = SM.table(w)
    eff = [r["weak_scaling_efficiency"] for r in rows]
    assert eff[0] == 1.0 and all(a > b for a, b in zip(eff, eff[1:]))
    # what row e-3 bought: round 4's model (graph and dictionary replicated) had 0.31 at N = 8 with 77 % of the step replicated
    assert rows[-1]["weak_scaling_efficiency"] > 0.6 and rows[-1]["replicated_share"] < 0.15


def test_efficiency_is_also_quoted_against_the_one_gpu_path():
    """the sharded path at world 1 is itself slower than the one-GPU path on the same reads; the table says what N GPUs buy
    against THAT"""
    t_one = SM.one_gpu_ms()
    assert t_one is not None and 100 < t_one < 200
    rows = SM.table()
    assert rows[0]["efficiency_vs_one_gpu"] < 1.0                      # world 1 of the sharded path against the one-GPU path
    for r in rows:
        assert abs(r["efficiency_vs_one_gpu"] - t_one / r["ms_per_step"]) < 2e-3
        assert r["efficiency_vs_one_gpu"] < r["weak_scaling_efficiency"]
