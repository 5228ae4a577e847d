#!/usr/bin/env python3
"""count-phase tuning probe: W2RAP_K3 / W2RAP_KPB sweeps at 50 M reads"""
import os, sys, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    from w2rap_contigger_amd import step2, synth
    n = int(float(os.environ.get("N_READS", "5e7")))
    d = synth.generate_reads_device(n, n * 5, 42, device="cuda")
    torch.cuda.synchronize(); torch.cuda.empty_cache()
    with step2.Step2Context(0) as ctx:
        ctx.set_reads_device(d["n"], d["packed"].data_ptr(), d["byte_off"].data_ptr(), d["read_len"].data_ptr(), d["quals"].data_ptr(), d["qual_off"].data_ptr(), keepalive=d)
        for it in range(2):
            st = ctx.count_kmers(7, 4)
            prof = ctx.profile()
        print(json.dumps({k: round(v[0], 2) for k, v in prof.items()}), "S", st["S"], "D", st["D"], f"count {st['ms']:.1f} ms")
else:
    for k3 in sys.argv[1].split(","):
        for kpb in sys.argv[2].split(","):
            env = dict(os.environ, W2RAP_K3=k3, W2RAP_KPB=kpb)
            out = subprocess.run([sys.executable, __file__, "child"], env=env, capture_output=True, text=True)
            print(f"K3={k3} KPB={kpb}:", out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-300:], flush=True)
