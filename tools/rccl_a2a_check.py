"""all_to_all_single of 0.3 ... 6 GiB at world 1 on RCCL: which sizes come back intact (the k-mer shuffle is cut into rounds of dist.A2A_MAX_BYTES because the large ones do not)."""
import os, torch, torch.distributed as dist
ROW = 32        # bytes of a super-k-mer record (w2rap_step2_record_bytes(); 36 in a -DW2RAP_REC36 build)
os.environ.setdefault("MASTER_ADDR","127.0.0.1"); os.environ.setdefault("MASTER_PORT","29590")
dev=torch.device("cuda",0); torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
for nrows in (10_000_000, 59_652_323, 59_652_324, 120_000_000, 184_000_000):
    x=torch.randint(0,255,(nrows,ROW),dtype=torch.uint8,device=dev)
    y=torch.empty_like(x)
    dist.all_to_all_single(y,x,output_split_sizes=[nrows],input_split_sizes=[nrows])
    torch.cuda.synchronize()
    bad=(y!=x).any(dim=1)
    nb=int(bad.sum().item())
    first=int(bad.nonzero()[0].item()) if nb else -1
    print(nrows, nrows*ROW/2**30, "GiB bad rows", nb, "first", first, flush=True)
    del x,y,bad
# all_gather_into_tensor of 250M int64
t=torch.arange(250_000_000,dtype=torch.int64,device=dev); out=torch.empty_like(t)
dist.all_gather_into_tensor(out,t); torch.cuda.synchronize(); print("gather ok", bool((out==t).all().item()))
dist.destroy_process_group()
