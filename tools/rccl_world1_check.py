import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
os.environ["RANK"] = "0"; os.environ["WORLD_SIZE"] = "1"
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=dev)
t = torch.ones(4, device=dev, dtype=torch.float64)
dist.all_reduce(t, op=dist.ReduceOp.MAX); dist.barrier()
src = torch.arange(6, device=dev, dtype=torch.bfloat16).view(2, 3); out = torch.empty(2, 3, device=dev, dtype=torch.bfloat16)
dist.all_gather_into_tensor(out, src)
g = [None]; dist.all_gather_object(g, {"a": 1})
torch.cuda.synchronize(); print("rccl world-1 ok", t.tolist(), out.tolist(), g)
dist.destroy_process_group()
