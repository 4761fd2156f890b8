import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR","127.0.0.1"); os.environ.setdefault("MASTER_PORT","29545"); os.environ.setdefault("RANK","0"); os.environ.setdefault("WORLD_SIZE","1")
import torch, torch.distributed as dist
dist.init_process_group("gloo", rank=0, world_size=1)
from groove_amd import entities as E
ctx = E.Context(0)
try:
    uid = ctx.comm_unique_id(); print("uid ok")
    ctx.comm_init(uid, 0, 1); print("comm ok, ranks", ctx.comm_ranks())
except Exception as e:
    print("FAILED:", e)
bus = ctx.bus(512); ctx.bus_reduce(bus, 512, 0); ctx.synchronize(); print("reduce ok")
dist.destroy_process_group(); ctx.close()
