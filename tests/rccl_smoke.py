"""RCCL smoke on the GPU box: the `nccl` backend (= RCCL on ROCm) initialised the way bench.py / the sharded forward do it (device_id at
init), one all-reduce of a node-aggregate-sized tensor and a barrier -- with ONE rank, which is all a 1-GPU box allows (RCCL refuses two
ranks on one device).  Run by tests/test_gpu_parity.py::test_rccl_backend_single_rank."""
import os

import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29555")
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
t = torch.ones(10002, 880, device=dev)
dist.all_reduce(t)
dist.barrier()
torch.cuda.synchronize()
assert float(t.sum()) == 10002 * 880
print("RCCL_OK", torch.cuda.nccl.version())
dist.destroy_process_group()
