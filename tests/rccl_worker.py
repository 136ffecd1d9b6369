"""Worker of tests/test_gpu_dist.py: one rank of a torch.distributed.run job on the GPU box, backend "nccl" (= RCCL on
ROCm).  Exercises exactly what bench.py and the training step use from sgcdet_amd/dist.py: init_from_env, max_over_ranks,
one OverlappedGradAllReduce step and one SyncBatchNorm3d layer, each checked against the value computed from all ranks'
inputs (every rank can rebuild them: the inputs are seeded by rank).  With WORLD_SIZE == 1 (a 1-GPU box) the process
group is still an RCCL group of one rank, so the collectives really go through librccl."""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sgcdet_amd import dist as sd  # noqa: E402


def main():
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    rank, world, _ = sd.init_from_env(backend="nccl", device=dev)
    if world == 1 and not dist.is_initialized():               # a group of one: init_from_env skips it, RCCL should still load
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    assert dist.get_backend() == "nccl" and dist.get_world_size() == world
    # 1. the bench's timing reduction
    assert sd.max_over_ranks(1.0 + rank, device=dev) == float(world) if world > 1 else True
    t = torch.full((1024,), float(rank + 1), device=dev)
    dist.all_reduce(t)
    assert float(t[0]) == world * (world + 1) / 2
    # 2. one overlapped gradient all-reduce step == mean over ranks of the per-rank gradients
    def grads_of(r):
        torch.manual_seed(0)
        m = torch.nn.Sequential(torch.nn.Linear(64, 64), torch.nn.ReLU(), torch.nn.Linear(64, 8)).to(dev)
        x = torch.randn(16, 64, generator=torch.Generator().manual_seed(10 + r)).to(dev)
        return m, x
    m, x = grads_of(rank)
    sync = sd.OverlappedGradAllReduce(m.parameters(), bucket_bytes=64 * 64 * 4)
    m(x).square().mean().backward()
    sync.finish()
    got = torch.cat([p.grad.reshape(-1) for p in m.parameters()])
    want = 0
    for r in range(world):
        mr, xr = grads_of(r)
        mr(xr).square().mean().backward()
        want = want + torch.cat([p.grad.reshape(-1) for p in mr.parameters()])
    want = want / world
    assert torch.allclose(got, want, rtol=1e-5, atol=1e-7), float((got - want).abs().max())
    sync.remove()
    # 3. one SyncBatchNorm3d layer == nn.BatchNorm3d on the concatenated batch of all ranks
    torch.manual_seed(1)
    bn = sd.SyncBatchNorm3d(8).to(dev).train()
    xs = [torch.randn(1, 8, 4, 4, 4, generator=torch.Generator().manual_seed(50 + r)).to(dev) for r in range(world)]
    y = bn(xs[rank])
    ref = torch.nn.BatchNorm3d(8).to(dev).train()
    yr = ref(torch.cat(xs))
    assert torch.allclose(y, yr[rank:rank + 1], rtol=1e-4, atol=1e-5)
    assert torch.allclose(bn.running_var, ref.running_var, rtol=1e-5, atol=1e-6)
    dist.barrier()
    if rank == 0:
        print(f"RCCL_OK world={world} backend={dist.get_backend()}")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
