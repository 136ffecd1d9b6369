"""World-size-2 gloo tests (CPU) of the N>1 plumbing: scene sharding, MAX timing reduction and the
bucketed gradient all-reduce on the (torch-path) neck in training mode."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from sgcdet_amd import dist as sd
    from sgcdet_amd.plugin.neck3d import FastIndoorImVoxelNeck
    torch.set_num_threads(1)
    r, w, _ = sd.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    mine = sd.shard_scenes(5, rank, world)
    slowest = sd.max_over_ranks(1.0 + rank)
    # one scene per rank through the (CPU, torch) training path of the neck, then gradient averaging
    torch.manual_seed(0)
    neck = FastIndoorImVoxelNeck(in_channels=8, n_blocks=[1, 1, 1], out_channels=4).train()
    scenes = [torch.randn(1, 8, 8, 8, 4, generator=torch.Generator().manual_seed(100 + i)) for i in range(world)]
    loss = sum(o.square().mean() for o in neck(scenes[rank]))
    loss.backward()
    sd.BucketedGradAllReduce(neck.parameters(), bucket_bytes=1 << 12)()
    grads = torch.cat([p.grad.reshape(-1) for p in neck.parameters()])
    # the overlapped form (hooks fire during backward, finish() after it) on a fresh copy, one parameter left unused
    torch.manual_seed(0)
    neck2 = FastIndoorImVoxelNeck(in_channels=8, n_blocks=[1, 1, 1], out_channels=4).train()
    unused = torch.nn.Parameter(torch.ones(3))
    sync = sd.OverlappedGradAllReduce(list(neck2.parameters()) + [unused], bucket_bytes=1 << 12)
    assert len(sync.buckets) > 3
    for step in range(2):                           # twice: the bookkeeping resets between steps
        for p in neck2.parameters():
            p.grad = None
        sum(o.square().mean() for o in neck2(scenes[rank])).backward()
        sync.finish()
    grads2 = torch.cat([p.grad.reshape(-1) for p in neck2.parameters()])
    assert unused.grad is not None and float(unused.grad.abs().sum()) == 0.0
    # a second backward() before finish() must not be silently lost
    for p in neck2.parameters():
        p.grad = None
    sum(o.square().mean() for o in neck2(scenes[rank])).backward()
    with pytest.raises(RuntimeError, match="second backward"):
        sum(o.square().mean() for o in neck2(scenes[rank])).backward()
    sync.finish()
    sync.remove()
    # the set of parameters that receive a gradient DIFFERS per rank (find_unused_parameters semantics): rank r leaves
    # branch r of a two-branch model unused, so its buckets fill in a different order -- collectives must still pair up
    torch.manual_seed(1)
    branches = torch.nn.ModuleList([torch.nn.Linear(16, 16) for _ in range(4)])
    sync3 = sd.OverlappedGradAllReduce(branches.parameters(), bucket_bytes=16 * 16 * 4)     # ~one bucket per layer
    assert len(sync3.buckets) >= 4
    x = torch.randn(3, 16, generator=torch.Generator().manual_seed(7))
    used = [i for i in range(4) if i != rank]                 # rank 0 skips branch 0, rank 1 skips branch 1
    sum(branches[i](x).square().mean() * (i + 1) for i in used).backward()
    sync3.finish()
    grads3 = torch.cat([p.grad.reshape(-1) for p in branches.parameters()])
    sync3.remove()
    if rank == 0:
        torch.save(dict(mine=mine, slowest=slowest, grads=grads, grads2=grads2, grads3=grads3, tiny=sd.shard_scenes(1, rank, 4)), out)
    else:
        torch.save(dict(mine=mine), out + ".1")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_gloo_sharding_and_grad_allreduce(tmp_path):
    out = str(tmp_path / "r0.pt")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    r0, r1 = torch.load(out), torch.load(out + ".1")
    assert r0["mine"] == [0, 2, 4] and r1["mine"] == [1, 3, 0]      # padded strided sharding
    assert r0["slowest"] == 2.0                                      # MAX over ranks
    # single-process reference: mean of the two per-scene gradients (BatchNorm sees one scene at a time)
    from sgcdet_amd.plugin.neck3d import FastIndoorImVoxelNeck
    ref = []
    for i in range(2):
        torch.manual_seed(0)
        neck = FastIndoorImVoxelNeck(in_channels=8, n_blocks=[1, 1, 1], out_channels=4).train()
        x = torch.randn(1, 8, 8, 8, 4, generator=torch.Generator().manual_seed(100 + i))
        sum(o.square().mean() for o in neck(x)).backward()
        ref.append(torch.cat([p.grad.reshape(-1) for p in neck.parameters()]))
    want = (ref[0] + ref[1]) / 2
    assert torch.allclose(r0["grads"], want, rtol=1e-5, atol=1e-7)
    assert torch.allclose(r0["grads2"], want, rtol=1e-5, atol=1e-7)        # OverlappedGradAllReduce: same means
    # per-rank different unused branches: mean over ranks with zeros for the rank that skipped the branch
    torch.manual_seed(1)
    branches = torch.nn.ModuleList([torch.nn.Linear(16, 16) for _ in range(4)])
    x = torch.randn(3, 16, generator=torch.Generator().manual_seed(7))
    per_rank = []
    for rank in range(2):
        for p in branches.parameters():
            p.grad = None
        sum(branches[i](x).square().mean() * (i + 1) for i in range(4) if i != rank).backward()
        per_rank.append(torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1)
                                   for p in branches.parameters()]))
    assert torch.allclose(r0["grads3"], (per_rank[0] + per_rank[1]) / 2, rtol=1e-5, atol=1e-7)
    assert r0["tiny"] == [0]                                                 # 1 scene, 4 ranks: nobody is left empty


def _syncbn_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from sgcdet_amd import dist as sd
    from sgcdet_amd.plugin.neck3d import FastIndoorImVoxelNeck
    torch.set_num_threads(1)
    sd.init_from_env(backend="gloo")
    torch.manual_seed(0)
    neck = sd.convert_sync_batchnorm(FastIndoorImVoxelNeck(in_channels=8, n_blocks=[1, 1, 1], out_channels=4)).train()
    n_bn = sum(isinstance(m, sd.SyncBatchNorm3d) for m in neck.modules())
    x = torch.randn(1, 8, 8, 8, 4, generator=torch.Generator().manual_seed(100 + rank))
    outs = neck(x)
    sum(o.square().mean() for o in outs).backward()
    sd.BucketedGradAllReduce(neck.parameters(), bucket_bytes=1 << 12)()
    grads = torch.cat([p.grad.reshape(-1) for p in neck.parameters()])
    stats = torch.cat([b.reshape(-1).float() for n, b in neck.named_buffers() if "running" in n])
    torch.save(dict(grads=grads, stats=stats, n_bn=n_bn, outs=[o.detach() for o in outs]), out + f".{rank}")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_sync_batchnorm_two_ranks_equal_a_single_process_batch_of_two(tmp_path):
    """SyncBN (reference: pl.Trainer(sync_batchnorm=True), main.py:81) with ONE scene per rank == nn.BatchNorm3d on the
    two-scene batch in one process: activations, running statistics and -- after the gradient all-reduce (mean over
    ranks, as DDP) -- the gradients of the loss 0.5 * (L_0 + L_1)."""
    out = str(tmp_path / "sbn")
    mp.spawn(_syncbn_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    r0, r1 = torch.load(out + ".0"), torch.load(out + ".1")
    assert r0["n_bn"] == 15                                    # every BatchNorm3d of the neck was converted
    from sgcdet_amd.plugin.neck3d import FastIndoorImVoxelNeck
    torch.manual_seed(0)
    neck = FastIndoorImVoxelNeck(in_channels=8, n_blocks=[1, 1, 1], out_channels=4).train()
    x = torch.cat([torch.randn(1, 8, 8, 8, 4, generator=torch.Generator().manual_seed(100 + i)) for i in range(2)])
    outs = neck(x)
    # rank r's loss is the mean over ITS scene; the step's gradient is the mean over ranks
    loss = 0.5 * sum(o[0].square().mean() + o[1].square().mean() for o in outs)
    loss.backward()
    want = torch.cat([p.grad.reshape(-1) for p in neck.parameters()])
    stats = torch.cat([b.reshape(-1).float() for n, b in neck.named_buffers() if "running" in n])
    for o, a, b in zip(outs, r0["outs"], r1["outs"]):
        assert torch.allclose(o[0:1], a, rtol=1e-4, atol=1e-5) and torch.allclose(o[1:2], b, rtol=1e-4, atol=1e-5)
    assert torch.allclose(r0["stats"], stats, rtol=1e-5, atol=1e-6) and torch.equal(r0["stats"], r1["stats"])
    assert torch.allclose(r0["grads"], want, rtol=2e-4, atol=1e-6)
    assert torch.equal(r0["grads"], r1["grads"])


def test_shard_scenes_pads_by_repetition():
    from sgcdet_amd.dist import shard_scenes
    for n, world in [(1, 4), (2, 8), (3, 8), (5, 2), (8, 8), (9, 4)]:
        shards = [shard_scenes(n, r, world) for r in range(world)]
        assert len({len(s) for s in shards}) == 1 and len(shards[0]) == -(-n // world)
        assert set(i for s in shards for i in s) == set(range(n))
    assert shard_scenes(0, 0, 4) == []
