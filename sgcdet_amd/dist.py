"""Multi-GPU plumbing for the hot path: one process per GPU, ``torch.distributed`` over RCCL.

The path shards by SCENE (the reference hard-wires one scene per GPU per step,
mmdet3d_plugin/models/im2voxel/AdaptiveSparseHead.py:45, and shards scenes with a
DistributedSampler, LightningTools/dataset_dm.py:30-37):

* inference / benchmark: scenes are dealt round-robin to ranks, NO data-path collective; the only
  collectives are the barrier and the MAX-reduction of the elapsed time;
* training: one exchange per step -- the gradient all-reduce (reference: torch DDP under Lightning,
  main.py:64-70).  ``BucketedGradAllReduce`` does it with few, large flat buckets: xGMI is
  point-to-point (7 links x ~153 GB/s per GPU), so ring collectives are per-link bound and per-call
  latency matters more than on a switched fabric; ~80 M hot-path parameters = 320 MB fp32 go out in
  <= 5 asynchronous all-reduces that overlap with the remaining backward.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None, device=None):
    """Reads RANK / WORLD_SIZE / LOCAL_RANK / MASTER_ADDR / MASTER_PORT (torch.distributed.run).
    Returns (rank, world, local_rank); a no-op for world == 1."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"   # "nccl" IS RCCL on ROCm
        kwargs = {}
        if backend == "nccl" and device is not None:
            kwargs["device_id"] = device
        dist.init_process_group(backend, rank=rank, world_size=world, **kwargs)
    return rank, world, local_rank


def shard_scenes(n_scenes, rank, world, pad=True):
    """Indices of the scenes rank ``rank`` processes: strided like DistributedSampler(shuffle=False);
    with ``pad`` every rank gets ceil(n/world) items (wrap-around) so ranks stay in lock-step."""
    idx = list(range(n_scenes))
    if pad and n_scenes and n_scenes % world:
        # pad by repetition (DistributedSampler does the same): with fewer scenes than the pad length a single
        # wrap-around slice would leave some ranks empty and the next barrier would hang
        total = -(-n_scenes // world) * world
        idx = (idx * (-(-total // n_scenes)))[:total]
    return idx[rank::world]


def max_over_ranks(value, device="cpu"):
    """MAX-reduce a python float over all ranks (bench timing contract)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


class BucketedGradAllReduce:
    """Averages ``.grad`` of the given parameters over all ranks with flat buckets.

    Parameters without a gradient (the reference runs DDP with ``find_unused_parameters=True``:
    e.g. heads whose loss is switched off) contribute zeros, exactly like DDP."""

    def __init__(self, params, bucket_bytes=64 << 20):
        self.params = [p for p in params if p.requires_grad]
        self.buckets, cur, size = [], [], 0
        for p in self.params:
            nbytes = p.numel() * p.element_size()
            if cur and size + nbytes > bucket_bytes:
                self.buckets.append(cur)
                cur, size = [], 0
            cur.append(p)
            size += nbytes
        if cur:
            self.buckets.append(cur)

    def __call__(self):
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
            return
        world = dist.get_world_size()
        pending = []
        for bucket in self.buckets:
            flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in bucket])
            pending.append((bucket, flat, dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True)))
        for bucket, flat, work in pending:
            work.wait()
            flat.div_(world)
            off = 0
            for p in bucket:
                n = p.numel()
                g = flat[off:off + n].view_as(p)
                if p.grad is None:
                    p.grad = g.clone()
                else:
                    p.grad.copy_(g)
                off += n


class OverlappedGradAllReduce:
    """The same averaging, started DURING the backward pass (one gradient exchange per step over
    RCCL, overlapped with the neck's backward).  Parameters are bucketed in reverse registration order -- the order
    their gradients become ready in; a post-accumulate hook counts a bucket's gradients in and launches its
    asynchronous all-reduce as soon as the last one lands AND every earlier bucket is out (strict index order on
    every rank, as DDP: ranks whose used-parameter sets differ still issue identical collective sequences), so the
    exchange of the head / neck gradients runs under the view transform's backward.  ``finish()`` (after ``loss.backward()``) flushes buckets whose parameters got no
    gradient this step (zeros, as DDP with ``find_unused_parameters=True``), waits, and writes the means back.

        sync = OverlappedGradAllReduce(model.parameters())
        for batch in data:
            loss(model, batch).backward()
            sync.finish()
            optimizer.step(); optimizer.zero_grad(set_to_none=True)
    """

    def __init__(self, params, bucket_bytes=64 << 20):
        self.params = [p for p in params if p.requires_grad][::-1]
        self.buckets, cur, size = [], [], 0
        for p in self.params:
            nbytes = p.numel() * p.element_size()
            if cur and size + nbytes > bucket_bytes:
                self.buckets.append(cur)
                cur, size = [], 0
            cur.append(p)
            size += nbytes
        if cur:
            self.buckets.append(cur)
        self._bucket_of = {id(p): b for b, bucket in enumerate(self.buckets) for p in bucket}
        self._seen = [set() for _ in self.buckets]   # ids of the parameters whose gradient has landed this step
        self._next = 0                               # buckets [0, _next) have been launched (always in index order)
        self._pending = {}
        self._hooks = [p.register_post_accumulate_grad_hook(self._on_grad) for p in self.params]

    def _active(self):
        return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1

    def _launch(self, b):
        bucket = self.buckets[b]
        flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in bucket])
        self._pending[b] = (flat, dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True))

    def _launch_ready_prefix(self):
        """Collectives are matched across ranks by ISSUE ORDER, and which parameters receive a gradient may differ
        from rank to rank (``find_unused_parameters=True`` semantics): bucket b goes out only once buckets 0 .. b-1
        have, so every rank issues the same sequence 0, 1, 2, ... whatever order its buckets fill up in."""
        while self._next < len(self.buckets) and len(self._seen[self._next]) == len(self.buckets[self._next]):
            self._launch(self._next)
            self._next += 1

    def _on_grad(self, p):
        if not self._active():
            return
        b = self._bucket_of[id(p)]
        if id(p) in self._seen[b]:
            # a bucket that is already on the wire was reduced from the FIRST micro-batch; finish() would overwrite
            # the accumulated gradient with it.  Gradient accumulation needs finish() once per backward().
            raise RuntimeError("OverlappedGradAllReduce: a second backward() reached a parameter before finish(); "
                               "call finish() after every backward() (or use BucketedGradAllReduce once after the "
                               "accumulation steps)")
        self._seen[b].add(id(p))
        self._launch_ready_prefix()

    def finish(self):
        if not self._active():
            return
        world = dist.get_world_size()
        for b in range(self._next, len(self.buckets)):     # the rest, in index order (buckets with unused parameters)
            self._launch(b)
        self._next = len(self.buckets)
        for b in range(len(self.buckets)):
            flat, work = self._pending[b]
            work.wait()
            flat.div_(world)
            off = 0
            for p in self.buckets[b]:
                n = p.numel()
                g = flat[off:off + n].view_as(p)
                if p.grad is None:
                    p.grad = g.clone()
                else:
                    p.grad.copy_(g)
                off += n
        self._pending.clear()
        self._seen = [set() for _ in self.buckets]
        self._next = 0

    def remove(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []


# ------------------------------------------------------------------------------------------------------------------
# SyncBN.  The reference trains with ``sync_batchnorm=True`` (main.py:81: Lightning converts the 15 BatchNorm3d layers of
# FastIndoorImVoxelNeck to torch.nn.SyncBatchNorm) at ONE scene per GPU, so the batch statistics of every layer are
# taken over the scenes of all ranks.  torch's SyncBatchNorm only accepts CUDA tensors and does three collectives per
# layer; this one is plain tensor arithmetic around one all_gather in forward ([mean, biased var, count] per rank, 2C+1
# floats, merged with the parallel-variance formula) and one all_reduce in backward ([sum dy, sum dy*(x-mean)], 2C
# floats), works on any backend (RCCL on the GPU, gloo in the CPU tests), and is bit-compatible with nn.BatchNorm3d
# in a single process.
class _SyncBatchNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, eps, momentum, group):
        C = x.shape[1]
        dims = [0] + list(range(2, x.dim()))
        n_local = x.numel() // C
        xf = x.float()
        mean_l = xf.mean(dims)
        var_l = xf.var(dims, unbiased=False)
        packed = torch.cat([mean_l, var_l, xf.new_tensor([float(n_local)])])
        world = dist.get_world_size(group) if _dist_on() else 1
        if world > 1:
            gathered = [torch.empty_like(packed) for _ in range(world)]
            dist.all_gather(gathered, packed, group=group)
            allp = torch.stack(gathered)
        else:
            allp = packed[None]
        means, vars_, counts = allp[:, :C], allp[:, C:2 * C], allp[:, 2 * C:]
        n_tot = counts.sum()
        mean = (means * counts).sum(0) / n_tot
        var = ((vars_ + (means - mean) ** 2) * counts).sum(0) / n_tot            # biased, over all ranks
        invstd = torch.rsqrt(var + eps)
        if running_mean is not None:
            with torch.no_grad():
                unbiased = var * (n_tot / (n_tot - 1).clamp(min=1.0))
                running_mean.mul_(1 - momentum).add_(mean.to(running_mean.dtype), alpha=momentum)
                running_var.mul_(1 - momentum).add_(unbiased.to(running_var.dtype), alpha=momentum)
        shape = [1, C] + [1] * (x.dim() - 2)
        xhat = (xf - mean.view(shape)) * invstd.view(shape)
        y = xhat
        if weight is not None:                                  # affine=False: no scale / shift, no parameter gradients
            y = y * weight.float().view(shape)
        if bias is not None:
            y = y + bias.float().view(shape)
        ctx.affine = (weight is not None, bias is not None)
        ctx.save_for_backward(xhat, weight if weight is not None else invstd.new_ones(C), invstd, n_tot)
        ctx.group = group
        return y.to(x.dtype)

    @staticmethod
    def backward(ctx, gy):
        xhat, weight, invstd, n_tot = ctx.saved_tensors
        C = gy.shape[1]
        dims = [0] + list(range(2, gy.dim()))
        shape = [1, C] + [1] * (gy.dim() - 2)
        gyf = gy.float()
        sum_dy = gyf.sum(dims)
        sum_dy_xhat = (gyf * xhat).sum(dims)
        packed = torch.cat([sum_dy, sum_dy_xhat])
        if _dist_on() and dist.get_world_size(ctx.group) > 1:
            dist.all_reduce(packed, op=dist.ReduceOp.SUM, group=ctx.group)
        g_sum, g_dot = packed[:C], packed[C:]
        w = weight.float()
        gx = (gyf - (g_sum / n_tot).view(shape) - xhat * (g_dot / n_tot).view(shape)) * (invstd * w).view(shape)
        # weight / bias gradients are LOCAL sums: the gradient all-reduce of the step averages them like every other parameter
        has_w, has_b = ctx.affine
        return (gx.to(gy.dtype), sum_dy_xhat.to(weight.dtype) if has_w else None, sum_dy.to(weight.dtype) if has_b else None,
                None, None, None, None, None)


def _dist_on():
    return dist.is_available() and dist.is_initialized()


class SyncBatchNorm3d(torch.nn.BatchNorm3d):
    """nn.BatchNorm3d whose TRAINING statistics are taken over the batch elements of all ranks (see above).  Same
    parameters / buffers / state-dict keys; eval mode is the parent's."""

    process_group = None

    def forward(self, x):
        if not self.training:
            return super().forward(x)
        if self.momentum is None:
            raise NotImplementedError("SyncBatchNorm3d: cumulative moving average (momentum=None) is not used by SGCDet")
        if self.num_batches_tracked is not None:
            self.num_batches_tracked.add_(1)
        return _SyncBatchNormFn.apply(x, self.weight, self.bias, self.running_mean, self.running_var, self.eps,
                                      self.momentum, self.process_group)


def convert_sync_batchnorm(module, process_group=None):
    """Replaces every nn.BatchNorm3d under ``module`` (the neck's 15 layers) by ``SyncBatchNorm3d`` sharing its parameters
    and buffers -- what ``pl.Trainer(sync_batchnorm=True)`` does in the reference (main.py:81).  Returns ``module``."""
    for name, child in list(module.named_children()):
        if isinstance(child, torch.nn.BatchNorm3d) and not isinstance(child, SyncBatchNorm3d):
            sbn = SyncBatchNorm3d(child.num_features, child.eps, child.momentum, child.affine, child.track_running_stats)
            sbn.weight, sbn.bias = child.weight, child.bias
            sbn.running_mean, sbn.running_var, sbn.num_batches_tracked = child.running_mean, child.running_var, child.num_batches_tracked
            sbn.process_group = process_group
            sbn.train(child.training)
            setattr(module, name, sbn)
        else:
            convert_sync_batchnorm(child, process_group)
    return module
