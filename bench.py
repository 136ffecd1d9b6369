#!/usr/bin/env python3
"""Hot-path benchmark: scenes/sec of the SGCDet view transformation on MI355X.

One "step" = one BATCH of scenes (--scenes-per-step, default one per stream = 4; the JSON
line says how many in config.scenes_per_step_per_gpu and also carries ms_per_scene) through
the path BASELINE.json names: FPN maps + depth distributions (already resident in HBM) ->
AdaptiveSparseHead (geometry/context-aware aggregation, coarse-to-fine refinement) ->
FastIndoorImVoxelNeck -> ImVoxelHeadV2 head tensors.  `value` = scenes / s over all ranks.
(Until the middle of round 4 a step was ONE scene: ms_per_step of BENCH_r01..r03 is per
scene, from round 4 on per batch -- compare ms_per_scene.)
Default workload = BASELINE.json configs[1]: 40 views x 256 ch, 40x40x16 voxels.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Multi-GPU: scenes are independent (batch size is 1 scene per GPU in the reference too), so
ranks shard scenes with NO data-path collective; scaling is weak (one scene per rank per
step).  Rank 0 prints ONE JSON line; ``roofline`` is for the deformable-gather kernel at the
finest level, timed with HIP events on its launch stream (with the default whole-scene hipGraph
replay the events bracket the same kernel in an eager pass over the same scenes right after the
timed region -- ``roofline.measured`` says which), ``roofline_mfma`` for the 90-GFLOP convolution, ``path_roofline`` for the whole path against its composite floor,
``cpu_baseline`` is the CPU oracle (a port: the reference has no CPU implementation of this path)
on a bounded sample on the host cores, ``self_check`` re-runs the scenes-in-flight configuration
against serial eager launches bit for bit.
"""
import argparse
import json
import os
import sys
import time

# Scenes in flight replay on separate HIP streams; ROCm maps streams onto GPU_MAX_HW_QUEUES hardware queues
# (default 4) round-robin, and torch's own streams take slots too -- two of the bench's streams then share a queue
# and their graphs serialise (measured: 3 streams 322 scenes/s at 4 queues, 380 at 8; 2 queues: 260).  The variable
# is read when the HIP runtime initialises, hence before torch is imported; an explicit setting wins.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
MFMA_SUSTAINED_TFLOPS = 1882.0   # dense bf16 MFMA rate the chip holds at its power limit on random operands (profiles/r06_mfma_power.txt:
                                 # 1882 TFLOP/s at 1.87 GHz / 1.34 kW on the round-6 tree's box; round 4 read 1847 at 1.89 GHz / 1.31 kW)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100,
                    help="timed steps; a step is one batch of --scenes-per-step scenes (default 100 x 4 scenes = 0.8 s at "
                         "config 2: the clocks need about a second of load to settle)")
    ap.add_argument("--warmup", type=int, default=25)
    ap.add_argument("--scenes-per-step", type=int, default=0,
                    help="scenes in the batch one step processes (default: one per stream, i.e. --streams; 1 without streams)")
    ap.add_argument("--prewarm", type=float, default=1.0,
                    help="seconds of untimed scenes during setup, before the --warmup steps (clock settle; 0 = none)")
    ap.add_argument("--workload", default="cfg2_scannet")
    ap.add_argument("--views", type=int, default=None, help="override the number of views")
    ap.add_argument("--img", default=None,
                    help="HxW of the resized input images.  Default: 256x320 for cfg2_scannet (the size BASELINE.json's "
                         "north star quotes); `config` = the reference config's own 239x320 (ScanNet) / 240x320 (ARKit), "
                         "which is also the default of the other workloads")
    ap.add_argument("--conv-mode", default="bf16x3", choices=["bf16x3", "f32", "bf16", "fp16"],
                    help="convolution / Linear arithmetic: 3-way bf16 split on the bf16 MFMA (fp32-faithful to ~1e-5, "
                         "default, the headline), exact fp32 products on the fp32 MFMA, or -- opt-in reduced precision of "
                         "BASELINE.json configs #2 / #5, their own lines, never the headline -- plain bf16 or fp16 products "
                         "(operands rounded to bfloat16 / IEEE half, fp32 accumulate: 1/3 of the matrix work)")
    ap.add_argument("--streams", type=int, default=4,
                    help="scenes in flight per GPU: consecutive steps alternate over this many HIP streams so the host "
                         "syncs / launch gaps of one scene overlap the kernels of the other")
    ap.add_argument("--graph", default="scene", choices=["scene", "tail", "none"],
                    help="scene: one hipGraph replay per scene (device-side pair counts, no host read-backs); "
                         "tail: eager view transform with a host read-back per level + hipGraph replay of neck/head; "
                         "none: everything launched eagerly")
    ap.add_argument("--no-graph", action="store_true", help="same as --graph none")
    ap.add_argument("--input-layout", default="nchw", choices=["nchw", "nhwc"],
                    help="memory layout of the FPN / depth maps handed to the path: nchw = the reference's producer "
                         "(default, what the metric is quoted on); nhwc = channels-last producer contract "
                         "(SURVEY.md 8 f-1): consumed in place, no transpose pass")
    ap.add_argument("--storage", default="f32", choices=["f32", "bf16"],
                    help="value-map storage of the tiled gather: f32 (parity mode, the headline) or bf16 (opt-in storage mode of "
                         "BASELINE.json configs #2/#5: bf16 value map, fp32 accumulate / outputs; its own line, never the headline)")
    ap.add_argument("--masked-tail", action="store_true",
                    help="output-masked finest decoder tail + head convolutions (north star's 'sparse 3D convolution over the "
                         "occupancy-masked voxels'): head tensors are then defined only where the head's valid pyramid is 1")
    ap.add_argument("--occupancy", default="predicted", choices=["predicted", "clustered"],
                    help="which scores the refined levels' top-k ranks: the predicted occupancy (the reference's behaviour, the "
                         "headline) or a seeded SURFACE-CLUSTERED override (sgcdet_amd.scene.clustered_occupancy; SURVEY.md 8d "
                         "allows a controlled mask): the workload on which --masked-tail can skip bricks.  Its own line, never "
                         "the headline")
    ap.add_argument("--winograd", default="auto", choices=["auto", "off"],
                    help="auto (default): the wide 3x3x3 stride-1 layers run through the Winograd F(2,3)-along-z form (2/3 of their "
                         "multiply-adds; <= 2e-5 of the tensor scale from the direct form, tested); off: every layer on the direct kernel")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--sustain", type=float, default=2.0,
                    help="seconds of the extra `sustained` leg (same configuration, back to back; 0 = skip)")
    ap.add_argument("--no-strict-fp32", action="store_true", help="skip the short `strict_fp32` leg (--conv-mode f32)")
    ap.add_argument("--breakdown", action="store_true", help="print a per-kernel HIP-event breakdown to stderr")
    ap.add_argument("--eager-geometry", default="as-timed", choices=["as-timed", "latency"],
                    help="launch geometry of the eager one-scene-at-a-time pass behind the `roofline` objects and --breakdown: "
                         "as-timed = the geometry of the timed region (with scenes in flight: the throughput geometry, which sizes "
                         "the layers with few voxels for CU-time, not for their own latency), latency = every kernel alone at its "
                         "latency-optimal split (what a per-layer table of kernel-alone times should be read with)")
    return ap.parse_args()


def build_path(w, device):
    import sgcdet_amd.plugin  # noqa: F401
    from sgcdet_amd.mmcv_lite import build_detector
    from sgcdet_amd.scene import model_config
    torch.manual_seed(0)
    det = build_detector(model_config(w)).eval()
    gen = torch.Generator().manual_seed(1)
    with torch.no_grad():   # random-init weights; perturb the zero-initialised offset/attention Linears
        for n, p in det.voxel_head.named_parameters():
            p.add_(torch.randn(p.shape, generator=gen) * 0.02)
    return det.to(device)


def algorithmic_bytes(n_views, hw, C, D, M, P, pairs, s=4):
    """compulsory traffic of the deformable gather with perfect on-chip
    reuse = projected value map + depth map + per-pair raw offsets/logits + per-pair output."""
    return n_views * hw * C * s + n_views * hw * D * s + pairs * (M * P * 4 * 4) + pairs * C * s


# entry points whose launches the eager pass brackets with events: the gathers (HBM roofline) and every GEMM-shaped launch
# (MFMA roofline); together they are the work the whole-path floor of `path_roofline` is made of
PATH_KERNELS = {"sgc_pairs_deform_gather", "sgc_pairs_deform_gather_tiled", "sgc_pairs_geometry_sample", "sgc_conv3d_cl_bf16x3",
                "sgc_conv3d_cl_bf16x3_masked", "sgc_conv3d_cl_bf16x3_act", "sgc_conv3d_winograd_z_bf16x3", "sgc_conv3d_cl_f32", "sgc_linear_rows_bf16x3",
                "sgc_linear_rows_zrow_bf16x3", "sgc_linear_rows_headmajor_bf16x3", "sgc_pairs_geometry_linear_bf16x3",
                "sgc_linear_rows_blockdiag_bf16x3"}


def usable_cores():
    """Cores this process may actually run on: the affinity mask capped by the cgroup CPU quota.  os.cpu_count() reports
    the machine (256 on the GPU hosts) even when the container is limited to a few cores, and an OpenMP team of 256
    spinning threads on such a quota is 10-1000x slower than one thread (measured in round 2)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    quota = None
    try:                                                 # cgroup v2
        q, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q) / float(period)
    except (OSError, ValueError):
        try:                                             # cgroup v1
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / period
        except (OSError, ValueError):
            pass
    if quota is not None:
        n = min(n, max(1, int(quota)))
    return max(1, n)


def cpu_baseline(w, n_views, seed):
    """The CPU oracle (a port: the reference has no CPU implementation of this path) timed on the host cores on a
    bounded sample of scenes of the same workload: all cores (~12 s), then one thread (one scene), plus the per-stage
    split SURVEY.md 8(d) asks for.  OpenMP threads are set through omp_set_num_threads in the oracle library (torch has
    already initialised its own pool by now; OMP_NUM_THREADS in os.environ would come too late)."""
    import ctypes
    import oracle
    from oracle.ref_path import RefPath
    from sgcdet_amd.scene import make_scene, model_config
    from sgcdet_amd.mmcv_lite import build_detector
    import sgcdet_amd.plugin  # noqa: F401
    oracle.build()
    all_cores = int(os.environ.get("SGC_CPU_THREADS", usable_cores()))
    try:
        gomp = ctypes.CDLL("libgomp.so.1")
    except OSError:
        gomp = None

    # torch's CPU ops (MHA over views, slot scatter, LayerNorm ...) get SLOWER beyond a few dozen threads on the 256-core
    # GPU hosts (measured: view pooling 1.1 s on 1 thread, 26.7 s on 256): torch intra-op threads are capped at 32, the
    # oracle's own OpenMP kernels use every core
    torch_cap = int(os.environ.get("SGC_CPU_TORCH_THREADS", 32))

    def set_threads(n):
        torch.set_num_threads(max(1, min(n, torch_cap)))
        if gomp is not None:
            gomp.omp_set_num_threads(n)

    torch.manual_seed(0)
    det = build_detector(model_config(w)).eval()
    feats, dpt, meta = make_scene(n_views, w["embed_dims"], kind=w["kind"], seed=seed)
    cfg = dict(embed_dims=w["embed_dims"], n_voxels_list=w["n_voxels_list"], voxel_size_list=w["voxel_size_list"],
               topk_list=w["topk_list"], dbound=(0.2, 5.0), num_heads=8, num_points=4,
               head="scannet" if w["head"].startswith("ScanNet") else "sunrgbd", n_classes=w["n_classes"], nms_pre=1000)
    sd = dict(det.voxel_head.state_dict())
    sd.update({"neck." + k: v for k, v in det.neck_3d.state_dict().items()})
    sd.update({"head." + k: v for k, v in det.bbox_head.state_dict().items()})
    rp = RefPath(sd, cfg, omp=True)
    import torch.nn.functional as F
    dpts = [dpt, F.interpolate(dpt, scale_factor=(1, 0.5, 0.5), mode="nearest"),
            F.interpolate(dpt, scale_factor=(1, 0.25, 0.25), mode="nearest")]

    def one_scene():
        vol, valid, occ = rp.adaptive_sparse_head(feats, meta, dpts)
        with rp._t("neck"):
            f3 = rp.neck(vol, prefix="neck.")
        with rp._t("head"):
            rp.head(f3, prefix="head.")

    def run(threads, budget_s, max_scenes):
        set_threads(threads)
        one_scene()                                     # warm-up (thread pools, oneDNN primitive caches)
        rp.timing = {}
        n, t0 = 0, time.perf_counter()
        while n < max_scenes and (n < 1 or time.perf_counter() - t0 < budget_s):
            one_scene()
            n += 1
        dt = time.perf_counter() - t0
        stages, rp.timing = rp.timing, None
        return n, dt, {k: round(v / n * 1e3, 1) for k, v in sorted(stages.items(), key=lambda kv: -kv[1])}

    def record(threads, n, dt, stages):
        return dict(value=n / dt, unit="scenes/sec", cores=threads, kind="port",
                    sample=f"{n} scene(s) of {w['name']} ({n_views} views) after 1 warm-up, CPU oracle (OpenMP C kernels "
                           f"on {threads} thread(s) + torch-CPU ops on {min(threads, torch_cap)}), {dt:.1f} s",
                    stages_ms_per_scene=stages)

    runs = [record(all_cores, *run(all_cores, 12.0, 64))]
    if all_cores > 1 and not os.environ.get("SGC_CPU_SKIP_1T"):
        runs.append(record(1, *run(1, 0.0, 1)))         # one scene on one thread (seconds at config 2)
    # the headline CPU figure is the FASTER configuration (more threads is not always faster for this path: the
    # torch-CPU ops over the [views, voxels] slots scale badly); the other run is kept beside it
    runs.sort(key=lambda r: -r["value"])
    out = dict(runs[0], host_cpus=os.cpu_count(), usable_cores=usable_cores())
    if len(runs) > 1:
        out["one_thread" if runs[1]["cores"] == 1 else "all_cores"] = {k: v for k, v in runs[1].items() if k != "kind"}
    set_threads(all_cores)
    return out


def launch_ranks(args):
    """``python bench.py --gpus N`` without a launcher: start N ranks (one per GPU) as CHILD processes through
    ``torch.distributed.run`` and exit with their code.  Runs before anything in this process has touched the GPU
    (``torch.cuda.device_count()`` does not initialise it on this image; a process that has initialised HIP must never
    exec or fork GPU work).  Mirrors the reference's launch, DDP over all visible GPUs (main.py:64-70)."""
    import socket
    import subprocess
    have = torch.cuda.device_count()
    if have < args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} requested but only {have} GPU(s) are visible -- refusing to report "
                 f"a {args.gpus}-GPU number from fewer ranks")
    with socket.socket() as sk:                     # a free rendezvous port on the loopback interface
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # the host driver only supports dmabuf IPC (RCCL needs it)
    sys.exit(subprocess.run(cmd, env=env).returncode)


def main():
    args = parse()
    if args.gpus < 1:
        sys.exit("bench.py: --gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        launch_ranks(args)                          # never returns
    env_world = int(os.environ.get("WORLD_SIZE", "1"))
    if env_world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={env_world} rank(s); "
                 "the line would misreport n_gpus")
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU (the product path has no CPU fallback)")
    from sgcdet_amd import dist as sgc_dist
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    rank, world, _ = sgc_dist.init_from_env(backend="nccl", device=device)   # "nccl" is RCCL on ROCm
    dist = None
    if world > 1:
        import torch.distributed as dist
        world = dist.get_world_size()              # the ranks RCCL actually joined: what `n_gpus` reports
    if world != args.gpus:
        sys.exit(f"bench.py: {world} rank(s) joined the process group, --gpus says {args.gpus}")

    from sgcdet_amd.scene import make_scene, workload
    from sgcdet_amd import ext
    from sgcdet_amd.plugin.conv_plan import set_conv_mode
    set_conv_mode(args.conv_mode)
    if args.winograd == "off":
        from sgcdet_amd.plugin import conv_plan as _cp
        _cp.set_winograd_z(False)
    w = workload(args.workload)
    n_views = args.views or w["n_views"]
    det = build_path(w, device)
    if args.no_graph:
        args.graph = "none"
    if args.graph == "scene" and args.conv_mode == "f32":
        args.graph = "tail"                      # the device-count GEMM entry point exists for the bf16x3 path only
    det.masked_tail = os.environ.get("SGC_MASKED_TAIL", "0") == "1" or args.masked_tail
    if args.occupancy == "clustered":
        from sgcdet_amd.scene import clustered_occupancy
        det.voxel_head.occupancy_override = clustered_occupancy(w["n_voxels_list"], seed=0, device=device)
    from sgcdet_amd.plugin import voxformer as _vf
    _vf.TILED_GATHER["storage"] = args.storage
    det.use_graph = args.graph != "none"
    det.scene_graph = args.graph == "scene"
    if args.img is None and args.workload.startswith("cfg2_scannet"):
        args.img = "256x320"          # BASELINE.json's north star: "40 views x 256x320 x 256ch -> 40x40x16 voxels"
    img_hw = tuple(int(v) for v in args.img.lower().split("x")) if args.img not in (None, "config") else None
    # a few distinct scenes per rank, resident in HBM before the timed region
    n_scenes = 3
    if args.graph == "scene":
        # every (scene buffers, stream) pair owns a graph whose pair-list buffers are sized for the worst case
        # (N * Nq pairs x 3C floats live at once): keep the total under ~1/3 of the 288 GB
        nq = max(w["topk_list"] + [w["n_voxels_list"][0][0] * w["n_voxels_list"][0][1] * w["n_voxels_list"][0][2]])
        per_graph = n_views * nq * 3 * w["embed_dims"] * 4 * 1.5
        while n_scenes * args.streams * per_graph > 96e9 and (n_scenes > 1 or args.streams > 1):
            if n_scenes > 2 or args.streams == 1:
                n_scenes -= 1
            else:
                args.streams -= 1
    scenes = []
    for s in range(n_scenes):
        feats, dpt, meta = make_scene(n_views, w["embed_dims"], kind=w["kind"], seed=1000 * rank + s, device=device,
                                      img_hw=img_hw)
        if args.input_layout == "nhwc":     # same logical [1,N,C,H,W] tensors, channels-last in memory
            cl = lambda t: t[0].contiguous(memory_format=torch.channels_last).unsqueeze(0)   # noqa: E731
            feats, dpt = [cl(f) for f in feats], cl(dpt)
        scenes.append((feats, dpt, [meta]))

    ops = ext.ops()                 # honours the development knobs of SGC_TUNE="tile_nw=8,tile_hg=2" (variant selection only)
    from sgcdet_amd.plugin.conv_plan import set_throughput_mode
    set_throughput_mode(args.streams > 1)      # launch geometry for scenes in flight (conv_plan.set_throughput_mode)

    streams = [torch.cuda.Stream(device=device) for _ in range(args.streams)] if args.streams > 1 else None

    def step(i):
        feats, dpt, metas = scenes[i % n_scenes]
        with torch.no_grad():
            if streams is None:
                return det.forward_features(feats, metas, dpt)
            with torch.cuda.stream(streams[i % len(streams)]):
                return det.forward_features(feats, metas, dpt)

    if det.scene_graph:
        # set-up, not a step: capture the scene graph of every (input buffers, stream) combination the loop will
        # visit, so that no capture (~100 ms, like a compile) lands in the warm-up or the timed region
        n_combo = n_scenes * max(1, args.streams)
        det.scene_graph_capacity = max(det.scene_graph_capacity, n_combo)
        for i in range(n_combo):
            step(i)
        torch.cuda.synchronize()
    if args.prewarm > 0:                 # setup, like the captures above: the clocks settle within about a second of load
        tp = time.perf_counter()
        i = 0
        while time.perf_counter() - tp < args.prewarm:
            for _ in range(8):
                step(i)
                i += 1
            torch.cuda.synchronize()
    spp = args.scenes_per_step if args.scenes_per_step > 0 else max(1, args.streams)   # scenes per step: one per stream
    n_timed = args.steps * spp
    for i in range(args.warmup * spp):
        step(i)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    lib_calls_per_scene = None
    # Kernel times for the `roofline` objects never come from the timed region: inside a replayed graph a kernel cannot carry
    # host-side events, and with several scenes in flight an event bracket on one stream also holds the time the kernel waits
    # for CUs the other scenes occupy (round 3's strict-fp32 line read 855 us for a 92-us gather that way).  They are taken in a
    # separate eager pass, one scene at a time, right after the timed region.
    t0 = time.perf_counter()
    for i in range(n_timed):              # args.steps batches of spp scenes, batch k = scenes k * spp .. k * spp + spp - 1
        step(i)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    log, ops.event_log = ops.event_log, None
    elapsed = sgc_dist.max_over_ranks(elapsed, device=device)
    # ---- sustained leg (untimed for `value`): the same configuration for >= 2 s, so that DVFS shows (a 50 ms burst of
    #      bf16 MFMA work runs at boost clocks) ----
    sustained = None
    if args.sustain > 0:
        n_sus = max(n_timed, int(args.sustain / max(elapsed / n_timed, 1e-6)) + 1)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        ts = time.perf_counter()
        for i in range(n_sus):
            step(i)
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        el_s = sgc_dist.max_over_ranks(time.perf_counter() - ts, device=device)
        sustained = dict(value=round(world * n_sus / el_s, 3), unit="scenes/sec", scenes=n_sus, seconds=round(el_s, 2),
                         ms_per_scene=round(el_s / n_sus * 1e3, 3))
    # same scenes, same kernels launched eagerly right after the timed region, one scene at a time: the deformable
    # gather is bracketed by HIP events on its launch stream (its pair count comes back to the host here)
    scene_graph_was, det.scene_graph = det.scene_graph, False
    tail_graph, det.use_graph = det.use_graph, False        # events cannot be recorded inside a capture / replay
    if args.eager_geometry == "latency":
        set_throughput_mode(False, keep_winograd=True)   # the splits of the latency geometry, the SAME convolution forms as the timed graphs
    ops.event_log = []
    ops.event_names = None if args.breakdown else PATH_KERNELS
    calls0 = ops.n_calls
    with torch.no_grad():
        for i in range(max(6, min(n_timed, 20))):         # one scene at a time: the kernel on its own
            feats, dpt, metas = scenes[i % n_scenes]
            det.forward_features(feats, metas, dpt)
            torch.cuda.synchronize()
    log, ops.event_log = ops.event_log, None
    lib_calls_per_scene = (ops.n_calls - calls0) / max(6, min(n_timed, 20))
    det.scene_graph, det.use_graph = scene_graph_was, tail_graph
    if args.eager_geometry == "latency":
        set_throughput_mode(args.streams > 1, keep_winograd=True)   # back to the geometry the graphs were captured with (the self check compares bits)
    roofline_pass = ("HIP events on the launch stream in an eager pass over the same scenes, one scene at a time, right "
                     "after the timed region (the timed region replays hipGraphs, which cannot carry events, with several "
                     "scenes in flight; inside it the kernel shares the chip with the other scenes and runs 0-3 % longer, see "
                     "profiles/r02_kernels_from_trace.json)")

    # ---- self check (untimed): the scenes-in-flight configuration reproduces the serial, graph-free results ----
    self_check = None
    if rank == 0:
        graph_was = det.use_graph, det.scene_graph
        det.use_graph = det.scene_graph = False
        with torch.no_grad():
            serial = []
            for feats, dpt, metas in scenes:
                r = det.forward_features(feats, metas, dpt)
                serial.append((r["volume"].clone(), r["occ"].clone(),
                               [t.clone() for t in r["centerness"] + r["bbox_pred"] + r["cls_score"]]))
        torch.cuda.synchronize()
        # the same scenes with every convolution on the DIRECT kernel: what the Winograd-z layers cost in accuracy end to end (per layer
        # <= 2e-5 of the tensor scale against the oracle's direct convolution, tested; here through neck + head on this run's scenes)
        wino_diff = None
        from sgcdet_amd.plugin import conv_plan as _cpw
        if _cpw.WINOGRAD_Z is not False and args.conv_mode != "f32":
            was = _cpw.WINOGRAD_Z
            _cpw.set_winograd_z(False)
            worst = torch.zeros((), device=device)
            with torch.no_grad():
                for (feats, dpt, metas), ref in zip(scenes, serial):
                    r = det.forward_features(feats, metas, dpt)
                    for a_, b_ in zip(r["centerness"] + r["bbox_pred"] + r["cls_score"], ref[2]):
                        worst = torch.maximum(worst, (a_ - b_).abs().max() / b_.abs().max().clamp(min=1.0))
            _cpw.set_winograd_z(was)
            wino_diff = float(worst)
        det.use_graph, det.scene_graph = graph_was
        # every run is compared on its own stream, right behind the replay that produced it (no host sync, nothing
        # kept): per run one device-side flag and the largest deviations; SGC_SELF_CHECK_RUNS lengthens the check
        n_runs = int(os.environ.get("SGC_SELF_CHECK_RUNS", 10 * n_scenes))
        flags = torch.zeros(n_runs, dtype=torch.int32, device=device)
        devs = torch.zeros((n_runs, 2), dtype=torch.float32, device=device)
        if streams:                                     # the fills above ran on the default stream
            for st in streams:
                st.wait_stream(torch.cuda.current_stream())
        for i in range(n_runs):
            r = step(i)
            ref = serial[i % n_scenes]
            with torch.cuda.stream(streams[i % len(streams)] if streams else torch.cuda.current_stream()):
                heads = r["centerness"] + r["bbox_pred"] + r["cls_score"]
                ne = (r["volume"] != ref[0]).any() | (r["occ"] != ref[1]).any()
                hrel = torch.zeros((), device=device)
                for a_, b_ in zip(heads, ref[2]):
                    ne = ne | (a_ != b_).any()
                    hrel = torch.maximum(hrel, (a_ - b_).abs().max() / b_.abs().max().clamp(min=1.0))
                flags[i] = ne.to(torch.int32)
                devs[i, 0] = (r["volume"] - ref[0]).abs().max()
                devs[i, 1] = hrel
        torch.cuda.synchronize()
        bad = torch.nonzero(flags).reshape(-1).tolist()
        self_check = dict(scene_runs=n_runs, mismatching=len(bad), max_abs_diff=float(devs[:, 0].max()),
                          head_max_rel_diff=float(devs[:, 1].max()),
                          compared="volume, occupancy and all head tensors bit-exact (elementwise ==) vs serial eager launches",
                          winograd_vs_direct_head_max_rel_diff=wino_diff,
                          winograd_vs_direct_note=("largest |head tensor (as timed, Winograd-z layers) - head tensor (every layer on the direct kernel)| "
                                                   "/ max(1, |direct|) over this run's scenes, serial eager launches" if wino_diff is not None else None))
        if os.environ.get("SGC_BENCH_DEBUG"):
            print("self-check mismatching runs:", bad, file=sys.stderr)

    # ---- roofline of the dominant hand-written kernel: finest-level deformable gather ----
    per_kernel = {}
    for name, meta, e0, e1 in log:
        per_kernel.setdefault(name, []).append((e0.elapsed_time(e1) * 1e-3, meta))
    roofline = None
    dg = per_kernel.get("sgc_pairs_deform_gather", []) + per_kernel.get("sgc_pairs_deform_gather_tiled", [])
    tiled_finest = False
    if dg:
        big = max(m["n_pairs"] * m["C"] + m["N"] * m["H"] * m["W"] * m["C"] for _, m in dg)
        finest = [(t, m) for t, m in dg if m["n_pairs"] * m["C"] + m["N"] * m["H"] * m["W"] * m["C"] >= 0.5 * big]
        tiled_finest = "bin" in finest[0][1]
        t_avg = sum(t for t, _ in finest) / len(finest)
        # bf16 storage mode: the value map and the depth maps cost 2 bytes per element, everything else stays fp32
        b_avg = sum(algorithmic_bytes(m["N"], m["H"] * m["W"], m["C"], m["D"], m["M"], m["P"], m["n_pairs"])
                    - (4 - m.get("value_bytes", 4)) * m["N"] * m["H"] * m["W"] * m["C"]
                    - (4 - m.get("depth_bytes", 4)) * m["N"] * m["H"] * m["W"] * m["D"]
                    for _, m in finest) / len(finest)
        achieved = b_avg / t_avg / 1e9
        # HBM bytes per launch from the PMC passes of the same command (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in
        # separate runs, FETCH_SIZE doubled per the gfx950 correction); collected offline, committed under profiles/
        traffic, traffic_source = None, None
        pmc_name = "none"
        if args.img == "256x320":        # the newest committed PMC pass of this kernel
            pmc_name = next((n for n in ("r05_gather_tile_pmc_hbm.json", "r04_gather_tile_pmc_hbm.json")
                             if os.path.exists(os.path.join(ROOT, "profiles", n))), "none") if tiled_finest else "r01_gather_pmc_v5.json"
        pmc_file = os.path.join(ROOT, "profiles", pmc_name)
        if args.workload == "cfg2_scannet" and args.views in (None, 40) and args.storage == "f32" and os.path.exists(pmc_file):
            traffic = json.load(open(pmc_file)).get("hbm_bytes_per_launch")
            traffic_source = (f"offline PMC (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes of this command, "
                              f"FETCH_SIZE doubled per the gfx950 correction), profiles/{pmc_name}; NOT measured in this run")
        roofline = dict(bound="hbm", achieved=round(achieved, 1), peak=HBM_PEAK_GBS, unit="GB/s",
                        frac=round(achieved / HBM_PEAK_GBS, 4), traffic=traffic, traffic_source=traffic_source,
                        kernel=("sgc::dfa3d_fwd_tile_kernel (LDS-staged head windows, finest level)" if tiled_finest else
                                "sgc::dfa3d_fwd_wave_kernel<kPairsDeform, P=4, M=8> (finest level)"),
                        measured=roofline_pass, eager_geometry=args.eager_geometry,
                        avg_launch_us=round(t_avg * 1e6, 1), algorithmic_bytes=int(b_avg), launches=len(finest))
    # ---- second object: the MFMA-bound kernel that takes the most time, the largest 3x3x3 convolution of the neck ----
    roofline_mfma = None
    cv = per_kernel.get("sgc_conv3d_cl_bf16x3" if args.conv_mode != "f32" else "sgc_conv3d_cl_f32", [])
    if args.conv_mode != "f32":
        cv = cv + per_kernel.get("sgc_conv3d_winograd_z_bf16x3", [])      # the same layers through the Winograd-z form (2/3 of the MACs issued)
    cv = [(t, m) for t, m in cv if m.get("taps") == 27]
    if cv:
        flops = lambda m: 2.0 * m["taps"] * m["Cin"] * m["Cout"] * m["OV"]      # noqa: E731
        top = max(flops(m) for _, m in cv)
        big_c = [(t, m) for t, m in cv if flops(m) >= 0.99 * top]
        t_c = sum(t for t, _ in big_c) / len(big_c)
        bf = args.conv_mode != "f32"
        nprod = 3 if args.conv_mode == "bf16x3" else 1
        peak = 2500.0 if bf else 157.0      # dense bf16 / fp32 MFMA peak (MI355X_MICROARCH.md)
        m0 = big_c[0][1]
        mac_frac = m0.get("mac_frac", 1.0)                 # Winograd F(2,3) along z issues 18 of the 27 tap-GEMMs
        issued = top * mac_frac * nprod / t_c / 1e12
        roofline_mfma = dict(bound="mfma", achieved=round(issued, 1), peak=peak, unit="TFLOP/s", frac=round(issued / peak, 4),
                             kernel=(("sgc_conv3d_winograd_z_bf16x3 = sgc::conv3d_halo_bf16x3_kernel (2-D form on 4 transform-domain positions, the "
                                      "input transform fused into its staging) + the output-transform launch, both launches inside the bracket" if mac_frac < 1.0 else
                                      "sgc::conv3d_halo_bf16x3_kernel") if bf else "sgc::conv3d_igemm_f32_kernel") +
                                    f" ({m0['Cin']}->{m0['Cout']} ch, 3x3x3, {m0['OV']} voxels)",
                             mac_frac_issued=round(mac_frac, 4),
                             algorithmic_gflop=round(top / 1e9, 1), fp32_equivalent_tflops=round(top / t_c / 1e12, 1),
                             fp32_equivalent_note=("algorithmic fp32 FLOPs / time; it can exceed the 157 TFLOP/s fp32 MFMA peak "
                                                   "because the work runs as three bf16 products on the bf16 pipe" if bf else None),
                             avg_launch_us=round(t_c * 1e6, 1), launches=len(big_c),
                             note=("achieved = MFMA work actually issued (three bf16 products per fp32 multiply-add: lo*hi + "
                                   "hi*lo + hi*hi); fp32_equivalent_tflops = algorithmic FLOPs of the fp32 convolution / time")
                             if nprod == 3 else f"one {'fp16' if args.conv_mode == 'fp16' else 'bf16'} product per multiply-add (opt-in mode)" if bf
                             else "exact fp32 products on v_mfma_f32_32x32x2_f32")
    # ---- the whole path against its own roofline (north star: "as a fraction of the HBM roofline"): compulsory gather bytes at
    #      8 TB/s + the matrix work actually issued at the dense MFMA peak, per scene, over the measured time per scene ----
    path_roofline = None
    if per_kernel:
        n_e = max(6, min(n_timed, 20))
        # (a block-diagonal Linear run as a dense GEMM -- the per-voxel V projection of the projected-query attention -- counts with
        #  its structurally non-zero fraction `useful`: the zero blocks are launch geometry, not work the path has to do)
        # (ConvTranspose3d(2, 2): every OUTPUT voxel receives exactly one of the 8 taps -- 2 Cin Cout per output voxel, SURVEY.md 8d;
        #  `flop_taps` = 1.  Rounds 3 - 4 multiplied by the 8 taps of the weight tensor and overstated the path's GEMM work by 70 GF
        #  (11 %) at config 2: floors / `path_roofline.frac` of those rounds are that much too high.)
        ftaps = lambda m: m.get("flop_taps", m.get("taps") or 1)      # noqa: E731
        gemm = sum(2.0 * ftaps(m) * m["Cin"] * m["Cout"] * (m.get("OV") or m["V"]) * m.get("useful", 1.0)
                   for items in per_kernel.values() for _, m in items if "Cin" in m) / n_e
        # multiply-adds actually issued: the Winograd-z layers issue 2/3 of their algorithmic count (`mac_frac`)
        gemm_issued = sum(2.0 * ftaps(m) * m["Cin"] * m["Cout"] * (m.get("OV") or m["V"]) * m.get("useful", 1.0) * m.get("mac_frac", 1.0)
                          for items in per_kernel.values() for _, m in items if "Cin" in m) / n_e
        gbytes = 0.0
        for name, items in per_kernel.items():
            for _, m in items:
                if name in ("sgc_pairs_deform_gather", "sgc_pairs_deform_gather_tiled"):
                    gbytes += algorithmic_bytes(m["N"], m["H"] * m["W"], m["C"], m["D"], m["M"], m["P"], m["n_pairs"]) \
                        - (4 - m.get("value_bytes", 4)) * m["N"] * m["H"] * m["W"] * m["C"] \
                        - (4 - m.get("depth_bytes", 4)) * m["N"] * m["H"] * m["W"] * m["D"]
                elif name == "sgc_pairs_geometry_sample":     # raw map + depth map + (u, v, z) per pair + output (B_gs)
                    gbytes += m["N"] * m["H"] * m["W"] * (m["C"] + m["D"]) * 4 + m["n_pairs"] * (12 + m["C"] * 4)
                elif name == "sgc_pairs_geometry_linear_bf16x3":   # the sample fused with its Linear: B_gs WITHOUT the sampled rows, which
                    gbytes += m["N"] * m["H"] * m["W"] * (m["C"] + m["D"]) * 4 + m["n_pairs"] * 12      # no longer exist (the floor shrinks)
        gbytes /= n_e
        bf = args.conv_mode != "f32"
        issued = gemm_issued * (3 if args.conv_mode == "bf16x3" else 1)
        floor_ms = (issued / ((2500.0 if bf else 157.0) * 1e12) + gbytes / (HBM_PEAK_GBS * 1e9)) * 1e3
        # the same floor with the matrix pipe priced at what it SUSTAINS under the package power limit on non-zero data
        # (1882 TFLOP/s at 1.87 GHz / 1.34 kW, tools/probe/mfma_power.hip, profiles/r06_mfma_power.txt; the 2.5 PFLOP/s of
        # the guide is reached with all-zero operands only: 2477 TFLOP/s at 2.39 GHz / 0.86 kW) -- reported beside `frac`
        floor_pl_ms = (issued / ((MFMA_SUSTAINED_TFLOPS if bf else 157.0) * 1e12) + gbytes / (HBM_PEAK_GBS * 1e9)) * 1e3
        path_roofline = dict(gather_mb_algorithmic=round(gbytes / 1e6, 1), gemm_gflop_algorithmic=round(gemm / 1e9, 1),
                             gemm_gflop_issued=round(issued / 1e9, 1), floor_ms_per_scene=round(floor_ms, 3),
                             frac=round(floor_ms / (elapsed / n_timed * 1e3) * world, 4),
                             frac_of_power_limited_floor=round(floor_pl_ms / (elapsed / n_timed * 1e3) * world, 4),
                             mfma_sustained_tflops=MFMA_SUSTAINED_TFLOPS,
                             note="floor = compulsory bytes of the geometry-sample + deformable gathers at 8 TB/s + the MFMA work "
                                  "issued by every convolution / Linear (3 bf16 products per fp32 multiply-add) at the dense MFMA "
                                  "peak, per scene; frac = floor / measured time per scene (timed region)")
    if args.breakdown and rank == 0:
        for name, items in sorted(per_kernel.items(), key=lambda kv: -sum(t for t, _ in kv[1])):
            tot = sum(t for t, _ in items)
            print(f"  {name:32s} {len(items):5d} launches  {tot * 1e3 / max(6, min(n_timed, 20)):8.3f} ms/scene", file=sys.stderr)
            shapes = {}
            for t, m in items:                                   # per-shape split of the GEMM / convolution entry points
                if "Cin" in m:
                    key = (m.get("V"), m.get("Cin"), m.get("Cout"), m.get("taps"), m.get("OV"))
                    a = shapes.setdefault(key, [0, 0.0])
                    a[0] += 1
                    a[1] += t
            n_eager = max(6, min(n_timed, 20))
            for key, (cnt, tt) in sorted(shapes.items(), key=lambda kv: -kv[1][1])[:24]:
                V, Cin, Cout, taps, OV = key
                gf = 2.0 * (1 if taps == 8 else (taps or 1)) * Cin * Cout * (OV or V) / 1e9       # taps = 8: ConvTranspose3d(2, 2), one tap per output voxel
                print(f"      V={V:7d} Cin={Cin:5d} Cout={Cout:5d} taps={taps:3d} -> {cnt / n_eager:4.1f}/scene {tt / cnt * 1e6:8.1f} us  "
                      f"{gf:7.1f} GF  {gf / (tt / cnt) / 1e3:7.1f} TF-equiv", file=sys.stderr)

    # ---- strict-fp32 leg: the same scenes with exact fp32 products on the fp32 MFMA (--conv-mode f32), so that the
    #      bf16x3 headline never hides what IEEE-fp32 arithmetic costs ----
    strict = None
    if args.conv_mode == "bf16x3" and not args.no_strict_fp32:
        set_conv_mode("f32")
        sg, ug = det.scene_graph, det.use_graph
        det.scene_graph, det.use_graph = False, True        # eager view transform + hipGraph tail (no device-count GEMMs in f32)
        n_f32 = max(4, min(n_timed, 10))
        for i in range(max(3, n_scenes * max(1, args.streams))):     # every (scene, stream) tail graph of this mode captured before the clock starts
            step(i)
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        tf = time.perf_counter()
        for i in range(n_f32):
            step(i)
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        el_f = sgc_dist.max_over_ranks(time.perf_counter() - tf, device=device)
        # this leg's step() is ONE scene per call: report it per scene, in the unit of the top-level ms_per_scene
        strict = dict(value=round(world * n_f32 / el_f, 3), unit="scenes/sec", scenes=n_f32, ms_per_scene=round(el_f / n_f32 * 1e3, 3),
                      dtype="f32 (exact fp32 products on v_mfma_f32_32x32x2_f32 for every convolution and Linear; LDS-tiled gather behind a permuting copy of the value map)")
        set_conv_mode(args.conv_mode)
        det.scene_graph, det.use_graph = sg, ug

    tail_stats = None
    if rank == 0 and (det.masked_tail or args.occupancy != "predicted"):
        # what the output masks of the decoder tail keep alive on the last scene (tools/valid_stats.py prints more): fractions of
        # voxels in valid / dilate(valid) / dilate^2(valid) and of the halo kernel's 256-voxel bricks holding one such voxel
        import torch.nn.functional as F
        with torch.no_grad():
            v = det.forward_features(*[scenes[0][k] for k in (0, 2, 1)])["valid"].float()
            d1 = F.max_pool3d(v, 3, 1, 1)
            d2 = F.max_pool3d(d1, 3, 1, 1)
            bshape = (8, 8, 4) if (v.shape[-1] >= 16 and v.shape[-3] % 8 == 0 and v.shape[-2] % 8 == 0) else (4, 4, 16)
            live = lambda m: round(float(F.max_pool3d(m, bshape, bshape, ceil_mode=True).mean()), 3)   # noqa: E731
            tail_stats = dict(voxels=dict(valid=round(float(v.mean()), 3), dilate1=round(float(d1.mean()), 3), dilate2=round(float(d2.mean()), 3)),
                              live_bricks={"brick": "x".join(map(str, bshape)), "head_conv(valid)": live(v),
                                           "out_block_0(dilate1)": live(d1), "up_block_1(dilate2)": live(d2)})
    from sgcdet_amd.plugin import conv_plan
    winograd_on = conv_plan.WINOGRAD_Z is not False and args.conv_mode != "f32"
    if rank == 0:
        out = {
            "metric": "scenes/sec (40-view ScanNet volume) at 1/2/4/8 MI355X; mAP@0.25 parity",
            "value": round(world * n_timed / elapsed, 3),
            "unit": "scenes/sec",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "ms_per_scene": round(elapsed / n_timed * 1e3 / world, 4),
            "step_definition": f"one step = a batch of {spp} scenes per GPU (one per stream); ms_per_step is per batch, ms_per_scene per scene",
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": ("bf16 (opt-in reduced precision: every convolution / Linear with operands rounded to bfloat16, one MFMA product, "
                      "fp32 accumulate" + ("; value map and depth maps of the gather stored in bfloat16" if args.storage == "bf16" else "") +
                      "; NOT parity-exact, not the headline)" if args.conv_mode == "bf16" else
                      "fp16 (opt-in reduced precision: every convolution / Linear with operands rounded to IEEE half -- saturated at "
                      "+-65504 --, one v_mfma_f32_32x32x16_f16 product, fp32 accumulate"
                      + ("; value map and depth maps of the gather stored in bfloat16" if args.storage == "bf16" else "") +
                      "; NOT parity-exact, not the headline)" if args.conv_mode == "fp16" else
                      "bf16 storage (value map and depth maps of the deformable gather in bfloat16, fp32 accumulate and outputs; opt-in, not parity-exact)"
                      if args.storage == "bf16" else
                      "f32" if args.conv_mode == "f32" else
                      "f32 (neck/head conv: 3xbf16-split MFMA, fp32 accumulate, ~1e-5 of fp32" +
                      ("; wide 3x3x3 stride-1 layers through a Winograd F(2,3) transform along z: <= 2e-5 of the tensor scale per layer against "
                       "the direct fp32 convolution" if winograd_on else "") + ")"),
            "data": "synthetic",
            "config": {"workload": f"{w['name']}: {n_views} views x {w['embed_dims']} ch, images "
                                   f"{'x'.join(str(v) for v in scenes[0][2][0]['img_shape'][:2])}, FPN maps "
                                   f"{'/'.join(f'{f.shape[-2]}x{f.shape[-1]}' for f in scenes[0][0][:3])}, "
                                   f"D=12, voxels {'x'.join(map(str, w['n_voxels_list'][-1]))}, top-k {w['topk_list']}, "
                                   f"neck 3-scale -> {w['head']}",
                       "occupancy": ("predicted" if args.occupancy == "predicted" else
                                     "clustered override (seeded floor + two walls + a box shell; top-k ranks the distance to them)"),
                       "masked_tail": bool(det.masked_tail), "masked_tail_stats": tail_stats,
                       "winograd_z": (f"auto: 3x3x3 stride-1 layers with >= {conv_plan.WINOGRAD_Z_MIN_CH} input channels and a z extent that is a "
                                      "multiple of 8 (sgc_conv3d_winograd_z_bf16x3)" if winograd_on else "off"),
                       "input_layout": args.input_layout, "scenes_per_step_per_gpu": spp, "scenes_in_flight_per_gpu": args.streams, "prewarm_s": args.prewarm,
                       "launch_geometry": ("throughput (row GEMMs on half the CUs; fewer reduction splits in the layers with few voxels; "
                                            "conv_plan.set_throughput_mode)"
                                           if args.streams > 1 else "latency"),
                       "launch": {"scene": "one hipGraph replay per scene (device-side pair counts, no host read-back)",
                                  "tail": "eager view transform (one host read-back per level) + hipGraph replay of neck/head",
                                  "none": "eager"}[args.graph],
                       "sharding": "scenes across ranks, no collective",
                       "library_calls_per_scene": lib_calls_per_scene},
            "roofline": roofline,
            "roofline_mfma": roofline_mfma,
            "path_roofline": path_roofline,
            "strict_fp32": strict,
            "sustained": sustained,
            "self_check": self_check,
        }
        if not args.no_cpu_baseline and world == 1:
            try:
                out["cpu_baseline"] = cpu_baseline(w, n_views, seed=0)
            except Exception as e:  # the baseline is reported, never required for the GPU number
                out["cpu_baseline"] = {"value": None, "unit": "scenes/sec", "cores": os.cpu_count(), "kind": "port",
                                       "sample": f"failed: {e}"}
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()
    if rank == 0 and self_check and self_check["mismatching"] > 0:
        sys.exit(f"bench.py: {self_check['mismatching']} of {self_check['scene_runs']} overlapped scene runs differ from "
                 "the serial eager result -- the number above is invalid")


if __name__ == "__main__":
    main()
