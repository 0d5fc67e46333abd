#!/usr/bin/env python
"""bench.py -- frames/sec of the per-tracklet forward-and-match hot path on MI355X.

Workload (BASELINE.json configs[1], the config the metric is quoted on): MARS-shaped synthetic clips, seq_len 8,
32 tracklets per GPU per step (256 frames of 256x128), VMGN eval forward (ResNet50 two-branch + 2 graph layers +
attention pooling) in bf16, then the cosine distance of the step's embeddings against the resident gallery
(12 180 x 4096, row-sharded over the ranks). One "step" = forward + [RCCL all-gather of embeddings, N > 1] +
distance matrix against the rank's gallery shard. Inputs are resident in HBM before the timed region.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line (see the task contract): value = whole-job frames/s (all ranks), plus
  roofline     : the dominant kernel (implicit-GEMM conv, MFMA-bound), achieved = algorithmic flops per launch /
                 average launch duration measured with HIP events on the launch stream, vs the dense MFMA peak
  cpu_baseline : the CPU oracle (oracle/vmgn_oracle.py, torch CPU kernels) timed on this host on a bounded sample
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "agrl.pytorch_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK_TFLOPS = {"bf16": 2500.0, "fp32": 157.3, "bf16x3": 2500.0 / 3}  # dense MFMA peaks, MI355X_MICROARCH.md (split mode: 3 bf16 MFMAs per product)
PEAK_HBM_GBS = 8000.0
GALLERY_ROWS = 12180  # MARS gallery of the reference tree
FEATURE_DIM = 4096


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=32, help="tracklets per GPU per step")
    ap.add_argument("--seq-len", type=int, default=8)
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32", "bf16x3"])
    ap.add_argument("--metric", default="cosine", choices=["cosine", "euclidean"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="lower bound of CPU-baseline work")
    ap.add_argument("--profile-steps", type=int, default=3)
    ap.add_argument("--graph", action="store_true",
                    help="replay the forward from a captured HIP graph (host issue cost 1.2 ms -> 0.08 ms per step; GPU time "
                         "unchanged within 1.5 %%, tools/graph_probe.py)")
    return ap.parse_args()


def build_model(device, precision):
    from recipe import recipe_state_dict
    from torchreid import models
    m = models.init_model("vmgn", num_classes=625, loss={"xent", "htri"}, last_stride=1, num_split=4, num_gb=2,
                          num_scale=1, pyramid_part=True, use_pose=True, learn_graph=True, consistent_loss=False,
                          num_parts=3, bnneck=True)
    sd = recipe_state_dict(m.state_dict(), seed=0)
    m.load_state_dict(sd)
    m.eval()
    m.hip_precision = precision
    m.hip_static_weights = True
    return m.to(device), sd


def cpu_baseline(sd, S, metric, gallery_cpu, min_seconds):
    """The oracle on this host's cores: same step (forward + distmat against the full gallery), fewer tracklets."""
    from oracle import vmgn_oracle as O
    from recipe import synthetic_adj, synthetic_clips
    bs = 4
    x, adj = synthetic_clips(bs, S, seed=123), synthetic_adj(bs, S, seed=123)
    fn = O.cosine if metric == "cosine" else O.euclidean_squared
    with torch.no_grad():
        O.vmgn_eval(x[:1], adj[:1], sd)  # warm the thread pool / allocator
        frames, t0 = 0, time.time()
        while time.time() - t0 < min_seconds:
            emb = O.vmgn_eval(x, adj, sd)
            fn(emb, gallery_cpu)
            frames += bs * S
        dt = time.time() - t0
    return {"value": round(frames / dt, 2), "unit": "frames/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "%d frames (batches of %d tracklets x %d frames, fwd+GCN+%s distmat vs %d gallery rows) in %.1f s, "
                      "oracle/vmgn_oracle.py on torch CPU fp32" % (frames, bs, S, metric, gallery_cpu.size(0), dt)}


def main():
    args = parse()
    from torchreid import _hip, parallel
    from torchreid.metrics.distance import hip_distmat_device
    from torchreid import hip_ops as ops

    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    rank, world, local_rank = parallel.init_from_env()
    assert world == args.gpus, "launch with torch.distributed.run --nproc-per-node %d (WORLD_SIZE=%d)" % (args.gpus, world)
    device = torch.device("cuda", local_rank % torch.cuda.device_count())
    torch.cuda.set_device(device)
    _hip.lib()

    from recipe import synthetic_adj
    B, S = args.batch, args.seq_len
    model, sd = build_model(device, args.precision)
    gen = torch.Generator(device=device)
    gen.manual_seed(0xFF + rank)
    clips = torch.randn((B, S, 3, 256, 128), device=device, generator=gen)
    adj = synthetic_adj(B, S, seed=rank).to(device)
    g_all = torch.Generator().manual_seed(7)
    gallery_cpu = torch.randn((GALLERY_ROWS, FEATURE_DIM), generator=g_all)
    lo, hi = parallel.shard_bounds(GALLERY_ROWS, rank, world)
    lp = args.precision == "bf16"
    dt_g = torch.bfloat16 if lp else torch.float32
    # resident gallery shard, prepared once (normalised rows for cosine / norms for euclidean)
    g_shard = gallery_cpu[lo:hi].to(device)
    if args.metric == "cosine":
        g_op, g_norm = ops.row_l2_normalize(g_shard, True, dt_g), None
    else:
        g_norm = ops.row_sqnorm(g_shard)
        g_op = ops.row_l2_normalize(g_shard, False, dt_g) if lp else g_shard
    dist_out = torch.empty((B * world, hi - lo), dtype=torch.float32, device=device)

    def match(emb):
        q_all = parallel.all_gather_rows(emb)         # RCCL all-gather over xGMI when world > 1
        if args.metric == "cosine":
            q_op = ops.row_l2_normalize(q_all, True, dt_g)
            return ops.distmat(q_op, g_op, "cosine", out=dist_out)
        qn = ops.row_sqnorm(q_all)
        q_op = ops.row_l2_normalize(q_all, False, dt_g) if lp else q_all
        return ops.distmat(q_op, g_op, "euclidean", qn, g_norm, out=dist_out)

    main_stream = torch.cuda.current_stream(device)
    match_stream = torch.cuda.Stream(device=device) if world > 1 else None

    if args.graph:
        side = torch.cuda.Stream(device=device)
        side.wait_stream(main_stream)
        with torch.cuda.stream(side):
            model(clips, adj)                         # warm every lazy allocation / weight pack outside the capture
        main_stream.wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            emb_static = model(clips, adj)

        def forward(eager=False):
            if eager:                                 # the instrumented steps time the same kernels launch by launch
                return model(clips, adj)
            graph.replay()
            return emb_static.clone()                 # the match stage may still read it when the next replay starts
    else:
        def forward(eager=False):
            return model(clips, adj)

    def step(eager=False):
        emb = forward(eager)                          # (B, 4096) fp32
        if match_stream is None:
            return match(emb)
        # N > 1: the exchange + match of this batch run on their own HIP stream, so the collective (and the skew
        # between ranks it absorbs) overlaps the next batch's forward instead of stalling it; every step still does
        # the same work, and the closing synchronize() waits for both streams.
        ready = torch.cuda.Event()
        ready.record(main_stream)
        with torch.cuda.stream(match_stream):
            match_stream.wait_event(ready)
            emb.record_stream(match_stream)
            return match(emb)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()
    frames = B * S * world * args.steps
    result = {
        "metric": "frames/sec (fwd+GCN+distmat), MARS seq_len=8",
        "value": round(frames / elapsed, 1),
        "unit": "frames/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(1e3 * elapsed / args.steps, 3),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": args.precision,
        "data": "synthetic",
        "config": {"workload": "MARS seq_len=8 bs=32/GPU %s fwd+distmat (BASELINE configs[1]): 256x128 frames, VMGN "
                               "ResNet50x2-branch + 2 graph layers, %s distmat vs resident %d x 4096 gallery" %
                               (args.precision, args.metric, GALLERY_ROWS),
                   "global_batch": B * world, "seq_len": S, "frames_per_step": B * S * world,
                   "gallery_rows_per_gpu": hi - lo, "parallelism": "dp%d" % world, "hip_graph": bool(args.graph)},
    }

    # ---- live per-kernel timing (HIP events on the launch stream). Every rank runs the extra steps (they contain
    # the collective); only rank 0 records and reports.
    if rank == 0:
        _hip.PROFILE = []
    for _ in range(max(1, args.profile_steps)):
        step(eager=True)
    sync()
    if rank == 0:
        prof, _hip.PROFILE = _hip.PROFILE, None
        agg = {}
        dom = {"ms": 0.0, "launches": 0, "flops": 0.0}  # the dominant kernel: conv3x3_wide_kernel (see below)
        for name, s_ev, e_ev, tag in prof:
            a = agg.setdefault(name, {"ms": 0.0, "launches": 0, "flops": 0.0, "bytes": 0.0})
            ms = s_ev.elapsed_time(e_ev)
            a["ms"] += ms
            a["launches"] += 1
            if tag:
                a["flops"] += tag["flops"]
                a["bytes"] += tag["bytes"]
                c = tag.get("conv")
                # dispatch rule of agrl_conv2d_bn_act (csrc/igemm.hip): bf16 3x3 stride-1 convs with >= 256 input channels
                # on 16 x 8 maps go to conv3x3_wide_kernel -- the 3x3 convs of layers 3 and 4
                if lp and c and c[0] == 3 and c[1] == 1 and c[2] >= 256:
                    dom["ms"] += ms
                    dom["launches"] += 1
                    dom["flops"] += tag["flops"]
        kernels = {}
        for name, a in sorted(agg.items(), key=lambda kv: -kv[1]["ms"]):
            sec = a["ms"] * 1e-3
            kernels[name] = {"ms_per_step": round(a["ms"] / max(1, args.profile_steps), 4),
                             "launches_per_step": a["launches"] // max(1, args.profile_steps),
                             "avg_launch_us": round(1e3 * a["ms"] / a["launches"], 2)}
            if a["flops"]:
                kernels[name]["tflops"] = round(a["flops"] / sec / 1e12, 2)
                kernels[name]["gbs"] = round(a["bytes"] / sec / 1e9, 1)
        # the conv family: every conv launch (generic / persistent / wide implicit GEMM, 3x3 patch kernels, the fused
        # layer-1 block and layer-2 tail, the pool-fused last conv)
        a = {"ms": 0.0, "launches": 0, "flops": 0.0}
        for fam in ("agrl_conv2d_bn_act", "agrl_conv1x1_bn_act_pool", "agrl_bottleneck_tail", "agrl_bottleneck_block"):
            if fam in agg:
                for key in a:
                    a[key] += agg[fam][key]
        achieved = a["flops"] / (a["ms"] * 1e-3) / 1e12
        peak = PEAK_TFLOPS[args.precision]
        traffic = fam_traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic_r01.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath)).get(args.precision, {})
                fam_traffic = tj.get("igemm_bytes_per_launch")
                k3 = tj.get("other_kernels", {}).get("conv3x3_wide_kernel")
                if k3:
                    traffic = k3["fetch_bytes_per_launch"] + (k3["write_bytes_per_launch"] or 0.0)
            except Exception:
                traffic = fam_traffic = None
        family = {"bound": "mfma (layers 3-4) / hbm (layers 1-2)",
                  "kernel": "conv implicit-GEMM family (agrl_conv2d_bn_act + agrl_conv1x1_bn_act_pool + agrl_bottleneck_tail + agrl_bottleneck_block)",
                  "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(achieved / peak, 4),
                  "traffic": fam_traffic, "flops_per_launch": round(a["flops"] / a["launches"], 1),
                  "avg_launch_us": round(1e3 * a["ms"] / a["launches"], 2), "ms_per_step": round(a["ms"] / max(1, args.profile_steps), 4)}
        if dom["launches"]:
            # THE dominant kernel by time (profiles/r01_bench_kernel_stats.csv): conv3x3_wide_kernel<0>
            ach = dom["flops"] / (dom["ms"] * 1e-3) / 1e12
            result["roofline"] = {"bound": "mfma", "kernel": "conv3x3_wide_kernel<0> (3x3 stride-1 convs of layers 3-4, csrc/conv3x3_wide.hip)",
                                  "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4),
                                  "traffic": traffic, "flops_per_launch": round(dom["flops"] / dom["launches"], 1),
                                  "avg_launch_us": round(1e3 * dom["ms"] / dom["launches"], 2),
                                  "launches_per_step": dom["launches"] // max(1, args.profile_steps),
                                  "ms_per_step": round(dom["ms"] / max(1, args.profile_steps), 4)}
            result["roofline_conv_family"] = family
        else:  # fp32 / split modes: one generic kernel serves every conv
            result["roofline"] = family
        for name, label in (("agrl_graph_propagate", "gcn_message_pass"), ("agrl_distmat", "distmat")):
            if name in agg and agg[name]["bytes"]:
                gbs = agg[name]["bytes"] / (agg[name]["ms"] * 1e-3) / 1e9
                result["roofline_" + label] = {"bound": "hbm", "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                               "frac": round(gbs / PEAK_HBM_GBS, 4)}
        result["kernels"] = kernels
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(sd, S, args.metric, gallery_cpu, args.cpu_seconds)
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
