#!/usr/bin/env python3
"""bench.py -- headline benchmark: knn_points B=32 N=M=65536 K=16 D=3 fp32 per GPU
(BASELINE.json configs[1]; metric "Mpoint-pairs/s (+ %HBM roofline) knn_points ...").

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path (one `knn_points_idx` call through the C ABI,
outputs allocated inside the timed region as the op does) over one resident batch of
32 synthetic clouds per GPU.  With N GPUs every rank owns its own 32 clouds (batch
sharding, no data-path collective -- clouds are independent), so
scaling is "weak" and `value` is the whole-job aggregate.

One JSON line on rank 0.  `roofline` prices the dominant kernel (the KNN scan)
against the HBM roofline with ALGORITHMIC bytes (SURVEY.md section 8d:
4*D*(P1+P2) + P1*K*12 bytes per cloud = 452,984,832 B per launch) over the HIP-event
duration of the launch, measured live on the launch stream.  `cpu_baseline` is the
reference's own CPU kernel (oracle/_ref, kind "reference") or the C oracle (kind
"port") timed single-threaded on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

B, P, K, D = 32, 65536, 16, 3
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
VALU_LANE_OPS = 256 * 4 * 32 * 2.4e9  # 7.86e13 fp32 lane-ops/s


def log(msg):
    print(f"[bench] {msg}", file=sys.stderr, flush=True)


def make_clouds(first_cloud: int, n_clouds: int):
    from pytorch3d_pointops_amd import synth

    p1 = np.empty((n_clouds, P, D), np.float32)
    p2 = np.empty((n_clouds, P, D), np.float32)
    for b in range(n_clouds):
        g = first_cloud + b
        p1[b] = synth.uniform_f32(7001 + 10 * g, (P, D))  # cloud 0 == tests' pinned cfg2 cloud
        p2[b] = synth.uniform_f32(7002 + 10 * g, (P, D))
    return p1, p2


def cpu_baseline(p1, p2, sample_queries: int):
    """Reference CPU kernel (or its C port) on `sample_queries` queries of cloud 0 vs all of p2[0]."""
    from oracle.oracle import Oracle, load_ref

    ora = load_ref() or Oracle()
    q = np.ascontiguousarray(p1[:1, :sample_queries])
    r = np.ascontiguousarray(p2[:1])
    l1 = np.array([sample_queries])
    l2 = np.array([P])
    t0 = time.perf_counter()
    idx, _ = ora.knn_points_idx(q, r, l1, l2, 2, K)
    dt = time.perf_counter() - t0
    pairs = float(sample_queries) * P
    return {
        "value": pairs / dt / 1e6,
        "unit": "Mpoint-pairs/s",
        "cores": 1,
        "kind": ora.kind,
        "sample": f"{sample_queries} queries of cloud 0 x all {P} points of p2[0], K={K} "
                  f"({pairs:.3g} pairs, {dt:.1f} s, single thread; host has {os.cpu_count()} cpus)",
    }, idx


def main():
    # stdout carries exactly ONE line, the JSON result: libraries that write banners to fd 1 (RCCL prints
    # its version block there when the first communicator is created) go to stderr for the whole run
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--cpu-sample-queries", type=int, default=24576,
                    help="queries of the CPU-baseline sample (0 disables); ~10-20 s of CPU work")
    ap.add_argument("--version", type=int, default=-1, help="knn kernel family (-1 auto)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch multi-GPU runs with torch.distributed.run (one process per GPU)")
    assert torch.cuda.is_available(), "bench.py needs a GPU (there is no CPU fallback)"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # POINTOPS_BENCH_FORCE_DIST=1 exercises the RCCL path with a single rank (1-GPU box rehearsal)
    use_dist = world > 1 or os.environ.get("POINTOPS_BENCH_FORCE_DIST") == "1"
    if use_dist:
        import torch.distributed as dist

        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=dev)  # nccl == RCCL on ROCm

    from pytorch3d_pointops_amd import _C

    p1_h, p2_h = make_clouds(rank * B, B)
    p1 = torch.from_numpy(p1_h).to(dev)
    p2 = torch.from_numpy(p2_h).to(dev)
    l1 = torch.full((B,), P, dtype=torch.int64, device=dev)
    l2 = torch.full((B,), P, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    log(f"rank {rank}/{world}: inputs resident ({p1.numel() * 8 / 1e6:.1f} MB)")

    def step():
        return _C.knn_points_idx(p1, p2, l1, l2, 2, K, args.version)

    for _ in range(args.warmup):
        out = step()
    torch.cuda.synchronize()

    def barrier():
        if use_dist:
            dist.barrier()

    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in range(args.steps):
        ev[s][0].record()  # current stream == the stream the C ABI launches on
        out = step()
        ev[s][1].record()
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    kern_ms = [a.elapsed_time(b) for a, b in ev]
    avg_kern_s = float(np.mean(kern_ms)) / 1e3

    pairs_per_step_rank = float(B) * P * P
    total_pairs = pairs_per_step_rank * args.steps * world
    value = total_pairs / elapsed / 1e6
    algo_bytes = B * (4 * D * (P + P) + P * K * 12)  # 452,984,832 B per launch
    achieved = algo_bytes / avg_kern_s / 1e9

    # HBM-side bytes per step from the committed rocprofv3 PMC passes of this same command
    # (tools/pmc_traffic.py; separate FETCH_SIZE / WRITE_SIZE passes, gfx950 x2 read correction)
    traffic, traffic_src = None, None
    tj = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(tj) and args.version == -1:
        try:
            traffic = float(json.load(open(tj))["traffic_bytes_per_step"])
            traffic_src = "profiles/pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of bench.py)"
        except Exception:  # noqa: BLE001
            traffic = None

    result = {
        "metric": "Mpoint-pairs/s (+ %HBM roofline) knn_points B=32 N=65536 K=16 @1/2/4/8 GPU",
        "value": value,
        "unit": "Mpoint-pairs/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": "knn_points B=32 N=M=65536 K=16 D=3 fp32 per GPU (BASELINE.json configs[1]); "
                        "splitmix64 uniform [0,1)^3 clouds, p1 != p2, full lengths, norm=2",
            "clouds_per_gpu": B,
            "sharding": "batch-sharded clouds, one process per GPU, no data-path collective",
            "kernel_version": args.version,
        },
        "roofline": {
            "bound": "hbm",
            "achieved": achieved,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS,
            "traffic": traffic,
            "traffic_source": traffic_src,
            "kernel": "knn_points_idx = grid build (bbox, histogram, scan, counting sort) + knn_grid_lane_kernel "
                      "(dominant, ~75 % of the step) + exact fallbacks; `achieved` prices the WHOLE op, "
                      "HIP-event timed on the launch stream",
            "algorithmic_bytes_per_launch": algo_bytes,
            "avg_launch_ms": avg_kern_s * 1e3,
            "valu_frac_9ops_per_pair": pairs_per_step_rank * 9 / avg_kern_s / VALU_LANE_OPS,
        },
    }

    if rank == 0 and world == 1 and args.cpu_sample_queries > 0:
        log("timing the CPU baseline sample ...")
        cb, cpu_idx = cpu_baseline(p1_h, p2_h, args.cpu_sample_queries)
        result["cpu_baseline"] = cb
        # the checker doubles as a parity probe of the benchmarked output
        gpu_idx = out[0][0, : args.cpu_sample_queries].cpu().numpy()
        result["cpu_baseline"]["idx_equal_on_sample"] = bool(np.array_equal(gpu_idx, cpu_idx[0]))
    if use_dist:
        # Outside the timed region: the path's one real exchange -- chamfer's batch reduction over
        # clouds sharded across ranks = one RCCL all_gather of the per-cloud loss vectors.
        try:
            from pytorch3d_pointops_amd.sharded import sharded_chamfer_distance

            torch.cuda.synchronize()
            t1 = time.perf_counter()
            loss, _ = sharded_chamfer_distance(p1, p2, B * world)
            torch.cuda.synchronize()
            result["sharded_chamfer"] = {"clouds_total": B * world, "loss": float(loss),
                                         "first_call_ms": (time.perf_counter() - t1) * 1e3,
                                         "collective": "one RCCL all_gather of (B/G,) fp32 per-cloud losses"}
        except Exception as e:  # noqa: BLE001  (diagnostic only; never fails the bench line)
            result["sharded_chamfer"] = {"error": repr(e)[:200]}
    if rank == 0:
        sys.stdout.flush()
        os.write(result_fd, (json.dumps(result) + "\n").encode())
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
