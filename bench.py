#!/usr/bin/env python3
"""bench.py -- headline benchmark: knn_points B=32 N=M=65536 K=16 D=3 fp32 per GPU
(BASELINE.json configs[1]; metric "Mpoint-pairs/s (+ %HBM roofline) knn_points ...").

    python bench.py                                   # 1 GPU
    python bench.py --gpus 4 --steps 20 --warmup 5    # spawns 4 ranks itself (one process per GPU, RCCL); fails fast
    python bench.py --gpus 8 --scaling strong         # BASELINE.json configs[4]: B=256 split over the ranks
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W      # the same ranks, external launcher

A "step" is one pass of the hot path (one `knn_points_idx` call through the C ABI,
outputs allocated inside the timed region as the op does) over one resident batch of
synthetic clouds.  Clouds are independent, so ranks own disjoint
clouds and the data path has no collective:
  --scaling weak   (default) every rank owns 32 clouds       -> per-GPU work fixed;
  --scaling strong 256 clouds in total, 256/N per rank       -> total work fixed.
`value` is the whole-job aggregate in BRUTE-FORCE-EQUIVALENT point pairs (sum_n len1*len2 per
step, the nominal all-pairs count of SURVEY.md section 8d -- the grid search evaluates only a
small fraction of them, exactly).

One JSON line on rank 0.  `roofline` prices the op against the HBM roofline with ALGORITHMIC
bytes (4*D*(P1+P2) + P1*K*12 bytes per cloud = 452,984,832 B per 32-cloud
launch) over the HIP-event duration of the launch, measured live on the launch stream;
`roofline.traffic` and `roofline.valu_issue` come from committed rocprofv3 counter passes of this
same command and are nulled when the kernel sources changed since they were taken.
`cpu_baseline` is the reference's own CPU kernel (oracle/_ref, kind "reference") or the C oracle
(kind "port") timed on a bounded sample of the same workload.
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

B_PER_GPU, B_STRONG_TOTAL, P, K, D = 32, 256, 65536, 16, 3
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
VALU_ISSUE_PEAK = 256 * 4 * 2.4e9 / 2.0  # wave64 VALU instructions/s: 1024 SIMD-32s, 2 cycles per instruction


def log(msg):
    print(f"[bench] {msg}", file=sys.stderr, flush=True)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak")
    ap.add_argument("--cpu-sample-queries", type=int, default=24576,
                    help="queries of the CPU-baseline sample (0 disables); ~10-20 s of CPU work")
    ap.add_argument("--cpu-all-cores", action="store_true",
                    help="also time the CPU kernel on every host core (query blocks over a thread pool)")
    ap.add_argument("--version", type=int, default=-1, help="knn kernel family (-1 auto)")
    ap.add_argument("--reuse-steps", type=int, default=10,
                    help="1-GPU runs: timed steps of the opt-in grid reuse after the headline region (0 disables)")
    ap.add_argument("--chamfer-steps", type=int, default=3,
                    help="multi-rank runs: timed sharded chamfer fwd+bwd steps after the KNN region (0 disables)")
    return ap.parse_args(argv)


# ---------------------------------------------------------------------------------------------
# launcher: `python bench.py --gpus N` with N > 1 and no torchrun environment spawns the N ranks
# ---------------------------------------------------------------------------------------------
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def visible_gpus() -> int:
    """GPUs this process would see, WITHOUT touching the HIP runtime (the parent of the ranks must stay free of it:
    children are spawned from here).  KFD topology nodes with SIMDs are GPUs; *_VISIBLE_DEVICES narrows them."""
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            return len([x for x in v.split(",") if x.strip() != ""])
    n = 0
    base = "/sys/class/kfd/kfd/topology/nodes"
    try:
        for node in os.listdir(base):
            try:
                props = open(os.path.join(base, node, "properties")).read()
            except OSError:
                continue
            for line in props.splitlines():
                if line.startswith("simd_count") and int(line.split()[1]) > 0:
                    n += 1
    except OSError:
        return -1  # unknown: let the ranks find out
    return n


def launch_ranks(args, rank_cmd=None, wall_limit_s=None) -> int:
    """Parent of a self-launched multi-GPU run.  It never touches the GPU runtime, starts one fresh interpreter per
    rank, polls ALL of them, and fails fast: the first rank that exits non-zero (or the wall-clock limit) kills the
    others -- a rank that dies before or inside an RCCL barrier would otherwise leave rank 0 waiting forever.
    Relays rank 0's JSON line.  `rank_cmd` (tests) replaces the per-rank command line."""
    have = visible_gpus()
    if 0 <= have < args.gpus and rank_cmd is None:
        log(f"--gpus {args.gpus} needs {args.gpus} visible GPUs, this machine has {have}: not launching")
        return 2
    if wall_limit_s is None:
        wall_limit_s = float(os.environ.get("POINTOPS_BENCH_WALL_LIMIT_S", "1500"))
    env = dict(os.environ)
    env.update({"WORLD_SIZE": str(args.gpus), "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(_free_port()),
                "HSA_ENABLE_IPC_MODE_LEGACY": env.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")})
    cmd = rank_cmd or ([sys.executable, os.path.abspath(__file__)] + sys.argv[1:])
    import tempfile

    out0 = tempfile.TemporaryFile()  # rank 0's stdout (a pipe nobody reads while polling could fill up)
    procs = []
    for r in range(args.gpus):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen(cmd, env=e, stdout=out0 if r == 0 else subprocess.DEVNULL,
                                      start_new_session=True))
    t0 = time.monotonic()
    failed = None
    while True:
        rcs = [p.poll() for p in procs]
        bad = [r for r, rc in enumerate(rcs) if rc not in (None, 0)]
        if bad:
            failed = f"rank {bad[0]} exited with code {rcs[bad[0]]}"
            break
        if all(rc == 0 for rc in rcs):
            break
        if time.monotonic() - t0 > wall_limit_s:
            failed = f"wall-clock limit of {wall_limit_s:.0f} s reached"
            break
        time.sleep(0.05)
    if failed is not None:
        log(f"{failed}: stopping the other ranks")
        for p in procs:
            if p.poll() is None:
                try:
                    os.killpg(p.pid, 15)  # the rank's own session: exactly the processes this launcher started
                except OSError:
                    pass
        deadline = time.monotonic() + 5.0
        for p in procs:
            try:
                p.wait(timeout=max(0.1, deadline - time.monotonic()))
            except subprocess.TimeoutExpired:
                try:
                    os.killpg(p.pid, 9)
                except OSError:
                    pass
                p.wait()
        return 1
    out0.seek(0)
    sys.stdout.write(out0.read().decode())
    sys.stdout.flush()
    return 0


# ---------------------------------------------------------------------------------------------
# sources of the kernels that run in the headline step (knn_points_idx through the grid family and its fallbacks)
KNN_PATH_SOURCES = ("capi", "common", "debug", "grid", "knn.", "knn_common", "knn_grid", "knn_wide", "sort_net")


def kernel_source_digest():
    """sha256 over the HIP sources of the headline step without comments and whitespace: ties committed counter profiles to the code they
    were measured on (there is no .git on the GPU box)."""
    import glob

    import re

    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "pytorch3d_pointops_amd", "csrc", "*"))):
        if not os.path.basename(f).startswith(KNN_PATH_SOURCES):
            continue  # (the other operators' kernels do not run in the headline step)
        h.update(os.path.basename(f).encode())
        text = open(f, "r", errors="replace").read()
        # the CODE: comments and whitespace do not move a counter (no string literal of the sources holds "//")
        text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
        text = re.sub(r"//[^\n]*", " ", text)
        h.update(" ".join(text.split()).encode())
    return h.hexdigest()[:16]


def committed_counters(name, version):
    """profiles/<name> if it was measured on the current kernel sources (and the auto kernel family), else None."""
    path = os.path.join(ROOT, "profiles", name)
    if version != -1 or not os.path.exists(path):
        return None
    try:
        j = json.load(open(path))
    except Exception:  # noqa: BLE001
        return None
    if j.get("kernel_source_digest") != kernel_source_digest():
        return None
    return j


def make_clouds(first_cloud: int, n_clouds: int):
    import numpy as np

    from pytorch3d_pointops_amd import synth

    p1 = np.empty((n_clouds, P, D), np.float32)
    p2 = np.empty((n_clouds, P, D), np.float32)
    for b in range(n_clouds):
        g = first_cloud + b
        p1[b] = synth.uniform_f32(7001 + 10 * g, (P, D))  # cloud 0 == tests' pinned cfg2 cloud
        p2[b] = synth.uniform_f32(7002 + 10 * g, (P, D))
    return p1, p2


def cpu_baseline(p1, p2, sample_queries: int, threads: int = 1):
    """Reference CPU kernel (or its C port) on `sample_queries` queries of cloud 0 vs all of p2[0];
    `threads` > 1 splits the queries into blocks over a thread pool (the kernels release the GIL)."""
    import numpy as np

    from oracle.oracle import Oracle, load_ref

    ora = load_ref() or Oracle()
    r = np.ascontiguousarray(p2[:1])
    l2 = np.array([P])
    blocks = np.array_split(np.arange(sample_queries), threads)

    def run(ix):
        q = np.ascontiguousarray(p1[:1, ix[0]:ix[-1] + 1])
        return ora.knn_points_idx(q, r, np.array([len(ix)]), l2, 2, K)[0]

    t0 = time.perf_counter()
    if threads == 1:
        parts = [run(blocks[0])]
    else:
        from concurrent.futures import ThreadPoolExecutor

        with ThreadPoolExecutor(threads) as ex:
            parts = list(ex.map(run, blocks))
    dt = time.perf_counter() - t0
    idx = np.concatenate(parts, axis=1)
    pairs = float(sample_queries) * P
    return {
        "value": pairs / dt / 1e6,
        "unit": "Mpoint-pairs/s",
        "cores": threads,
        "kind": ora.kind,
        "sample": f"{sample_queries} queries of cloud 0 x all {P} points of p2[0], K={K} "
                  f"({pairs:.3g} pairs, {dt:.1f} s, {threads} thread(s); host has {os.cpu_count()} cpus)",
    }, idx


def main():
    args = parse_args()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args))

    # stdout carries exactly ONE line, the JSON result: libraries that write banners to fd 1 (RCCL prints
    # its version block there when the first communicator is created) go to stderr for the whole run
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)

    import numpy as np
    import torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} does not match WORLD_SIZE={world}")
    assert torch.cuda.is_available(), "bench.py needs a GPU (there is no CPU fallback)"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # POINTOPS_BENCH_FORCE_DIST=1 exercises the RCCL path with a single rank (1-GPU box rehearsal)
    use_dist = world > 1 or os.environ.get("POINTOPS_BENCH_FORCE_DIST") == "1"
    if use_dist:
        import torch.distributed as dist

        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        import datetime

        # nccl == RCCL on ROCm; a short timeout: a missing rank must become an error, not a hang
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev,
                                timeout=datetime.timedelta(seconds=int(os.environ.get("POINTOPS_BENCH_PG_TIMEOUT_S", "300"))))

    from pytorch3d_pointops_amd import _C
    from pytorch3d_pointops_amd.sharded import shard_bounds

    if args.scaling == "strong":
        first, last = shard_bounds(B_STRONG_TOTAL, world)[rank]
        total_clouds = B_STRONG_TOTAL
    else:
        first, last = rank * B_PER_GPU, (rank + 1) * B_PER_GPU
        total_clouds = B_PER_GPU * world
    nb = last - first
    p1_h, p2_h = make_clouds(first, nb)
    p1 = torch.from_numpy(p1_h).to(dev)
    p2 = torch.from_numpy(p2_h).to(dev)
    l1 = torch.full((nb,), P, dtype=torch.int64, device=dev)
    l2 = torch.full((nb,), P, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    log(f"rank {rank}/{world}: clouds [{first}, {last}) resident ({p1.numel() * 8 / 1e6:.1f} MB)")

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

    total_pairs = float(total_clouds) * P * P * args.steps
    value = total_pairs / elapsed / 1e6
    algo_bytes = nb * (4 * D * (P + P) + P * K * 12)  # 452,984,832 B per 32-cloud launch
    achieved = algo_bytes / avg_kern_s / 1e9

    # counter passes of this same command (tools/profile_round.sh): HBM-side bytes per launch (FETCH_SIZE /
    # WRITE_SIZE, separate passes) and executed VALU instructions per launch (SQ_INSTS_VALU)
    weak1 = args.scaling == "weak" or world == 8
    tj = committed_counters("pmc_traffic.json", args.version) if weak1 else None
    sq = committed_counters("sq_counters.json", args.version) if weak1 else None
    valu_issue = None
    if sq is not None:
        insts = float(sq["valu_insts_per_launch"])
        valu_issue = {
            "insts_per_launch": insts,
            "achieved_ginst_s": insts / avg_kern_s / 1e9,
            "peak_ginst_s": VALU_ISSUE_PEAK / 1e9,
            "frac": insts / avg_kern_s / VALU_ISSUE_PEAK,
            "valu_busy_frac_dominant_kernel": sq.get("valu_busy_frac_dominant_kernel"),
            "source": "profiles/sq_counters.json (rocprofv3 --pmc SQ_INSTS_VALU ... pass of bench.py)",
            "note": "wave64 VALU instructions executed by all kernels of one launch / time / (1024 SIMDs x 1.2 G "
                    "instructions/s); the exact search is bound by VALU issue, not by HBM (SURVEY.md section 8d)",
        }

    result = {
        "metric": "Mpoint-pairs/s (+ %HBM roofline) knn_points B=32 N=65536 K=16 @1/2/4/8 GPU",
        "value": value,
        "unit": "Mpoint-pairs/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": args.scaling,
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": ("knn_points B=32 N=M=65536 K=16 D=3 fp32 per GPU (BASELINE.json configs[1])"
                         if args.scaling == "weak" else
                         f"knn_points B={B_STRONG_TOTAL} N=M=65536 K=16 D=3 fp32 split over {world} GPU(s) "
                         "(BASELINE.json configs[4])")
                        + "; splitmix64 uniform [0,1)^3 clouds, p1 != p2, full lengths, norm=2",
            "clouds_per_gpu": nb,
            "clouds_total": total_clouds,
            "pairs": "brute-force-equivalent: sum_n len1[n]*len2[n] per step (the grid search visits ~0.3 % of them)",
            "sharding": "batch-sharded clouds, one process per GPU, no data-path collective",
            "kernel_version": args.version,
        },
        "roofline": {
            "bound": "hbm",
            "achieved": achieved,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS,
            "traffic": float(tj["traffic_bytes_per_step"]) if tj else None,
            "traffic_source": "profiles/pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of bench.py)"
                              if tj else None,
            "kernel": "knn_points_idx = grid build (bbox, two-level counting sort: count / scatter / per-bin sort) + knn_grid_lane_kernel "
                      "(dominant) + exact fallbacks; `achieved` prices the WHOLE op, HIP-event timed on the launch stream",
            "algorithmic_bytes_per_launch": algo_bytes,
            "avg_launch_ms": avg_kern_s * 1e3,
            "valu_issue": valu_issue,
        },
    }

    if world == 1 and args.reuse_steps > 0:
        # Outside the headline region (which rebuilds the grid every step, like the reference's stateless operator):
        # the opt-in grid reuse of repeated queries against an unmodified target (pointops_knn_points_idx_reuse).
        import pytorch3d_pointops_amd as po

        p1_alt = p1.flip(0).contiguous()  # other queries of the same shape

        def timed(fn, n):
            fn()
            torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(n):
                fn()
            torch.cuda.synchronize()
            return (time.perf_counter() - t) / n * 1e3

        po.set_grid_cache(True)
        try:
            step()
            same_ms = timed(step, args.reuse_steps)
            flip = [p1_alt, p1]

            def new_queries():
                flip.reverse()
                return _C.knn_points_idx(flip[0], p2, l1, l2, 2, K, args.version)

            new_ms = timed(new_queries, args.reuse_steps)
        finally:
            po.set_grid_cache(False)
        result["grid_reuse"] = {
            "same_target_same_queries_ms_per_step": same_ms,
            "same_target_new_queries_ms_per_step": new_ms,
            "note": "opt-in (set_grid_cache): the workspace of the previous call is handed back when the target tensors are "
                    "unmodified (object, data pointer, version counter); NOT part of `value`, which rebuilds every step",
        }
    if rank == 0 and world == 1 and args.cpu_sample_queries > 0:
        log("timing the CPU baseline sample ...")
        cb, cpu_idx = cpu_baseline(p1_h, p2_h, args.cpu_sample_queries)
        result["cpu_baseline"] = cb
        # the checker doubles as a parity probe of the benchmarked output
        gpu_idx = out[0][0, : args.cpu_sample_queries].cpu().numpy()
        result["cpu_baseline"]["idx_equal_on_sample"] = bool(np.array_equal(gpu_idx, cpu_idx[0]))
        if args.cpu_all_cores:
            n = os.cpu_count() or 1
            cb_all, _ = cpu_baseline(p1_h, p2_h, args.cpu_sample_queries * min(n, 8), threads=n)
            result["cpu_baseline_all_cores"] = cb_all
    if use_dist and args.chamfer_steps > 0:
        # Outside the KNN region: the path's one real exchange -- chamfer's batch reduction over clouds
        # sharded across ranks = one RCCL all_gather of the per-cloud loss vectors (fwd + bwd timed).
        from pytorch3d_pointops_amd.sharded import sharded_chamfer_distance

        x = p1.clone().requires_grad_(True)
        y = p2.clone().requires_grad_(True)

        def cham():
            loss, _ = sharded_chamfer_distance(x, y, total_clouds, cloud_counts=[e - s for s, e in (
                shard_bounds(B_STRONG_TOTAL, world) if args.scaling == "strong"
                else [(r * B_PER_GPU, (r + 1) * B_PER_GPU) for r in range(world)])])
            loss.backward()
            return loss

        loss = cham()  # warm-up (communicator creation, autograd graph)
        torch.cuda.synchronize()
        barrier()
        t1 = time.perf_counter()
        for _ in range(args.chamfer_steps):
            loss = cham()
        torch.cuda.synchronize()
        barrier()
        dt = (time.perf_counter() - t1) / args.chamfer_steps
        result["sharded_chamfer"] = {"clouds_total": total_clouds, "loss": float(loss), "fwd_bwd_ms_per_step": dt * 1e3,
                                     "collective": "one RCCL all_gather of the per-cloud fp32 loss vectors per step"}
    if rank == 0:
        sys.stdout.flush()
        os.write(result_fd, (json.dumps(result) + "\n").encode())
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
