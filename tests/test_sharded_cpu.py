"""world_size-2 gloo tests (CPU) of the batch-sharding layer: partitioning, the
differentiable all_gather of per-cloud losses, and the sharded batch reduction equal to
the single-process reduction.  The local per-cloud computation is injected (an oracle-based
numpy chamfer) because the product path has no CPU implementation."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import cases

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_bounds_and_balanced_assignment():
    from pytorch3d_pointops_amd.sharded import balanced_assignment, shard_bounds

    assert shard_bounds(256, 8) == [(32 * r, 32 * r + 32) for r in range(8)]
    assert shard_bounds(10, 4) == [(0, 3), (3, 6), (6, 8), (8, 10)]
    assert shard_bounds(2, 4) == [(0, 1), (1, 2), (2, 2), (2, 2)]
    costs = [9, 1, 8, 2, 7, 3, 6, 4]
    bins = balanced_assignment(costs, 2)
    assert sorted(sum(bins, [])) == list(range(8))
    loads = [sum(costs[i] for i in b) for b in bins]
    assert abs(loads[0] - loads[1]) <= 2


def _oracle_chamfer_per_cloud(x, y, xl, yl, w):
    """per-cloud bidirectional mean chamfer via the CPU oracle (torch tensors in/out, with a
    hand-written gradient so autograd flows through the all_gather)."""
    from oracle.oracle import Oracle

    ora = Oracle()

    class F(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x, y):
            xn, yn = x.detach().numpy(), y.detach().numpy()
            i1, d1 = ora.knn_points_idx(xn, yn, xl, yl, 2, 1)
            i2, d2 = ora.knn_points_idx(yn, xn, yl, xl, 2, 1)
            ctx.save = (xn, yn, i1, i2)
            a = torch.from_numpy(d1[..., 0].sum(1) / np.maximum(xl, 1)).float()
            b = torch.from_numpy(d2[..., 0].sum(1) / np.maximum(yl, 1)).float()
            return (a + b) * w

        @staticmethod
        def backward(ctx, g):
            xn, yn, i1, i2 = ctx.save
            gn = (g * w).numpy()
            g1 = (gn / np.maximum(xl, 1))[:, None, None] * np.ones_like(i1, np.float32)
            g2 = (gn / np.maximum(yl, 1))[:, None, None] * np.ones_like(i2, np.float32)
            ax, ay = ora.knn_points_backward(xn, yn, xl, yl, i1, 2, g1.astype(np.float32))
            by, bx = ora.knn_points_backward(yn, xn, yl, xl, i2, 2, g2.astype(np.float32))
            return torch.from_numpy(ax + bx), torch.from_numpy(ay + by)

    return F.apply(x, y)


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from pytorch3d_pointops_amd.sharded import all_gather_losses, shard_bounds, sharded_chamfer_distance

        B = 5  # uneven split: 3 + 2
        x = cases.cloud(1201, (B, 40, 3))
        y = cases.cloud(1202, (B, 55, 3))
        xl = np.array([40, 17, 33, 40, 5])
        yl = np.array([55, 55, 9, 21, 30])
        w = np.array([1.0, 0.5, 2.0, 1.5, 0.25], np.float32)
        s, e = shard_bounds(B, world)[rank]
        xt = torch.from_numpy(x[s:e]).requires_grad_(True)
        yt = torch.from_numpy(y[s:e]).requires_grad_(True)

        def local_fn(xx, yy, weights=None, **kw):
            return _oracle_chamfer_per_cloud(xx, yy, xl[s:e], yl[s:e], weights), None

        loss, lf = sharded_chamfer_distance(xt, yt, B, weights_local=torch.from_numpy(w[s:e]),
                                            batch_reduction="mean", local_fn=local_fn)
        loss.backward()
        # raw gather with uneven counts
        vec = all_gather_losses(torch.arange(s, e, dtype=torch.float32), [b - a for a, b in shard_bounds(B, world)])
        # cost-balanced placement of the ragged batch (cost = len1 * len2): rank r owns clouds plan[r]
        from pytorch3d_pointops_amd.sharded import plan_shards

        plan = plan_shards(B, world, costs=(xl * yl).tolist())
        ids = plan[rank]
        xb = torch.from_numpy(x[ids]).requires_grad_(True)
        yb = torch.from_numpy(y[ids]).requires_grad_(True)

        def local_fn_b(xx, yy, weights=None, **kw):
            return _oracle_chamfer_per_cloud(xx, yy, xl[ids], yl[ids], weights), None

        per_b, _ = sharded_chamfer_distance(xb, yb, B, weights_local=torch.from_numpy(w[ids]), batch_reduction=None,
                                            local_fn=local_fn_b, assignment=plan)
        loss_b, _ = sharded_chamfer_distance(xb, yb, B, weights_local=torch.from_numpy(w[ids]),
                                             batch_reduction="mean", local_fn=local_fn_b, assignment=plan)
        loss_b.backward()
        q.put((rank, float(loss), xt.grad.numpy().copy(), yt.grad.numpy().copy(), vec.numpy().copy(),
               plan, per_b.detach().numpy().copy(), float(loss_b), xb.grad.numpy().copy()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_sharded_chamfer_matches_single_process_gloo(world):
    """world_size 2 (3 + 2 clouds) and 4 (2 + 1 + 1 + 1 clouds, a cost-balanced assignment with ranks of 1 and 2
    clouds): the same loss on every rank, gradients equal to the single-process run."""
    port = 29500 + (os.getpid() % 2000) + 7 * world
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(world):
        r = q.get(timeout=240)
        res[r[0]] = r[1:]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    # single-process expectation on the full batch
    B = 5
    x = cases.cloud(1201, (B, 40, 3))
    y = cases.cloud(1202, (B, 55, 3))
    xl = np.array([40, 17, 33, 40, 5])
    yl = np.array([55, 55, 9, 21, 30])
    w = np.array([1.0, 0.5, 2.0, 1.5, 0.25], np.float32)
    xt = torch.from_numpy(x).requires_grad_(True)
    yt = torch.from_numpy(y).requires_grad_(True)
    per = _oracle_chamfer_per_cloud(xt, yt, xl, yl, torch.from_numpy(w))
    full = per.sum() / float(w.sum())
    full.backward()
    assert abs(res[0][0] - float(full)) <= 1e-6 and all(res[r][0] == res[0][0] for r in range(world))  # identical on every rank
    gx = np.concatenate([res[r][1] for r in range(world)])
    gy = np.concatenate([res[r][2] for r in range(world)])
    assert np.allclose(gx, xt.grad.numpy(), atol=1e-7) and np.allclose(gy, yt.grad.numpy(), atol=1e-7)
    for r in range(world):
        assert np.array_equal(res[r][3], np.arange(5, dtype=np.float32))
    # balanced placement: a non-contiguous plan, per-cloud vector back in GLOBAL cloud order, same loss / grads
    plan = res[0][4]
    assert all(res[r][4] == plan for r in range(world)) and sorted(sum(plan, [])) == list(range(B))
    from pytorch3d_pointops_amd.sharded import shard_bounds

    assert plan != [list(range(a, b)) for a, b in shard_bounds(B, world)]
    cost = (xl * yl).astype(np.float64)
    loads = [cost[ids].sum() for ids in plan]
    assert max(loads) - min(loads) <= cost.max()
    for r in range(world):
        assert np.allclose(res[r][5], per.detach().numpy(), atol=1e-7)
        assert abs(res[r][6] - float(full)) <= 1e-6
        assert np.allclose(res[r][7], xt.grad.numpy()[plan[r]], atol=1e-7)


def test_bench_launcher_fails_fast_when_a_rank_dies(tmp_path):
    """`bench.py --gpus N` run as its own launcher: a rank that dies must end the run with exit code 1 within seconds,
    even while rank 0 sits in a barrier it can never leave (stub rank body: rank 1 exits 3 at once, the others sleep)."""
    import subprocess
    import sys
    import time
    import types

    import bench

    stub = tmp_path / "rank.py"
    stub.write_text("import os, sys, time\n"
                    "if os.environ['RANK'] == '1':\n    sys.exit(3)\n"
                    "time.sleep(120)\nprint('{}')\n")
    args = types.SimpleNamespace(gpus=3)
    t0 = time.monotonic()
    rc = bench.launch_ranks(args, rank_cmd=[sys.executable, str(stub)], wall_limit_s=60)
    assert rc == 1 and time.monotonic() - t0 < 20
    # all ranks healthy: rank 0's line is relayed, exit code 0
    ok = tmp_path / "ok.py"
    ok.write_text("import os\nif os.environ['RANK'] == '0':\n    print('{\"ok\": 1}')\n")
    r = subprocess.run([sys.executable, "-c",
                        "import sys, types; sys.path.insert(0, %r); import bench; "
                        "sys.exit(bench.launch_ranks(types.SimpleNamespace(gpus=2), rank_cmd=[sys.executable, %r]))"
                        % (str(bench.ROOT), str(ok))], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0 and '{"ok": 1}' in r.stdout
    # the wall-clock limit also ends a run in which nobody fails
    hang = tmp_path / "hang.py"
    hang.write_text("import time\ntime.sleep(120)\n")
    t0 = time.monotonic()
    assert bench.launch_ranks(types.SimpleNamespace(gpus=2), rank_cmd=[sys.executable, str(hang)], wall_limit_s=1.0) == 1
    assert time.monotonic() - t0 < 20
