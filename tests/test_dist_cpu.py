"""CPU, gloo, world_size 2: the N>1 aggregation used by bench.py (units summed, elapsed maxed) and the image sharding."""
import importlib
import os
import socket
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    D = importlib.import_module("u-llava_amd.dist")
    r, w, _ = D.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    lo, hi = D.shard_range(257, r, w)                 # 257 images over 2 ranks: 129 + 128, contiguous, disjoint
    units = float(hi - lo)
    elapsed = 1.0 + 0.5 * rank                        # rank 1 is the slow one
    rate, total, tmax = D.global_rate(units, elapsed)
    dist.barrier()
    q.put((rank, lo, hi, rate, total, tmax))
    dist.destroy_process_group()


def test_global_rate_and_sharding_gloo_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, lo0, hi0, rate0, tot0, t0), (_, lo1, hi1, rate1, tot1, t1) = out
    assert (lo0, hi0, lo1, hi1) == (0, 129, 129, 257)
    assert tot0 == tot1 == 257.0 and t0 == t1 == 1.5
    assert abs(rate0 - 257.0 / 1.5) < 1e-9 and rate0 == rate1


def test_single_process_rate_needs_no_group():
    D = importlib.import_module("u-llava_amd.dist")
    assert D.global_rate(64.0, 2.0) == (32.0, 64.0, 2.0)
    assert [D.shard_range(10, r, 4) for r in range(4)] == [(0, 3), (3, 6), (6, 8), (8, 10)]


def test_bench_launches_n_ranks_itself_gloo_stub():
    """`python bench.py --gpus 2` must start 2 ranks on its own (no external torchrun), run the barrier-bracketed timing protocol
    over them and report n_gpus = 2 with the MAX-over-ranks time and the SUM-over-ranks images (stub step, gloo, CPU)."""
    import json
    import subprocess
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--stub", "--steps", "3",
                        "--warmup", "1"], capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout                     # rank 0 only, ONE line
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["steps"] == 3 and rec["warmup"] == 1 and rec["scaling"] == "weak"
    assert rec["total_images"] == 2 * 32 * 3 and rec["config"]["global_batch"] == 64
    assert rec["ms_per_step"] >= 20.0                    # rank 1 sleeps 20 ms per step: the MAX over ranks
    assert abs(rec["value"] - rec["total_images"] / (rec["ms_per_step"] * 3e-3)) < 0.05 * rec["value"]


def test_bench_rejects_world_size_mismatch():
    import subprocess
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--stub", "--backend", "gloo"], capture_output=True,
                       text=True, timeout=120, env=env, cwd=ROOT)
    assert r.returncode != 0 and "launcher started 1 rank" in (r.stderr + r.stdout)


def _grad_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    D = importlib.import_module("u-llava_amd.dist")
    D.init_from_env(backend="gloo")
    g = torch.Generator().manual_seed(100 + rank)
    params = [torch.nn.Parameter(torch.zeros(s)) for s in ((7, 5), (33,), (4, 4, 3), (1,))]
    for p in params:
        p.grad = torch.randn(p.shape, generator=g)
    params.append(torch.nn.Parameter(torch.zeros(3)))            # no gradient: skipped
    nb = D.allreduce_gradients(params, bucket_bytes=200)         # forces several buckets
    q.put((rank, nb, [p.grad.clone() for p in params[:-1]]))
    dist.barrier()
    dist.destroy_process_group()


def test_gradient_allreduce_gloo_world2():
    """data-parallel gradient averaging (bucketed) over 2 CPU ranks: every rank ends with the mean of the per-rank gradients."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_grad_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = sorted((q.get(timeout=120) for _ in procs), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    shapes = ((7, 5), (33,), (4, 4, 3), (1,))
    gens = [torch.Generator().manual_seed(100 + r) for r in range(2)]
    per_rank = [[torch.randn(s, generator=gens[r]) for s in shapes] for r in range(2)]
    for (rank, nb, grads) in out:
        assert nb >= 2
        for i, g in enumerate(grads):
            torch.testing.assert_close(g, (per_rank[0][i] + per_rank[1][i]) / 2, rtol=1e-6, atol=1e-6)
