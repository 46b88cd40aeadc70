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
    # a head that only rank 0's batch exercised (no [SEG] row on rank 1: modeling_ullava._select never touches seg_projector there):
    # rank 1 has NO .grad for it, the bucket layout must not depend on that and both ranks must end with g0 / 2
    head = torch.nn.Parameter(torch.zeros(6, 2))
    if rank == 0:
        head.grad = torch.full((6, 2), 3.0)
    params.insert(2, head)
    frozen = torch.nn.Parameter(torch.zeros(3), requires_grad=False)      # frozen on every rank: skipped, stays without a gradient
    params.append(frozen)
    nb = D.allreduce_gradients(params, bucket_bytes=200)         # forces several buckets
    assert frozen.grad is None
    q.put((rank, nb, [p.grad.clone() for p in params[:-1]]))
    dist.barrier()
    dist.destroy_process_group()


def test_gradient_allreduce_gloo_world2():
    """data-parallel gradient averaging (bucketed) over 2 CPU ranks: every rank ends with the mean of the per-rank gradients, a
    parameter that has a gradient on one rank only is averaged against zeros (same bucket layout on both ranks)."""
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
        head = grads.pop(2)
        torch.testing.assert_close(head, torch.full((6, 2), 1.5))
        for i, g in enumerate(grads):
            torch.testing.assert_close(g, (per_rank[0][i] + per_rank[1][i]) / 2, rtol=1e-6, atol=1e-6)


def test_bench_launch_and_aggregation_gloo_world8_stub():
    """The 8-rank launch the driver's scaling run uses (one rank per GPU of an 8-GPU node), on CPU with the stub step: 8 ranks start,
    pin themselves (bench.pin_rank_to_cpus), the timing protocol aggregates SUM(images) / MAX(elapsed), one JSON line comes back."""
    import json
    import subprocess
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env["OMP_NUM_THREADS"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--backend", "gloo", "--stub", "--steps", "2",
                        "--warmup", "1"], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 8 and rec["total_images"] == 8 * 32 * 2 and rec["config"]["global_batch"] == 256
    assert rec["ms_per_step"] >= 80.0                    # rank 7 sleeps 80 ms per step: the MAX over ranks
    aff = rec["cpu_affinity"]                            # 8 CPUs here -> one per rank; hosts with < 8 usable CPUs do not pin
    assert aff is None or (aff["cpus_per_rank"] >= 1 and aff["cpus_per_rank"] * 8 <= os.cpu_count())


def test_rank_cpu_sets_partition_the_host():
    """per-rank CPU pinning (host-side contention is what threatens >= 6x at 8 GPUs): disjoint, contiguous, covers the
    allowed CPUs, and follows NUMA nodes when the rank count divides them."""
    sys.path.insert(0, ROOT)
    import bench
    allowed = list(range(0, 96)) + list(range(128, 224))            # 192 usable CPUs with a hole
    sets = [bench.rank_cpu_set(r, 8, allowed, None) for r in range(8)]
    assert all(len(s) == 24 for s in sets) and sorted(c for s in sets for c in s) == allowed
    numa = {0: list(range(0, 64)), 1: list(range(64, 128))}
    sets = [bench.rank_cpu_set(r, 4, list(range(128)), numa) for r in range(4)]
    assert sets[0] == list(range(0, 32)) and sets[1] == list(range(32, 64)) and sets[2] == list(range(64, 96))
    assert bench.rank_cpu_set(0, 1, list(range(16)), None) == list(range(16))
    assert bench.rank_cpu_set(2, 3, [0, 1], None) == [0, 1]         # fewer CPUs than ranks: no pinning


def _zero2_worker(rank, world, port, q):
    """ShardedAdamW's N > 1 cycle on CPU ranks.  The three HIP kernels it calls are replaced by test-local torch restatements (this
    process only: the product refuses CPU tensors), so what runs here is the package's own bucket / shard layout, gradient exchange,
    clipping all-reduce, owner update, all-gather and scatter back -- over gloo, which has no all_to_all: fp32 parameters take the
    all_reduce form of the exchange."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    D = importlib.import_module("u-llava_amd.dist")
    OPT = importlib.import_module("u-llava_amd.optim")
    D.init_from_env(backend="gloo")

    class _TorchKernels:
        @staticmethod
        def sumsq(g, out):
            out += g.float().pow(2).sum()

        @staticmethod
        def sum_slabs(x, scale):
            return (x.float().sum(0) * scale).to(x.dtype)

        @staticmethod
        def adamw_step(master, m, v, grad, param_out, lr, b1, b2, eps, wd, step, grad_scale):
            g = grad.float() * grad_scale                      # torch.optim.AdamW (single-tensor form), fp32
            master.mul_(1 - lr * wd)
            m.lerp_(g, 1 - b1)
            v.mul_(b2).addcmul_(g, g, value=1 - b2)
            denom = (v.sqrt() / (1 - b2 ** step) ** 0.5).add_(eps)
            master.addcdiv_(m, denom, value=-lr / (1 - b1 ** step))
            param_out.copy_(master)
    OPT.ops = _TorchKernels

    class _CpuShardedAdamW(OPT.ShardedAdamW):
        def _check_device(self):
            pass
    g0 = torch.Generator().manual_seed(7)
    shapes = ((7, 5), (33,), (4, 4, 3), (1,), (6, 2))
    params = [torch.nn.Parameter(torch.randn(s, generator=g0)) for s in shapes]
    opt = _CpuShardedAdamW(params, lr=1e-2, weight_decay=0.1, max_grad_norm=0.5, bucket_bytes=200)
    assert len(opt.buckets) >= 2 and opt.world == 2
    norms = []
    for step in range(3):
        g = torch.Generator().manual_seed(1000 * step + rank)
        for i, p in enumerate(params):
            p.grad = None if (i == 4 and rank == 1) else torch.randn(p.shape, generator=g)     # a head rank 1 never touched
        norms.append(opt.step())
    q.put((rank, norms, [p.detach().clone() for p in params], opt.state_bytes()))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_adamw_cycle_gloo_world2():
    """ZeRO-2 AdamW over 2 CPU ranks (kernels replaced by torch restatements in the workers): both ranks end every step with the same
    parameters, equal to single-process torch.optim.AdamW on the rank-averaged, norm-clipped gradients; each rank holds half the state."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_zero2_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = sorted((q.get(timeout=180) for _ in procs), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    g0 = torch.Generator().manual_seed(7)
    shapes = ((7, 5), (33,), (4, 4, 3), (1,), (6, 2))
    ref = [torch.nn.Parameter(torch.randn(s, generator=g0)) for s in shapes]
    topt = torch.optim.AdamW(ref, lr=1e-2, weight_decay=0.1)
    ref_norms = []
    for step in range(3):
        gens = [torch.Generator().manual_seed(1000 * step + r) for r in range(2)]
        for i, p in enumerate(ref):
            g_r0 = torch.randn(p.shape, generator=gens[0])
            g_r1 = torch.randn(p.shape, generator=gens[1])
            p.grad = (g_r0 + (torch.zeros_like(g_r1) if i == 4 else g_r1)) / 2
        ref_norms.append(float(torch.nn.utils.clip_grad_norm_(ref, 0.5)))
        topt.step()
    (r0, n0, p0, sb0), (r1, n1, p1, sb1) = out
    total = sum(int(torch.tensor(s).prod()) for s in shapes)
    assert sb0 == sb1 and 12 * total / 2 <= sb0 <= 12 * (total / 2 + 8 * 3)          # 3 fp32 words per owned element, shards padded to 8
    for a, b, c in zip(n0, n1, ref_norms):
        assert abs(a - b) < 1e-6 and abs(a - c) < 1e-5 * max(1.0, c)
    for a, b, c in zip(p0, p1, ref):
        assert torch.equal(a, b)
        torch.testing.assert_close(a, c.detach(), rtol=2e-6, atol=2e-7)
