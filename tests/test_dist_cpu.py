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
    q.put((rank, nb, [p.grad.tolist() for p in params[:-1]]))      # plain lists: a tensor crosses mp.Queue as a file descriptor the
    dist.barrier()                                                 # parent may only open after this process has gone (the round-3 flake)
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
        grads = [torch.tensor(g) for g in grads]
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
    # what the collective layer saw, and where the skew is: rank 0 sleeps 10 ms per step, rank 7 80 ms
    assert rec["process_group"] == {"backend": "gloo", "rccl_world_size": 8, "rank": 0}
    sp = rec["per_rank_ms_per_step"]
    assert 10.0 <= sp["min"] < 40.0 and 80.0 <= sp["max"] <= rec["ms_per_step"] + 0.01      # (+ the closing barrier)
    # rank 0's single-rank probes run after the process group is gone (ranks 1..7 have left: nobody waits in a barrier, nobody
    # shares the node with the probes)
    assert rec["probes"] == {"order": "after destroy_process_group", "process_group_alive": False}
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


def _zero2_worker(rank, world, port, q, dtype=torch.float32):
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
    params = [torch.nn.Parameter(torch.randn(s, generator=g0).to(dtype)) for s in shapes]
    opt = _CpuShardedAdamW(params, lr=1e-2, weight_decay=0.1, max_grad_norm=0.5, bucket_bytes=200)
    assert len(opt.buckets) >= 2 and opt.world == 2
    norms = []
    for step in range(3):
        g = torch.Generator().manual_seed(1000 * step + rank)
        for i, p in enumerate(params):
            p.grad = None if (i == 4 and rank == 1) else torch.randn(p.shape, generator=g).to(dtype)     # a head rank 1 never touched
        norms.append(opt.step())
    q.put((rank, norms, [p.detach().float().tolist() for p in params], opt.state_bytes()))
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
        a, b = torch.tensor(a), torch.tensor(b)
        assert torch.equal(a, b)
        torch.testing.assert_close(a, c.detach(), rtol=2e-6, atol=2e-7)


class _SlabKernels:
    """test-local restatement of `ull_sum_slabs` (fp32 accumulation over the slabs in rank order, one rounding to the element type)."""
    @staticmethod
    def sum_slabs(x, scale=1.0):
        acc = torch.zeros(x.shape[1], dtype=torch.float32)
        for r in range(x.shape[0]):
            acc += x[r].float()
        return (acc * scale).to(x.dtype)


def _direct_worker(rank, world, port, q):
    """The direct-exchange branch ITSELF (all_to_all_single -> sum of the received slabs -> all_gather_into_tensor) at N = 2 over gloo, bf16
    gradients, with the one HIP kernel of that branch replaced by the restatement above (this process only)."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    D = importlib.import_module("u-llava_amd.dist")
    D.init_from_env(backend="gloo")
    D._KERNELS = _SlabKernels
    calls = {"a2a": 0, "ag": 0, "ar": 0}
    real = (dist.all_to_all_single, dist.all_gather_into_tensor, dist.all_reduce)

    def a2a(*a, **k):
        calls["a2a"] += 1
        return real[0](*a, **k)

    def ag(*a, **k):
        calls["ag"] += 1
        return real[1](*a, **k)

    def ar(*a, **k):
        calls["ar"] += 1
        return real[2](*a, **k)
    D.dist.all_to_all_single, D.dist.all_gather_into_tensor, D.dist.all_reduce = a2a, ag, ar
    g = torch.Generator().manual_seed(300 + rank)
    shapes = ((64, 37), (129,), (8, 8, 5), (3,))
    params = [torch.nn.Parameter(torch.zeros(s, dtype=torch.bfloat16)) for s in shapes]
    for i, p in enumerate(params):
        p.grad = (torch.randn(p.shape, generator=g) * (1.0 + 3.0 * rank + i)).to(torch.bfloat16)       # unequal magnitudes per rank
    head = torch.nn.Parameter(torch.zeros(11, 2, dtype=torch.bfloat16))          # only rank 0's batch reached this head
    if rank == 0:
        head.grad = torch.full((11, 2), 3.0, dtype=torch.bfloat16)
    params.insert(1, head)
    master = torch.nn.Parameter(torch.zeros(9, dtype=torch.float32))             # an fp32 gradient: the all_reduce branch, its own bucket
    master.grad = torch.full((9,), float(rank + 1))
    params.append(master)
    nb = D.allreduce_gradients(params, bucket_bytes=3000, force_direct=True)
    q.put((rank, nb, dict(calls), [p.grad.float().tolist() for p in params]))
    dist.barrier()
    D.dist.all_to_all_single, D.dist.all_gather_into_tensor, D.dist.all_reduce = real
    dist.destroy_process_group()


def test_direct_exchange_branch_gloo_world2():
    """`allreduce_gradients(force_direct=True)` at N = 2: every 16-bit bucket goes all_to_all_single -> slab sum -> all_gather_into_tensor
    (one of each per bucket, no all_reduce), both ranks end with bf16((g0 + g1) / 2) computed in fp32, a head without a gradient on
    rank 1 is averaged against zeros, and the fp32 bucket takes the all_reduce form."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_direct_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = sorted((q.get(timeout=180) for _ in procs), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    shapes = ((64, 37), (129,), (8, 8, 5), (3,))
    gens = [torch.Generator().manual_seed(300 + r) for r in range(2)]
    per_rank = [[(torch.randn(s, generator=gens[r]) * (1.0 + 3.0 * r + i)).to(torch.bfloat16) for i, s in enumerate(shapes)] for r in range(2)]
    want = [((a.float() + b.float()) * 0.5).to(torch.bfloat16).float() for a, b in zip(*per_rank)]
    want.insert(1, torch.full((11, 2), 1.5))
    want.append(torch.full((9,), 1.5))
    (_, nb0, c0, g0), (_, nb1, c1, g1) = out
    assert nb0 == nb1 and nb0 >= 3
    assert c0 == c1 and c0["a2a"] == nb0 - 1 and c0["ag"] == nb0 - 1 and c0["ar"] == 1, c0      # the fp32 bucket is the one all_reduce
    for a, b, w in zip(g0, g1, want):
        a, b = torch.tensor(a), torch.tensor(b)
        assert torch.equal(a, b), "ranks disagree after the exchange"
        assert torch.equal(a, w), "direct exchange != bf16(mean of the per-rank gradients)"


def test_sharded_adamw_direct_exchange_bf16_gloo_world2():
    """The 16-bit form of the ZeRO-2 cycle at N = 2: all_to_all_single of the gradient shards, fp32 mean of the slabs, owner update, all-gather
    (gloo carries all three).  Both ranks end with identical bf16 parameters = the rounded fp32 masters of single-process AdamW on
    bf16(mean of the per-rank gradients), to bf16 rounding."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_zero2_worker, args=(r, 2, port, q, torch.bfloat16)) for r in range(2)]
    for p in procs:
        p.start()
    out = sorted((q.get(timeout=180) for _ in procs), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    g0 = torch.Generator().manual_seed(7)
    shapes = ((7, 5), (33,), (4, 4, 3), (1,), (6, 2))
    ref = [torch.nn.Parameter(torch.randn(s, generator=g0).to(torch.bfloat16).float()) for s in shapes]
    topt = torch.optim.AdamW(ref, lr=1e-2, weight_decay=0.1)
    ref_norms = []
    for step in range(3):
        gens = [torch.Generator().manual_seed(1000 * step + r) for r in range(2)]
        for i, p in enumerate(ref):
            g_r0 = torch.randn(p.shape, generator=gens[0]).to(torch.bfloat16).float()
            g_r1 = torch.randn(p.shape, generator=gens[1]).to(torch.bfloat16).float()
            p.grad = ((g_r0 + (torch.zeros_like(g_r1) if i == 4 else g_r1)) * 0.5).to(torch.bfloat16).float()      # the exchange rounds the mean once
        ref_norms.append(float(torch.nn.utils.clip_grad_norm_(ref, 0.5)))
        topt.step()
    (_, n0, p0, _), (_, n1, p1, _) = out
    for a, b, c in zip(n0, n1, ref_norms):
        assert abs(a - b) < 1e-6 and abs(a - c) < 1e-4 * max(1.0, c)
    for a, b, c in zip(p0, p1, ref):
        a, b = torch.tensor(a), torch.tensor(b)
        assert torch.equal(a, b)
        assert torch.equal(a, c.detach().to(torch.bfloat16).float()) or float((a - c.detach()).abs().max()) <= 2.0 ** -7 * float(c.detach().abs().max())


def test_sharded_adamw_param_groups_scheduler_and_state_dict():
    """torch.optim's surface on ShardedAdamW (kernels replaced by torch restatements, one process): `param_groups` drives the step (a
    LambdaLR attached to it changes the update), HF Trainer's no-decay split is honoured per group, and state_dict / load_state_dict resume
    bit-exactly (master, both moments, step count, the scheduled lr)."""
    sys.path.insert(0, ROOT)
    OPT = importlib.import_module("u-llava_amd.optim")

    class _K:
        @staticmethod
        def sumsq(g, out):
            out += g.float().pow(2).sum()

        @staticmethod
        def adamw_step(master, m, v, grad, param_out, lr, b1, b2, eps, wd, step, grad_scale):
            g = grad.float() * grad_scale
            master.mul_(1 - lr * wd)
            m.lerp_(g, 1 - b1)
            v.mul_(b2).addcmul_(g, g, value=1 - b2)
            denom = (v.sqrt() / (1 - b2 ** step) ** 0.5).add_(eps)
            master.addcdiv_(m, denom, value=-lr / (1 - b1 ** step))
            param_out.copy_(master)
    saved = OPT.ops
    OPT.ops = _K
    try:
        class _Cpu(OPT.ShardedAdamW):
            def _check_device(self):
                pass

        def make():
            g = torch.Generator().manual_seed(21)
            lin = torch.nn.Linear(6, 5)
            norm = torch.nn.LayerNorm(5)
            for p in list(lin.parameters()) + list(norm.parameters()):
                p.data.copy_(torch.randn(p.shape, generator=g))
            named = [("proj.weight", lin.weight), ("proj.bias", lin.bias), ("norm.weight", norm.weight), ("norm.bias", norm.bias)]
            return named

        def grads(named, step):
            g = torch.Generator().manual_seed(500 + step)
            for _, p in named:
                p.grad = torch.randn(p.shape, generator=g)
        named = make()
        groups = OPT.ShardedAdamW.no_decay_groups(named, 0.1)
        assert [len(g["params"]) for g in groups] == [1, 3] and groups[1]["weight_decay"] == 0.0
        opt = _Cpu(groups, lr=1e-2, max_grad_norm=None, bucket_bytes=64)
        assert isinstance(opt.param_groups, list) and opt.param_groups[0]["lr"] == 1e-2
        sched = torch.optim.lr_scheduler.LambdaLR(opt, lambda s: 1.0 / (1 + s))         # attaches: ShardedAdamW is a torch.optim.Optimizer
        ref_named = make()
        ref = torch.optim.AdamW([{"params": [ref_named[0][1]], "weight_decay": 0.1}, {"params": [p for _, p in ref_named[1:]], "weight_decay": 0.0}],
                                lr=1e-2)
        rs = torch.optim.lr_scheduler.LambdaLR(ref, lambda s: 1.0 / (1 + s))
        mid = None
        for step in range(4):
            grads(named, step); grads(ref_named, step)
            assert [g_["lr"] for g_ in opt.param_groups] == [rg["lr"] for rg in ref.param_groups]
            opt.step(); ref.step(); rs.step(); sched.step()
            if step == 1:
                mid = opt.state_dict()
                mid_params = [p.detach().clone() for _, p in named]
        for (_, a), (_, b) in zip(named, ref_named):
            torch.testing.assert_close(a.detach(), b.detach(), rtol=2e-6, atol=2e-7)
        # resume from the step-1 state on a fresh optimizer over fresh parameters holding the step-1 values: steps 2 and 3 reproduce
        named2 = make()
        for (_, p), v in zip(named2, mid_params):
            p.data.copy_(v)
        opt2 = _Cpu(OPT.ShardedAdamW.no_decay_groups(named2, 0.1), lr=123.0, max_grad_norm=None, bucket_bytes=64)
        opt2.load_state_dict(mid)
        assert opt2.step_count == 2 and opt2.param_groups[0]["lr"] == mid["param_groups"][0]["lr"] != 123.0
        rs2 = [1e-2 / (1 + s) for s in range(4)]
        for step in (2, 3):
            grads(named2, step)
            for g_ in opt2.param_groups:
                g_["lr"] = rs2[step]
            opt2.step()
        for (_, a), (_, b) in zip(named2, named):
            assert torch.equal(a.detach(), b.detach())
    finally:
        OPT.ops = saved
