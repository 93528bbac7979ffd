"""CPU BASELINE — TEST / BENCH INFRASTRUCTURE ONLY (kind = "port").

The reference has no CPU forward path (forward_functional always launches Triton, gemlite/core.py:184-190),
so the CPU baseline BASELINE.md §3 names is the reference's *test oracle* restated with torch CPU ops:
    unpack along K  (bitpack.py:146-162 unpack_over_cols_torch semantics)
 -> W = (W_u.float() - zeros) * scales          (tests/test_gemlitelineartriton.py:36-37)
 -> y = x.float() @ W.T                         (:141)
run on all host cores (torch.set_num_threads(os.cpu_count())).  Imported only by bench.py's cpu_baseline leg
and by tests; never by gemlite_amd/.
"""
import os
import time

import torch


def unpack_over_k(packed_kn: torch.Tensor, W_nbits: int) -> torch.Tensor:
    """packed int32 [K/e, N] -> uint8 [N, K]."""
    e = 32 // W_nbits
    shifts = torch.arange(e, dtype=torch.int32) * W_nbits
    q = (packed_kn.t().unsqueeze(-1) >> shifts) & ((1 << W_nbits) - 1)  # [N, K/e, e]
    return q.reshape(packed_kn.shape[1], -1).to(torch.uint8)


def forward_cpu(x: torch.Tensor, packed_kn: torch.Tensor, scales_g: torch.Tensor, zeros_g: torch.Tensor, W_nbits: int,
                group_size: int) -> torch.Tensor:
    """x [M,K] (any float), packed [K/e,N] int32, scales/zeros [N*K/group, 1] (the un-laid-out originals)."""
    N = packed_kn.shape[1]
    W_u = unpack_over_k(packed_kn, W_nbits)
    W = ((W_u.reshape(-1, group_size).float() - zeros_g.float()) * scales_g.float()).reshape(N, -1)
    return x.float() @ W.t()


def _make_problem(M, N, K, W_nbits, group_size, seed=0):
    g = torch.Generator().manual_seed(seed)
    W_q = torch.randint(0, 2 ** W_nbits, (N, K), generator=g, dtype=torch.int32)
    e = 32 // W_nbits
    shifts = torch.arange(e, dtype=torch.int32) * W_nbits
    packed = (W_q.reshape(N, K // e, e) << shifts).sum(-1).to(torch.int32).t().contiguous()
    scales = torch.rand(N * K // group_size, 1, generator=g) * 0.01 + 0.001
    zeros = torch.rand(N * K // group_size, 1, generator=g) * (2 ** W_nbits - 1)
    x = torch.randn(M, K, generator=g) / 10
    return x, packed, scales, zeros


def _time_calls(fn, budget_s, max_calls=2000, min_calls=1):
    fn()  # warm-up
    calls, t0 = 0, time.perf_counter()
    while True:
        fn()
        calls += 1
        el = time.perf_counter() - t0
        if (el >= budget_s and calls >= min_calls) or calls >= max_calls:
            return el / calls, calls


def time_cpu_baseline(M: int, N: int, K: int, W_nbits: int, group_size: int, budget_s: float = 12.0, seed: int = 0):
    """Bounded timing of the CPU path.  The thread count is SWEPT (oversubscribing the elementwise unpack / dequant ops
    with every hardware thread of a big host is several times slower than a moderate count) and the best setting is
    timed for the remaining budget.  Also times the "matmul only on pre-dequantised fp32 W" variant (SURVEY.md §8 d).
    Returns dict(sec_per_call, calls, threads, sweep={threads: sec}, matmul_only_sec, matmul_only_threads)."""
    ncpu = os.cpu_count() or 1
    x, packed, scales, zeros = _make_problem(M, N, K, W_nbits, group_size, seed)
    run = lambda: forward_cpu(x, packed, scales, zeros, W_nbits, group_size)  # noqa: E731
    cands = sorted({t for t in (4, 8, 16, 32, 64, 128, ncpu) if 1 <= t <= ncpu})
    sweep, t_start = {}, time.perf_counter()
    for t in cands:
        torch.set_num_threads(t)
        sec, _ = _time_calls(run, budget_s=0.25, max_calls=3)
        sweep[t] = sec
        if time.perf_counter() - t_start > budget_s * 0.4:
            break
    best = min(sweep, key=sweep.get)
    torch.set_num_threads(best)
    left = max(1.0, budget_s * 0.75 - (time.perf_counter() - t_start))
    sec, calls = _time_calls(run, budget_s=left)
    # matmul only: W dequantised once, outside the timed region
    N_ = packed.shape[1]
    W = ((unpack_over_k(packed, W_nbits).reshape(-1, group_size).float() - zeros) * scales).reshape(N_, -1)
    xf = x.float()
    mm, mm_threads = None, None
    for t in cands:
        torch.set_num_threads(t)
        s_mm, _ = _time_calls(lambda: xf @ W.t(), budget_s=0.15, max_calls=20, min_calls=3)
        if mm is None or s_mm < mm:
            mm, mm_threads = s_mm, t
    torch.set_num_threads(best)
    return dict(sec_per_call=sec, calls=calls, threads=best, sweep={int(k): round(v, 5) for k, v in sweep.items()},
                matmul_only_sec=mm, matmul_only_threads=mm_threads, host_cpus=ncpu)
