"""Timing helpers shared by bench.py: replica-parallel launch contract (barrier, max over ranks, whole-job sum).

The hot path does not shard (SURVEY.md §8 e): `--gpus N` runs N independent replicas.  torch.distributed is used
only to line the ranks up and to reduce the elapsed time — backend "nccl" (= RCCL over xGMI) on GPUs, "gloo" in the
CPU tests.  No tensor of the data path ever crosses ranks.
"""
import os
import time

import torch


class ReplicaGroup:
    def __init__(self, backend: str = "nccl"):
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.dist = None
        if self.world > 1:
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29500")
            dist.init_process_group(backend, rank=self.rank, world_size=self.world)
            self.dist = dist

    def barrier(self, device_sync=None):
        if device_sync is not None:
            device_sync()
        if self.dist is not None:
            self.dist.barrier()
        if device_sync is not None:
            device_sync()

    def max_over_ranks(self, seconds: float, device=None) -> float:
        if self.dist is None:
            return seconds
        t = torch.tensor([seconds], dtype=torch.float64, device=device if device is not None else "cpu")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def close(self):
        if self.dist is not None:
            self.dist.destroy_process_group()


def timed_steps(group: ReplicaGroup, run_step, steps: int, warmup: int, device_sync=None, device=None) -> float:
    """W untimed steps, then exactly K steps bracketed by barrier + device sync; returns MAX over ranks (seconds)."""
    for _ in range(warmup):
        run_step()
    group.barrier(device_sync)
    t0 = time.perf_counter()
    for _ in range(steps):
        run_step()
    if device_sync is not None:
        device_sync()
    elapsed = time.perf_counter() - t0
    group.barrier(device_sync)
    return group.max_over_ranks(elapsed, device)


def whole_job_rate(units_per_step_per_rank: float, steps: int, world: int, elapsed_max: float) -> float:
    """Weak scaling: every rank does the same work; the job rate is the sum of the units over the slowest time."""
    return world * steps * units_per_step_per_rank / elapsed_max


class HipEvents:
    """Minimal hipEvent access through libamdhip64 (the runtime torch already loaded): start / stop events that
    `gemlite_hip_set_profile_events` attaches to ONE kernel launch (hipExtLaunchKernel), i.e. device time of the
    kernel itself, without launch gaps or host overhead."""

    def __init__(self):
        import ctypes
        self._c = ctypes
        self.hip = ctypes.CDLL("libamdhip64.so")
        self.hip.hipEventCreate.argtypes = [ctypes.POINTER(ctypes.c_void_p)]
        self.hip.hipEventElapsedTime.argtypes = [ctypes.POINTER(ctypes.c_float), ctypes.c_void_p, ctypes.c_void_p]
        self.hip.hipEventSynchronize.argtypes = [ctypes.c_void_p]
        self.hip.hipEventDestroy.argtypes = [ctypes.c_void_p]

    def create(self):
        e = self._c.c_void_p()
        assert self.hip.hipEventCreate(self._c.byref(e)) == 0
        return e

    def destroy(self, e):
        self.hip.hipEventDestroy(e)

    def elapsed_ms(self, a, b):
        self.hip.hipEventSynchronize(b)
        ms = self._c.c_float()
        rc = self.hip.hipEventElapsedTime(self._c.byref(ms), a, b)
        return ms.value if rc == 0 else float("nan")


def kernel_device_us(launch, iters: int = 30, warmup: int = 3, before_each=None) -> float:
    """Mean device duration (us) of the library kernel that `launch()` starts, from per-launch HIP events.
    `before_each()` (optional) runs before every timed launch, e.g. a cache flush."""
    import torch
    from . import _hip
    lib = _hip.load()
    for _ in range(warmup):
        launch()
    ev = HipEvents()
    pairs = [(ev.create(), ev.create()) for _ in range(iters)]
    for a, b in pairs:
        if before_each is not None:
            before_each()
        lib.gemlite_hip_set_profile_events(a, b)
        launch()
    torch.cuda.synchronize()
    durs = [ev.elapsed_ms(a, b) * 1e3 for a, b in pairs]
    for a, b in pairs:
        ev.destroy(a)
        ev.destroy(b)
    durs = [d for d in durs if d == d and d > 0]
    return sum(durs) / len(durs) if durs else float("nan")
