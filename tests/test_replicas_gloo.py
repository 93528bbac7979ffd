"""N > 1 path of bench.py on CPU: two processes over gloo exercise the replica timing contract
(barrier, MAX over ranks, whole-job = sum of the replicas' units / slowest time).  No data-path collective exists."""
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import json, os, sys, time
    sys.path.insert(0, %r)
    from gemlite_amd.bench_utils import ReplicaGroup, timed_steps, whole_job_rate
    g = ReplicaGroup("gloo")
    calls = []
    def step():
        calls.append(1)
        time.sleep(0.01 * (1 + g.rank))      # rank 1 is the slow replica
    el = timed_steps(g, step, steps=5, warmup=2)
    out = dict(rank=g.rank, world=g.world, calls=len(calls), elapsed_max=el, rate=whole_job_rate(100.0, 5, g.world, el))
    print("RESULT " + json.dumps(out), flush=True)
    g.close()
""") % ROOT


def test_two_replicas_over_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    port = 29500 + (os.getpid() % 2000)
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    results = []
    for p in procs:
        out, err = p.communicate(timeout=120)
        assert p.returncode == 0, err[-2000:]
        line = [l for l in out.splitlines() if l.startswith("RESULT ")][0]
        import json
        results.append(json.loads(line[7:]))
    assert {r["rank"] for r in results} == {0, 1} and all(r["world"] == 2 for r in results)
    assert all(r["calls"] == 7 for r in results)                      # W + K steps on every rank
    assert abs(results[0]["elapsed_max"] - results[1]["elapsed_max"]) < 1e-9   # MAX over ranks is shared
    assert results[0]["elapsed_max"] >= 5 * 0.02 * 0.95                # the slow replica defines the time
    expected = 2 * 5 * 100.0 / results[0]["elapsed_max"]
    assert abs(results[0]["rate"] - expected) < 1e-6


def test_single_process_needs_no_process_group():
    sys.path.insert(0, ROOT)
    from gemlite_amd.bench_utils import ReplicaGroup, timed_steps, whole_job_rate
    env_backup = {k: os.environ.pop(k, None) for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    try:
        g = ReplicaGroup("gloo")
        assert g.world == 1 and g.dist is None
        n = []
        el = timed_steps(g, lambda: n.append(1), steps=3, warmup=1)
        assert len(n) == 4 and el >= 0 and whole_job_rate(10, 3, 1, 1.0) == 30
    finally:
        for k, v in env_backup.items():
            if v is not None:
                os.environ[k] = v
