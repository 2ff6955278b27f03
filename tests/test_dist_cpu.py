"""world_size-2 gloo test of the multi-process plumbing bench.py uses for --gpus N
(sharding by independent image pairs, barrier, max-over-ranks timing, whole-job rate)."""
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import json, os, sys, time
    sys.path.insert(0, %r)
    from stereo_amd import dist as D
    rank, local_rank, world, dist = D.init("gloo")
    assert world == 2 and dist is not None
    mine = D.shard(5, rank, world)
    D.barrier(dist)
    t0 = time.perf_counter()
    time.sleep(0.05 * (rank + 1))          # rank 1 is the slow one
    dt = time.perf_counter() - t0
    rate, tmax = D.throughput(dist, len(mine), dt)
    D.barrier(dist)
    if rank == 0:
        print(json.dumps({"mine": mine, "rate": rate, "tmax": tmax, "dt0": dt}))
    dist.destroy_process_group()
""") % ROOT


def test_two_rank_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    port = 29500 + (os.getpid() % 400)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    import json
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    r = json.loads(line)
    assert r["mine"] == [0, 2, 4]
    assert r["tmax"] >= 0.09 and r["tmax"] >= r["dt0"]        # the max over ranks, not rank 0's time
    assert abs(r["rate"] - 5 / r["tmax"]) < 1e-9               # all ranks' units / slowest time


def test_single_process_degenerates():
    from stereo_amd import dist as D
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        os.environ.pop(k, None)
    rank, local_rank, world, dist = D.init()
    assert (rank, local_rank, world, dist) == (0, 0, 1, None)
    assert D.shard(3, 0, 1) == [0, 1, 2]
    assert D.throughput(None, 4, 2.0) == (2.0, 2.0)
