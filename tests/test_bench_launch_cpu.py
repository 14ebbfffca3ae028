"""CPU: `python bench.py --gpus N` without a launcher becomes its own launcher (torch.distributed.run, one process per rank, 127.0.0.1 rendezvous) and rank 0's
ONE JSON line is the command's stdout.  ASQ_BENCH_LAUNCH_PROBE=1 stops each rank after the rendezvous (gloo), before anything needs a HIP device."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_self_launches_n_ranks():
    env = dict(os.environ, ASQ_BENCH_LAUNCH_PROBE="1")
    env.pop("WORLD_SIZE", None), env.pop("RANK", None), env.pop("LOCAL_RANK", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d == {"probe": True, "n_gpus": 2, "max_rank_plus_1": 2.0, "gpus_arg": 2}


def test_bench_rejects_a_world_size_that_contradicts_gpus():
    env = dict(os.environ, ASQ_BENCH_LAUNCH_PROBE="1", WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stderr + r.stdout)
