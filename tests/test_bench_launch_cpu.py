"""`python bench.py --gpus N` as the driver may invoke it (no torch.distributed.run around it): the parent spawns its N
ranks itself.  Dry mode (PIO_BENCH_DRY=1: gloo, no GPU work) so that the self-launch, the rendezvous, the id all-gathers,
the MAX-over-ranks time and rank 0's single JSON line are covered in a container without a GPU."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra_env, *argv):
    env = dict(os.environ, PIO_BENCH_DRY="1", **extra_env)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(argv), env=env, capture_output=True,
                          text=True, timeout=300)


def test_bench_gpus_2_launches_its_own_ranks():
    r = _run({}, "--gpus", "2", "--steps", "3", "--warmup", "1")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout                      # ONE line, from rank 0
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["warmup"] == 1 and line["dry"] is True


def test_bench_self_launch_propagates_a_failing_rank():
    r = _run({"PIO_BENCH_DRY_FAIL": "1"}, "--gpus", "2", "--steps", "1", "--warmup", "0")
    assert r.returncode != 0
