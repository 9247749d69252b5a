"""bench.py's own code paths on the hardware at hand (one GPU): the plain N = 1 run, the N > 1 path with one rank (process
group, RCCL gatherv, barriers: CM_BENCH_FORCE_DIST) and two ranks on one card over gloo (rank start-up, the index shared through
memory-mapped files, max-over-ranks timing).  No scaling figure comes out of this; it makes the driver's first multi-GPU run mean
something.  SURVEY 8(e)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _bench(extra_args=(), env=None, timeout=600):
    e = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    e.pop("CM_LIB", None)
    if env:
        e.update(env)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--pairs", "65536", "--workload", "chr21",
                        "--batches", "3", *extra_args], env=e, capture_output=True, text=True, cwd=ROOT, timeout=timeout)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-1500:]
    return json.loads(lines[0])


def test_bench_line_plain_and_through_the_process_group():
    plain = _bench()
    assert plain["n_gpus"] == 1 and plain["value"] > 0 and plain["config"]["distinct_batches"] == 3
    # the line carries its own parity evidence and the per-stage figures
    assert plain["parity"]["equal"] is True and plain["parity"]["pairs"] > 1000
    assert set(plain["stages"]) >= {"seed", "chain", "pair"} and plain["stages"]["pair"]["avg_launch_ms"] > 0
    assert plain["roofline"]["frac"] > 0 and plain["cpu_baseline"]["value"] > 0
    dist1 = _bench(env={"CM_BENCH_FORCE_DIST": "1", "MASTER_PORT": "29533"})
    assert dist1["n_gpus"] == 1
    # same reads, same batches: the records rank 0 ends up with are the same whether they came through cm_collect_records or
    # through cm_collect_records_device + gather over RCCL
    assert dist1["config"]["bsj_records_last_step"] == plain["config"]["bsj_records_last_step"] > 0
    assert dist1["parity"]["equal"] is True


def test_bench_two_ranks_on_one_card():
    two = _bench(["--gpus", "2", "--backend", "gloo", "--no-cpu-baseline"], env={"CM_BENCH_ONE_DEVICE": "1"})
    assert two["n_gpus"] == 2 and two["config"]["world_size"] == 2
    assert two["config"]["total_pairs"] == 2 * 65536 * 2
    assert two["config"]["bsj_records_last_step"] > 0            # both ranks' records, gathered on rank 0
