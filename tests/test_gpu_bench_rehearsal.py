"""GPU: `python bench.py --gpus N` rehearsed on ONE GPU (SPARTAN_BENCH_REHEARSAL=1: every rank on device 0, gloo, the callback exchange): the whole
N > 1 control flow — self-launch, barriers, max-over-ranks timing, the sharded legs under their watchdogs, the sharded config-4 prove with its
exchanges — runs and prints ONE well-formed line. The figures are not measurements (the ranks share the GPU); the line says so."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("workload,world", [("c2", 2), ("c4", 2), ("c5", 2)])
def test_bench_rehearsal(workload, world):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["SPARTAN_BENCH_REHEARSAL"] = "1"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--workload", workload, "--steps", "2", "--warmup", "1", "--concurrent", "0",
           "--no-cpu-baseline"]
    if workload == "c5":
        cmd += ["--instances", "8"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == world and d["steps"] == 2 and d["value"] > 0 and d["scaling"] in ("weak", "strong")
    if workload == "c2":
        legs = d["sharded"]
        assert legs["rccl_ranks"] == world and "error" not in legs
        assert legs["c4_commit"]["rows_per_rank"] == 2048 // world and legs["c4_prove"]["exchanges_per_prove"] > 0
        assert "rehearsal" in d


def test_the_line_survives_a_process_that_dies_inside_a_sharded_leg():
    """SPARTAN_BENCH_DIE_IN=<leg>: bench.py kills itself (SIGKILL) at the start of that leg — what a fault in native collective code, or the launcher ending
    the ranks after another one died, does to rank 0. Its guardian (a child holding the latest snapshot of the line) prints the ONE line: the headline,
    the legs finished before, the leg it died in marked."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["SPARTAN_BENCH_DIE_IN"] = "msm_general"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--concurrent", "0", "--no-cpu-baseline"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode != 0
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["roofline"]["frac"] > 0
    assert d["sharded"]["c4_commit"]["rows_per_rank"] == 2048 and "ended inside this leg" in d["sharded"]["msm_general"]["error"]
