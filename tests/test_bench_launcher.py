"""bench.py's own launcher (`python bench.py --gpus N` with no torchrun around it): the N ranks must start and fail INSIDE rank code on a node
without N devices, not in argument checking (the driver invokes bench.py exactly this way)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_self_launch_reaches_rank_code():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["HIP_VISIBLE_DEVICES"] = ""  # also on a GPU box: the ranks must see fewer devices than they need
    env["CUDA_VISIBLE_DEVICES"] = ""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], capture_output=True, text=True,
                       timeout=600, env=env)
    assert r.returncode != 0
    text = r.stdout + r.stderr
    # the launcher ends the other rank as soon as one has failed, so on a cold machine (first `import torch` of a rank takes a minute) only the faster
    # rank gets to print: one rank's message is the evidence that rank code was reached
    assert any(f"rank {r}: bench.py needs 2 MI355X device(s)" in text for r in (0, 1)), text[-2000:]
    assert "launch with torch.distributed.run" not in text


def test_under_a_launcher_world_must_match():
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="4", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode != 0 and "--gpus 2 but WORLD_SIZE=4" in (r.stdout + r.stderr)
