"""bench.py's rank handling, checked without a GPU: `--gpus N` with no launcher must START N child ranks (each of which
then reports that no GPU is visible here), and a launcher whose WORLD_SIZE disagrees with --gpus must be refused."""
import json
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(extra)
    return env


@pytest.mark.skipif(torch.cuda.is_available(), reason="CPU-only check of the spawn path")
def test_gpus_flag_spawns_child_ranks():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=_env(), capture_output=True, text=True, timeout=600)
    errs = [json.loads(l) for l in p.stdout.splitlines() if l.startswith("{")]
    assert p.returncode != 0
    # (the launcher stops the other rank as soon as one has failed, so one or two rank records arrive) ...
    ranks = [e for e in errs if "rank" in e]
    assert 1 <= len(ranks) <= 2 and all("no GPU visible" in e["error"] and e["world"] == 2 for e in ranks), p.stdout + p.stderr[-1500:]
    # ... followed by the parent's own record: exit status and the tail of the children's stderr (round 4: a multi-GPU run that dies
    # must be diagnosable from the one line a driver keeps)
    last = errs[-1]
    assert "rank" not in last and last["n_gpus"] == 2 and "child ranks exited with status" in last["error"]
    assert "exitcode" in last["stderr_tail"] and len(last["stderr_tail"]) <= 1500
    assert last["stderr_tail"][-200:] in p.stderr          # relayed as well as recorded


def test_world_size_mismatch_is_refused():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], env=_env(WORLD_SIZE="2", RANK="0"),
                       capture_output=True, text=True, timeout=300)
    assert p.returncode == 2
    assert "--gpus 4 but WORLD_SIZE=2" in p.stdout
