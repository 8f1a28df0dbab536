"""CPU: bench.py's multi-rank entry.  `python bench.py --gpus N` must start N ranks by itself (the round-1 bench parsed
--gpus and ignored it).  Only the launcher / barrier / corner-gather plumbing is exercised here (gloo, world size 2,
`--cpu-plumbing`): the data path has no CPU form and stays GPU-only.  Replaces the reference's pickle + gloo gather
(/root/reference/src/utils/comm.py:179-219)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env():
    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID")}
    env["OMP_NUM_THREADS"] = "1"
    return env


def test_bench_spawns_its_own_ranks():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--cpu-plumbing",
                        "--steps", "3", "--warmup", "1", "--batch", "4"], capture_output=True, text=True, env=_env(),
                       timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout                       # rank 0 prints ONE json line
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["gathered_rows"] == 8 and j["gather_ok"] and j["steps"] == 3
    assert len(j["per_rank_ms_per_step"]) == 2


def test_world_size_must_match_gpus():
    env = _env()
    env.update(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--cpu-plumbing"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=1 but --gpus 2" in (r.stderr + r.stdout)
