"""GPU: the multi-GPU leg's collective path EXECUTED on RCCL.  A gpurun box has one GPU, so the world size is 1;
what runs is everything else of the N > 1 sweep: `torch.distributed.run` as the launcher, `init_process_group("nccl", device_id=...)`,
the polled barrier, `dist.gather_corners` (`all_gather_into_tensor`) on a device tensor, the per-rank timing gather, and the
configs[3] leg under the same process group.  Replaces the reference's pickle + gloo `comm.gather` / `dist.barrier`
(/root/reference/src/utils/comm.py:84-92, 179-219; src/lightning/BoxDreamer_lightning_model.py:248-289)."""
import importlib.util
import json
import math
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _env():
    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    return env


def _bench():
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _keep(name, obj):
    out = os.path.join(ROOT, "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, name), "w") as f:
            json.dump(obj, f, indent=1)
    except OSError:
        pass


def test_rccl_probe_world1(hip):
    """The probe the default bench line carries as `rccl_world1`: communicator creation, the corner all-gather (equal and ragged shards)
    and the barrier on RCCL, through the sweep's own launcher."""
    j = _bench().rccl_world1_block()
    _keep("r6_rccl_world1.json", j)
    assert j.get("executed"), j
    assert j["backend"] == "nccl" and j["world_size_seen_by_the_collective"] == 1
    assert len(j["nccl_version"].split(".")) >= 2 and int(j["nccl_version"].split(".")[0]) >= 2
    assert j["gather_ok"] and j["ragged_gather_ok"]
    assert math.isfinite(j["corner_allgather_ms"]) and 0 < j["corner_allgather_ms"] < 50
    assert j["per_rank"] == 1 and j["timed_steps_ms_per_step"] > 0


def test_bench_sweep_flow_on_rccl_world1(hip):
    """`bench.py --gpus 1 --force-dist` under torch.distributed.run: the timed configs[1] step with the corner all-gather inside it and
    the configs[3] shard, every collective on RCCL; the line must say what the collective layer saw."""
    m = _bench()
    port = str(m.free_port())
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
           "--master-port", port, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-dist", "--config3", "--steps", "3", "--warmup", "1",
           "--no-strict", "--no-fp8", "--no-latency", "--no-cpu-baseline", "--no-inline-counters", "--no-h2d", "--no-pnp", "--no-power",
           "--no-rccl-probe", "--no-parity", "--dist-timeout", "300"]
    r = subprocess.run(cmd, capture_output=True, text=True, env=_env(), timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    j = json.loads(lines[0])
    d = j["distributed"]
    _keep("r6_rccl_world1_sweep.json", {k: j[k] for k in ("value", "ms_per_step", "n_gpus", "distributed", "corner_allgather_ms",
                                                          "per_rank_ms_per_step", "config3") if k in j})
    assert d["backend"] == "nccl" and d["world_size_seen_by_the_collective"] == 1 and d["nccl_version"]
    assert len(d["devices"]) == 1 and "cuda:0" in d["devices"][0]
    assert j["n_gpus"] == 1 and j["value"] > 0 and len(j["per_rank_ms_per_step"]) == 1
    assert math.isfinite(j["corner_allgather_ms"]) and 0 < j["corner_allgather_ms"] < 50
    c3 = j["config3"]
    assert c3["views"] == 17 and c3["value"] > 0 and len(c3["per_rank_ms_per_step"]) == 1
