"""GPU: sanity of the box's RCCL through torch.distributed ("nccl" backend, a world of one) next to the library's own
RCCL binding (tests/test_gpu_comm.py), and bench.py's line contract."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

SCRIPT = r"""
import os, sys, torch, torch.distributed as dist
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=dev)
x = torch.arange(3 * 32, dtype=torch.uint8, device=dev).reshape(3, 32)
out = torch.empty((3, 32), dtype=torch.uint8, device=dev)
dist.all_gather_into_tensor(out, x)
assert torch.equal(out, x)
t = torch.tensor([1.5], dtype=torch.float64, device=dev)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
dist.barrier()
dist.destroy_process_group()
print("RCCL OK")
"""


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _env():
    env = dict(os.environ)
    env.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(_free_port()), "RANK": "0", "LOCAL_RANK": "0",
                "WORLD_SIZE": "1", "WF_ROOT": ROOT, "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    return env


def test_rccl_world_of_one(capi):
    capi.load()
    out = subprocess.run([sys.executable, "-c", SCRIPT], env=_env(), capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "RCCL OK" in out.stdout


def test_bench_line_contract(capi):
    """bench.py prints one JSON line carrying the contract's keys (short run, no CPU baseline)."""
    capi.load()
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--no-cpu-baseline"],
                         capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stdout + out.stderr
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    j = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in j, k
    assert j["n_gpus"] == 1 and j["steps"] == 3 and j["roofline"]["bound"] == "hbm" and 0 < j["roofline"]["frac"] < 1
    assert j["roots_gathered"] == 3


@pytest.mark.parametrize("config", ["cfg5", "dowork"])
def test_bench_other_configs_print_the_same_line(capi, config):
    """bench.py --config: the f128 workloads print the contract's line with in-run `alu` rates and pass their CPU-oracle
    root gate (cfg3 -- 2^22 x 64 -- runs the same code at a size kept for the profiling scripts)."""
    capi.load()
    env = dict(os.environ)
    env["WF_BENCH_CPU_QUICK"] = "1"  # one CPU commitment (the root gate below needs no thread sweep: it took 47 s of the suite for dowork)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", config, "--steps", "2", "--warmup", "1"],
                         capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stdout + out.stderr
    j = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert j["config"]["name"] == config and j["config"]["field"] == "f128"
    assert j["roofline"]["bound"] == "hbm" and 0 < j["roofline"]["frac"] < 1
    assert j["alu"]["yardstick_butterflies_per_s"] > 1e10 and 0 < j["alu"]["frac"] < 1.5
    assert j["cpu_baseline"]["root_matches_gpu"] is True
    assert set(j["cpu_baseline"]["phase_ms"]) == {"interpolate", "evaluate", "hash_rows", "merkle"}
