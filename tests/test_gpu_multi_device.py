"""GPU, multi-process: the N > 1 path with REAL ranks.

  * On a box with >= 2 GPUs (skipped cleanly otherwise): RCCL inside libwf_lde.so across processes, one rank per GPU --
    `bench.py --gpus W` in both sharding modes (its root-parity gates must pass and the line must say so) and the
    collective query service against the oracle (tests/ranks_sharded_query.py).  W = the largest power of two <= the
    device count, capped by WF_TEST_RANKS (default 4: the GPU boxes of this pool allow few processes per card).
  * On any GPU box: the very same rank script and the same bench gates over the gloo rehearsal transport, the ranks
    sharing the device(s) -- so the code the multi-GPU run executes is exercised by every `pytest -m gpu`.
  * The gates themselves: a corrupted root must take the run down with a non-zero exit code and no benchmark line."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _clean_env(extra=None):
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_USE_AGENT_STORE"):
        env.pop(k, None)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    env.update(extra or {})
    return env


def real_world(capi):
    n = capi.device_count()
    if n < 2:
        pytest.skip(f"{n} HIP device(s): RCCL needs one device per rank (the same ranks run over gloo below)")
    w = 1
    while 2 * w <= min(n, int(os.environ.get("WF_TEST_RANKS", "4"))):
        w *= 2
    return w


def run_ranks_script(world, env_extra=None, timeout=600):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "ranks_sharded_query.py")]
    return subprocess.run(cmd, env=_clean_env(env_extra), capture_output=True, text=True, timeout=timeout, cwd=ROOT)


def run_bench(*args, env_extra=None, timeout=900):
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], env=_clean_env(env_extra), capture_output=True,
                          text=True, timeout=timeout, cwd=ROOT)


def bench_line(out):
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    return json.loads(lines[0])


# ---- real RCCL, one rank per GPU (needs >= 2 devices) ------------------------------------------------------------------
def test_rccl_sharded_commitment_and_queries_across_processes(capi):
    w = real_world(capi)
    out = run_ranks_script(w)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    for r in range(w):
        assert f"RANK {r} OK" in out.stdout


@pytest.mark.parametrize("mode", ["proofs", "packed"])
def test_rccl_bench_verifies_itself(capi, mode):
    w = real_world(capi)
    j = bench_line(run_bench("--gpus", str(w), "--steps", "3", "--warmup", "1", "--mode", mode))
    assert j["n_gpus"] == w and j["config"]["mode"] == mode
    assert "RCCL inside libwf_lde.so" in j["collective"]["transport"]
    assert j["collective"]["verified"]["ok"] is True
    assert j["collective"]["rccl_path"] and j["collective"]["rccl_version"] > 0
    assert j["roots_gathered"] == (3 * w if mode == "proofs" else 3)


# ---- the same code over the gloo rehearsal transport (any GPU box) -----------------------------------------------------
@pytest.mark.parametrize("world", [2, 4])
def test_gloo_sharded_commitment_and_queries_across_processes(capi, world):
    capi.load()
    out = run_ranks_script(world, {"WF_BENCH_BACKEND": "gloo"})
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    for r in range(world):
        assert f"RANK {r} OK" in out.stdout


@pytest.mark.parametrize("mode", ["proofs", "packed"])
def test_gloo_bench_gates_pass_and_are_reported(capi, mode):
    capi.load()
    j = bench_line(run_bench("--gpus", "2", "--steps", "2", "--warmup", "1", "--mode", mode, env_extra={"WF_BENCH_BACKEND": "gloo"}))
    v = j["collective"]["verified"]
    assert v["ok"] is True and ("re-committed" in v["checked"] or "unsharded" in v["checked"])


@pytest.mark.parametrize("mode", ["proofs", "packed"])
def test_gloo_bench_gate_fires_on_a_corrupted_root(capi, mode):
    """--inject-fault flips one byte of a root after the timed region (rank 1's gathered root / rank 1's sharded root): the
    gate must end every rank with a non-zero exit code and print no benchmark line."""
    capi.load()
    out = run_bench("--gpus", "2", "--steps", "2", "--warmup", "1", "--mode", mode, "--inject-fault", "root",
                    env_extra={"WF_BENCH_BACKEND": "gloo"})
    assert out.returncode != 0
    assert "PARITY FAILURE" in out.stderr
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]


def test_launcher_gives_up_on_ranks_that_hang(capi):
    """The parent of `bench.py --gpus N` ends its ranks' process group at its wall-clock limit and exits non-zero."""
    capi.load()
    out = run_bench("--gpus", "2", "--steps", "1", "--warmup", "0", "--inject-fault", "hang",
                    env_extra={"WF_BENCH_BACKEND": "gloo", "WF_BENCH_LAUNCH_TIMEOUT_S": "20"}, timeout=300)
    assert out.returncode == 124, out.stdout[-2000:] + out.stderr[-2000:]
    assert "did not finish within" in out.stderr
