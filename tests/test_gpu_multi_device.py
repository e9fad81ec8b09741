"""GPU, multi-process: the N > 1 path with REAL ranks.

  * On a box with >= 2 GPUs (skipped cleanly otherwise): RCCL inside libwf_lde.so across processes, one rank per GPU --
    `bench.py --gpus W` in both sharding modes (its root-parity gates must pass and the line must say so) and the
    collective query service against the oracle (tests/ranks_sharded_query.py).  W = the largest power of two <= the
    device count, capped by WF_TEST_RANKS (default 4 -- the GPU boxes of this pool allow few processes per card -- and 8 on a box with >= 8 devices).
  * On any GPU box: the very same rank script and the same bench gates over the gloo rehearsal transport, the ranks
    sharing the device(s) -- so the code the multi-GPU run executes is exercised by every `pytest -m gpu`.
  * The gates themselves: a corrupted root must take the run down with a non-zero exit code and no benchmark line.
  * The ONE-PROCESS route (`bench.py --gpus W --ranks threads`: a host thread, a context and a communicator per GPU): on
    >= 2 devices over real RCCL with W = ALL devices (largest power of two; no cap -- it costs one process whatever W is),
    on any box over the in-process rehearsal transport; its record must carry what RCCL says about the communicator
    (ncclCommCount, user ranks, devices) and every rank's own step time."""
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
    # WF_TEST_RANKS caps the world (default: 4 processes -- the one-GPU boxes of this pool allow few processes per card; a
    # whole node, >= 8 devices, runs all eight: one process per card is within every limit there)
    cap = int(os.environ.get("WF_TEST_RANKS", "8" if n >= 8 else "4"))
    w = 1
    while 2 * w <= min(n, cap):
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


# ---- one process, one host thread per GPU ---------------------------------------------------------------------------------
def all_devices_world(capi):
    n = capi.device_count()
    if n < 2:
        pytest.skip(f"{n} HIP device(s): RCCL needs one device per rank (the thread route runs over the loopback transport below)")
    w = 1
    while 2 * w <= n:
        w *= 2
    return w


@pytest.mark.parametrize("mode", ["proofs", "packed"])
def test_rccl_thread_route_verifies_itself_on_all_devices(capi, mode):
    w = all_devices_world(capi)
    j = bench_line(run_bench("--gpus", str(w), "--ranks", "threads", "--steps", "3", "--warmup", "1", "--mode", mode))
    c = j["collective"]
    assert j["n_gpus"] == w and "RCCL inside libwf_lde.so" in c["transport"] and "thread" in c["ranks"]
    assert c["verified"]["ok"] is True
    assert c["nccl_comm_count"] == w and c["nccl_comm_counts"] == [w] * w     # RCCL itself: one communicator of w ranks
    assert c["nccl_user_ranks"] == list(range(w)) and sorted(c["nccl_devices"]) == list(range(w))   # ... on w distinct devices
    assert len(c["ms_per_step_per_rank"]) == w and all(x > 0 for x in c["ms_per_step_per_rank"])
    assert j["roots_gathered"] == (3 * w if mode == "proofs" else 3)


@pytest.mark.parametrize("mode,world", [("proofs", 2), ("packed", 2), ("proofs", 4), ("packed", 8), ("proofs", 8)])  # (proofs, 8) = configs[3]'s rank count
def test_thread_route_over_the_loopback_transport(capi, mode, world):
    """The same thread-per-rank control flow with the ranks sharing the device(s): gates pass, the record says what ran."""
    capi.load()
    j = bench_line(run_bench("--gpus", str(world), "--ranks", "threads", "--steps", "2", "--warmup", "1", "--mode", mode,
                             env_extra={"WF_BENCH_BACKEND": "loopback"}))
    c = j["collective"]
    assert j["n_gpus"] == world and c["verified"]["ok"] is True and "thread" in c["ranks"] and "rehearsal" in c["transport"]
    assert c["nccl_comm_counts"] == [world] * world and c["nccl_user_ranks"] == list(range(world))
    assert len(c["ms_per_step_per_rank"]) == world
    assert j["ms_per_step"] >= max(c["ms_per_step_per_rank"]) * 0.999    # the line's time is the MAX over the ranks
    assert j["roots_gathered"] == (2 * world if mode == "proofs" else 2)


@pytest.mark.parametrize("mode", ["proofs", "packed"])
def test_thread_route_gate_fires_on_a_corrupted_root(capi, mode):
    capi.load()
    out = run_bench("--gpus", "2", "--ranks", "threads", "--steps", "2", "--warmup", "1", "--mode", mode, "--inject-fault", "root",
                    env_extra={"WF_BENCH_BACKEND": "loopback"})
    assert out.returncode != 0
    assert "PARITY FAILURE" in out.stderr
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]


def test_thread_route_gives_up_on_a_rank_that_hangs(capi):
    capi.load()
    out = run_bench("--gpus", "2", "--ranks", "threads", "--steps", "1", "--warmup", "0", "--inject-fault", "hang",
                    env_extra={"WF_BENCH_BACKEND": "loopback", "WF_BENCH_LAUNCH_TIMEOUT_S": "15", "WF_COMM_TIMEOUT_S": "60"}, timeout=300)
    assert out.returncode == 124, out.stdout[-2000:] + out.stderr[-2000:]
    assert "did not finish within" in out.stderr


def test_failed_process_launcher_falls_back_to_the_thread_route_and_says_so(capi, tmp_path):
    """`python bench.py --gpus N` whose process launcher dies without a line (stand-in launcher: exit 7) must still print a
    verified line -- from a fresh `--ranks threads` child -- and the line must say that the fallback ran and why."""
    capi.load()
    f = tmp_path / "fake_launcher.py"
    f.write_text("import sys\nsys.stderr.write('process guard: too many processes on the card\\n')\nsys.exit(7)\n")
    out = run_bench("--gpus", "2", "--steps", "2", "--warmup", "1",
                    env_extra={"WF_BENCH_BACKEND": "loopback", "WF_BENCH_LAUNCHER": f"{sys.executable} {f}"})
    j = bench_line(out)
    c = j["collective"]
    assert j["n_gpus"] == 2 and c["verified"]["ok"] is True
    assert "thread" in c["ranks"] and "FALLBACK" in c["ranks"] and "exited 7" in c["ranks"]
    assert "starting a fresh child with --ranks threads" in out.stderr


def test_thread_route_refuses_rccl_with_fewer_devices_than_ranks(capi):
    n = capi.device_count()
    out = run_bench("--gpus", str(2 * n), "--ranks", "threads", "--steps", "1", "--warmup", "0")
    assert out.returncode == 1 and "one device per rank" in out.stderr
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]


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
                    env_extra={"WF_BENCH_BACKEND": "gloo", "WF_BENCH_LAUNCH_TIMEOUT_S": "15"}, timeout=300)
    assert out.returncode == 124, out.stdout[-2000:] + out.stderr[-2000:]
    assert "did not finish within" in out.stderr
