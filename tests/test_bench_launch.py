"""bench.py --gpus N as a PLAIN command: the parent starts the ranks (torch.distributed.run as a child) before it touches
the GPU and passes their JSON line and exit code through.

CPU: without a GPU the ranks must fail loudly (no CPU fallback) and the launcher must report it.
GPU: two ranks sharing the one device of the test box, WF_BENCH_BACKEND=gloo (RCCL refuses two ranks on one device):
the multi-rank control flow, the wf_comm calls and both sharding modes run end to end."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*args, env_extra=None, timeout=900):
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_USE_AGENT_STORE"):
        env.pop(k, None)
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], env=env, capture_output=True,
                          text=True, timeout=timeout, cwd=ROOT)


def test_plain_multi_gpu_command_starts_ranks_and_fails_loudly_without_gpus(capi):
    if capi.device_count() > 0:
        pytest.skip("a HIP device is present")
    out = run_bench("--gpus", "2", "--steps", "1", "--warmup", "0")
    assert out.returncode != 0
    assert "must be launched with" not in out.stderr + out.stdout      # the round-1 refusal is gone
    assert "torch.distributed" in out.stderr or "ChildFailedError" in out.stderr, out.stderr[-2000:]
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]  # no number without a GPU


def test_thread_route_fails_loudly_without_gpus(capi):
    """`--ranks threads` on a box without a HIP device: no line, a non-zero exit code, the reason on stderr."""
    if capi.device_count() > 0:
        pytest.skip("a HIP device is present")
    out = run_bench("--gpus", "2", "--ranks", "threads", "--steps", "1", "--warmup", "0", timeout=300)
    assert out.returncode == 1, out.stderr[-2000:]
    assert "no HIP device" in out.stderr
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]


def test_cpu_budget_reads_the_cgroup_quota(tmp_path, monkeypatch):
    """bench.cpu_budget(): the thread count the CPU baseline may use = affinity capped by the cgroup's CPU quota."""
    sys.path.insert(0, ROOT)
    import bench
    usable, det = bench.cpu_budget()
    assert usable >= 1 and det["affinity"] >= usable and "os_cpu_count" in det
    real_open = open

    def fake_open(path, *a, **k):
        if path == "/sys/fs/cgroup/cpu.max":
            f = tmp_path / "cpu.max"
            f.write_text("300000 100000\n")
            return real_open(f, *a, **k)
        return real_open(path, *a, **k)

    monkeypatch.setattr("builtins.open", fake_open)
    usable, det = bench.cpu_budget()
    assert det["quota_cpus"] == 3.0 and usable == min(3, det["affinity"])


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["proofs", "packed"])
def test_two_ranks_on_one_device_gloo_rehearsal(capi, mode):
    capi.load()
    out = run_bench("--gpus", "2", "--steps", "2", "--warmup", "1", "--mode", mode, "--no-cpu-baseline",
                    env_extra={"WF_BENCH_BACKEND": "gloo"})
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                                              # rank 0 only
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["steps"] == 2 and j["config"]["mode"] == mode
    assert j["scaling"] == ("weak" if mode == "proofs" else "strong")
    assert j["roots_gathered"] == (4 if mode == "proofs" else 2)
    assert "gloo" in j["collective"]["transport"]
    assert j["value"] > 0 and j["ms_per_step"] > 0


@pytest.mark.gpu
def test_packed_mode_single_gpu_root_matches_sharded(capi):
    """The packed workload on one rank (plain wf_trace_commit_dev) and on two (wf_trace_commit_sharded_dev) commit the
    same traces: the roots must agree."""
    capi.load()
    one = run_bench("--gpus", "1", "--steps", "1", "--warmup", "1", "--mode", "packed", "--no-cpu-baseline")
    assert one.returncode == 0, one.stdout[-2000:] + one.stderr[-2000:]
    two = run_bench("--gpus", "2", "--steps", "1", "--warmup", "1", "--mode", "packed", "--no-cpu-baseline",
                    env_extra={"WF_BENCH_BACKEND": "gloo"})
    assert two.returncode == 0, two.stdout[-2000:] + two.stderr[-2000:]
    r1 = json.loads([l for l in one.stdout.splitlines() if l.startswith("{")][-1])["root"]
    r2 = json.loads([l for l in two.stdout.splitlines() if l.startswith("{")][-1])["root"]
    assert r1 == r2


# ---- the fallback of the plain multi-GPU command (round 5): process launcher fails without a line -> fresh --ranks threads child
def _fake_launcher(tmp_path, body):
    f = tmp_path / "fake_launcher.py"
    f.write_text("import sys\n" + body)
    return f"{sys.executable} {f}"


def test_failed_process_launcher_falls_back_to_a_fresh_thread_child(capi, tmp_path):
    """A process launcher that dies without a benchmark line (a per-card process limit, say) must not lose the N > 1 record:
    the parent starts `bench.py --gpus N --ranks threads` as a NEW child.  Without a GPU that child fails loudly in turn --
    what this test sees is that it was started, with the reason handed over."""
    if capi.device_count() > 0:
        pytest.skip("a HIP device is present (the GPU form of this test is in test_gpu_multi_device.py)")
    launcher = _fake_launcher(tmp_path, "sys.stderr.write('process guard: too many processes on the card\\n'); sys.exit(7)\n")
    out = run_bench("--gpus", "2", "--steps", "1", "--warmup", "0", env_extra={"WF_BENCH_LAUNCHER": launcher}, timeout=300)
    assert out.returncode == 1, out.stderr[-2000:]
    assert "process launcher exited 7 without a benchmark line" in out.stderr
    assert "starting a fresh child with --ranks threads" in out.stderr
    assert "no HIP device" in out.stderr                                  # the thread child ran (and refused: no GPU here)
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]


@pytest.mark.parametrize("body,rc", [
    ("sys.stderr.write('PARITY FAILURE: root mismatch\\n'); sys.exit(1)\n", 1),         # a parity gate is a result, not a launch failure
    ("print('{\"metric\": \"x\"}'); sys.exit(3)\n", 3),                                    # a line was printed
    ("sys.exit(0)\n", 0),
])
def test_no_fallback_after_a_parity_failure_a_line_or_success(capi, tmp_path, body, rc):
    out = run_bench("--gpus", "2", "--steps", "1", "--warmup", "0", env_extra={"WF_BENCH_LAUNCHER": _fake_launcher(tmp_path, body)}, timeout=120)
    assert out.returncode == rc
    assert "--ranks threads" not in out.stderr


def test_fallback_can_be_switched_off(capi, tmp_path):
    launcher = _fake_launcher(tmp_path, "sys.exit(7)\n")
    out = run_bench("--gpus", "2", env_extra={"WF_BENCH_LAUNCHER": launcher, "WF_BENCH_NO_FALLBACK": "1"}, timeout=120)
    assert out.returncode == 7 and "--ranks threads" not in out.stderr
