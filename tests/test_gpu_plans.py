"""GPU: the parity suites again under forced pass plans.  By default a transform of 2^L rows takes 1 pass (L <= 10) or
2 (3 only from 2^22 f64 / 2^20 f128 on), so the 3- and 4-pass code paths of the segment kernels would only ever see the
largest inputs.  WF_EXP_MAX_DIGIT (a tuning switch of the library, read by its planner) caps the digit size, which sends
the small and medium shapes of the parity and golden suites through many-pass plans; results must stay bit-exact."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


# The forced-plan runs.  CORE runs on every invocation (the switches that reach the most code that default shapes never see);
# of the ROTATING ones a subset of ROTATE_PICK runs per day -- chosen by the date, printed in the test ids, so that a failure
# names the switch -- unless WF_TEST_ALL_PLANS=1 (every switch: what scripts/random_soak.py and the end-of-round soaks in
# profiles/ run).  The suite had grown to 14 child suites and 95 s of a 900 s limit; the oracle comparisons of the BASELINE
# configurations (tests/test_gpu_fullsize.py, test_gpu_production_size.py) are not part of the rotation and always run.
PARITY_CORE = ["WF_EXP_MAX_DIGIT=5", "WF_EXP_NO_FUSED_HASH=1", "WF_EXP_PERSISTENT_ALWAYS=1", "WF_EXP_NO_SPECIALIZED=1",
               # the alternatives the round-5 defaults replaced (per-tile factor tables of the f128 passes, a lane walking its own row in the
               # separate chunk hashing): product code for f64 / odd widths, reached for the other shapes only through these switches
               "WF_EXP_NO_GTAB1=1 WF_EXP_NO_FTAB=1 WF_EXP_NO_STAGED_CHUNKS=1 WF_EXP_NO_CHUNKED=1 WF_EXP_NO_GTAB1_WIDE=1 WF_EXP_MAX_DIGIT=6"]
PARITY_ROTATING = ["WF_EXP_MAX_DIGIT=7", "WF_EXP_MAX_DIGIT=5 WF_EXP_WIDE_TI=8", "WF_EXP_MAX_DIGIT=7 WF_EXP_WIDE_TI=1", "WF_EXP_NO_PERSISTENT=1",
                   "WF_EXP_NO_CHUNKED=1", "WF_EXP_NO_COSET_INNER=1 WF_EXP_MAX_DIGIT=7",
                   "WF_EXP_NO_GTAB=1 WF_EXP_MAX_DIGIT=5 WF_EXP_WIDE_TI=8", "WF_EXP_GTAB1_F64=1 WF_EXP_MAX_DIGIT=7"]
RESIDENT_CORE = ["WF_EXP_PIPELINE_MIN_BYTES=0 WF_EXP_MAX_DIGIT=5"]
RESIDENT_ROTATING = ["WF_EXP_PIPELINE_MIN_BYTES=0", "WF_EXP_NO_PIPELINE=1"]
ROTATE_PICK = 3


def _todays(rotating, pick):
    if os.environ.get("WF_TEST_ALL_PLANS") == "1":
        return list(rotating)
    import datetime
    day = int(os.environ.get("WF_TEST_PLAN_DAY", datetime.date.today().toordinal()))
    return [rotating[(day * pick + k) % len(rotating)] for k in range(min(pick, len(rotating)))]


PARITY_SWITCHES = PARITY_CORE + _todays(PARITY_ROTATING, ROTATE_PICK)
PARITY_FILES = ["test_gpu_coset_shard.py", "test_gpu_parity.py", "test_gpu_golden.py"]
RESIDENT_SWITCHES = RESIDENT_CORE + _todays(RESIDENT_ROTATING, 1)
RESIDENT_FILES = ["test_gpu_queries.py", "test_gpu_deep.py", "test_gpu_pipeline.py", "test_gpu_wide_resident.py"]


def _run_suite(switches, files):
    env = dict(os.environ)
    env["WF_EXP_ENABLE"] = "1"  # the library reads its WF_EXP_* switches only under this one
    for sw in switches.split():
        name, value = sw.split("=")
        env[name] = value
    return subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider"]
                          + [os.path.join(ROOT, "tests", f) for f in files],
                          env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)


@pytest.fixture(scope="module")
def forced_runs(capi):
    """Every forced-plan run of this module, four child processes at a time (with this process: five on the GPU, within
    the pool's limit of six) -- most of a child's time is the CPU oracle, so the runs overlap well: the module takes a
    third of the wall clock of running them one after the other."""
    from concurrent.futures import ThreadPoolExecutor
    capi.load()
    jobs = [(sw, PARITY_FILES) for sw in PARITY_SWITCHES] + [(sw, RESIDENT_FILES) for sw in RESIDENT_SWITCHES]
    with ThreadPoolExecutor(max_workers=4) as pool:
        results = list(pool.map(lambda j: _run_suite(*j), jobs))
    return {sw: r for (sw, _), r in zip(jobs, results)}


@pytest.mark.parametrize("switch", PARITY_SWITCHES)
def test_parity_under_forced_plans(forced_runs, switch):
    """WF_EXP_NO_FUSED_HASH: one-segment, one-trace matrices normally get their leaves from the last evaluation pass;
    with the switch they go through k_hash_rows like every other shape -- both routes must give the same bytes.
    WF_EXP_NO_PERSISTENT: the fused last pass as one work-group per tile instead of the persistent ticket kernel.
    WF_EXP_NO_CHUNKED: rows longer than one BLAKE3 chunk hashed by the separate chunk kernels instead of chunk by chunk
    inside the persistent pass.  WF_EXP_PERSISTENT_ALWAYS: the ticket kernel also on the small shapes that normally take
    one work-group per tile.  WF_EXP_NO_SPECIALIZED: every tile size on the generic kernels (the tile-size-specialised
    instantiations are the default for 2^7 .. 2^10-row tiles, so without this run the generic code would see few shapes).
    WF_EXP_NO_COSET_INNER: the first strided evaluation pass with the coset as the outermost tile index (the default walks the
    cosets of 64 neighbouring tiles back to back on one XCD whenever the tile count allows).  WF_EXP_NO_GTAB: the later wide strided
    passes rebuild their output factors in LDS per tile (the default reads them from a per-context table in global memory; the
    WF_EXP_MAX_DIGIT=5 WF_EXP_WIDE_TI=8 run above takes that default on the small shapes of the parity suites)."""
    out = forced_runs[switch]
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]


@pytest.mark.parametrize("switches", RESIDENT_SWITCHES)
def test_resident_suites_with_pipelined_upload(forced_runs, switches):
    """Resident trace commitments of several segments upload segment by segment under the kernels of the previous segments
    (trace_commit_pipelined) once a column is a MiB or more; WF_EXP_PIPELINE_MIN_BYTES=0 sends the small multi-segment
    shapes of the resident test suites down that route (with WF_EXP_MAX_DIGIT=5: also those below 2^11 rows, which are
    single-pass otherwise), WF_EXP_NO_PIPELINE switches it off: same roots, rows, proofs and polynomials every way."""
    out = forced_runs[switches]
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
