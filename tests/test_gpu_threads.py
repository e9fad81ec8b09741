"""GPU: threading contract of the boundary (SURVEY.md §8b "Threading"): a context is used from one thread at a time,
but distinct contexts are independent -- here four host threads, each with its own wf_ctx on the same device, run
commitments of different shapes concurrently (ctypes releases the GIL inside the calls); every result must equal the
oracle's, and wf_last_error() is per thread."""
import threading

import numpy as np
import pytest

from conftest import rand_cols

pytestmark = pytest.mark.gpu
F64, F128 = 1, 2

SHAPES = [(F64, 1, 12, 3, 8, 1), (F128, 1, 10, 2, 5, 2), (F64, 2, 11, 3, 3, 1), (F64, 1, 13, 2, 10, 1)]


def test_contexts_on_concurrent_threads(orc, capi):
    capi.load()
    jobs = []
    for k, (field, ext, logR, logB, n_cols, n_traces) in enumerate(SHAPES):
        rng = np.random.default_rng(4242 + k)
        traces = [rand_cols(rng, field, n_cols, (1 << logR) * ext) for _ in range(n_traces)]
        want = orc.build_trace_commitment(field, traces, ext, logR, logB, 7 if field == F64 else 3)
        jobs.append((capi.make_params(field, ext, logR, logB, n_cols, n_traces), traces, want))
    errors = []
    start = threading.Barrier(len(jobs))

    def worker(k):
        try:
            params, traces, want = jobs[k]
            ctx = capi.Context(0)
            try:
                start.wait()
                for _ in range(3):
                    got = ctx.trace_commit(params, [c for t in traces for c in t])
                    assert got["root"] == want["root"], f"thread {k}: root"
                    assert np.array_equal(got["nodes"], want["nodes"]), f"thread {k}: nodes"
                    for t in range(len(traces)):
                        assert np.array_equal(got["lde"][t], want["lde"][t]), f"thread {k}: lde {t}"
                # an error raised in this thread is reported to this thread
                bad = capi.make_params(F64, 1, 2, 3, 1, 1)
                with pytest.raises(capi.WfError) as e:
                    ctx.trace_commit(bad, [np.zeros(4, dtype=np.uint64)])
                assert e.value.code == -12 and "trace length" in str(e.value)
            finally:
                ctx.close()
        except BaseException as exc:  # noqa: BLE001 - reported to the main thread
            errors.append((k, repr(exc)))

    threads = [threading.Thread(target=worker, args=(k,)) for k in range(len(jobs))]
    for t in threads:
        t.start()
    for t in threads:
        t.join(600)
    assert not errors, errors
