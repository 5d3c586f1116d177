import os
import sys

os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")     # the configuration patchioner_amd/__init__.py and bench.py run (see there)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def usable_cpus() -> int:
    """CPUs this process may actually use: the cgroup quota (a GPU box shows 256 CPUs and grants 16: torch's default of
    128 threads then spends its time being throttled -- one oracle-heavy test took 48 s with 128 threads and 7 s with 16),
    the affinity mask, the CPU count, whichever is smallest."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except AttributeError:
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        pass
    return max(1, n)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    try:
        import torch
        torch.set_num_threads(min(torch.get_num_threads(), usable_cpus()))
    except ImportError:
        pass


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLDEN, name + ".npz"))

    return load


@pytest.fixture(autouse=True)
def _oracle_decoder_keeps_keys_and_values(request):
    """The -m gpu tests decode hundreds of prefixes with the CPU oracle (twice: on the fp32 path and on the HIP path's own
    prefixes).  They use the oracle's key/value-cached greedy loop -- the same sums per position, held to the reference's
    golden ids by tests/test_oracle_golden.py::test_decoder_ids_bit_exact[*-True] -- so the suite takes minutes, not a
    quarter of an hour.  The CPU tests keep the reference's literal full re-forward."""
    if request.node.get_closest_marker("gpu") is None:
        yield
        return
    from oracle import patchioner_oracle as O
    old = O.DeCapOracle.fast
    O.DeCapOracle.fast = True
    yield
    O.DeCapOracle.fast = old


def pytest_terminal_summary(terminalreporter, exitstatus, config):
    """The parity ledger survives ``pytest -q``: one line per caption set that went through the fp16 / bf16 backbone --
    identical / total, departures, the largest reference margin at a departure, the largest prefix error -- and the same
    records in tests/_parity_report.json (and gpurun_out/, when the run is on a GPU box)."""
    try:
        import parity_helpers as ph
    except ImportError:
        return
    if not ph.REPORT:
        return
    tr = terminalreporter
    tr.write_line("")
    tr.write_line("parity report (%d caption sets; captions identical to the fp32 reference / total):" % len(ph.REPORT))
    tot = dep = 0
    for r in ph.REPORT:
        tot, dep = tot + r["total"], dep + r["departures"]
        tr.write_line("  %4d / %4d  dep %d  margin %.1e  shift %.1e  prefix-err %s  [%s] %s" % (
            r["identical"], r["total"], r["departures"], r["worst_margin_at_departure"], r["max_logit_shift_at_departure"],
            "%.1e" % r["max_prefix_rel_err"] if r["max_prefix_rel_err"] is not None else "n/a", r["bound"], r["label"]))
    tr.write_line("  total: %d / %d identical, %d departures (each one a near-tie of the reference, asserted)" % (tot - dep, tot, dep))
    out = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out):
        import json
        with open(os.path.join(out, "parity_report.json"), "w") as f:
            json.dump(ph.REPORT, f, indent=1)
