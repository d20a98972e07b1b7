"""The driver runs `pytest -m gpu` under a 1200 s step limit with -x: a suite that drifts towards the limit turns one slow board into a red
GPUTEST and everything behind the timeout into "untested" (round-5 verdict: 763 s).  The budget is 600 s.  This CPU test reads the most
recent recorded run — profiles/r0N/pytest_gpu_durations.txt, written on the GPU box with `--durations=0` by tools/dev/run_final_evidence.sh —
and fails when that run was over budget, so that a round cannot end on a suite that is."""
import glob
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUDGET_S = 600.0


def latest_record():
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r0*", "pytest_gpu_durations.txt")))
    return files[-1] if files else None


def parse(path):
    text = open(path).read()
    m = re.search(r"(\d+) passed.* in ([0-9.]+)s", text)
    assert m, f"{path}: no pytest summary line"
    listed = [float(x) for x in re.findall(r"^([0-9.]+)s (?:call|setup|teardown) ", text, flags=re.M)]
    return int(m.group(1)), float(m.group(2)), listed


def test_the_recorded_gpu_run_is_within_budget():
    path = latest_record()
    assert path, "no profiles/r0N/pytest_gpu_durations.txt: record one (pytest -m gpu --durations=0 on the GPU box)"
    passed, wall, listed = parse(path)
    assert passed > 300, f"{path}: {passed} tests passed — not the whole suite"
    assert sum(listed) <= wall + 1.0
    if os.path.basename(os.path.dirname(path)) >= "r06":
        assert wall <= BUDGET_S, f"{path}: pytest -m gpu took {wall:.0f} s, the budget is {BUDGET_S:.0f} s (the driver's limit is 1200 s)"
        slow = [x for x in listed if x > 20.0]
        assert not slow, f"{path}: single tests above 20 s: {slow}"
