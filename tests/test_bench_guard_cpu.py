"""bench.py's guard around the extra legs of a multi-rank run: whatever happens to them, rank 0
still prints the one JSON line with the headline and the process ends with status 0."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = """
import sys, time
sys.path.insert(0, %r)
import bench
extras = {}
def run():
    extras["api"] = {"ok": 1}
    %s
bench.run_guarded(run, extras, {"metric": "m", "value": 1.0}, %d, %g)
print("finished")
"""


def _run(body, rank, limit):
    return subprocess.run([sys.executable, "-c", SCRIPT % (ROOT, body, rank, limit)], capture_output=True,
                          text=True, timeout=120)


def test_legs_that_finish_leave_the_printing_to_the_caller():
    out = _run("pass", 0, 30.0)
    assert out.returncode == 0 and out.stdout.strip() == "finished", out.stderr[-2000:]


def test_a_leg_that_hangs_ends_in_the_headline_line_without_it():
    out = _run("time.sleep(60)", 0, 1.0)
    assert out.returncode == 0, out.stderr[-2000:]
    j = json.loads(out.stdout.strip())
    assert j["value"] == 1.0 and j["api"] == {"ok": 1} and "did not finish" in j["extras_error"]
    # another rank prints nothing and leaves with status 0 as well
    out = _run("time.sleep(60)", 1, 1.0)
    assert out.returncode == 0 and out.stdout.strip() == "", out.stderr[-2000:]


def test_a_leg_that_raises_is_reported_in_the_line():
    out = _run("raise RuntimeError('boom')", 0, 30.0)
    assert out.returncode == 0, out.stderr[-2000:]
    j = json.loads(out.stdout.strip())
    assert j["value"] == 1.0 and j["extras_error"] == "RuntimeError: boom"
    # a failing rank other than 0 stays until its limit, then leaves quietly
    out = _run("raise RuntimeError('boom')", 3, 1.0)
    assert out.returncode == 0 and out.stdout.strip() == "", out.stderr[-2000:]
