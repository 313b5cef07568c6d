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
    assert j["value"] == 1.0 and j["extras_error"] == "rank 0: RuntimeError: boom"
    assert "Traceback" in out.stderr and "RuntimeError: boom" in out.stderr
    # a failing rank other than 0 says why on stderr at once (exception and traceback), stays until
    # its limit -- no rank disappears under the others -- and leaves with status 0, printing no line
    out = _run("raise RuntimeError('boom')", 3, 1.0)
    assert out.returncode == 0 and out.stdout.strip() == "", out.stderr[-2000:]
    assert "Traceback" in out.stderr and "RuntimeError: boom" in out.stderr and "rank 3" in out.stderr


def test_a_hang_dumps_the_stacks_of_all_threads():
    out = _run("time.sleep(60)", 0, 1.0)
    assert "stacks of all threads follow" in out.stderr and "time.sleep" not in out.stdout
    assert "Thread 0x" in out.stderr or "Current thread" in out.stderr, out.stderr[-2000:]


GROUP_SCRIPT = """
import os, sys, time
sys.path.insert(0, %r)
import torch.distributed as dist
import bench
rank = int(os.environ["RANK"])
dist.init_process_group("gloo")
extras = {}
def run():
    extras["api"] = {"ok": 1}
    extras["_leg"] = "train_step"
    if rank == 1:
        raise RuntimeError("boom on the second rank")
    dist.barrier()                      # the other rank never arrives
bench.run_guarded(run, extras, {"metric": "m", "value": 1.0}, rank, 60.0, poll_s=0.2)
print("finished")
"""


def test_another_ranks_error_reaches_the_line_of_rank_0_at_once():
    """Two gloo ranks: rank 1 raises in a leg while rank 0 waits for it in a collective.  Rank 0
    must print the line with rank 1's exception long before the 60 s limit; both leave with 0."""
    import time
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29613", WORLD_SIZE="2")
    t0 = time.time()
    procs = [subprocess.Popen([sys.executable, "-c", GROUP_SCRIPT % ROOT], env=dict(env, RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in (0, 1)]
    outs = [p.communicate(timeout=110) for p in procs]
    assert time.time() - t0 < 50, "the ranks waited for their limit"
    assert [p.returncode for p in procs] == [0, 0], [o[1][-1500:] for o in outs]
    lines = [l for l in outs[0][0].splitlines() if l.startswith("{")]       # gloo greets on stdout
    assert len(lines) == 1, outs[0][0]
    j = json.loads(lines[0])
    assert j["value"] == 1.0 and j["api"] == {"ok": 1} and "_leg" not in j
    assert j["extras_error"] == "rank 1: RuntimeError: boom on the second rank"
    assert "{" not in outs[1][0] and "Traceback" in outs[1][1] and "train_step" in outs[1][1]
