#!/bin/bash
# One GPU-box pass: parity tests, smoke, bench line, rocprof kernel stats.  Every GPU step has
# its own time limit and the chain stops at the first failure.
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 600 python -u -m pytest tests -m gpu -v --timeout 180 --timeout-method thread > gpurun_out/gpu_tests.log 2>&1
rc=$?; echo "pytest_exit=$rc" >> gpurun_out/gpu_tests.log; grep -E "passed|failed|PASS|FAIL|Error" gpurun_out/gpu_tests.log | tail -60
[ $rc -eq 0 ] || [ $rc -eq 1 ] || exit $rc
timeout -k 10 120 python -u -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1 || { tail -20 gpurun_out/smoke.log; exit 1; }
tail -1 gpurun_out/smoke.log
timeout -k 10 300 python -u bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err || { tail -20 gpurun_out/bench.err; exit 1; }
cat gpurun_out/bench.json
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o r1 -- python bench.py --no-cpu-baseline --no-extras > gpurun_out/prof_bench.json 2> gpurun_out/prof.err || { tail -20 gpurun_out/prof.err; exit 1; }
find gpurun_out/prof -name "*stats*" | head; f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cat "$f"
