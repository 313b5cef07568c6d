#!/bin/bash
# rocprofv3 passes over the default bench.py run: kernel-trace stats, SQ counters, HBM read and
# write bytes (one counter group per pass).  Outputs under gpurun_out/prof_<tag>/.
set -o pipefail
# usage: profile_round.sh <tag> [extra bench.py arguments, e.g. --batch 64]
tag=${1:-r01}
shift
out=gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
B="python bench.py --no-cpu-baseline --no-extras --steps 50 --warmup 10 --kernel-launches 10 $*"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o s -- $B > $out/bench_stats.json 2> $out/stats.err || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_SALU --output-format csv -d $out/sq -o c -- $B > /dev/null 2> $out/sq.err || exit 1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -o c -- $B > /dev/null 2> $out/fetch.err || exit 1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -o c -- $B > /dev/null 2> $out/write.err || exit 1
timeout -s KILL 120 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $out/lds -o c -- $B > /dev/null 2> $out/lds.err || exit 1
python tools/summarize_profile.py $out > $out/summary.txt && cat $out/summary.txt
