#!/bin/bash
# rocprofv3 passes over the Chamfer forward at the sizes bench.py reports: kernel-trace stats, SQ
# counters, HBM read and write bytes (one counter group per pass, one process per size).
# usage: profile_chamfer.sh <tag>     outputs under gpurun_out/prof_chamfer_<tag>/
set -o pipefail
tag=${1:-r02}
out=gpurun_out/prof_chamfer_$tag
mkdir -p $out
export TMPDIR=/tmp
for size in "4 642" "64 642" "64 2048" "16 8192"; do
  name=$(echo $size | tr ' ' 'x')
  P="python tools/chamfer_prof_run.py $size"
  timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $out/$name/stats -o s -- $P > /dev/null 2> $out/$name.stats.err || exit 1
  timeout -s KILL 90 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_SALU --output-format csv -d $out/$name/sq -o c -- $P > /dev/null 2> $out/$name.sq.err || exit 1
  timeout -s KILL 90 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/$name/fetch -o c -- $P > /dev/null 2> $out/$name.fetch.err || exit 1
  timeout -s KILL 90 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/$name/write -o c -- $P > /dev/null 2> $out/$name.write.err || exit 1
  timeout -s KILL 90 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $out/$name/lds -o c -- $P > /dev/null 2> $out/$name.lds.err || exit 1
  echo "== chamfer forward, $size clouds x points ==" >> $out/summary.txt
  python tools/summarize_profile.py $out/$name >> $out/summary.txt || exit 1
done
cat $out/summary.txt
