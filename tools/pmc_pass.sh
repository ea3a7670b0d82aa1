#!/bin/bash
# extra SQ counter passes for a bench config: tools/pmc_pass.sh <cfg> <spp> "<counters pass 2>" "<counters pass 3>"
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof
CFG=${1:-c2}; SPP=${2:-64}
mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
CMD="python $ROOT/bench.py --config $CFG --steps 1 --warmup 1 --spp $SPP --no-cpu-baseline --no-exclusive-pass ${BENCH_EXTRA:-}"
[ -n "${3:-}" ] && rocprofv3 --pmc $3 --kernel-trace -d $OUT/pmc_sq2 -o sq2 -- $CMD > $OUT/pmc_sq2.log 2>&1
[ -n "${4:-}" ] && rocprofv3 --pmc $4 --kernel-trace -d $OUT/pmc_sq3 -o sq3 -- $CMD > $OUT/pmc_sq3.log 2>&1
cd $ROOT
python $ROOT/tools/summarize_prof.py $OUT > $OUT/summary2.txt 2>&1
grep -A2 "^k_extend\|^k_shadow\|^k_shade" $OUT/summary2.txt
