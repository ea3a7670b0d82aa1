#!/bin/bash
# rocprofv3 recipe for the bench workload (run on the GPU box through gpurun):
#   pass 1: kernel trace + stats (per-kernel time)      -> gpurun_out/prof/stats
#   pass 2: SQ counters (issue / wait breakdown)        -> gpurun_out/prof/pmc_sq
#   pass 3/4: FETCH_SIZE, WRITE_SIZE (HBM-side traffic) -> gpurun_out/prof/pmc_fetch, pmc_write
# PMC passes never combine with sys/hip/hsa tracing (the pool refuses that combination).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof
CFG=${1:-c2}
SPP=${2:-64}
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
CMD="python $ROOT/bench.py --config $CFG --steps 1 --warmup 1 --spp $SPP --no-cpu-baseline --no-exclusive-pass ${BENCH_EXTRA:-}"
rocprofv3 --kernel-trace --stats -d $OUT/stats -o stats -- $CMD > $OUT/stats.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --kernel-trace -d $OUT/pmc_sq -o sq -- $CMD > $OUT/pmc_sq.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc_fetch -o fetch -- $CMD > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/pmc_write -o write -- $CMD > $OUT/pmc_write.log 2>&1
cd $ROOT
find $OUT -name "*.csv" | head -20
python $ROOT/tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
