#!/bin/bash
# PMC + kernel-trace passes of one bench config for THIS round's kernels (run on the GPU box through gpurun):
#   tools/pmc_round.sh <config> <spp> [round]
#     -> gpurun_out/prof_<config>/{stats,pmc_sq,pmc_fetch,pmc_write}   rocprofv3 output (scratch)
#     -> gpurun_out/profiles/r0N_<config>_rocprofv3.txt                per-kernel time table + counters
#     -> gpurun_out/profiles/r0N_<config>_counters.json                what bench.py attaches to its roofline
#        (gpurun merges gpurun_out/ back; copy both into profiles/, which is tracked)
# One render lane, so that every kernel has the GPU to itself (bench.py's exclusive pass measures the same thing with HIP events).
# Counter passes never combine with sys/hip/hsa tracing (the pool refuses that combination).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
CFG=${1:-c2}; SPP=${2:-64}; RND=${3:-2}
OUT=$ROOT/gpurun_out/prof_$CFG
rm -rf $OUT; mkdir -p $OUT $ROOT/gpurun_out/profiles
export TMPDIR=/tmp
cd /tmp
CMD="python $ROOT/bench.py --config $CFG --steps 1 --warmup 1 --spp $SPP --lanes 1 --no-cpu-baseline --no-profile"
timeout 420 rocprofv3 --kernel-trace --stats -d $OUT/stats -o stats -- $CMD > $OUT/stats.log 2>&1
timeout 420 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --kernel-trace -d $OUT/pmc_sq -o sq -- $CMD > $OUT/pmc_sq.log 2>&1
timeout 420 rocprofv3 --pmc SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_INST_LEVEL_VMEM --kernel-trace -d $OUT/pmc_sq2 -o sq2 -- $CMD > $OUT/pmc_sq2.log 2>&1
timeout 420 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc_fetch -o fetch -- $CMD > $OUT/pmc_fetch.log 2>&1
timeout 420 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/pmc_write -o write -- $CMD > $OUT/pmc_write.log 2>&1
cd $ROOT
TXT=$ROOT/gpurun_out/profiles/$(printf "r%02d" $RND)_${CFG}_rocprofv3.txt
{ echo "# rocprofv3 passes of: bench.py --config $CFG --steps 1 --warmup 1 --spp $SPP --lanes 1 --no-cpu-baseline --no-profile"; echo "# (1 + $SPP + $SPP spp rendered per pass: kernel-load render, warm-up step, timed step; tools/pmc_round.sh)"; python $ROOT/tools/summarize_prof.py $OUT; } > $TXT 2>&1
python $ROOT/tools/make_counters_json.py $CFG $SPP $OUT $RND
cp $OUT/stats.log $ROOT/gpurun_out/profiles/$(printf "r%02d" $RND)_${CFG}_bench_stdout.log 2>/dev/null
rm -rf $OUT            # the rocprofv3 databases are tens of MiB per config; gpurun only merges 64 MiB back
