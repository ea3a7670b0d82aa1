#!/bin/bash
# VALU instruction mix, instruction-cache and issue-side counters of one bench config (GPU box; separate from tools/pmc_round.sh, whose
# output bench.py reads): tools/pmc_mix.sh <config> <spp> [round]  ->  gpurun_out/profiles/r0N_<config>_valu_mix.txt
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
CFG=${1:-c2}; SPP=${2:-64}; RND=${3:-2}
OUT=$ROOT/gpurun_out/mix_$CFG
rm -rf $OUT; mkdir -p $OUT $ROOT/gpurun_out/profiles
export TMPDIR=/tmp
cd /tmp
CMD="python $ROOT/bench.py --config $CFG --steps 1 --warmup 1 --spp $SPP --lanes 1 --no-cpu-baseline --no-exclusive-pass"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT --kernel-trace -d $OUT/pmc_sq -o mix1 -- $CMD > $OUT/mix1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_BRANCH SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU --kernel-trace -d $OUT/pmc_sq2 -o mix2 -- $CMD > $OUT/mix2.log 2>&1
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS --kernel-trace -d $OUT/pmc_fetch -o mix3 -- $CMD > $OUT/mix3.log 2>&1
rocprofv3 --kernel-trace --stats -d $OUT/stats -o stats -- $CMD > $OUT/stats.log 2>&1
cd $ROOT
TXT=$ROOT/gpurun_out/profiles/$(printf "r%02d" $RND)_${CFG}_valu_mix.txt
{ echo "# VALU instruction mix / instruction cache / issue counters: rocprofv3 --pmc passes of: bench.py --config $CFG --steps 1 --warmup 1 --spp $SPP --lanes 1 --no-cpu-baseline --no-exclusive-pass (tools/pmc_mix.sh)"; python $ROOT/tools/summarize_prof.py $OUT; } > $TXT 2>&1
rm -rf $OUT
