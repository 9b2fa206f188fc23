#!/bin/bash
# rocprofv3 counter passes over `bench.py --config N` (run on the GPU box through gpurun).
# usage: scripts/pmc_config.sh <config> <tag>     outputs under gpurun_out/prof_<tag>/
set -e
CFG=${1:-2}
TAG=${2:-c$CFG}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
(cd $ROOT && python3 -m kmc_amd.build --source-sha256) > $OUT/source_sha256.txt
cd /tmp
export TMPDIR=/tmp
BENCH="python3 $ROOT/bench.py --config $CFG --steps 6 --warmup 2 --no-cpu-baseline --no-long-launch"
timeout -s KILL 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --config $CFG --steps 100 --warmup 10 --no-cpu-baseline --no-long-launch > $OUT/trace.log 2>&1
echo trace done
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $BENCH > $OUT/pmc_fetch.log 2>&1
echo fetch done
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $BENCH > $OUT/pmc_write.log 2>&1
echo write done
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY --output-format csv -d $OUT/pmc_sq1 -- $BENCH > $OUT/pmc_sq1.log 2>&1
echo sq1 done
timeout -s KILL 120 rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/pmc_sq2 -- $BENCH > $OUT/pmc_sq2.log 2>&1
echo sq2 done
timeout -s KILL 120 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_tcc -- $BENCH > $OUT/pmc_tcc.log 2>&1
echo tcc done
find $OUT -name "*.csv" | head -50
