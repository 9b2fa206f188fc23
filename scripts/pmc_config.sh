#!/bin/bash
# rocprofv3 passes over `bench.py --config N` (run on the GPU box through gpurun): kernel trace + separate counter passes.
# usage: scripts/pmc_config.sh <config 1..5> <tag>     outputs under gpurun_out/prof_<tag>/
# then (in the build container): scripts/summarize_profile.py <tag> <config> <kernel substring>
set -e
CFG=${1:-5}
TAG=${2:-c$CFG}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
# what the counters are taken on: bench.py refuses a pass whose hash differs from the sources it is timing
(cd $ROOT && python3 -m kmc_amd.build --source-sha256) > $OUT/source_sha256.txt
cd /tmp
export TMPDIR=/tmp
TRACE_STEPS=400; [ "$CFG" = 4 ] && TRACE_STEPS=40
BENCH="python3 $ROOT/bench.py --config $CFG --steps 6 --warmup 2 --no-cpu-baseline --no-long-launch --launch-steps 1"
# the trace pass runs many steps (launch groups need a few steps to drift apart, the first launches are cold); the
# counter passes serialise the kernels anyway and take 6 + 2
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --config $CFG --steps $TRACE_STEPS --warmup 20 --no-cpu-baseline --no-long-launch --launch-steps 1 > $OUT/trace.log 2>&1
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
# the shader clock DURING a counter pass: busy cycles of the GPU / the dispatch's duration (timestamps of the same rows)
timeout -s KILL 120 rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_grbm -- $BENCH > $OUT/pmc_grbm.log 2>&1
echo grbm done
find $OUT -name "*.csv" | wc -l
