#!/bin/bash
# One SQ counter pass over scripts/time_big.py (one 1024 x 1024 lattice, 20000-hop launches); per-hop values to stdout.
# usage (GPU box): bash scripts/pmc_big.sh <tag>
TAG=${1:-big}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/pmcq_$TAG
mkdir -p $OUT
cd /tmp
export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD --output-format csv -d $OUT -- python3 $ROOT/scripts/time_big.py > $OUT/run.log 2>&1 || { tail -5 $OUT/run.log; exit 1; }
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(list)
for f in glob.glob(out + '/**/*counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        if 'hop_big_kernel' in row['Kernel_Name']:
            acc[row['Counter_Name']].append(float(row['Counter_Value']))
for k, v in sorted(acc.items()):
    v = [x for x in v if x > 0.5 * max(v)]              # the 20000-hop launches (the 2000-hop presimulation aside)
    print('%-22s %10.1f per hop  (%d launches; all waves of the workgroup)' % (k, sum(v) / len(v) / 20000, len(v)))
PY
