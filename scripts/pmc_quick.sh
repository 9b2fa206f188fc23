#!/bin/bash
# One SQ counter pass over scripts/time_calls.py (4096 x 64x64, 2048-hop launches); per-hop values to stdout.
# usage (GPU box): bash scripts/pmc_quick.sh <tag> [kernel-name-substring]
TAG=${1:-q}
KERN=${2:-hop_q64_kernel}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/pmcq_$TAG
mkdir -p $OUT
cd /tmp
export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD --output-format csv -d $OUT -- python3 $ROOT/scripts/time_calls.py > $OUT/run.log 2>&1 || { tail -5 $OUT/run.log; exit 1; }
python3 - "$OUT" "$KERN" <<'PY'
import csv, glob, sys, collections
out, kern = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(list)
for f in glob.glob(out + '/**/*counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        if kern in row['Kernel_Name']:
            acc[row['Counter_Name']].append(float(row['Counter_Value']))
H = 4096 * 2048
for k, v in sorted(acc.items()):
    print('%-22s %8.2f per hop  (%d launches)' % (k, sum(v) / len(v) / H, len(v)))
PY
