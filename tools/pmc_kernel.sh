#!/bin/bash
# usage (on the GPU box): bash tools/pmc_kernel.sh <out tag> <kernel name substring> -- <program and args>
# Separate rocprofv3 --pmc passes (kernel-trace only beside them), then per-kernel averages of every counter.
TAG=$1; PAT=$2; shift 3
OUT=gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-$OLDPWD}
i=0
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA" \
           "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_WAVES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_FLAT"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $SET -d $OUT/p$i -o c --output-format csv -- "$@" > $OUT/p$i.log 2>&1
done
python3 - "$OUT" "$PAT" <<'PY'
import csv, glob, sys, collections
out, pat = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(out + '/p*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if pat in r['Kernel_Name']:
            a = agg[r['Counter_Name']]
            a[0] += float(r['Counter_Value']); a[1] += 1
for k in sorted(agg):
    print(f'{k:36s} {agg[k][0] / agg[k][1]:16.1f}  (n={agg[k][1]})')
PY
