#!/bin/bash
# Memory-instruction latency diagnosis of the step kernel: SQ_INST_LEVEL_* accumulate the number of instructions in flight per
# cycle, so LEVEL / INSTS = average latency of that instruction class; plus how the issue cycles split over the classes.
# Usage: bash tools/pmc_latency.sh <tag> [wg|wave|pm]
set -e
( while sleep 45; do echo "[pmc_latency] alive"; done ) &
HB=$!
trap "kill $HB 2>/dev/null" EXIT
TAG=${1:-lat}
export MD_STEP_KERNEL=${2:-wg}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
CACHE_DIR=$(mktemp -d); CACHE=$CACHE_DIR/host_cache.pkl   # a private scratch dir (the pickle is keyed; gpurun_out/ only carries results back)
LEAN="--no-cpu-baseline --no-lane-follow --no-env-api --no-shared-maps --sub-batches 0 --host-cache $CACHE"
rocprofv3 -L > $OUT/avail.txt 2>&1 || true
python bench.py --steps 20 --warmup 5 $LEAN > $OUT/bench_lean.json 2> $OUT/bench_lean.err
i=0
for SET in "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_FLAT" \
           "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_LDS" \
           "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_MISC" \
           "SQ_INSTS_BRANCH SQ_INSTS_SENDMSG SQ_INSTS_VSKIPPED SQ_INSTS_VALU_TRANS_F32 SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_ICACHE_REQ SQC_ICACHE_MISSES" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL" \
           "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum" \
           "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $SET --output-format csv -d $OUT/p$i -- python bench.py --steps 10 --warmup 5 --preroll 60 $LEAN > $OUT/p$i.log 2>&1 || echo "set $i failed: $SET"
done
python - <<PY
import csv, glob, os
acc = {}
for f in glob.glob("$OUT/p*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "env_kernel<511" not in k and "step_kernel" not in k and "pm_" not in k:
            continue
        a = acc.setdefault((k[:48], r["Counter_Name"]), [0.0, 0])
        a[0] += float(r["Counter_Value"]); a[1] += 1
for (k, c), (v, n) in sorted(acc.items()):
    print("%-50s %-32s %16.0f per launch (%d launches)" % (k, c, v / n, n))
PY
