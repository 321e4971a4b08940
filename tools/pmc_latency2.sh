#!/bin/bash
# Average memory-instruction latencies of the step kernel through rocprofv3's derived counters (accumulate() over the
# SQ_INST_LEVEL_* counters): VmemLatency, SmemLatency, LdsLatency, InstrFetchLatency, plus occupancy and unit busy figures.
# Usage: bash tools/pmc_latency2.sh <tag> [wg|wave|pm] [envs]
set -e
( while sleep 45; do echo "[pmc_latency] alive"; done ) &
HB=$!
trap "kill $HB 2>/dev/null" EXIT
TAG=${1:-lat}
export MD_STEP_KERNEL=${2:-wg}
ENVS=${3:-4096}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
CACHE_DIR=$(mktemp -d); CACHE=$CACHE_DIR/host_cache.pkl   # a private scratch dir (the pickle is keyed; gpurun_out/ only carries results back)
LEAN="--envs $ENVS --no-cpu-baseline --no-lane-follow --no-env-api --no-shared-maps --sub-batches 0 --host-cache $CACHE"
python bench.py --steps 20 --warmup 5 $LEAN > $OUT/bench_lean.json 2> $OUT/bench_lean.err
i=0
for SET in "VmemLatency" "SmemLatency" "LdsLatency" "InstrFetchLatency" "MeanOccupancyPerCU" "VALUBusy SALUBusy" "MemUnitStalled" "VALUUtilization"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $SET --output-format csv -d $OUT/q$i -- python bench.py --steps 10 --warmup 5 --preroll 60 $LEAN > $OUT/q$i.log 2>&1 || echo "set $i failed: $SET"
done
python - <<PY
import csv, glob, os
acc = {}
for f in glob.glob("$OUT/q*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "env_kernel<511" not in k and "step_kernel" not in k and "pm_" not in k:
            continue
        a = acc.setdefault((k[:48], r["Counter_Name"]), [0.0, 0])
        a[0] += float(r["Counter_Value"]); a[1] += 1
for (k, c), (v, n) in sorted(acc.items()):
    print("%-50s %-32s %16.2f avg (%d launches)" % (k, c, v / n, n))
PY
