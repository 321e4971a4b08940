# Instruction-cache / scalar-cache / issue-level counters of one bench workload (separate passes, counters only).
# Usage (GPU box): bash tools/pmc_icache.sh <workload> <tag>
set -e
( while sleep 45; do echo "[pmc_icache] alive"; done ) &
HB=$!
trap "kill $HB 2>/dev/null" EXIT
WL=${1:-metadrive}
TAG=${2:-r03_icache_$WL}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
CACHE_DIR=$(mktemp -d); CACHE=$CACHE_DIR/host_cache.pkl
LEAN="--workload $WL --no-cpu-baseline --no-lane-follow --no-env-api --no-shared-maps --sub-batches 0 --host-cache $CACHE"
python bench.py --steps 20 --warmup 5 $LEAN > $OUT/bench.json 2> $OUT/bench.err || { tail -5 $OUT/bench.err; exit 1; }
P="--steps 20 --warmup 5 --preroll 100 $LEAN"
timeout -k 10 240 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE --output-format csv -d $OUT/p1 -- python bench.py $P > $OUT/p1.log 2>&1
timeout -k 10 240 rocprofv3 --pmc SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS SQ_INSTS_BRANCH SQ_BUSY_CYCLES --output-format csv -d $OUT/p2 -- python bench.py $P > $OUT/p2.log 2>&1
timeout -k 10 240 rocprofv3 --pmc SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_TC_STALL SQC_TC_INST_REQ --output-format csv -d $OUT/p3 -- python bench.py $P > $OUT/p3.log 2>&1
timeout -k 10 240 rocprofv3 --pmc SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/p4 -- python bench.py $P > $OUT/p4.log 2>&1 || true
python - $OUT <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for p in ("p1", "p2", "p3", "p4"):
    fs = glob.glob(out + "/" + p + "/*/*counter_collection.csv")
    if not fs:
        print(p, "no output"); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        if "env_kernel<511" in k or "scenario_step" in k or "wave_step" in k:
            print(p, k)
            for c, vals in sorted(v.items()):
                print("    %-32s n=%d mean=%.1f" % (c, len(vals), sum(vals) / len(vals)))
PY
