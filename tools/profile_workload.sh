# PMC passes (instruction mix, wait shares, HBM traffic) + kernel trace of one bench workload other than the headline one.
# Usage: bash tools/profile_workload.sh <workload> <tag>     (outputs under gpurun_out/<tag>/; run on the GPU box)
set -e
( while sleep 45; do echo "[profile_workload] alive"; done ) &   # gpurun treats 7 silent minutes as a hang
HB=$!
trap "kill $HB 2>/dev/null" EXIT
WL=${1:-scenario}
TAG=${2:-r03_$WL}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
CACHE_DIR=$(mktemp -d); CACHE=$CACHE_DIR/host_cache.pkl
LEAN="--workload $WL --no-cpu-baseline --no-lane-follow --no-env-api --no-shared-maps --sub-batches 0 --host-cache $CACHE"
python bench.py --steps 300 --warmup 30 $LEAN > $OUT/bench.json 2> $OUT/bench.err || { tail -5 $OUT/bench.err; exit 1; }   # writes the cache
cat $OUT/bench.json
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python bench.py --steps 100 --warmup 10 $LEAN > $OUT/trace.log 2>&1
timeout -k 10 240 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python bench.py --steps 20 --warmup 5 --preroll 100 $LEAN > $OUT/pmc_fetch.log 2>&1
timeout -k 10 240 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python bench.py --steps 20 --warmup 5 --preroll 100 $LEAN > $OUT/pmc_write.log 2>&1
timeout -k 10 240 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVES --output-format csv -d $OUT/pmc_insts -- python bench.py --steps 20 --warmup 5 --preroll 100 $LEAN > $OUT/pmc_insts.log 2>&1
timeout -k 10 240 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES --output-format csv -d $OUT/pmc_wait -- python bench.py --steps 20 --warmup 5 --preroll 100 $LEAN > $OUT/pmc_wait.log 2>&1 || true
python tools/summarize_profile.py $OUT $OUT/pmc_traffic.json > $OUT/summary.txt
cat $OUT/summary.txt
