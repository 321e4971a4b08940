#!/bin/bash
# Collect the round's evidence on the GPU box: bench line, rocprofv3 kernel-trace stats, HBM PMC passes.
# The profiled runs skip the extra operating points (--no-lane-follow --no-env-api --sub-batches 0), so that every
# step-kernel dispatch in them is one whole-batch launch, and load the host scenes from a cache written by the first
# (unprofiled) run: a process whose GPU the profiler has initialised never forks map-builder workers.
# Usage: bash tools/profile_round.sh <tag>     (outputs under gpurun_out/<tag>/)
set -e
( while sleep 45; do echo "[profile_round] alive"; done ) &   # gpurun treats 7 silent minutes as a hang
HB=$!
trap "kill $HB 2>/dev/null" EXIT
TAG=${1:-r02}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
CACHE_DIR=$(mktemp -d); CACHE=$CACHE_DIR/host_cache.pkl   # a private scratch dir (the pickle is keyed; gpurun_out/ only carries results back)
LEAN="--no-cpu-baseline --no-lane-follow --no-env-api --no-shared-maps --sub-batches 0 --host-cache $CACHE"
python bench.py --steps 300 --warmup 30 > $OUT/bench.json 2> $OUT/bench.err || { tail -5 $OUT/bench.err; exit 1; }
cat $OUT/bench.json
python bench.py --steps 20 --warmup 5 $LEAN > $OUT/bench_lean.json 2> $OUT/bench_lean.err     # writes the cache
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python bench.py --steps 100 --warmup 10 $LEAN > $OUT/trace.log 2>&1
timeout -k 10 240 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python bench.py --steps 20 --warmup 5 --preroll 100 $LEAN > $OUT/pmc_fetch.log 2>&1
timeout -k 10 240 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python bench.py --steps 20 --warmup 5 --preroll 100 $LEAN > $OUT/pmc_write.log 2>&1
# instruction mix of the same launches (SQ counters, one pass)
timeout -k 10 240 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVES --output-format csv -d $OUT/pmc_insts -- python bench.py --steps 20 --warmup 5 --preroll 100 $LEAN > $OUT/pmc_insts.log 2>&1
timeout -k 10 240 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES --output-format csv -d $OUT/pmc_wait -- python bench.py --steps 20 --warmup 5 --preroll 100 $LEAN > $OUT/pmc_wait.log 2>&1 || true
python tools/summarize_profile.py $OUT $OUT/pmc_traffic.json > $OUT/summary.txt
cat $OUT/summary.txt
