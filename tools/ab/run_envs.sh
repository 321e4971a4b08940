#!/bin/bash
# A/B of library builds at several env counts: bash tools/ab/run_envs.sh <out> "1792 2048 2304" lib1.so ...
OUT=$1; ENVS=$2; shift; shift
mkdir -p $OUT
CACHE_DIR=$(mktemp -d)
for E in $ENVS; do
  LEAN="--envs $E --no-cpu-baseline --no-lane-follow --no-env-api --no-shared-maps --sub-batches 0 --steps 100 --warmup 10 --host-cache $CACHE_DIR/host_cache_$E.pkl"
  python bench.py $LEAN > $OUT/base_$E.json 2> $OUT/base_$E.err
  python - <<PY
import json; d=json.loads(open("$OUT/base_$E.json").read().strip().splitlines()[-1]); print("$E base", d["ms_per_step"], d["value"])
PY
  for L in "$@"; do
    MD_LIB_PATH=$PWD/$L python bench.py $LEAN > $OUT/$(basename $L)_$E.json 2> $OUT/$(basename $L)_$E.err
    python - <<PY
import json; d=json.loads(open("$OUT/$(basename $L)_$E.json").read().strip().splitlines()[-1]); print("$E $L", d["ms_per_step"], d["value"])
PY
  done
done
