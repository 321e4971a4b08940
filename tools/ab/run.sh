#!/bin/bash
# A/B of library builds on the bench workload: bash tools/ab/run.sh <out> lib1.so lib2.so ...
OUT=$1; shift
mkdir -p $OUT
CACHE_DIR=$(mktemp -d); CACHE=$CACHE_DIR/host_cache.pkl   # a private scratch dir (the pickle is keyed; gpurun_out/ only carries results back)
LEAN="--no-cpu-baseline --no-lane-follow --no-env-api --no-shared-maps --host-cache $CACHE"
python bench.py --steps 100 --warmup 10 $LEAN > $OUT/base.json 2> $OUT/base.err || { tail -5 $OUT/base.err; exit 1; }
python - <<PY
import json; d=json.loads(open("$OUT/base.json").read().strip().splitlines()[-1]); print("base", d["ms_per_step"], d["value"], (d.get("double_buffered") or {}).get("ms_per_step"))
PY
for L in "$@"; do
  MD_LIB_PATH=$PWD/$L python bench.py --steps 100 --warmup 10 $LEAN > $OUT/$(basename $L).json 2> $OUT/$(basename $L).err || { tail -5 $OUT/$(basename $L).err; continue; }
  python - <<PY
import json; d=json.loads(open("$OUT/$(basename $L).json").read().strip().splitlines()[-1]); print("$L", d["ms_per_step"], d["value"], (d.get("double_buffered") or {}).get("ms_per_step"))
PY
done
