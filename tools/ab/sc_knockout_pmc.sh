export TMPDIR=/tmp
C=$(mktemp -d)/c.pkl
LEAN="--workload scenario --no-cpu-baseline --no-lane-follow --no-env-api --no-shared-maps --sub-batches 0 --host-cache $C"
python bench.py --steps 20 --warmup 5 $LEAN > /dev/null 2>&1
for b in 0 1 2 16 32 64 128; do
  O=gpurun_out/pmc_ko/$b; mkdir -p $O
  if [ $b = 0 ]; then L=""; else L=$PWD/metadrive_ped_amd/lib/ab/skip_$b.so; fi
  MD_LIB_PATH=$L timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVES --output-format csv -d $O -- python bench.py --steps 20 --warmup 5 --preroll 100 $LEAN > $O.log 2>&1
  echo "done $b"
done
python - <<'PY'
import csv, glob, collections
for b in (0, 1, 2, 16, 32, 64, 128):
    fs = glob.glob("gpurun_out/pmc_ko/%d/*/*counter_collection.csv" % b)
    if not fs:
        print(b, "no data"); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.defaultdict(set)
    for r in csv.DictReader(open(fs[0])):
        if "scenario_step" in r["Kernel_Name"]:
            acc[r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
    out = {c: sum(v.values()) / len(v) for c, v in acc.items()}
    print("skip %3d" % b, " ".join("%s %.2fM" % (c.replace("SQ_INSTS_", ""), v / 1e6) for c, v in sorted(out.items())))
PY
