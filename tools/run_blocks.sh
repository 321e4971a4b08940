# A/B of the env workgroup size: builds of the same source with -DMD_ENV_BLOCK=128 / 64 next to the default 256
# (build them first: hipcc ... -DMD_ENV_BLOCK=128 -o metadrive_ped_amd/lib/libmdstep_b128.so).
set -e
mkdir -p gpurun_out/ab
python bench.py --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/ab/b256.json
for b in 128; do
  MD_LIB_PATH=$PWD/metadrive_ped_amd/lib/libmdstep_b$b.so python bench.py --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/ab/b$b.json
done
MD_LIB_PATH=$PWD/metadrive_ped_amd/lib/libmdstep_b128.so timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x 2>&1 | tail -3
python - <<'PY'
import json
for b in (256, 128):
    d = json.load(open("gpurun_out/ab/b%d.json" % b))
    print("block", b, "value", d["value"], "ms/step", d["ms_per_step"], "launch us", d["roofline"]["avg_launch_us"], "copy GB/s", d["roofline"]["peak_attainable"])
PY
