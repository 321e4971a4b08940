set -e
bash tools/profile_round.sh r03 > gpurun_out/r03_profile.log 2>&1 || { tail -20 gpurun_out/r03_profile.log; exit 1; }
tail -30 gpurun_out/r03_profile.log
for W in safe marl scenario; do
  timeout -k 10 300 python bench.py --workload $W --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/r03/bench_$W.json 2> gpurun_out/r03/bench_$W.err || tail -5 gpurun_out/r03/bench_$W.err
  python -c "import json;d=json.loads(open('gpurun_out/r03/bench_$W.json').read().strip().splitlines()[-1]);print('$W',d['ms_per_step'],d['value'])"
done
