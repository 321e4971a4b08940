# Marginal cost of the fused env kernel's stages: builds with one stage left out each (-DMD_ENV_SKIP=bit; NOT the product, results
# differ), timed on the headline bench.  Build here (no GPU needed), run on the box:
#   bash tools/ab/env_knockout.sh build ; gpurun -- 'bash tools/ab/env_knockout.sh run > gpurun_out/env_knockout.txt 2>&1'
D=metadrive_ped_amd/lib/ab
BITS="1 2 4 8 16"
if [ "$1" = build ]; then
  mkdir -p $D
  for b in $BITS; do
    /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -fPIC -shared -fvisibility=hidden -std=c++17 -DMD_ENV_SKIP=$b \
      -Iinclude metadrive_ped_amd/csrc/mdstep.hip -o $D/eskip_$b.so &
    [ $(jobs -r | wc -l) -ge 4 ] && wait -n
  done
  wait
  ls -la $D
else
  echo "1 lidar | 2 IDM of the traffic (waves 1-3, beside the observation) | 4 localisation | 8 agent contacts | 16 observation (wave 0)"
  L=""
  for b in $BITS; do L="$L $D/eskip_$b.so"; done
  bash tools/ab/run.sh gpurun_out/env_knockout $L
fi
