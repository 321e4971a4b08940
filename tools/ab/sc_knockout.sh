# Marginal cost of the scenario kernel's stages: builds with one stage left out each (-DMD_SC_SKIP=bit; NOT the product, results
# differ), timed on the scenario bench.  Build here (no GPU needed), run on the box:
#   bash tools/ab/sc_knockout.sh build ; gpurun -- 'bash tools/ab/sc_knockout.sh run > gpurun_out/sc_knockout.txt 2>&1'
D=metadrive_ped_amd/lib/ab
BITS="1 2 4 8 16 32 64 128"
if [ "$1" = build ]; then
  mkdir -p $D
  for b in $BITS; do
    /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -fPIC -shared -fvisibility=hidden -std=c++17 -DMD_SC_SKIP=$b \
      -Iinclude metadrive_ped_amd/csrc/mdstep.hip -o $D/skip_$b.so &
    [ $(jobs -r | wc -l) -ge 4 ] && wait -n
  done
  wait
  ls -la $D
else
  echo "1 lidar | 2 detectors (waves 2-3) | 4 agent contacts (wave 1) | 8 agent projection | 16 front-search pairs | 32 whole per-vehicle preparation | 64 traffic manager after_step | 128 integration"
  L=""
  for b in $BITS; do L="$L $D/skip_$b.so"; done
  bash tools/ab/run_sc.sh gpurun_out/sc_knockout $L
fi
