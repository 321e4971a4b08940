#!/bin/bash
# Register / spill / LDS figures of the kernels: bash tools/kernel_regs.sh [extra -D flags] (device-only compile, ~40 s)
set -e
ROOT=$(cd $(dirname $0)/.. && pwd)
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -std=c++17 -I$ROOT/include "$@" --cuda-device-only -c $ROOT/metadrive_ped_amd/csrc/mdstep.hip -o /tmp/md_dev.o
cd /tmp
/opt/rocm/lib/llvm/bin/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=/tmp/md_dev.o --output=/tmp/md_dev.co --unbundle
/opt/rocm/lib/llvm/bin/llvm-readelf --notes /tmp/md_dev.co | python3 -c '
import sys, re
cur = {}
for line in sys.stdin:
    m = re.match(r"\s*-?\s*\.(\w+):\s*(.*)", line)
    if not m: continue
    k, v = m.group(1), m.group(2).strip()
    if k == "agpr_count" and cur.get("name"):
        cur = {}
    cur[k] = v
    if k == "wavefront_size":
        print("%-60s vgpr %3s sgpr %3s spill v%s s%s lds %s" % (cur.get("name", "?")[:60], cur.get("vgpr_count"), cur.get("sgpr_count"), cur.get("vgpr_spill_count"), cur.get("sgpr_spill_count"), cur.get("group_segment_fixed_size")))
        cur = {}
'
