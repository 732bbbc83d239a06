#!/bin/bash
# tools/ab_build.sh "<extra hipcc flags for A>" "<extra hipcc flags for B>"  ->  tools/_libA.so, tools/_libB.so
cd "$(dirname "$0")/.."
for v in A B; do
  if [ $v = A ]; then F="$1"; else F="$2"; fi
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared $F -Iinclude -o tools/_lib$v.so opendrift_amd/csrc/odrift.hip 2>/dev/null &
done
wait
ls -la tools/_lib?.so
