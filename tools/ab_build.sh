#!/bin/bash
# tools/ab_build.sh "<extra hipcc flags for A>" "<extra hipcc flags for B>"  ->  tools/_libA.so, tools/_libB.so
# (the four translation units of each variant compile in parallel; objects under tools/_objA, tools/_objB)
cd "$(dirname "$0")/.."
for v in A B; do
  if [ $v = A ]; then F="$1"; else F="$2"; fi
  python - "$v" $F <<'PY' &
import sys
sys.path.insert(0, '.')
from opendrift_amd import build as b
v = sys.argv[1]
print(b.build(force=True, extra_flags=sys.argv[2:], lib='tools/_lib%s.so' % v, objdir='tools/_obj%s' % v))
PY
done
wait
ls -la tools/_lib?.so
