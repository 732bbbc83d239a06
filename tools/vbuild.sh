#!/bin/bash
# tools/vbuild.sh TAG "<extra hipcc flags>"  ->  tools/_lib<TAG>.so  (objects under tools/_obj<TAG>; what-if / A-B builds)
cd "$(dirname "$0")/.."
TAG=$1; shift
python - "$TAG" $@ <<'PY'
import sys
sys.path.insert(0, '.')
from opendrift_amd import build as b
v = sys.argv[1]
print(b.build(force=True, extra_flags=sys.argv[2:], lib='tools/_lib%s.so' % v, objdir='tools/_obj%s' % v))
PY
