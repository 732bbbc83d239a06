#!/bin/bash
# tools/vbuild_tu.sh TAG "unit.hip [unit.hip ...]" <extra hipcc flags>  ->  tools/_lib<TAG>.so with only those translation units
# rebuilt under the flags (the rest: the default build's objects)
cd "$(dirname "$0")/.."
TAG=$1; UNITS=$2; shift 2
python - "$TAG" "$UNITS" "$@" <<'PY'
import sys
sys.path.insert(0, '.')
from opendrift_amd import build as b
print(b.build_variant(sys.argv[1], sys.argv[2].split(), sys.argv[3:]))
PY
