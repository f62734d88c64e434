#!/bin/bash
# Compile the working tree's gpdla.hip into build/ab/<name>.so with the product flags (see ab.sh);
# further arguments are extra compiler flags (e.g. -DGPDLA_SLIM_NOWAIT).
set -e
cd "$(dirname "$0")/.."
mkdir -p build/ab
python - "$@" <<'PY'
import subprocess, sys, os
sys.path.insert(0, ".")
from gp_dla_detection_amd import _lib
out = os.path.join("build", "ab", sys.argv[1] + ".so")
subprocess.check_call(["hipcc", *_lib.HIPCC_FLAGS, *sys.argv[2:], os.path.join(_lib.CSRC, "gpdla.hip"), "-o", out])
print("built", out)
PY
