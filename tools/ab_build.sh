#!/bin/bash
# Compile the working tree's gpdla.hip into build/ab/<name>.so with the product flags (see ab.sh).
set -e
cd "$(dirname "$0")/.."
mkdir -p build/ab
python - "$1" <<'PY'
import subprocess, sys, os
sys.path.insert(0, ".")
from gp_dla_detection_amd import _lib
out = os.path.join("build", "ab", sys.argv[1] + ".so")
subprocess.check_call(["hipcc", *_lib.HIPCC_FLAGS, os.path.join(_lib.CSRC, "gpdla.hip"), "-o", out])
print("built", out)
PY
