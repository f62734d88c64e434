#!/bin/bash
# Developer aid: compile gpdla.hip to /tmp (the in-tree libgpdla.so is left alone), print the sweep kernel's resource usage, and dump its ISA to
# /tmp/gpdla_asm/sweep.s with MFMA / scratch line numbers.
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
SRC=$ROOT/gp_dla_detection_amd/csrc/gpdla.hip
FLAGS="--offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared -std=c++17 -no-hip-rt -Wno-inline-asm"
KERNEL=${1:-_ZN5gpdla7k_sweepIdLi8ELi14ELi1ELi8ELi13ELi3EEEvNS_9SweepArgsE}
mkdir -p /tmp/gpdla_asm
hipcc $FLAGS -Rpass-analysis=kernel-resource-usage $SRC -o /tmp/gpdla_asm/lib_remarks.so 2>&1 \
  | grep -E "error|$KERNEL" -A9 | grep -E "error|Name|VGPRs:|Spill|ScratchSize" || true
mkdir -p /tmp/gpdla_asm && cd /tmp/gpdla_asm
hipcc $FLAGS -save-temps $SRC -o /tmp/gpdla_asm/lib.so 2>/dev/null
S=gpdla-hip-amdgcn-amd-amdhsa-gfx950.s
awk -v k="^$KERNEL:" '$0 ~ k {f=1} f{print} /^\.Lfunc_end/{if(f)exit}' $S > sweep.s
echo "lines: $(wc -l < sweep.s)"
echo "MFMA: $(grep -n 'v_mfma' sweep.s | awk -F: '{print $1}' | tr '\n' ' ')"
echo "SCRATCH: $(grep -n 'scratch_' sweep.s | awk -F: '{print $1}' | tr '\n' ' ')"
echo "CALLS: $(grep -n 's_swappc' sweep.s | awk -F: '{print $1}' | tr '\n' ' ')"
