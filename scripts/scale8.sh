#!/bin/bash
# The first multi-GPU box: everything in one go (scripts/scale8.py has the details).  On the GPU box:
#     bash scripts/scale8.sh            -> gpurun_out/scale8.json (copy it to profiles/rNN/scale8.json)
# Smaller dry run (a one-GPU box, minutes):  SCALE8_GB=4 SCALE8_DEPTH=1 bash scripts/scale8.sh
set -u
cd "$(dirname "$0")/.."
export TMPDIR=${TMPDIR:-/tmp} HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out
python3 scripts/scale8.py "${SCALE8_SHM:-/dev/shm}" gpurun_out/scale8.json
