#!/bin/bash
# One GPU-box round: parity tests, then benches.  Stops after any step that timed out or was killed.
# Usage (through gpurun): bash tools/gpu_round.sh [tag] [bench rows...]
set -o pipefail
tag=${1:-r}; shift
mkdir -p gpurun_out
step() { # name timeout cmd...
  local name=$1 to=$2; shift 2
  echo "=== $name" | tee -a gpurun_out/${tag}_summary.log
  timeout -k 10 "$to" "$@" > gpurun_out/${tag}_${name}.log 2>&1
  local rc=$?
  echo "rc=$rc" | tee -a gpurun_out/${tag}_summary.log
  tail -n 25 gpurun_out/${tag}_${name}.log
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step $name timed out/killed: stopping"; exit $rc; fi
  return 0
}
step smoke 300 python -c "import __graft_entry__ as g; g.smoke()"
step pytest 900 python -m pytest tests -q -m gpu -x --timeout 600
for rows in "$@"; do
  step bench_$rows 600 python bench.py --rows $rows --steps 20 --warmup 5
done
