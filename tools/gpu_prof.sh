#!/bin/bash
# rocprofv3 passes for the bench (kernel trace + stats, then PMC counters in their own runs).
# Usage: bash tools/gpu_prof.sh <tag> [bench args...]         PMC_LEGS="..." overrides the legs of the counter passes
set -o pipefail
tag=$1; shift
export TMPDIR=/tmp
out=$PWD/gpurun_out/prof_$tag
mkdir -p "$out"
echo "=== kernel trace"
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- python3 bench.py "$@" --no-cpu-baseline --check-rows 0 > "$out/trace.log" 2>&1
rc=$?; echo "rc=$rc"; tail -n 3 "$out/trace.log"
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
extra=()
if [ -n "$PMC_LEGS" ]; then extra=(--legs "$PMC_LEGS"); fi
# counter passes: the encoder leg waits for its stream every 16 batches (<= ~1400 dispatches in flight).  With the whole leg queued at
# once (~27 000 dispatches) all three counter passes of round 3 died with SIGSEGV inside the profiler's dispatch interception, at the
# same instruction, faulting on a page-aligned address (profiles/README.md); the kernel-trace pass above has no such limit.
export CODERAG_BENCH_EMBED_SYNC_EVERY=${CODERAG_BENCH_EMBED_SYNC_EVERY:-16}
export CODERAG_BENCH_DUMP_MAPS="$out/maps.txt"      # (load addresses of the profiled process: a crash's raw frames can then be named)
for pmc in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE" ; do
  name=$(echo $pmc | tr ' ' '_')
  echo "=== pmc $pmc"
  timeout -k 10 500 rocprofv3 --pmc $pmc --output-format csv -d "$out/pmc_$name" -- python3 bench.py "$@" "${extra[@]}" --no-cpu-baseline --check-rows 0 > "$out/pmc_$name.log" 2>&1
  rc=$?; echo "rc=$rc"; grep -v "^    @" "$out/pmc_$name.log" | tail -n 2
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
done
# keep only the small CSVs (stats + counter rows of the scan kernel)
find "$out" -name "*.csv" -size +20M -delete
ls -R "$out" | head -50
