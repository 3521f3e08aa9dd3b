#!/bin/bash
# The measurement build for tools/fused_stamps.py: the product's sources with -DCRH_FUSED_STAMPS (phase clocks inside
# k_scan_fused + the entry point that reads them) -> code-rag_amd/lib/libcoderag_hip_stamps.so.  Not part of build.sh.
set -e
cd "$(dirname "$0")/../code-rag_amd"
mkdir -p build/stamps lib
for b in crh_index crh_encoder crh_rerank; do
  hipcc -O3 --offload-arch=gfx950 -fPIC -ffp-contract=off -std=c++17 -DCRH_FUSED_STAMPS -c csrc/$b.hip -o build/stamps/$b.o &
done
wait
g++ -shared -fPIC -o lib/libcoderag_hip_stamps.so build/stamps/*.o
