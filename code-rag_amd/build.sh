#!/bin/bash
# Builds libcoderag_hip.so for gfx950 (cross-compiles without a GPU).  Usage: ./build.sh [extra hipcc flags]
#
# Two libraries come out of the same sources:
#   lib/libcoderag_hip.so        the product: exactly the entry points include/coderag_hip.h declares
#   lib/libcoderag_hip_debug.so  the same + the crh_debug_* entry points (-DCRH_ENABLE_DEBUG): timing ablations and
#                                kernel-selection overrides for tools/ and a few kernel tests; never loaded by the package
# Neither is linked against libamdhip64: the HIP runtime they use is whichever one is already in the process (ffi.py
# preloads PyTorch's bundled runtime, so torch streams / RCCL and these kernels share one runtime; a C/C++ host links
# -lamdhip64 itself -- see INTEGRATION.md).
set -e
cd "$(dirname "$0")"
mkdir -p lib build build/debug
pids=()
objs=()
dobjs=()
for src in csrc/*.hip; do
  base=$(basename "${src%.hip}")
  hipcc -O3 --offload-arch=gfx950 -fPIC -ffp-contract=off -std=c++17 "$@" -c "$src" -o "build/$base.o" &
  pids+=($!)
  hipcc -O3 --offload-arch=gfx950 -fPIC -ffp-contract=off -std=c++17 -DCRH_ENABLE_DEBUG "$@" -c "$src" -o "build/debug/$base.o" &
  pids+=($!)
  objs+=("build/$base.o")
  dobjs+=("build/debug/$base.o")
done
# host-side native tokenizer (plain C++, no GPU code): lib/libcoderag_tok.so
g++ -O2 -std=c++17 -shared -fPIC -pthread csrc_host/bpe_tokenizer.cpp -o lib/libcoderag_tok.so &
pids+=($!)
for p in "${pids[@]}"; do wait "$p"; done
g++ -shared -fPIC -o lib/libcoderag_hip.so "${objs[@]}"
g++ -shared -fPIC -o lib/libcoderag_hip_debug.so "${dobjs[@]}"
