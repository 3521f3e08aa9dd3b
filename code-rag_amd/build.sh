#!/bin/bash
# Builds libcoderag_hip.so for gfx950 (cross-compiles without a GPU).  Usage: ./build.sh [extra hipcc flags]
#
# The library is deliberately NOT linked against libamdhip64: the HIP runtime it uses is whichever one is
# already in the process (ffi.py preloads PyTorch's bundled runtime, so torch streams / RCCL and these kernels
# share one runtime; a C/C++ host links -lamdhip64 itself -- see INTEGRATION.md).
set -e
cd "$(dirname "$0")"
mkdir -p lib build
objs=()
for src in csrc/*.hip; do
  obj=build/$(basename "${src%.hip}").o
  hipcc -O3 --offload-arch=gfx950 -fPIC -ffp-contract=off -std=c++17 "$@" -c "$src" -o "$obj"
  objs+=("$obj")
done
g++ -shared -fPIC -o lib/libcoderag_hip.so "${objs[@]}"
# host-side native tokenizer (plain C++, no GPU code): lib/libcoderag_tok.so
g++ -O2 -std=c++17 -shared -fPIC -pthread csrc_host/bpe_tokenizer.cpp -o lib/libcoderag_tok.so
