#!/bin/bash
# Builds libcoderag_hip.so for gfx950 (cross-compiles without a GPU).  Usage: ./build.sh [extra hipcc flags]
set -e
cd "$(dirname "$0")"
mkdir -p lib
hipcc -O3 --offload-arch=gfx950 -fPIC -shared -ffp-contract=off -std=c++17 "$@" -o lib/libcoderag_hip.so csrc/*.hip
