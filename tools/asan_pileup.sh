#!/bin/bash
# CPU AddressSanitizer pass over the native loader / pileup encoder: builds libdl4vc_loader.so with -fsanitize=address
# into a scratch directory and runs the native-pileup and native-loader tests against it.  CPU only (GPU sanitizers are not
# available on the pool).  usage: tools/asan_pileup.sh
set -e
cd "$(dirname "$0")/.."
out=$(mktemp -d)
g++ -O1 -g -std=c++17 -fPIC -shared -fsanitize=address -fno-omit-frame-pointer dl4vc_amd/csrc/dan_loader.cpp dl4vc_amd/csrc/dan_pileup.cpp \
    -o "$out/libdl4vc_loader.so" -lz -ldl -lpthread
asan=$(g++ -print-file-name=libasan.so)
DL4VC_LOADER_LIB="$out/libdl4vc_loader.so" LD_PRELOAD="$asan" ASAN_OPTIONS=detect_leaks=0 \
    python -m pytest tests/test_pileup_native.py tests/test_native_loader.py -q -x -k "not faster" "$@"
rm -rf "$out"
