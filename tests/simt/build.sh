#!/bin/sh
# TEST INFRASTRUCTURE ONLY: compile the product's kernel sources against the SIMT
# emulator (tests/simt/hip/hip_runtime.h) into tests/simt/libzipnn_simt.so, so that the
# CPU test-suite can exercise kernel logic where no GPU exists.  Never loaded by zipnn_amd/.
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
SRC="$HERE/../../zipnn_amd/csrc"
FILES=""
for f in "$SRC"/*.hip; do FILES="$FILES -x c++ $f"; done
g++ -O1 -g -std=c++17 -fPIC -shared -w -I"$HERE" -DZN_SIMT_EMUL=1 ${ZN_SIMT_EXTRA:-} $FILES -o "$HERE/libzipnn_simt.so"      # (ZN_SIMT_EXTRA: -D switches of a variant under test)
echo "built $HERE/libzipnn_simt.so"
