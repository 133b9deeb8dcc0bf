#!/bin/bash
# Developer tool (CPU only): the emulated suite — the product's kernel sources and host code compiled by g++ against tests/simt — under UBSan and under ASan.
# GPU AddressSanitizer is not available on the pool; this is the sanitizer coverage the kernels' index arithmetic and the host paths get.
#   scripts/simt_sanitize.sh [ubsan|asan] [pytest -k expression]
# The sanitized library takes the place of tests/simt/libzipnn_simt.so for the run and the plain one is rebuilt afterwards.
set -u
KIND="${1:-ubsan}"; KEXPR="${2:-}"
R="$(cd "$(dirname "$0")/.." && pwd)"; cd "$R/tests/simt"
FILES=""; for f in ../../zipnn_amd/csrc/*.hip; do FILES="$FILES -x c++ $f"; done
if [ "$KIND" = asan ]; then
  SAN="-fsanitize=address -fno-omit-frame-pointer"; PRE="$(gcc -print-file-name=libasan.so)"
  export ASAN_OPTIONS=detect_leaks=0:detect_stack_use_after_return=0:halt_on_error=1      # (the lanes are ucontext fibers on heap stacks; python itself is not instrumented)
else
  # (alignment: the EMULATED build's stand-ins for the device's unaligned vector stores are plain struct stores — x86 does not care, the device code uses aligned(1) types)
  SAN="-fsanitize=undefined,bounds -fno-sanitize=alignment"; PRE=""
  export UBSAN_OPTIONS=print_stacktrace=0:halt_on_error=0
fi
g++ -O1 -g -std=c++17 -fPIC -shared -w -I. -DZN_SIMT_EMUL=1 $SAN $FILES -o libzipnn_simt.so || exit 1
cd "$R"
LOG="$(mktemp)"
if [ -n "$KEXPR" ]; then LD_PRELOAD="$PRE" python -m pytest tests/test_kernels_simt.py tests/test_plugin_simt.py tests/test_legacy_weights.py -x -q -k "$KEXPR" > "$LOG" 2>&1
else LD_PRELOAD="$PRE" python -m pytest tests/test_kernels_simt.py tests/test_plugin_simt.py tests/test_legacy_weights.py -x -q > "$LOG" 2>&1; fi
RC=$?
echo "pytest rc=$RC"; tail -2 "$LOG"
echo "sanitizer reports:"; grep -E "runtime error|ERROR: AddressSanitizer" "$LOG" | sed 's/0x[0-9a-f]*/ADDR/g' | sort | uniq -c | sort -rn | head -20
sh tests/simt/build.sh > /dev/null && echo "plain emulated library rebuilt"
