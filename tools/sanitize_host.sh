#!/bin/bash
# Host sources of the library (planner, reference-order model, .graph files, object constructors, C-ABI glue) built with
# -fsanitize=address,undefined by g++, linked with tests/support/solver_nodevice.cpp in place of the HIP translation unit, and the
# CPU tests that exercise them run against that build ("Build: ASan/UBSan for host C++").
#     tools/sanitize_host.sh            every host-side CPU test (three minutes under the sanitizers)
#     tools/sanitize_host.sh --quick    the slice tests/test_sanitized_host.py runs under -m "not gpu" (under a minute)
set -euo pipefail
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=${SANITIZE_OUT:-$ROOT/build/sanitize}
mkdir -p "$OUT"
LIB=$OUT/libaprilsam_amd_hostsan.so
SRC="capi.cpp host_objects.cpp ordering.cpp symbolic.cpp refmodel.cpp graph_io.cpp errors.cpp"
NEWEST=$(ls -t $(for f in $SRC; do echo $ROOT/aprilsam_amd/csrc/$f; done) $ROOT/aprilsam_amd/csrc/*.h $ROOT/tests/support/solver_nodevice.cpp $ROOT/include/aprilsam_amd.h | head -1)
if [ ! -e "$LIB" ] || [ "$NEWEST" -nt "$LIB" ]; then
    g++ -std=c++17 -O1 -g -fno-omit-frame-pointer -fsanitize=address,undefined -fno-sanitize-recover=undefined -fPIC -shared -pthread -Wall -Wno-unused-function \
        $(for f in $SRC; do echo $ROOT/aprilsam_amd/csrc/$f; done) $ROOT/tests/support/solver_nodevice.cpp -o "$LIB"
fi
SELECT="not fail_loudly_without_gpu and not selftest and not exports_every_declared and not sanitized"
if [ "${1:-}" = "--quick" ]; then shift; SELECT="$SELECT and not m3500 and not prefix1088 and not step_by_step and not number_of_planner_threads"; fi
ASAN_RT=$(g++ -print-file-name=libasan.so)
cd "$ROOT"
# (leak detection off: python itself "leaks" at exit; everything else on, first error ends the run)
LD_PRELOAD=$ASAN_RT ASAN_OPTIONS=detect_leaks=0:halt_on_error=1:abort_on_error=0 UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1 \
APRILSAM_AMD_LIB=$LIB APRILSAM_AMD_SANITIZED=1 \
    python -m pytest -x -q -p no:cacheprovider tests/test_plan.py tests/test_refmodel.py tests/test_graph_io.py tests/test_abi.py \
        -k "$SELECT" "$@"
