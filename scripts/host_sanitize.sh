#!/bin/bash
# CPU only: builds the library's HOST code (device code compiled as usual, not instrumented) and tests/native/host_sanitize_check.cpp
# with AddressSanitizer + UndefinedBehaviorSanitizer and runs the check on host-only contexts.  ~4 min on 8 cores.
# Output: build/host_sanitize.log (copy under profiles/rNN/ when it is evidence).
set -u
R=$(cd "$(dirname "$0")/.." && pwd)
cd $R/pbrt-rs_amd/csrc
SAN="-Xarch_host -fsanitize=address -Xarch_host -fsanitize=undefined -Xarch_host -fno-omit-frame-pointer"
make -j8 OBJDIR=../../build/obj_san OUT=../../build/libpbrt_mi355_san.so EXTRA="$SAN -g" > $R/build/host_sanitize_build.log 2>&1 || { tail -20 $R/build/host_sanitize_build.log; exit 1; }
cd $R
/opt/rocm/bin/hipcc -O1 -g -std=c++17 -fsanitize=address -fsanitize=undefined -fno-omit-frame-pointer -Iinclude -o build/host_sanitize_check tests/native/host_sanitize_check.cpp \
  -Lbuild -lpbrt_mi355_san -Wl,-rpath,$R/build >> build/host_sanitize_build.log 2>&1 || { tail -20 build/host_sanitize_build.log; exit 1; }
ASAN_OPTIONS=detect_leaks=1:abort_on_error=0 UBSAN_OPTIONS=print_stacktrace=1 ./build/host_sanitize_check pbrt-rs_amd/data/cie_tables.bin > build/host_sanitize.log 2>&1
echo "exit code $?" >> build/host_sanitize.log
tail -30 build/host_sanitize.log
