#!/bin/bash
# Host side of libdungeon_maps_amd.so under AddressSanitizer + UBSan (CPU only: no GPU needed).
#   make -C dungeon_maps_amd/csrc asan      builds csrc/build_asan/libdungeon_maps_amd_asan.so
# then the host-only tests that drive the library's geometry code (window bounds, strip covers,
# launch bounds, ABI surface) run against it with the sanitizer runtime preloaded.
set -e
here=$(cd "$(dirname "$0")/.." && pwd)
make -C $here/dungeon_maps_amd/csrc asan -j4
rt=$(/opt/rocm/lib/llvm/bin/clang++ -print-file-name=libclang_rt.asan-x86_64.so)
[ -f "$rt" ] || rt=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
cd $here
# (python itself is not instrumented: leak reports from the interpreter are switched off)
ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1 \
LD_PRELOAD=$rt DUNGEON_MAPS_AMD_LIB=$here/dungeon_maps_amd/csrc/build_asan/libdungeon_maps_amd_asan.so \
  python -m pytest tests/test_window_geometry.py tests/test_strip_geometry.py tests/test_abi.py -q -x -m "not gpu" "$@"
