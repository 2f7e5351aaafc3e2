#!/bin/bash
# Builds the kernel-logic simulator (TEST INFRASTRUCTURE, see tests/hostsim/hostsim.h):
# the same source as libcdbg.so, compiled with g++ and -DCDBG_HOSTSIM so that workgroup
# threads run as fibers on the CPU.  Never loaded by bcalm_amd/.
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
ROOT="$(cd "$HERE/../.." && pwd)"
mkdir -p "$HERE/_build"
g++ -O2 -g -std=c++17 -Wall -Wno-unused-function -Wno-unknown-pragmas -Wno-unused-variable -DCDBG_HOSTSIM -shared -fPIC \
    -I"$ROOT/include" -I"$HERE" "$ROOT/bcalm_amd/csrc/cdbg_impl.cpp" -o "$HERE/_build/libcdbg_hostsim.so"
g++ -O2 -std=c++17 -I"$ROOT/include" "$ROOT/bcalm_amd/host/bcalm_main.cpp" -o "$HERE/_build/bcalm_hostsim" \
    -L"$HERE/_build" -lcdbg_hostsim -lz -Wl,-rpath,'$ORIGIN'
echo "built $HERE/_build/libcdbg_hostsim.so"
