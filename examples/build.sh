#!/bin/sh
# builds examples/multi_gpu_c_abi against the in-tree libdspgn.so (g++ only: the example speaks the C ABI, nothing else)
set -e
cd "$(dirname "$0")/.."
g++ -std=c++17 -O2 -Iinclude examples/multi_gpu_c_abi.cpp -o examples/multi_gpu_c_abi -Ldsp_slam_amd/lib -ldspgn -Wl,-rpath,"$PWD/dsp_slam_amd/lib" -pthread
