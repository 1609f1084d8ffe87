#!/bin/sh
# builds tests/embed/embed_harness (g++, pybind11 embed); no Eigen/OpenCV needed
set -e
cd "$(dirname "$0")"
g++ -O1 -std=c++17 embed_harness.cpp $(python3 -m pybind11 --includes) $(python3-config --embed --ldflags) -o embed_harness
