#!/bin/sh
# builds the SIMT-emulator flavours (float32 and float64) of the engine (tests only)
set -e
cd "$(dirname "$0")"
mkdir -p _build
CXX="g++ -std=c++17 -O2 -fPIC -shared -DLPC_SIMT_EMU -I../../lenslesspicam_amd/csrc -I../../include"
$CXX -x c++ ../../lenslesspicam_amd/csrc/lpc_engine.cpp emu.cpp -o _build/liblpc_emu.so -lpthread &
$CXX -DLPC_DOUBLE -x c++ ../../lenslesspicam_amd/csrc/lpc_engine.cpp emu.cpp -o _build/liblpc_emu_f64.so -lpthread &
wait
test -f _build/liblpc_emu.so && test -f _build/liblpc_emu_f64.so
