#!/bin/sh
# builds the SIMT-emulator flavour of the engine (tests only)
set -e
cd "$(dirname "$0")"
mkdir -p _build
g++ -std=c++17 -O2 -fPIC -shared -DLPC_SIMT_EMU -I../../lenslesspicam_amd/csrc -I../../include \
    -x c++ ../../lenslesspicam_amd/csrc/lpc_engine.cpp emu.cpp -o _build/liblpc_emu.so -lpthread
