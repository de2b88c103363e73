#!/bin/sh
# builds the SIMT-emulator flavours (float32 and float64) of the engine (tests only); one g++ job per
# translation unit and flavour, linked into _build/liblpc_emu[_f64].so
set -e
cd "$(dirname "$0")"
mkdir -p _build/o32 _build/o64
CSRC=../../lenslesspicam_amd/csrc
CXX="g++ -std=c++17 -O2 -fPIC -DLPC_SIMT_EMU -I$CSRC -I../../include"
pids=""
for f in $CSRC/*.cpp emu.cpp; do
  b=$(basename "$f" .cpp)
  $CXX -c -x c++ "$f" -o _build/o32/$b.o & pids="$pids $!"
  $CXX -DLPC_DOUBLE -c -x c++ "$f" -o _build/o64/$b.o & pids="$pids $!"
done
for p in $pids; do wait $p; done
g++ -shared _build/o32/*.o -o _build/liblpc_emu.so -lpthread
g++ -shared _build/o64/*.o -o _build/liblpc_emu_f64.so -lpthread
test -f _build/liblpc_emu.so && test -f _build/liblpc_emu_f64.so
