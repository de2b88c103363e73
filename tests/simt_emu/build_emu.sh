#!/bin/sh
# builds the SIMT-emulator flavours (float32 and float64) of the engine (tests only); one g++ job per
# translation unit and flavour, linked into _build/liblpc_emu[_f64].so.  lpc_module.cpp is not part of the library: it
# is the source of the plan modules, which the library compiles itself (lpc_jit.cpp) into _build/modules/.
set -e
cd "$(dirname "$0")"
mkdir -p _build/o32 _build/o64 _build/modules
CSRC=../../lenslesspicam_amd/csrc
FP=$(cat $CSRC/*.h $CSRC/*.cpp $CSRC/*.inc ../../include/lpc.h emu.cpp | sha1sum | cut -c1-12)
CXX="g++ -std=c++17 -O2 -fPIC -DLPC_SIMT_EMU -I$CSRC -I../../include -DLPC_SRC_FP=\"$FP\""
CRC=$(python3 -c "import sys; sys.path.insert(0, '../..'); from lenslesspicam_amd import build; print('0x%08xu' % build.sources_crc())")
CXX="$CXX -DLPC_SRC_CRC=$CRC"
CXX="$CXX -DLPC_CSRC_REL=\"../../../lenslesspicam_amd/csrc\" -DLPC_INCLUDE_REL=\"../../../include\""
pids=""
for f in $CSRC/*.cpp emu.cpp; do
  b=$(basename "$f" .cpp)
  [ "$b" = lpc_module ] && continue
  $CXX -c -x c++ "$f" -o _build/o32/$b.o & pids="$pids $!"
  $CXX -DLPC_DOUBLE -c -x c++ "$f" -o _build/o64/$b.o & pids="$pids $!"
done
for p in $pids; do wait $p; done
rm -f _build/o32/lpc_module.o _build/o64/lpc_module.o _build/o32/lpc_gd_update_fwd.o _build/o64/lpc_gd_update_fwd.o
g++ -shared -Wl,-soname,liblpc_emu.so _build/o32/*.o -o _build/liblpc_emu.so -lpthread -ldl
g++ -shared -Wl,-soname,liblpc_emu_f64.so _build/o64/*.o -o _build/liblpc_emu_f64.so -lpthread -ldl
# modules built from other sources are dead weight
find _build/modules -name 'lpcmod_*.so' ! -name "lpcmod_emu_${FP}_*" -delete 2>/dev/null || true
test -f _build/liblpc_emu.so && test -f _build/liblpc_emu_f64.so
