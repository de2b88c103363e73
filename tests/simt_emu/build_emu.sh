#!/bin/sh
# builds the SIMT-emulator flavours (float32 and float64) of the engine (tests only); one g++ job per
# translation unit and flavour, linked into $B/liblpc_emu[_f64].so.  lpc_module.cpp is not part of the library: it
# is the source of the plan modules, which the library compiles itself (lpc_jit.cpp) into $B/modules/.
# build_emu.sh --san: the same two libraries under AddressSanitizer + UndefinedBehaviorSanitizer in _build_san/ (SURVEY
# section 5's sanitizer row): every LDS / global index of the real kernel sources is bounds-checked, the plan modules the
# library compiles for itself included (LPC_MODULE_EXTRA_DEFS hands the flags to the JIT).  Load it with
# LD_PRELOAD=$(gcc -print-file-name=libasan.so) (tests/test_sanitizer.py does).
set -e
cd "$(dirname "$0")"
B=_build
SAN=""
if [ "$1" = "--san" ]; then
  B=_build_san
  SAN="-fsanitize=address,undefined -fno-sanitize-recover=all -fno-omit-frame-pointer -g1"
fi
mkdir -p $B/o32 $B/o64 $B/modules
CSRC=../../lenslesspicam_amd/csrc
FP=$( (cat $CSRC/*.h $CSRC/*.cpp $CSRC/*.inc ../../include/lpc.h emu.cpp; echo "$SAN") | sha1sum | cut -c1-12)
CXX="g++ -std=c++17 -O2 -fPIC $SAN -DLPC_SIMT_EMU -I$CSRC -I../../include -DLPC_SRC_FP=\"$FP\""
CRC=$(python3 -c "import sys; sys.path.insert(0, '../..'); from lenslesspicam_amd import build; print('0x%08xu' % build.sources_crc())")
CXX="$CXX -DLPC_SRC_CRC=$CRC"
CXX="$CXX -DLPC_CSRC_REL=\"../../../lenslesspicam_amd/csrc\" -DLPC_INCLUDE_REL=\"../../../include\""
if [ -n "$SAN" ]; then   # (a header, not -D: the value contains spaces)
  echo "#define LPC_MODULE_EXTRA_DEFS \"$SAN\"" > $B/san_defs.h
  CXX="$CXX -include $B/san_defs.h"
fi
pids=""
for f in $CSRC/*.cpp emu.cpp; do
  b=$(basename "$f" .cpp)
  [ "$b" = lpc_module ] && continue
  $CXX -c -x c++ "$f" -o $B/o32/$b.o & pids="$pids $!"
  [ -z "$SAN" ] && { $CXX -DLPC_DOUBLE -c -x c++ "$f" -o $B/o64/$b.o & pids="$pids $!"; }     # (--san: float32 only)
done
for p in $pids; do wait $p; done
rm -f $B/o32/lpc_module.o $B/o64/lpc_module.o $B/o32/lpc_gd_update_fwd.o $B/o64/lpc_gd_update_fwd.o
g++ -shared $SAN -Wl,-soname,liblpc_emu.so $B/o32/*.o -o $B/liblpc_emu.so -lpthread -ldl
[ -z "$SAN" ] && g++ -shared -Wl,-soname,liblpc_emu_f64.so $B/o64/*.o -o $B/liblpc_emu_f64.so -lpthread -ldl
# modules built from other sources are dead weight
find $B/modules -name 'lpcmod_*.so' ! -name "lpcmod_emu_${FP}_*" -delete 2>/dev/null || true
test -f $B/liblpc_emu.so && { [ -n "$SAN" ] || test -f $B/liblpc_emu_f64.so; }
