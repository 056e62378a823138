#!/bin/bash
# Build what can be built of ohm_amd/host/ref_adaptor (INTEGRATION.md Level 2) against the reference checkout, and SAY
# what was not: the translation units that need no glm -- the gputil backend, the device
# selection and the binding core that holds the adaptor's logic, 4 of 9, plus the reference's own gpuEventList.cpp -- are
# compiled on every box; the five that include an
# ohm header which includes glm are compiled where glm exists and listed by name as NOT COMPILED where it does not
# (exit 77, the "skipped" code; tests/test_ref_adaptor_build.py).  Never writes a stand-in for glm.
#
#   scripts/build_ref_adaptor.sh [reference checkout = /root/reference] [output dir = build/ref_adaptor]
#   GLM_INCLUDE_DIR=<dir containing glm/glm.hpp>   overrides the search
#   OHM_LIB_DIR=<dir with libohm / libohmutil / liblogutil built with real glm>   also LINKS libohmgpuhip.so
#
# Step 1 compiles the adaptor sources -- the gputil backend, OhmGpu, and with glm the ohm:: half (GpuMap / GpuNdtMap /
# GpuTsdfMap / GpuCache / HipMapBinding) -- to object files against the reference's headers where they lie; the four headers
# the reference's build generates (OhmConfig.h, OhmGpuConfig.h, gpuConfig.h and the export-macro headers) are produced
# from its own templates in the output directory, as ref_adaptor/CMakeLists.txt does.  Step 2 (only with OHM_LIB_DIR)
# links them into libohmgpuhip.so against the reference's core libraries and libohmhip.so.  The full recipe with the
# reference's own GPU tests is the CMakeLists.txt next to the sources.
set -u
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
REF="${1:-/root/reference}"
OUT="${2:-$ROOT/build/ref_adaptor}"
SRC="$ROOT/ohm_amd/host/ref_adaptor"
if [ ! -f "$REF/ohmgpu/GpuMap.h" ]; then
  echo "SKIPPED: no reference checkout at $REF (ohmgpu/GpuMap.h not found)"
  exit 77
fi
GLM=""
for d in "${GLM_INCLUDE_DIR:-}" /usr/include /usr/local/include /opt/rocm/include /usr/include/x86_64-linux-gnu; do
  if [ -n "$d" ] && [ -f "$d/glm/glm.hpp" ]; then GLM="$d"; break; fi
done
if [ -n "$GLM" ]; then echo "glm: $GLM/glm/glm.hpp"; else echo "glm: ABSENT (looked for glm/glm.hpp in \$GLM_INCLUDE_DIR, /usr/include, /usr/local/include, /opt/rocm/include); no stand-in is ever written"; fi
GEN="$OUT/generated"
mkdir -p "$GEN/ohm" "$GEN/ohmgpu" "$GEN/gputil" "$GEN/ohmutil" "$GEN/logutil" "$OUT/obj"
# configure_file(): #cmakedefine X -> /* #undef X */ (no optional feature is switched on), @VAR@ -> empty
configure() { sed -E 's|^#cmakedefine01 ([A-Za-z0-9_]+).*|#define \1 0|; s|^#cmakedefine ([A-Za-z0-9_]+).*|/* #undef \1 */|; s|@[A-Za-z0-9_]+@||g' "$1" > "$2"; }
configure "$REF/ohm/OhmConfig.in.h" "$GEN/ohm/OhmConfig.h"
configure "$REF/ohmgpu/OhmGpuConfig.in.h" "$GEN/ohmgpu/OhmGpuConfig.h"
configure "$REF/gputil/gpuConfig.in.h" "$GEN/gputil/gpuConfig.h"
configure "$REF/ohmutil/OhmUtilConfig.in.h" "$GEN/ohmutil/OhmUtilConfig.h"
configure "$REF/logutil/LogUtilConfig.in.h" "$GEN/logutil/LogUtilConfig.h"
# generate_export_header() for a shared build
export_header() { # file, macro, guard
  printf '#ifndef %s\n#define %s\n#define %s __attribute__((visibility("default")))\n#define %s_NO_EXPORT __attribute__((visibility("hidden")))\n#endif\n' "$3" "$3" "$2" "$2" > "$1"
}
export_header "$GEN/ohm/OhmExport.h" ohm_API OHM_EXPORT_H
export_header "$GEN/ohmgpu/OhmGpuExport.h" ohmgpu_API OHMGPU_EXPORT_H
export_header "$GEN/gputil/gputilExport.h" gputilAPI GPUTIL_EXPORT_H
export_header "$GEN/ohmutil/OhmUtilExport.h" ohmutil_API OHMUTIL_EXPORT_H
export_header "$GEN/logutil/LogUtilExport.h" logutil_API LOGUTIL_EXPORT_H
INC="-I$GEN -I$GEN/ohm -I$GEN/ohmgpu -I$GEN/gputil -I$GEN/ohmutil -I$GEN/logutil -I$REF -I$REF/ohmutil/3rdparty -I$ROOT/include -I$SRC -I$SRC/gputil_hip"
if [ -n "$GLM" ]; then INC="$INC -I$GLM"; fi
# Translation units that need NO glm (the gputil backend, device selection) are compiled on every box; the ones that
# include an ohm header which includes glm only where glm exists.  Every unit's fate is printed: nothing is "checked"
# anywhere else.
NO_GLM_UNITS="OhmGpu.cpp gputil_hip/gputilHip.cpp gputil_hip/gputilHipBuffer.cpp private/HipBindingCore.cpp"
GLM_UNITS="GpuCache.cpp GpuMap.cpp GpuNdtMap.cpp GpuTsdfMap.cpp private/HipMapBinding.cpp"
OBJS=""
COMPILED=""
NOT_COMPILED=""
FAILED=""
compile_unit() { # source path relative to $SRC (or absolute), object name
  local src="$1" obj="$OUT/obj/$2"
  if g++ -std=c++14 -O1 -fPIC -Wall -Dgputil_EXPORTS -Dohmgpuhip_EXPORTS $INC -c "$src" -o "$obj" 2> "$obj.log"; then
    OBJS="$OBJS $obj"; COMPILED="$COMPILED $2"; echo "compiled      $1"
  else
    FAILED="$FAILED $1"; echo "FAILED        $1: $(grep -m1 error "$obj.log" | cut -c1-160)"
  fi
}
for f in $NO_GLM_UNITS; do
  compile_unit "$SRC/$f" "$(echo "$f" | tr '/' '_' | sed 's/\.cpp$/.o/')"
done
compile_unit "$REF/gputil/gpuEventList.cpp" gpuEventList.o   # the reference's own, backend independent
for f in $GLM_UNITS; do
  if [ -n "$GLM" ]; then
    compile_unit "$SRC/$f" "$(echo "$f" | tr '/' '_' | sed 's/\.cpp$/.o/')"
  else
    first=$(g++ -std=c++14 -fsyntax-only $INC "$SRC/$f" 2>&1 | grep -m1 'fatal error' | sed -E 's/.*fatal error: //' | cut -c1-80)
    NOT_COMPILED="$NOT_COMPILED $f"; echo "NOT COMPILED  $f  (stops at: $first)"
  fi
done
echo "COMPILED: $(echo $COMPILED | wc -w) objects under $OUT/obj:$COMPILED"
if [ -n "$FAILED" ]; then
  echo "FAILED:$FAILED"
  exit 1
fi
if [ -n "$NOT_COMPILED" ]; then
  echo "NOT COMPILED (need glm, have never been through a compiler on this box):$NOT_COMPILED"
  # How much source that is: every line of the glm units and of the header only they include, and of those the lines that
  # are statements (not blank, not comment, not a lone brace, not an #include) -- the rest of the adaptor's logic lives in
  # private/HipBindingCore.cpp, which was compiled above and runs on the GPU (tests/test_gpu_binding_core.py).
  TOTAL=0; STATEMENTS=0
  for f in $NOT_COMPILED private/HipMapBinding.h; do
    n=$(wc -l < "$SRC/$f"); st=$(grep -vcE '^[[:space:]]*(//.*|[{}][;]?|#include.*|)[[:space:]]*$' "$SRC/$f")
    TOTAL=$((TOTAL + n)); STATEMENTS=$((STATEMENTS + st))
    printf '   %-28s %4d lines, %4d statements\n' "$f" "$n" "$st"
  done
  echo "UNCOMPILED SOURCE: $TOTAL lines ($STATEMENTS statements) in $(echo $NOT_COMPILED | wc -w) translation units + 1 header; compiled adaptor logic: $(wc -l < "$SRC/private/HipBindingCore.cpp") lines in private/HipBindingCore.cpp"
  echo "SKIPPED: glm absent -- $(echo $NOT_COMPILED | wc -w) of 9 adaptor translation units not compiled"
  exit 77
fi
if [ -n "${OHM_LIB_DIR:-}" ]; then
  g++ -shared -o "$OUT/libohmgpuhip.so" $OBJS -L"$OHM_LIB_DIR" -lohm -lohmutil -llogutil -L"$ROOT/ohm_amd/lib" -lohmhip \
      -Wl,-rpath,"$ROOT/ohm_amd/lib" -Wl,-rpath,"$OHM_LIB_DIR"
  echo "LINKED: $OUT/libohmgpuhip.so"
else
  echo "NOT LINKED: set OHM_LIB_DIR to the reference's core libraries (built with real glm) to link libohmgpuhip.so; ref_adaptor/CMakeLists.txt is the full recipe incl. the reference's GPU tests"
fi
