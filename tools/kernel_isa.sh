#!/bin/bash
# Disassemble one kernel of the built library and print its register budget and instruction histogram.
#   tools/kernel_isa.sh <substring of the mangled kernel name> [out.s]
# Works on the CPU-only build box (hipcc cross-compiles gfx950); scratch files go to /tmp/odo_isa.
set -e
NAME="$1"
OUT="${2:-/tmp/odo_isa/kernel.s}"
LIB="$(cd "$(dirname "$0")/.." && pwd)/odometry_amd/lib/libodometry_hip.so"
LLVM=/opt/rocm/lib/llvm/bin
mkdir -p /tmp/odo_isa
cd /tmp/odo_isa
rm -f libodometry_hip.so*
cp "$LIB" .
$LLVM/llvm-objdump --offloading libodometry_hip.so > /dev/null
for co in libodometry_hip.so.*gfx950; do
  $LLVM/llvm-readelf --notes "$co" > notes.txt
  if grep -q "name:.*$NAME" notes.txt; then
    grep -A14 "\.name:.*$NAME" notes.txt | grep -E "\.name|vgpr_count|sgpr_count|private_segment_fixed|group_segment_fixed|vgpr_spill"
    $LLVM/llvm-objdump -d "$co" > all.s
    awk -v n="$NAME" '$0 ~ "^[0-9a-f]+ <.*" n {p=1} p {print} p && /s_endpgm/ {exit}' all.s > "$OUT"
    echo "lines: $(wc -l < "$OUT")"
    grep -oE "^\s+[a-z_0-9]+" "$OUT" | sort | uniq -c | sort -rn | head -${TOP:-24}
  fi
done
