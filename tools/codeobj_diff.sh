#!/bin/bash
# Developer tool: is the device code of two builds of libatcstep.so the same?  Extracts the gfx950 code object of each library and
# compares the disassembly (llvm-objdump -d) — used when source is refactored without a functional change (round 6 prune).
#   bash tools/codeobj_diff.sh <a.so> <b.so>
set -e
OBJDUMP=/opt/rocm/lib/llvm/bin/llvm-objdump
T=$(mktemp -d)
for n in a b; do
  f=$1; shift
  cp "$f" $T/$n.so
  (cd $T && $OBJDUMP --offloading $n.so > /dev/null && $OBJDUMP -d $n.so.0.hipv4-amdgcn-amd-amdhsa--gfx950 | tail -n +3 > $n.dis)
  sha256sum $T/$n.dis | cut -d' ' -f1
done
if cmp -s $T/a.dis $T/b.dis; then echo "device code identical ($(grep -c '^[0-9a-f]* <' $T/a.dis) symbols, $(wc -l < $T/a.dis) lines)"; else echo "device code DIFFERS"; diff $T/a.dis $T/b.dis | head -40; fi
rm -rf $T
