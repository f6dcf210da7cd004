#!/bin/bash
# Does a refactoring change the shipped kernels?  Compiles the file's committed version (git HEAD) and the working
# copy to device assembly for gfx950 and compares them instruction for instruction.
#   tools/scratch/isa_same.sh csrc/kernels/conv2.hip
set -e
f=$1
d=$(mktemp -d)
git show HEAD:$f > $d/old_$(basename $f)
for tag in old new; do
  src=$([ $tag = old ] && echo $d/old_$(basename $f) || echo $f)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -I csrc/kernels -S --cuda-device-only -x hip $src -o $d/$tag.s 2>/dev/null
  # drop comments, debug / file directives and the metadata that embeds names of temporaries
  grep -v "__hip_cuid\|^\s*;\|^\s*\.file\|^\s*\.ident\|^\s*\.loc\|\.note\|^\s*\.section\s*\.debug" $d/$tag.s | sed -e 's/;.*$//' > $d/$tag.txt
  echo "$tag: $(grep -c "v_mfma" $d/$tag.txt) MFMAs, $(wc -l < $d/$tag.txt) lines, md5 $(md5sum < $d/$tag.txt | cut -c1-16)"
done
if cmp -s $d/old.txt $d/new.txt; then echo "IDENTICAL device code"; else echo "DIFFERENT"; diff $d/old.txt $d/new.txt | head -20; exit 1; fi
