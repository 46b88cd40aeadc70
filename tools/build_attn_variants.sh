#!/bin/bash
# Kernel A/B experiments only: copies of the library whose attention.hip was compiled with the given -D switches.
#   tools/build_attn_variants.sh name1:"-DA -DB" name2:"-DC" ...   ->  tools/probes/libatt_<name>.so   (loaded through ULL_LIB_PATH)
# (round 6: the switches live in the LAB COPY tools/probes/lab/attention_lab_r05.hip -- the product attention.hip carries no ablation / stamp macro; the two
# compile to byte-identical device code when no switch is given)
set -e
cd "$(dirname "$0")/../u-llava_amd/csrc"
make -j8 >/dev/null
OTHERS=$(ls *.o | grep -E '^[a-z_0-9]+(\.f16)?\.o$' | grep -v '^attention\.o$')
for spec in "$@"; do
  name=${spec%%:*}; defs=${spec#*:}
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value $defs -I. -c ../../tools/probes/lab/attention_lab_r05.hip -o /tmp/att_$name.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/probes/libatt_$name.so /tmp/att_$name.o $OTHERS && echo built $name ) &
done
wait
