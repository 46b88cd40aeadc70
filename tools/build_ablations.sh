#!/bin/bash
# Kernel A/B experiments only: links copies of the library whose GEMM was compiled with one ULL_ABL_* switch (results are garbage with
# most of them; they answer "what does the K-loop cost without X").  Output: tools/probes/lib_<name>.so, loaded through ULL_LIB_PATH.
# (round 6: the switches live in the LAB COPY tools/probes/lab/gemm_lab_r05.hip -- the product gemm.hip carries no ablation / stamp macro; the two
# compile to byte-identical device code when no switch is given)
set -e
cd "$(dirname "$0")/../u-llava_amd/csrc"
OTHERS=$(ls *.o | grep -E '^[a-z_0-9]+(\.f16)?\.o$' | grep -v '^gemm\.o$')     # the Makefile's objects only (no -save-temps leftovers)
for abl in "$@"; do
  # A+B = both switches; a switch with '=' is a schedule knob: BAR_A=16 -> -DULL_W4_BAR_A=16
  DEFS=$(echo $abl | tr '+' '\n' | sed -e '/=/s/^/-DULL_W4_/' -e '/^W4_/s/^/-DULL_/' -e '/^-D/!s/^/-DULL_ABL_/' | tr '\n' ' ')
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value $DEFS -I. -c ../../tools/probes/lab/gemm_lab_r05.hip -o /tmp/gemm_$abl.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/probes/lib_$abl.so /tmp/gemm_$abl.o $OTHERS
done
