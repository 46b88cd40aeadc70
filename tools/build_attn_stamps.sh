#!/bin/bash
# Debug build for tools/attn_phase_times.py: the library with attention.hip compiled -DULL_ATTN_STAMPS (per-wave phase cycle totals).
# (round 6: the switches live in the LAB COPY tools/probes/lab/attention_lab_r05.hip -- the product attention.hip carries no ablation / stamp macro; the two
# compile to byte-identical device code when no switch is given)
set -e
cd "$(dirname "$0")/../u-llava_amd/csrc"
make -j8 >/dev/null
OTHERS=$(ls *.o | grep -E '^[a-z_0-9]+(\.f16)?\.o$' | grep -v '^attention\.o$')
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -DULL_ATTN_STAMPS "$@" -I. -c ../../tools/probes/lab/attention_lab_r05.hip -o /tmp/attention_stamps.o
mkdir -p ../../tools/debug
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/debug/libullava_attn_stamps.so /tmp/attention_stamps.o $OTHERS
echo built tools/debug/libullava_attn_stamps.so
