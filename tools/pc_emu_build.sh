#!/bin/bash
# Round 6: timing-only variants of conv3x3_pc_kernel with fewer fragment reads per MFMA (PC_EMU, see csrc/conv3x3_pc.hip), each linked with
# the tuning build's other objects into bihome_amd/libbihome_hip_emu<v>.so (BIHOME_LIB_VARIANT=emu<v>).  Run `make -C bihome_amd/csrc tuning` first.
set -e
cd "$(dirname "$0")/../bihome_amd/csrc"
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -Wall -Wno-unused-function -DBH_TUNING"
for v in ${@:-0 1 2 3}; do
  /opt/rocm/bin/hipcc $FLAGS -DPC_EMU=$v -c conv3x3_pc.hip -o tuning_obj/conv3x3_pc_emu$v.o &
done
wait
for v in ${@:-0 1 2 3}; do
  objs=$(ls tuning_obj/*.o | grep -v conv3x3_pc)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs tuning_obj/conv3x3_pc_emu$v.o -o ../libbihome_hip_emu$v.so
done
ls -la ../libbihome_hip_emu*.so
