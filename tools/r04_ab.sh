#!/bin/bash
# A/B of the default library against libbihome_hip_ab.so on the per-layer table: tools/r04_ab.sh <tag>
T=${1:-ab}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out; mkdir -p $O
python3 tools/step_detail.py zeng-bihome 64 > $O/${T}_detail_default.txt 2>&1
BIHOME_LIB_VARIANT=ab python3 tools/step_detail.py zeng-bihome 64 > $O/${T}_detail_ab.txt 2>&1
head -3 $O/${T}_detail_default.txt $O/${T}_detail_ab.txt
grep "halo_kernel<\(true\|false\),32" $O/${T}_detail_default.txt; echo ---; grep "halo_kernel<\(true\|false\),32" $O/${T}_detail_ab.txt
