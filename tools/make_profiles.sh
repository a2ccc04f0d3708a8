#!/bin/bash
# Round profile set, run ON THE GPU BOX from the repo root: tools/make_profiles.sh <tag>   (outputs under gpurun_out/<tag>_*; the
# summaries are then copied to profiles/ by hand).  Counter passes are separate rocprofv3 runs with nothing but --pmc (and the
# program directly after --), as MI355X_MICROARCH.md prescribes.
T=${1:-rXX}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out
last_json() { python3 -c "import sys; print([l for l in open(sys.argv[1]) if l.startswith('{')][-1].strip())" "$1"; }
python3 bench.py > $O/${T}_bench.log 2>&1; last_json $O/${T}_bench.log > $O/${T}_bench.json
python3 bench.py --no-alt --no-cpu-baseline --precision f32-mfma > $O/tmp.log 2>&1; last_json $O/tmp.log > $O/${T}_bench_f32_mfma.json
python3 bench.py --no-alt --no-cpu-baseline --no-overlap > $O/tmp.log 2>&1; last_json $O/tmp.log > $O/${T}_bench_no_overlap.json
python3 bench.py --no-alt --no-cpu-baseline --graph > $O/tmp.log 2>&1; last_json $O/tmp.log > $O/${T}_bench_graph.json
python3 bench.py --no-alt --no-cpu-baseline --config detone-bihome > $O/tmp.log 2>&1; last_json $O/tmp.log > $O/${T}_bench_detone.json
python3 bench.py --no-alt --no-cpu-baseline --config detone-bihome --precision bf16 > $O/tmp.log 2>&1; last_json $O/tmp.log > $O/${T}_bench_detone_bf16.json
python3 bench.py --no-alt --no-cpu-baseline --config detone-bihome --precision f32x2 > $O/tmp.log 2>&1; last_json $O/tmp.log > $O/${T}_bench_detone_f32x2.json
python3 bench.py --no-alt --no-cpu-baseline --precision f32x2 > $O/tmp.log 2>&1; last_json $O/tmp.log > $O/${T}_bench_f32x2.json
python3 bench.py --no-alt --no-cpu-baseline --precision f32x3 > $O/tmp.log 2>&1; last_json $O/tmp.log > $O/${T}_bench_f32x3.json
python3 bench.py --no-alt --no-cpu-baseline --steps 200 --warmup 20 --no-roofline > $O/tmp.log 2>&1; last_json $O/tmp.log > $O/${T}_bench_steps200.json
BIHOME_DETERMINISTIC=1 python3 bench.py --no-alt --no-cpu-baseline > $O/tmp.log 2>&1; last_json $O/tmp.log > $O/${T}_bench_deterministic.json
[ -x tools/mfma_clock_probe.bin ] && tools/mfma_clock_probe.bin > $O/${T}_mfma_clock_probe.txt 2>&1
[ -f bihome_amd/libbihome_hip_tuning.so ] && { for m in fwd bnr; do BIHOME_TUNING=1 python3 tools/pc_timeline.py 128,32,64,64 $m; done; BIHOME_TUNING=1 python3 tools/pc_timeline.py 128,8,256,256 fwd; BIHOME_TUNING=1 python3 tools/pc_ablate.py; } > $O/${T}_pc_timeline.txt 2>&1
python3 tools/step_detail.py zeng-bihome 64 > $O/${T}_step_detail.txt 2>&1
python3 bench.py --no-alt --no-cpu-baseline --config zeng-bihome-rgb256 > $O/tmp.log 2>&1; last_json $O/tmp.log > $O/${T}_bench_rgb256.json
python3 bench.py --no-alt --no-cpu-baseline --config zeng-bihome-pds > $O/tmp.log 2>&1; last_json $O/tmp.log > $O/${T}_bench_pds.json
# (--no-roofline since round 6: the roofline leg's stem micro-benchmark would put 184 cold-cache launches of stem7_dgrad_c1_kernel<true> into the
#  averages next to the 25 of the steps - tools/hbm_path_from_csv.py compares that kernel between the two traces)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${T}_kt -o ${T} -- python3 bench.py --no-alt --steps 20 --warmup 5 --no-cpu-baseline --no-overlap --no-roofline > $O/${T}_kt.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/${T}_mfma -o ${T} -- python3 bench.py --no-alt --steps 6 --warmup 2 --no-cpu-baseline --no-overlap > $O/tmp.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/${T}_fetch -o ${T} -- python3 bench.py --no-alt --steps 6 --warmup 2 --no-cpu-baseline --no-overlap > $O/tmp.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/${T}_write -o ${T} -- python3 bench.py --no-alt --steps 6 --warmup 2 --no-cpu-baseline --no-overlap > $O/tmp.log 2>&1
cp $(find $O/${T}_kt -name "*kernel_stats.csv" | head -1) $O/${T}_kernel_stats.csv
# (round 6) the same trace with the warp and its adjoint as their own launches again: the plain stem dgrad's duration, for tools/hbm_path_from_csv.py
BIHOME_WARP_IN_STEM_DGRAD=0 BIHOME_WARP_IN_STEM_FWD=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${T}_kt2 -o ${T} -- python3 bench.py --no-alt --steps 20 --warmup 5 --no-cpu-baseline --no-overlap --no-roofline > $O/${T}_kt2.log 2>&1
cp $(find $O/${T}_kt2 -name "*kernel_stats.csv" | head -1) $O/${T}_kernel_stats_unfolded.csv
python3 tools/hbm_path_from_csv.py $O/${T}_kernel_stats.csv $O/${T}_kernel_stats_unfolded.csv > $O/${T}_hbm_path.txt 2>&1
python3 tools/mfma_busy_summary.py $(find $O/${T}_mfma -name "*counter_collection.csv" | head -1) $O/${T}_mfma_busy.json
python3 tools/pmc_summary.py $(find $O/${T}_fetch -name "*counter_collection.csv" | head -1) $(find $O/${T}_write -name "*counter_collection.csv" | head -1) $O/${T}_pmc_traffic.json
rm -rf $O/${T}_kt $O/${T}_kt2 $O/${T}_mfma $O/${T}_fetch $O/${T}_write $O/tmp.log
ls -la $O | grep ${T}_
for f in $O/${T}_bench*.json; do python3 -c "import json,sys; j=json.load(open(sys.argv[1])); r=j.get('roofline') or {}; print(sys.argv[1], j['ms_per_step'], round(j['value'],1), r.get('kernel'), round(r.get('frac') or 0,3))" $f; done
