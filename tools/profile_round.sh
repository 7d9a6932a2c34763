#!/bin/bash
# One round's judged profile set, on the GPU box (through gpurun):   bash tools/profile_round.sh r05
# -> gpurun_out/prof_<tag>{,_resnet,_agegender,_f32}/ (tools/gpu_pmc.sh: kernel-trace stats + one PMC group per pass), the per-LAYER
# tables of the four configs and the default bench line of the same box.  Then, here:  bash tools/profile_round.sh r05 collect
# (bench.py marks its traffic figures stale unless profiles/<tag>_traffic.json carries the hash of the sources it runs on: for a bench line
# with fresh traffic run  `profile_round.sh <tag>; profile_round.sh <tag> collect; python bench.py > gpurun_out/prof_<tag>/bench_line.json`
# in ONE gpurun call -- gpurun_out/ does not travel to the box -- and collect again here.)
set -u
TAG=${1:-r05}
R=${GRAFT_REPO_ROOT:-$(pwd)}
if [ "${2:-}" = "collect" ]; then
  fw() { grep -m1 '^forwards' "gpurun_out/prof_${TAG}$1/layers.txt" | awk '{print $2}'; }
  python3 tools/make_profile_summary.py gpurun_out/prof_${TAG} ${TAG}
  python3 tools/make_profile_summary.py gpurun_out/prof_${TAG}_resnet ${TAG}_resnet50 "$(fw _resnet)"
  python3 tools/make_profile_summary.py gpurun_out/prof_${TAG}_agegender ${TAG}_agegender "$(fw _agegender)"
  python3 tools/make_profile_summary.py gpurun_out/prof_${TAG}_f32 ${TAG}_mobilenet_f32 "$(fw _f32)"
  [ -d gpurun_out/prof_${TAG}_rnf32 ] && python3 tools/make_profile_summary.py gpurun_out/prof_${TAG}_rnf32 ${TAG}_resnet50_f32 "$(fw _rnf32)" 32
  cp gpurun_out/prof_${TAG}/layers.txt profiles/${TAG}_layers.txt
  cp gpurun_out/prof_${TAG}_resnet/layers.txt profiles/${TAG}_resnet50_layers.txt
  cp gpurun_out/prof_${TAG}_agegender/layers.txt profiles/${TAG}_agegender_layers.txt
  cp gpurun_out/prof_${TAG}_f32/layers.txt profiles/${TAG}_mobilenet_f32_layers.txt
  [ -s gpurun_out/prof_${TAG}_rnf32/layers.txt ] && cp gpurun_out/prof_${TAG}_rnf32/layers.txt profiles/${TAG}_resnet50_f32_layers.txt
  [ -s gpurun_out/prof_${TAG}/bench_line.json ] && cp gpurun_out/prof_${TAG}/bench_line.json profiles/${TAG}_bench_line.json
  exit 0
fi
cd "$R"
HEAD="--steps 20 --warmup 5 --no-config5 --no-other-configs --no-pipeline --no-latency --no-cpu-baseline --no-sustained"
bash tools/gpu_pmc.sh gpurun_out/prof_${TAG} python3 $R/bench.py $HEAD
bash tools/gpu_pmc.sh gpurun_out/prof_${TAG}_resnet python3 $R/tools/bench_configs.py resnet50
bash tools/gpu_pmc.sh gpurun_out/prof_${TAG}_agegender python3 $R/tools/bench_configs.py agegender
bash tools/gpu_pmc.sh gpurun_out/prof_${TAG}_f32 python3 $R/tools/bench_configs.py mobilenet_f32
BC_STEPS=5 bash tools/gpu_pmc.sh gpurun_out/prof_${TAG}_rnf32 python3 $R/tools/bench_configs.py resnet50f32      # the fp32-grade ResNet mode, batch 32
cd "$R"
python3 tools/bench_configs.py mobilenet192 > gpurun_out/prof_${TAG}/layers.txt 2>&1
python3 tools/bench_configs.py resnet50 > gpurun_out/prof_${TAG}_resnet/layers.txt 2>&1
python3 tools/bench_configs.py agegender > gpurun_out/prof_${TAG}_agegender/layers.txt 2>&1
python3 tools/bench_configs.py mobilenet_f32 > gpurun_out/prof_${TAG}_f32/layers.txt 2>&1
BC_STEPS=5 python3 tools/bench_configs.py resnet50f32 > gpurun_out/prof_${TAG}_rnf32/layers.txt 2>&1
for d in "" _resnet _agegender _f32 _rnf32; do python3 -c "import bench; print(bench.csrc_hash())" > gpurun_out/prof_${TAG}$d/csrc_hash.txt; done      # the sources these counters belong to
