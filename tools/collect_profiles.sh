#!/bin/bash
# usage (on the GPU box, from the repo root): bash tools/collect_profiles.sh <round tag, e.g. r03>
# Writes everything under gpurun_out/<tag>_profiles/; copy the summaries into profiles/ afterwards (tools/finish_profiles.py does).
#   1. rocprofv3 --kernel-trace --stats of the headline bench (graph replay)
#   2. PMC passes of the headline bench, one counter group per run (FETCH_SIZE / WRITE_SIZE / MFMA busy), as MI355X_MICROARCH.md prescribes
#   3. bench records of every workload (+ per-launch table of the headline config), taken with 1. and 2. summarised in profiles/
#   4. the per-launch table with three passes in flight (tools/throughput_profile.py)
tag=${1:-r03}
out=$(realpath -m gpurun_out/${tag}_profiles)
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
# rocprofv3 passes: the headline's kernels (throughput-mode plan) ONE PASS AT A TIME -- counter collection serialises kernels anyway, and
# the per-kernel durations of the trace are then comparable with bench.py's own per-launch hip-event times (which are taken one
# launch after the other); stats_lanes = the default command (3 passes in flight: kernel durations overlap and stretch)
B="python bench.py --no-cpu-baseline --no-e2e --no-roofline --lanes 1 --plan-lanes 3"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o p -- $B > $out/stats.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_lanes -o p -- python bench.py --no-cpu-baseline --no-e2e --no-roofline > $out/stats_lanes.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -o p -- $B > $out/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -o p -- $B > $out/pmc_write.log 2>&1
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $out/pmc_mfma -o p -- $B > $out/pmc_mfma.log 2>&1
# the summaries of these passes into profiles/ ON THE BOX (stamped with the hash of the kernel sources + plans), so that the bench records taken
# next report the rocprofv3 clock and the PMC traffic beside their own numbers; run tools/finish_profiles.py again at home on the merged files
python tools/finish_profiles.py $tag > $out/finish_on_box.log 2>&1
timeout 400 python bench.py --layers > $out/bench.json 2> $out/layers_hipevents.txt
for spec in "config1_bootstrap:--workload bootstrap" "batch1:--batch 1" "batch8:--batch 8" "batch64:--batch 64" "config4_hires:--workload hires --layers" "v2:--workload v2 --layers"; do
  name=${spec%%:*}; args=${spec#*:}
  timeout 400 python bench.py $args --no-cpu-baseline > $out/bench_$name.json 2> $out/layers_$name.txt
done
timeout 200 python tools/throughput_profile.py > $out/throughput_profile.txt 2> $out/throughput_profile.err
# keep what travels back small: the per-dispatch traces are large
find $out -name "*kernel_trace.csv" -size +20M -delete
ls -la $out $out/stats/* | head -40
