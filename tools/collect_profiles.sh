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
PL=${PLAN_LANES:-4}
B="python bench.py --no-cpu-baseline --no-e2e --no-roofline --lanes 1 --plan-lanes $PL"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o p -- $B > $out/stats.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_lanes -o p -- python bench.py --no-cpu-baseline --no-e2e --no-roofline > $out/stats_lanes.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -o p -- $B > $out/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -o p -- $B > $out/pmc_write.log 2>&1
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $out/pmc_mfma -o p -- $B > $out/pmc_mfma.log 2>&1
# the summaries of these passes into profiles/ ON THE BOX (stamped with the hash of the kernel sources + plans), so that the bench records taken
# next report the rocprofv3 clock and the PMC traffic beside their own numbers; run tools/finish_profiles.py again at home on the merged files
python tools/finish_profiles.py $tag > $out/finish_on_box.log 2>&1
( time timeout 700 python bench.py --layers ) > $out/bench.json 2> $out/layers_hipevents.txt
for spec in "config1_bootstrap:--workload bootstrap" "batch1:--batch 1 --layers" "batch8:--batch 8 --layers" "batch64:--batch 64" "config4_hires:--workload hires --layers" "v2:--workload v2 --layers"; do
  name=${spec%%:*}; args=${spec#*:}
  timeout 400 python bench.py $args --no-cpu-baseline > $out/bench_$name.json 2> $out/layers_$name.txt
done
timeout 200 python tools/throughput_profile.py --lanes $PL > $out/throughput_profile.txt 2> $out/throughput_profile.err
# 5. what each group of layers costs with the headline's passes in flight
timeout 400 python tools/ablate_lanes.py --lanes $PL > $out/ablate_lanes.txt 2> $out/ablate_lanes.err
cp $out/ablate_lanes.txt profiles/${tag}_ablate_lanes.txt
# 6. BASELINE configs[4] evidence (640x480, "HBM-bound warp2d stress, rocprof GB/s"): counters + durations of the same command
H="python bench.py --workload hires --no-cpu-baseline --no-e2e --no-roofline --lanes 1 --steps 4 --warmup 2"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/hires_stats -o p -- $H > $out/hires_stats.log 2>&1
timeout 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/hires_fetch -o p -- $H > $out/hires_fetch.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/hires_write -o p -- $H > $out/hires_write.log 2>&1
python tools/hires_counters.py $out/hires_fetch $out/hires_write $out/hires_stats profiles/${tag}_hires_counters.json > $out/hires_counters.log 2>&1
find $out/hires_stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} profiles/${tag}_hires_kernel_stats.csv
# 6b. the headline's own protocol against the CPU oracle (every pair of every kept lane), and the lane-mode table (round 6)
timeout 400 python tools/lane_parity.py --lanes 5 > profiles/${tag}_lane_parity.json 2> $out/lane_parity.err
timeout 900 python tools/lane_modes.py --steps 40 --modes rr,pmask-4-4c,pmask-4-2c,group-4,group-3,rrside-4,quarter,rr > profiles/${tag}_lane_modes.txt 2> $out/lane_modes.err
# 7. the RCCL route as the driver launches a rank, and the whole -m gpu suite with the reasons of its skips
DEMON_FORCE_DIST=1 timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29577 bench.py --gpus 1 --no-cpu-baseline --no-roofline --no-e2e > $out/forcedist.out 2> $out/forcedist.err
grep "^{" $out/forcedist.out | tail -1 > profiles/${tag}_forcedist_rccl_1rank.json
timeout 1500 python -m pytest tests -m gpu -q -rs -p no:cacheprovider > $out/gputest_rs.log 2>&1; echo "pytest rc $?" >> $out/gputest_rs.log
cp $out/gputest_rs.log profiles/${tag}_gputest_rs.log
mkdir -p $out/profiles_on_box && cp profiles/${tag}_* $out/profiles_on_box/
# keep what travels back small: the per-dispatch traces are large
find $out -name "*kernel_trace.csv" -size +20M -delete
ls -la $out $out/stats/* | head -40
