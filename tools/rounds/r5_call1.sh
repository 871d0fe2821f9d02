#!/bin/bash
# round 5, GPU call 1: the whole -m gpu suite (with -rs: which tests skip), a baseline bench line of this box, PMC detail of the
# dominant kernels (stand-alone): wino4<t5,v4> on conv3y / conv4x, wino_deconv<32x16> on refine2
out=gpurun_out/r5a; mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q -rs -p no:cacheprovider --durations=15 > $out/gputest_rs.log 2>&1; echo "pytest rc $?" >> $out/gputest_rs.log
timeout 400 python bench.py --no-cpu-baseline > $out/bench.json 2> $out/bench.err
export DEMON_FORCE_PLAN=16,4,1
timeout 200 bash tools/pmc_kernel.sh $out/pmc_conv3y conv 32 64 48 64 128 5 1 2 1 -1 0 30 > $out/pmc_conv3y.txt 2>&1
timeout 200 bash tools/pmc_kernel.sh $out/pmc_conv4x conv 32 256 12 32 256 1 5 1 2 -1 0 30 > $out/pmc_conv4x.txt 2>&1
export DEMON_FORCE_PLAN=8,4,1
timeout 200 bash tools/pmc_kernel.sh $out/pmc_refine2 deconv 32 256 24 32 64 4 4 2 2 -1 0 30 > $out/pmc_refine2.txt 2>&1
unset DEMON_FORCE_PLAN
find $out -name "*.csv" -size +5M -delete
tail -5 $out/gputest_rs.log; tail -c 600 $out/bench.json
