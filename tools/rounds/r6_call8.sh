#!/bin/bash
# round 6, GPU call 8: lanes WITH the side branches of a pass on (round 4 switched them off inside a group when there were 4 hardware queues;
# with 8 / 16 queues four lanes x two streams can each have one), stream mapping calibrated; against the shipped form on this box
out=gpurun_out/r6h; mkdir -p $out
cd $GRAFT_REPO_ROOT
( time timeout 900 python tools/lane_modes.py --steps 40 --modes rr,rrside-4,rrside-3,rrside-5,rr ) > $out/lane_modes.txt 2> $out/lane_modes.err
DEMON_HW_QUEUES=16 timeout 600 python tools/lane_modes.py --steps 40 --modes rrside-4,rrside-5,rr >> $out/lane_modes.txt 2>> $out/lane_modes.err
DEMON_HW_QUEUES=12 timeout 600 python tools/lane_modes.py --steps 40 --modes rrside-4 >> $out/lane_modes.txt 2>> $out/lane_modes.err
cat $out/lane_modes.txt | cut -c1-400
