# under torch.distributed.run (RCCL communicator alive): which hardware-queue count has a good four-lane cell?
for nq in 6 10 12 16; do
GPU_MAX_HW_QUEUES=$nq DEMON_FORCE_DIST=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29577 bench.py --gpus 1 --no-cpu-baseline --no-roofline --no-e2e 2>/dev/null | grep "^{" | tail -1 > /tmp/fd.json
python -c "
import json; f=json.load(open('/tmp/fd.json')); t=f['config']['lanes_calibration_pairs_per_s']; print('queues $nq', round(f['value'],1), f['config']['lanes_mapping'])
for pad in range(6): print('   ', pad, [t.get('%d@%d'%(k,pad)) for k in range(1,6)])"
done
