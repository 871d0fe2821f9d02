#!/bin/bash
# usage: tools/pmc_kernel.sh <outdir> <one_layer.py args...>   -- SQ counters of one kernel variant, in passes of <= 8 counters
# (counter names are checked against `rocprofv3 -L` first: an unknown name fails the whole pass)
out=$(realpath -m $1); shift
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $out/counters.txt 2>&1
cd $GRAFT_REPO_ROOT
pass() {
  tag=$1; shift
  names=""
  for c in "$@"; do if grep -qw "$c" $out/counters.txt; then names="$names $c"; else echo "no counter $c" >> $out/missing.txt; fi; done
  rocprofv3 --pmc $names -d $out/$tag -o p -- python tools/one_layer.py "${ARGS[@]}" > $out/$tag.log 2>&1
}
ARGS=("$@")
pass a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES
pass b SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC
pass c SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INST_CYCLES_VMEM SQ_LDS_ADDR_CONFLICT
python tools/pmc_print.py $out
