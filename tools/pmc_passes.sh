#!/bin/bash
# rocprofv3 counter passes over the headline bench (run on the GPU box from the repo root):
#   bash tools/pmc_passes.sh <out-dir-under-gpurun_out> [extra bench.py args]
# Each --pmc set is its own run (8 SQ slots per pass; FETCH_SIZE and WRITE_SIZE cannot share one); the kernel
# trace is a separate run again (gpurun refuses --pmc combined with the trace domains).  Summaries:
#   python tools/pmc_summary.py <out-dir>/pmc_*     python tools/kernel_times.py <out-dir>/trace
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
mkdir -p $OUT
export TMPDIR=/tmp
BENCH="python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --timed-only $*"
cd $GRAFT_REPO_ROOT
run() { name=$1; shift; rocprofv3 --pmc "$@" -d $OUT/pmc_$name --output-format csv -- $BENCH > $OUT/pmc_$name.log 2>&1 || echo "pass $name failed (see pmc_$name.log)"; }
run insts   SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES
run fp32    SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU
run lds     SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
run occ     SQ_WAVES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH SQ_ACTIVE_INST_MISC SQ_CYCLES
run fetch   FETCH_SIZE
run write   WRITE_SIZE
rocprofv3 --kernel-trace --stats -d $OUT/trace --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 40 --warmup 3 --timed-only $* > $OUT/trace.log 2>&1 || echo "trace pass failed"
echo "pmc passes done: $OUT"
