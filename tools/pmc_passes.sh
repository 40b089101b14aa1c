#!/bin/bash
# rocprofv3 passes over one bench workload (run on the GPU box from the repo root):
#   bash tools/pmc_passes.sh <out-dir-under-gpurun_out> [--workload cfg5 ...]
# Every pass profiles the SAME command: bench.py --timed-only --warmup 20 --steps 30 (only the warm-up and the timed
# steps run, so every dispatch in the profile belongs to the workload; tools/pmc_to_json.py drops the 20 warm-up steps).
# Each --pmc set is its own run (8 SQ slots per pass; FETCH_SIZE and WRITE_SIZE cannot share one); the kernel trace is a
# separate run again (gpurun refuses --pmc combined with the trace domains).
#   python tools/pmc_to_json.py gpurun_out/<out-dir> <workload> profiles/r03_pmc.json
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
mkdir -p $OUT
export TMPDIR=/tmp
BENCH="python3 $GRAFT_REPO_ROOT/bench.py --steps 30 --warmup 20 --timed-only $*"
cd $GRAFT_REPO_ROOT
run() { name=$1; shift; rocprofv3 --pmc "$@" -d $OUT/pmc_$name --output-format csv -- $BENCH > $OUT/pmc_$name.log 2>&1 || echo "pass $name failed (see pmc_$name.log)"; }
run insts   SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES
run fp32    SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU
run lds     SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
run occ     SQ_WAVES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH SQ_ACTIVE_INST_MISC SQ_CYCLES
run fetch   FETCH_SIZE
run write   WRITE_SIZE
rocprofv3 --kernel-trace --stats -d $OUT/trace --output-format csv -- $BENCH > $OUT/trace.log 2>&1 || echo "trace pass failed"
echo "pmc passes done: $OUT"
