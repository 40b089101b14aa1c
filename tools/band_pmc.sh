#!/bin/bash
# PMC view of one rank's bands at N = 1 and 8: instructions per row and where the wave-cycles go
export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; OUT=$GRAFT_REPO_ROOT/gpurun_out/band_pmc; mkdir -p $OUT
for n in 1 8; do
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $OUT/a$n --output-format csv -- python3 tools/band_frames.py $n > $OUT/a$n.log 2>&1
  rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC -d $OUT/b$n --output-format csv -- python3 tools/band_frames.py $n > $OUT/b$n.log 2>&1
  rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_BUSY_CU_CYCLES SQ_CYCLES SQ_LDS_IDX_ACTIVE -d $OUT/c$n --output-format csv -- python3 tools/band_frames.py $n > $OUT/c$n.log 2>&1
done
python3 tools/pmc_summary.py $OUT/a1 $OUT/b1 $OUT/c1 $OUT/a8 $OUT/b8 $OUT/c8 > $OUT/summary.txt
cat $OUT/summary.txt
