# PMC passes over the headline-only encoder step (one MI355X).  Every pass is its own rocprofv3 run with --pmc only.
# usage: bash tools/pmc_r03.sh [tag]
set -e
TAG=${1:-r03}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_$TAG
mkdir -p $O
CMD="python3 $R/bench.py --steps 1 --warmup 1 --no-cpu --no-decode --no-extra-legs --no-ceiling"
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES" \
           "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum" \
           "SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_TRANS_F32" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $O/p$i -- $CMD > $O/p$i.log 2>&1 || echo "pass $i failed"
  echo "pass $i done: $set"
done
python3 $R/tools/pmc_table.py $O/${TAG}_pmc_table.json $O/p1 $O/p2 $O/p3 $O/p4 > $O/table.log 2>&1 || true
cat $O/table.log
F=$(find $O/p5 -name "*counter_collection.csv" | head -1); W=$(find $O/p6 -name "*counter_collection.csv" | head -1)
python3 $R/tools/pmc_summarize.py $F $W $O/${TAG}_hbm_traffic_pmc.json kernels_$TAG "headline-only encoder step, B=32" > $O/traffic.log 2>&1 || true
cat $O/traffic.log
find $O -name "*.csv" -size +20M -delete
