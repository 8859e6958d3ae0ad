# Dev tool, run ON THE GPU BOX: occupancy / VALU-pipe counters of every kernel for one config (usage: pmc_probe3.sh "5 16" tag)
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
CFG="${1:-2 256}"
O=gpurun_out/pmc3_$2
rm -rf $O; mkdir -p $O
rocprofv3 -L > $O/avail.txt 2>&1 || true
i=0
for set in "GRBM_GUI_ACTIVE SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INST_CYCLES_VALU" \
           "MeanOccupancyPerCU MeanOccupancyPerActiveCU" "VALUBusy VALUUtilization SALUBusy" "MemUnitStalled WriteUnitStalled MemUnitBusy" \
           "SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32" \
           "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT64 SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT"; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --pmc $set --output-format csv -d $O/s$i -- python3 tools/perf_probe.py $CFG 4 > $O/s$i.log 2>&1 && python3 tools/pmc_summary.py $O/s$i >> $O/summary.txt || echo "set $i failed: $(tail -2 $O/s$i.log)" >> $O/summary.txt
done
cat $O/summary.txt
