set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
CFG="${1:-2 256}"
O=gpurun_out/pmc_$2
rm -rf $O; mkdir -p $O
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAVE_CYCLES --output-format csv -d $O/sq1 -- python3 tools/perf_probe.py $CFG 4 > $O/sq1.log 2>&1 || exit 1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $O/sq2 -- python3 tools/perf_probe.py $CFG 4 > $O/sq2.log 2>&1 || exit 1
python3 tools/pmc_summary.py $O/sq1 > $O/summary.txt; python3 tools/pmc_summary.py $O/sq2 >> $O/summary.txt
cat $O/summary.txt
