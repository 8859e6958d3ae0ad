# Dev tool, run ON THE GPU BOX: SQ issue / wait counters of every kernel for one config (usage: pmc_probe2.sh "5 16" tag)
# (a pass with TA_* / TCP_* counters hung rocprofv3 on this pool — 7 minutes until the silence watchdog — and is not repeated here)
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
CFG="${1:-2 256}"
O=gpurun_out/pmc2_$2
rm -rf $O; mkdir -p $O
i=0
for set in "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY" \
           "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_IFETCH"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $O/s$i -- python3 tools/perf_probe.py $CFG 4 > $O/s$i.log 2>&1 && python3 tools/pmc_summary.py $O/s$i >> $O/summary.txt || echo "set $i failed: $(tail -2 $O/s$i.log)" >> $O/summary.txt
done
grep -E "k_shade|k_raster |failed" $O/summary.txt
