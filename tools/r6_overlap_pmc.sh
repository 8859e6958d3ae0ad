#!/bin/bash
# round 6: what the second lane's kernels find beside k_shade.  For configs 4 and 5:
#   (1) the two-lane timeline without counters (kernel trace only): what runs beside what, and how long each kernel takes there
#   (2) a --pmc pass (SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_INSTS_VALU) of the SAME two-lane command with the kernel trace:
#       per-kernel counters, and whether the profiler lets the two lanes' kernels overlap at all while it collects
#   (3) the same pass with one lane (kernels alone) for comparison
# results: gpurun_out/r6/ov_<wl>_{trace,pmc2,pmc1}.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r6; mkdir -p $O
SQ="SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_INSTS_VALU"
for wf in spot_x16_texture_2048:128 spot_x8_overdraw_4096:64; do
  w=${wf%%:*}; f=${wf##*:}
  CMD="python3 bench.py --no-cpu-baseline --no-extras --workload $w --frames $f --steps 10 --warmup 5"
  rm -rf $O/ovt $O/ovp2 $O/ovp1
  rocprofv3 --kernel-trace --output-format csv -d $O/ovt -- $CMD --lanes 2 > $O/ov_$w.trace.json 2> $O/ov_$w.trace.err || { tail -5 $O/ov_$w.trace.err; exit 1; }
  python3 tools/overlap_trace.py $O/ovt $O/ov_${w}_trace.json > $O/ov_${w}_trace.txt
  rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d $O/ovp2 -- $CMD --lanes 2 > $O/ov_$w.pmc2.json 2> $O/ov_$w.pmc2.err || { tail -5 $O/ov_$w.pmc2.err; exit 1; }
  python3 tools/overlap_trace.py $O/ovp2 $O/ov_${w}_pmc2_trace.json > $O/ov_${w}_pmc2.txt
  python3 tools/pmc_summary.py $O/ovp2 >> $O/ov_${w}_pmc2.txt
  rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d $O/ovp1 -- $CMD --lanes 1 > $O/ov_$w.pmc1.json 2> $O/ov_$w.pmc1.err || { tail -5 $O/ov_$w.pmc1.err; exit 1; }
  python3 tools/overlap_trace.py $O/ovp1 $O/ov_${w}_pmc1_trace.json > $O/ov_${w}_pmc1.txt
  python3 tools/pmc_summary.py $O/ovp1 >> $O/ov_${w}_pmc1.txt
  rm -rf $O/ovt $O/ovp2 $O/ovp1
  for k in trace pmc2 pmc1; do echo "=== $w $k"; cat $O/ov_${w}_$k.txt; done
done
