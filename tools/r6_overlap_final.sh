#!/bin/bash
# round 6, after the per-plane clear: the two-lane timeline once more (kernel trace only; tools/overlap_trace.py) for configs 2 / 4 / 5 —
# how long each kernel takes beside the others and what share of it each other kind runs beside it.  -> gpurun_out/r6/ovf_<wl>_trace.json
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r6; mkdir -p $O
for wf in spot_texture_1024:256 spot_x16_texture_2048:128 spot_x8_overdraw_4096:64; do
  w=${wf%%:*}; f=${wf##*:}
  rm -rf $O/ovt
  rocprofv3 --kernel-trace --output-format csv -d $O/ovt -- python3 bench.py --no-cpu-baseline --no-extras --workload $w --frames $f --steps 20 --warmup 5 --lanes 2 > $O/ovf_$w.bench.json 2> $O/ovf_$w.err || { tail -5 $O/ovf_$w.err; exit 1; }
  python3 tools/overlap_trace.py $O/ovt $O/ovf_${w}_trace.json 0.40 0.62 > $O/ovf_${w}_trace.txt  # (the second two-lane timed region: bench.py renders 2 x 50 launch sets on two lanes, then 51 on one stream)
  rm -rf $O/ovt
  echo "=== $w"; cat $O/ovf_${w}_trace.txt
done
