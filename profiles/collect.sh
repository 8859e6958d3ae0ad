#!/bin/bash
# Run ON THE GPU BOX (gpurun --timeout 1200 -- 'bash profiles/collect.sh rNN'): for EVERY workload of bench.py's line — the
# headline (BASELINE configs[1]) and the configs[] entries — the same bench.py command under rocprofv3: the kernel-trace
# statistics and five PMC passes (FETCH_SIZE / WRITE_SIZE / SQ counters / VALU instruction classes / resident waves: separate passes, FETCH_SIZE needs 3 TCC slots and
# WRITE_SIZE 2 of 4; never combined with tracing options other than the kernel trace).  The workload runs as ONE frameset on ONE
# stream (--lanes 1: the bytes and instructions per step are those of the default two-lane run, the kernels do not overlap, so
# per-kernel durations add up); the default two-lane command's own kernel statistics are kept beside them for the headline.
# Raw outputs land in gpurun_out/prof_<tag>/; profiles/summarize.py turns them into the committed summaries.
set -o pipefail
TAG=${1:-r03}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
SQ="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
# instruction classes (what a VALU instruction costs the pipe: plain 2 cycles, binary64 4, transcendental 8) and resident waves per SIMD
MIX="SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32"
python3 bench.py > $OUT/bench_plain.json 2> $OUT/bench_plain.log || exit 1
cp bench_details.json $OUT/bench_plain_details.json
python3 bench.py --steps 20 --warmup 5 > $OUT/bench_driver_args.json 2> $OUT/bench_driver_args.log || exit 1
cp bench_details.json $OUT/bench_driver_args_details.json
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_lanes -- python3 bench.py --no-cpu-baseline --no-extras > $OUT/bench_traced_lanes.json 2> $OUT/trace_lanes.log || exit 1
# workload : frames per step (= bench.py's EXTRA_CASES and its default)
for wf in spot_texture_1024:256 spot_bunny_phong_1080p:128 spot_x16_texture_2048:128 spot_x8_overdraw_4096:64 readme_spot_crate_1024:256 \
          spot_texture_1024_3lights:256 spot_texture_1024_p32:256 spot_texture_1024_p7.5:256 spot_bump_1024:256; do
  w=${wf%%:*}; f=${wf##*:}
  CMD="python3 bench.py --no-cpu-baseline --no-extras --lanes 1 --workload $w --frames $f --steps 8 --warmup 4"
  echo "== $w x $f"
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$w/trace -- $CMD > $OUT/$w.traced.json 2> $OUT/$w.trace.log || exit 1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/$w/pmc_fetch -- $CMD > /dev/null 2> $OUT/$w.pmc_fetch.log || exit 1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/$w/pmc_write -- $CMD > /dev/null 2> $OUT/$w.pmc_write.log || exit 1
  rocprofv3 --pmc $SQ --output-format csv -d $OUT/$w/pmc_sq -- $CMD > /dev/null 2> $OUT/$w.pmc_sq.log || exit 1
  rocprofv3 --pmc $MIX --output-format csv -d $OUT/$w/pmc_mix -- $CMD > /dev/null 2> $OUT/$w.pmc_mix.log || exit 1
  rocprofv3 --pmc MeanOccupancyPerCU --output-format csv -d $OUT/$w/pmc_occ -- $CMD > /dev/null 2> $OUT/$w.pmc_occ.log || exit 1
done
ls $OUT | head -60
