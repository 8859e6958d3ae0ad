#!/bin/bash
# Run ON THE GPU BOX (gpurun -- 'bash profiles/collect.sh rNN'): rocprofv3 kernel-trace stats + two PMC passes of the
# same bench.py command; raw outputs land in gpurun_out/prof_<tag>/, profiles/summarize.py turns them into the
# committed summaries.  PMC passes are separate (FETCH_SIZE needs 3 TCC slots, WRITE_SIZE 2, 4 per pass) and never
# combined with tracing options other than the kernel trace.
set -o pipefail
TAG=${1:-r02}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
# bench.py's default renders the batch as two lanes on two streams: their kernels overlap, so per-kernel durations and
# per-launch counters are taken from the same command with --lanes 1 (one frameset, one stream: the bytes per step are the
# same, the kernels do not overlap); the default command's own kernel statistics are kept beside them (trace_lanes)
CMD="python3 bench.py --no-cpu-baseline --no-extras --lanes 1"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_lanes -- python3 bench.py --no-cpu-baseline --no-extras > $OUT/bench_traced_lanes.json 2> $OUT/trace_lanes.log || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $CMD > $OUT/bench_traced.json 2> $OUT/trace.log || exit 1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $CMD > $OUT/bench_pmc_fetch.json 2> $OUT/pmc_fetch.log || exit 1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $CMD > $OUT/bench_pmc_write.json 2> $OUT/pmc_write.log || exit 1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAVE_CYCLES --output-format csv -d $OUT/pmc_sq -- $CMD > $OUT/bench_pmc_sq.json 2> $OUT/pmc_sq.log || exit 1
python3 bench.py > $OUT/bench_plain.json 2> $OUT/bench_plain.log || exit 1
# the other BASELINE configs: kernel-trace stats + SQ counters of the dev probe (whole frames on one GPU)
for c in "3 64" "4 32" "5 16"; do
  n=${c%% *}
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_c$n -- python3 tools/perf_probe.py $c 10 > $OUT/probe_c$n.log 2>&1 || exit 1
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAVE_CYCLES --output-format csv -d $OUT/pmc_sq_c$n -- python3 tools/perf_probe.py $c 4 > $OUT/probe_pmc_c$n.log 2>&1 || exit 1
done
ls -R $OUT | head -40
