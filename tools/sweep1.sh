cd $GRAFT_REPO_ROOT
out=gpurun_out/sweep1.log; rm -f $out
export BENCH_ARGS="--steps 20 --warmup 8"
bash tools/sweep_env.sh $out "SRZ_X=0" "SRZ_CLEAR_WGS=32" "SRZ_CLEAR_WGS=48" "SRZ_CLEAR_WGS=96" "SRZ_CLEAR_WGS=128" "SRZ_SHADE_GRID=1024" "SRZ_SHADE_GRID=4096" "SRZ_SHADE_GRID=16384"
export BENCH_ARGS="--steps 20 --warmup 8 --lanes 3"
bash tools/sweep_env.sh $out "SRZ_X=0" "SRZ_CLEAR_WGS=40"
export BENCH_ARGS="--steps 20 --warmup 8 --lanes 4"
bash tools/sweep_env.sh $out "SRZ_X=0" "SRZ_CLEAR_WGS=32"
export BENCH_ARGS="--steps 20 --warmup 8 --lanes 1"
bash tools/sweep_env.sh $out "SRZ_X=0" "SRZ_CLEAR_WGS=96"
export BENCH_ARGS="--steps 20 --warmup 8 --frames 128"
bash tools/sweep_env.sh $out "SRZ_X=0"
export BENCH_ARGS="--steps 20 --warmup 8 --frames 384"
bash tools/sweep_env.sh $out "SRZ_X=0"
cat $out
