#!/bin/bash
# round 6: the render-side imbalance of the N-GPU job (bench.py's emulate_shards) under band -> rank maps rank = (b + k * (b / N)) % N
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for lib in rot0.so - rot3.so rot5.so rot0.so -; do
  if [ "$lib" = "-" ]; then unset SRZ_LIB_PATH; else export SRZ_LIB_PATH=$PWD/software-rasterizer_amd/build/$lib; fi
  python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2>/dev/null
  python3 -c "
import json
d=json.load(open('bench_details.json')); me=d['roofline']['multi_gpu_emulated']
print('$lib', ' '.join(f\"{w.split('_')[1] if w!='spot_texture_1024' else 'c2'}:{k}={v['max_over_mean']:.3f}\" for w,rec in me.items() for k,v in rec.items() if isinstance(v,dict)))"
done
