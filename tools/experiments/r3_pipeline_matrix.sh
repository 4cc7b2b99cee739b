#!/bin/bash
# Round 3: launch structure of the head call (ovn_set_head_pipeline), measured on BASELINE configs[1] through bench.py.
# Every line of gpurun_out/r3_matrix.jsonl = {"tag": ..., "value": pairs/s, "ms_per_step": ..., "kernels": {...}}.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
OUT=gpurun_out/r3_matrix.jsonl
: > $OUT
run() {   # tag, env...
  tag=$1; shift
  env "$@" timeout 300 python bench.py --no-extras --no-cpu-baseline --steps 20 --warmup 5 2> gpurun_out/r3_matrix_$tag.err | \
    python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps({'tag':'$tag','value':d['value'],'ms_per_step':d['ms_per_step'],'maxerr':d.get('overlap_maxerr_vs_oracle'),'yaw':d.get('yaw_exact_rate'),'kernels':{k:round(v['ms_per_launch']*v['launches']/20,4) for k,v in d['kernels'].items()}}))" >> $OUT
}
run serial      OVN_YAW_SIDE=0
run yawside     OVN_YAW_SIDE=1
run sub256_s1   OVN_HEAD_SUBCHUNK=256 OVN_HEAD_STREAMS=1
run sub256_s2   OVN_HEAD_SUBCHUNK=256 OVN_HEAD_STREAMS=2
run sub128_s2   OVN_HEAD_SUBCHUNK=128 OVN_HEAD_STREAMS=2
run sub64_s1    OVN_HEAD_SUBCHUNK=64 OVN_HEAD_STREAMS=1
run sub64_s2    OVN_HEAD_SUBCHUNK=64 OVN_HEAD_STREAMS=2
run sub512_s2   OVN_HEAD_SUBCHUNK=512 OVN_HEAD_STREAMS=2
cat $OUT
