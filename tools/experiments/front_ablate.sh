# Timing-only ablations of leg_front_kernel: builds tools/bin/libovn_front_abl<N>.so HERE (no GPU); on the GPU box: bash tools/experiments/front_ablate.sh run
# N (bits): 1 constants instead of the strip's global loads, 2 no stage A MFMAs, 4 no stage B MFMAs, 8 no output stores, 16 no strip staging,
# 32 no intermediate-tile exchange
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
VARIANTS="1 2 4 8 16 32 6 62"
if [ "$1" = run ]; then
  # the kernel's own duration (rocprofv3 kernel trace): later layers see different data in the ablated builds, and a power-limited
  # matrix pipe runs faster on constant or zero operands, so whole-leg times would mix the two effects
  cd /tmp && export TMPDIR=/tmp
  for n in 0 $VARIANTS 0; do
    lib=$R/tools/bin/libovn_front_abl$n.so
    [ $n = 0 ] && lib=$R/overlapnet_amd/libovn_hip.so
    rm -rf $R/gpurun_out/front_abl_$n
    rocprofv3 --kernel-trace --stats -d $R/gpurun_out/front_abl_$n -o x -- python $R/tools/experiments/leg_time.py $lib > $R/gpurun_out/front_abl_$n.log 2>&1
    echo "ABL $n: $(grep 'leg ms' $R/gpurun_out/front_abl_$n.log | cut -d: -f2 | cut -c1-30) | $(python $R/tools/rocprof_top.py $(find $R/gpurun_out/front_abl_$n -name '*.db' | head -1) 12 2>/dev/null | grep -i front | cut -c90-140)"
  done
  exit 0
fi
mkdir -p $R/tools/bin
cd $R/overlapnet_amd/csrc
objs=$(ls *.o | grep -v '^leg_front.o$' | tr '\n' ' ')
for n in $VARIANTS; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DFRONT_ABL=$n -c leg_front.hip -o /tmp/leg_front_abl$n.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs /tmp/leg_front_abl$n.o -ldl -o $R/tools/bin/libovn_front_abl$n.so
done
ls $R/tools/bin/*.so
