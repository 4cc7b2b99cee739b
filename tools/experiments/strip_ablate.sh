# Timing-only ablations of conv_strip_kernel (s_conv3 / s_conv3a / s_conv4): builds tools/bin/libovn_strip_abl<N>.so HERE (no GPU),
# then on the GPU box:  bash tools/experiments/strip_ablate.sh run
# N: 1 no strip staging, 2 no K loop, 4 staging without the global loads, 3 neither (launch + epilogue only)
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
if [ "$1" = run ]; then
  cd /tmp && export TMPDIR=/tmp
  for n in 0 1 2 4 3; do
    lib=$R/tools/bin/libovn_strip_abl$n.so
    [ $n = 0 ] && lib=$R/overlapnet_amd/libovn_hip.so
    rocprofv3 --kernel-trace --stats -d $R/gpurun_out/strip_abl_$n -o x -- python $R/tools/experiments/leg_time.py $lib > $R/gpurun_out/strip_abl_$n.log 2>&1
    echo "== ABL $n: $(tail -1 $R/gpurun_out/strip_abl_$n.log)"
    python $R/tools/rocprof_top.py $(find $R/gpurun_out/strip_abl_$n -name "*.db" | head -1) 12 2>/dev/null | grep -i "strip\|tail\|front" || true
  done
  exit 0
fi
mkdir -p $R/tools/bin
cd $R/overlapnet_amd/csrc
for n in 1 2 4 3; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++20 -fPIC -DSTRIP_ABL=$n -c conv_strip.hip -o /tmp/conv_strip_abl$n.o
  objs=$(ls *.o | grep -v '^conv_strip.o$' | tr '\n' ' ')
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs /tmp/conv_strip_abl$n.o -ldl -o $R/tools/bin/libovn_strip_abl$n.so
done
ls -la $R/tools/bin
