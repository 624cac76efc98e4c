# Experiment helper (GPU box): per-kernel rocprof averages of one bench command under each set of -D flags.
#   bash tools/try_kernel_times.sh "<bench args>" "-DA" "-DB" ...
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
PKG=ray-tracer-challenge_amd
ARGS=$1; shift
for v in "$@"; do
  for f in rtc_kernels rtc_capi; do /opt/rocm/bin/hipcc --offload-arch=gfx950 -std=c++17 -O3 -ffp-contract=off -fPIC $v -c -o $PKG/lib/$f.o $PKG/csrc/$f.hip 2>/dev/null || exit 1; done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $PKG/lib/librtc_hip.so $PKG/lib/rtc_kernels.o $PKG/lib/rtc_capi.o || exit 1
  rm -rf gpurun_out/tkt; rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/tkt -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras $ARGS > gpurun_out/tkt.log 2>&1
  echo "== $v"; grep "rtc_render" gpurun_out/tkt/*/*kernel_stats.csv | cut -d, -f1-4
done
