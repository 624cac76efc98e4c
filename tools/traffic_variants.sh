# Diagnostic: HBM write traffic of the render kernel for build variants ("-DX -DY" per argument)
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
PKG=ray-tracer-challenge_amd
mkdir -p gpurun_out/traffic
i=0
for v in "$@"; do
  i=$((i+1))
  for f in rtc_kernels rtc_capi; do /opt/rocm/bin/hipcc --offload-arch=gfx950 -std=c++17 -O3 -ffp-contract=off -fPIC $v -c -o $PKG/lib/$f.o $PKG/csrc/$f.hip 2>/dev/null; done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $PKG/lib/librtc_hip.so $PKG/lib/rtc_kernels.o $PKG/lib/rtc_capi.o
  rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_EA0_WRREQ TCC_EA0_WRREQ_64B --output-format csv -d gpurun_out/traffic/v$i -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/traffic/v$i.log 2>&1
  echo "variant [$v]"; python3 tools/pmc_summary.py gpurun_out/traffic/v$i
done
