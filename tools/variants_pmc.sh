# Diagnostic: time and HBM traffic counters per build variant: bash tools/variants_pmc.sh "-DX" "-DY -DZ" ...
set -e
cd $GRAFT_REPO_ROOT
PKG=ray-tracer-challenge_amd
for v in "$@"; do
  for f in rtc_kernels rtc_capi; do /opt/rocm/bin/hipcc --offload-arch=gfx950 -std=c++17 -O3 -ffp-contract=off -fPIC $v -c -o $PKG/lib/$f.o $PKG/csrc/$f.hip 2>/dev/null; done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $PKG/lib/librtc_hip.so $PKG/lib/rtc_kernels.o $PKG/lib/rtc_capi.o
  echo "[$v] $(python tools/all_time.py cover reflection_and_refraction 2>/dev/null)"
  bash tools/pmc_probe.sh WRITE_SIZE | grep WRITE
  bash tools/pmc_probe.sh FETCH_SIZE | grep FETCH
done
