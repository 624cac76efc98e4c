# Diagnostic: predicted vs steady-state packet times (RTC_TIME_ALWAYS) per build variant
set -e
cd $GRAFT_REPO_ROOT
PKG=ray-tracer-challenge_amd
for v in "$@"; do
  for f in rtc_kernels rtc_capi; do /opt/rocm/bin/hipcc --offload-arch=gfx950 -std=c++17 -O3 -ffp-contract=off -fPIC $v -c -o $PKG/lib/$f.o $PKG/csrc/$f.hip 2>/dev/null; done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $PKG/lib/librtc_hip.so $PKG/lib/rtc_kernels.o $PKG/lib/rtc_capi.o
  echo "[$v] $(RTC_TIME_ALWAYS=1 python tools/packet_times.py 2>&1 | grep 'rtc packet times: all')"
done
