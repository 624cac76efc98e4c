# Diagnostic: -DRTC_PROFILE build (plus any extra flags in $RTC_EXTRA_FLAGS) on the GPU box, then tools/prof_sections.py "$@"
set -e
cd $GRAFT_REPO_ROOT
PKG=ray-tracer-challenge_amd
for f in rtc_kernels rtc_capi; do /opt/rocm/bin/hipcc --offload-arch=gfx950 -std=c++17 -O3 -ffp-contract=off -fPIC -DRTC_PROFILE $RTC_EXTRA_FLAGS -c -o $PKG/lib/$f.o $PKG/csrc/$f.hip 2>/dev/null; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $PKG/lib/librtc_hip.so $PKG/lib/rtc_kernels.o $PKG/lib/rtc_capi.o
python tools/prof_sections.py "$@"
