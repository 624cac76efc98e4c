# Diagnostic: one command per build variant: bash tools/variants_cmd.sh "<command>" "-DX" "-DY -DZ" ...
set -e
cd $GRAFT_REPO_ROOT
PKG=ray-tracer-challenge_amd
CMD="$1"; shift
for v in "$@"; do
  for f in rtc_kernels rtc_capi; do /opt/rocm/bin/hipcc --offload-arch=gfx950 -std=c++17 -O3 -ffp-contract=off -fPIC $v -c -o $PKG/lib/$f.o $PKG/csrc/$f.hip 2>/dev/null; done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $PKG/lib/librtc_hip.so $PKG/lib/rtc_kernels.o $PKG/lib/rtc_capi.o
  echo "[$v] $($CMD 2>/dev/null)"
done
