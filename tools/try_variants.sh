# Experiment helper (GPU box): builds librtc_hip with each set of -D flags in turn and times the BASELINE configs.
#   bash tools/try_variants.sh "-DA" "-DA -DB" ...
cd $GRAFT_REPO_ROOT
PKG=ray-tracer-challenge_amd
for v in "$@"; do
  for f in rtc_kernels rtc_capi; do /opt/rocm/bin/hipcc --offload-arch=gfx950 -std=c++17 -O3 -ffp-contract=off -fPIC $v -c -o $PKG/lib/$f.o $PKG/csrc/$f.hip 2>/dev/null || exit 1; done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $PKG/lib/librtc_hip.so $PKG/lib/rtc_kernels.o $PKG/lib/rtc_capi.o || exit 1
  echo "== $v"
  python3 tools/option_time_full.py pack_rounds=3 || exit 1
done
