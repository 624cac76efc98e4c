set -e
cd $GRAFT_REPO_ROOT
PKG=ray-tracer-challenge_amd
build() { /opt/rocm/bin/hipcc --offload-arch=gfx950 -std=c++17 -O3 -ffp-contract=off -fPIC $1 -c -o $PKG/lib/rtc_kernels.o $PKG/csrc/rtc_kernels.hip 2>/dev/null && /opt/rocm/bin/hipcc --offload-arch=gfx950 -std=c++17 -O3 -ffp-contract=off -fPIC $1 -c -o $PKG/lib/rtc_capi.o $PKG/csrc/rtc_capi.hip 2>/dev/null && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $PKG/lib/librtc_hip.so $PKG/lib/rtc_kernels.o $PKG/lib/rtc_capi.o; }
for v in "$@"; do
  echo "=== $v"; build "$v"; timeout -k 5 200 python tools/fuzz_debug.py ${FUZZ_SEEDS:-1 3 4 6 8} 2>&1 | grep -E "^seed" 
done
