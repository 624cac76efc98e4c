#!/bin/bash
# AddressSanitizer + UBSan + LeakSanitizer over everything that runs on the host (no GPU needed, runs in the build
# container): the host library's KATs, loading + flattening every golden scene, and the host half of librtc_hip
# (validateScene + buildTables: bounds, SAH build, four-wide collapse) on every golden scene.  The HIP sources are
# compiled --cuda-host-only; the device code object they would register is replaced by an empty one (no kernel can
# be launched: rtc_scene_create has to come back with NoDevice after the host work is done).
set -e
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=${1:-/tmp/rtc_sanitize}
mkdir -p $OUT && cd $OUT
PKG=$REPO/ray-tracer-challenge_amd
HOST="$PKG/host/rtc_scene.cpp $PKG/host/rtc_loader.cpp $PKG/host/rtc_flatten.cpp $PKG/host/rtc_api.cpp $PKG/host/rtc_host_capi.cpp"
SAN="-std=c++17 -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -I$REPO/include"
CLANG=/opt/rocm/lib/llvm/bin/clang++
for f in rtc_kernels rtc_capi; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 --cuda-host-only $SAN -ffp-contract=off -fPIC -c -o ${f}_host.o $PKG/csrc/$f.hip
done
# the symbols the host objects expect their device code under
: > no_device_code.cpp
for SYM in $(strings -a rtc_kernels_host.o rtc_capi_host.o | grep -o "__hip_fatbin_[0-9a-f]*" | sort -u); do
  echo "extern \"C\" const char $SYM[64] __attribute__((aligned(4096))) = {0};" >> no_device_code.cpp
done
$CLANG $SAN -o create_all $REPO/tools/sanitize/create_all.cpp no_device_code.cpp rtc_kernels_host.o rtc_capi_host.o $HOST -L/opt/rocm/lib -lamdhip64 -lz -Wl,-rpath,/opt/rocm/lib
$CLANG $SAN -o host_kat $REPO/tests/cpp/host_kat_main.cpp no_device_code.cpp rtc_kernels_host.o rtc_capi_host.o $HOST -L/opt/rocm/lib -lamdhip64 -lz -Wl,-rpath,/opt/rocm/lib
export ASAN_OPTIONS=detect_leaks=1
./host_kat | tail -1
./create_all $REPO/tests/golden/data $REPO/tests/golden/scenes/*.json
echo "sanitizers: clean"
