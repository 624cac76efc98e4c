#!/bin/bash
# ThreadSanitizer over the host half of rtc_scene_create (CPU only, runs in the build container): the candidate BVHs are
# built on threads - one top-level group each (dragons.json: six), the threads left over inside a group's own build
# (nefertiti.json: one group), the plain tables copied beside them - and over the loader in front of it, which builds the
# entries of a scene's "objects" side by side (parseScene: dragons.json's six dragons, ShapeIdScope).  Same trick as sanitize_host.sh: the HIP sources are
# compiled --cuda-host-only and the device code object is replaced by an empty one; rtc_scene_create comes back with
# NoDevice after the host work is done.
set -e
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=${1:-/tmp/rtc_tsan}
mkdir -p $OUT && cd $OUT
PKG=$REPO/ray-tracer-challenge_amd
HOST="$PKG/host/rtc_scene.cpp $PKG/host/rtc_loader.cpp $PKG/host/rtc_flatten.cpp $PKG/host/rtc_api.cpp $PKG/host/rtc_host_capi.cpp"
SAN="-std=c++17 -O1 -g -fsanitize=thread -fno-omit-frame-pointer -I$REPO/include"
CLANG=/opt/rocm/lib/llvm/bin/clang++
for f in rtc_kernels rtc_capi; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 --cuda-host-only $SAN -ffp-contract=off -fPIC -c -o ${f}_host.o $PKG/csrc/$f.hip
done
: > no_device_code.cpp
for SYM in $(strings -a rtc_kernels_host.o rtc_capi_host.o | grep -o "__hip_fatbin_[0-9a-f]*" | sort -u); do
  echo "extern \"C\" const char $SYM[64] __attribute__((aligned(4096))) = {0};" >> no_device_code.cpp
done
$CLANG $SAN -o create_all $REPO/tools/sanitize/create_all.cpp no_device_code.cpp rtc_kernels_host.o rtc_capi_host.o $HOST -L/opt/rocm/lib -lamdhip64 -lz -Wl,-rpath,/opt/rocm/lib
./create_all $REPO/tests/golden/data $REPO/tests/golden/scenes/dragons.json $REPO/tests/golden/scenes/nefertiti.json \
             $REPO/tests/golden/scenes/groups.json $REPO/tests/golden/scenes/teapot.json $REPO/tests/golden/scenes/csg_demo.json 2>&1 | tee tsan.log | grep -c "WARNING: ThreadSanitizer" | (read n; [ "$n" = "0" ] && echo "thread sanitizer: clean" || (grep -A12 "WARNING: ThreadSanitizer" tsan.log | head -60; exit 1))
