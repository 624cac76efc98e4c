# Diagnostic: rebuilds librtc_hip.so with extra -D flags on the GPU box and times bench.py per variant.
set -e
cd $GRAFT_REPO_ROOT
PKG=ray-tracer-challenge_amd
run() { python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>gpurun_out/err_$1.txt | python3 -c "import json,sys; d=json.load(sys.stdin); print('$1', round(d['ms_per_step'],3))"; grep "rtc prof" gpurun_out/err_$1.txt | tail -1 || true; }
build() { /opt/rocm/bin/hipcc --offload-arch=gfx950 -std=c++17 -O3 -ffp-contract=off -fPIC $1 -c -o $PKG/lib/rtc_kernels.o $PKG/csrc/rtc_kernels.hip 2>/dev/null && /opt/rocm/bin/hipcc --offload-arch=gfx950 -std=c++17 -O3 -ffp-contract=off -fPIC $1 -c -o $PKG/lib/rtc_capi.o $PKG/csrc/rtc_capi.hip 2>/dev/null && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $PKG/lib/librtc_hip.so $PKG/lib/rtc_kernels.o $PKG/lib/rtc_capi.o; }
mkdir -p gpurun_out
for v in "$@"; do
  name=$(echo "$v" | tr -c 'A-Za-z0-9=\n' '_')
  build "$v"; RTC_PROFILE_DUMP=1 run "$name"
done
