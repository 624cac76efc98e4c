#!/bin/bash
# AddressSanitizer + UBSan + LeakSanitizer over the oracle's own scene build (JSON reader, OBJ parser, group division,
# bounds: oracle/rtc_oracle_scene.hpp), every golden scene.  CPU only.
set -e
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=${1:-/tmp/rtc_sanitize_oracle}
mkdir -p $OUT && cd $OUT
g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -ffp-contract=off -pthread -I$REPO/include -I$REPO/oracle \
    -o oracle_build_all $REPO/tools/sanitize/oracle_build_all.cpp $REPO/oracle/oracle_capi.cpp
export ASAN_OPTIONS=detect_leaks=1
./oracle_build_all $REPO/tests/golden/data $REPO/tests/golden/scenes/*.json
echo "sanitizers: clean"
