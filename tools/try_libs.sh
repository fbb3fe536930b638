#!/bin/bash
# usage: tools/try_libs.sh lib1.so lib2.so ... : runs tools/bench_vit.py 448 64 128 with each library variant in place
cd semantic-slam-master_amd/csrc
cp libsslam_hip.so /tmp/libsslam_hip.keep
for l in "$@"; do cp $l libsslam_hip.so; echo "== $l"; (cd ../.. && timeout -k 10 200 python tools/bench_vit.py 448 64 128); done
cp /tmp/libsslam_hip.keep libsslam_hip.so
