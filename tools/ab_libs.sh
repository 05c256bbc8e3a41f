#!/bin/bash
# same-box A/B of whole-library builds: tools/libf_<name>.so ...; prints the train step per variant (tools/time_step.py), two rounds
cp coskad_amd/libcoskad_hip.so /tmp/lib_orig.so
for r in 1 2; do for v in "$@"; do cp tools/libf_$v.so coskad_amd/libcoskad_hip.so; AB_TAG=$v timeout -k 10 100 python tools/time_step.py 100 2>&1 | tail -1; done; done
cp /tmp/lib_orig.so coskad_amd/libcoskad_hip.so
