#!/bin/bash
# same-box A/B of whole-library builds: tools/libf_<name>.so ...; prints the train step per variant (tools/time_step.py), two rounds.
# Variants are selected through COSKAD_LIB (coskad_amd/_lib.py): the shipped library is never overwritten.
set -e
for r in 1 2; do for v in "$@"; do COSKAD_LIB=$PWD/tools/libf_$v.so AB_TAG=$v timeout -k 10 100 python tools/time_step.py 100 2>&1 | tail -1; done; done
