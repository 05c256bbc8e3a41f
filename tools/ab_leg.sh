#!/bin/bash
# usage: ab_leg.sh <leg> <variant> ...   -> ms_per_step of a bench leg per library variant, two rounds
leg=$1; shift
for r in 1 2; do for v in "$@"; do echo -n "$v: "; COSKAD_LIB=$PWD/tools/libf_$v.so python tools/bench_leg.py $leg 20 2>&1 | grep -o "ms_per_step.: [0-9.]*"; done; done
