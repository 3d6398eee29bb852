#!/bin/bash
export SLS_LAB=1      # diagnostic knobs are honoured in lab mode only
D=systemlevelcontrol.jl_amd
cp $D/libsls_mi355x.so /tmp/lib_A.so
run() { for w in chain512_d20 chain512_d28; do timeout 200 python tools/iters_hist.py $w | grep "^kernel avg\|iters histogram" | tr '\n' ' '; echo; done; }
echo "A (default build)"; run
cp $1 $D/libsls_mi355x.so; echo "B ($1)"; run
cp /tmp/lib_A.so $D/libsls_mi355x.so
