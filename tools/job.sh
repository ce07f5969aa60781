#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r04r; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
hipcc --offload-arch=gfx950 -O2 -o /tmp/valu_rate $R/tools/valu_rate.hip 2> $O/valu_build.log || { tail $O/valu_build.log; exit 1; }
timeout -k 10 600 /tmp/valu_rate mix_ > $O/valu_rate_mix2.txt 2> $O/valu_rate.err; echo "valu_rate rc=$?"; cat $O/valu_rate_mix2.txt | grep -v "^lanes\|^[ 0-9]* low\|^[ 0-9]* spread\|active lanes"
