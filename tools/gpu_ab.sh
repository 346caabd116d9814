#!/bin/bash
# A/B of one knob on scenario lines.  Usage: tools/gpu_ab.sh <tag> <KNOB> <v1> <v2> <scenario>...
tag=$1; knob=$2; v1=$3; v2=$4; shift 4
for v in $v1 $v2; do
  echo "== $knob=$v"
  env $knob=$v tools/gpu_scen.sh ${tag}_$v "$@" || exit 1
done
