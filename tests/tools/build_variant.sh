#!/bin/bash
# developer A/B build from a PATCHED copy of csrc/: video_prediction_amd/ab/libsavp_hip_<tag>.so = the shipped objects with the files a
# patch touches recompiled.  usage: build_variant.sh <tag> <patch file | -> [extra hipcc flags...]   ("-" = no patch, flags only;
# files to recompile are taken from the patch, or all of VARIANT_FILES="a.hip b.hip")
TAG=$1; PATCH=$2; shift 2
ROOT="$(cd "$(dirname "$0")/../.." && pwd)"
W=/tmp/variant_$TAG; rm -rf $W; mkdir -p $W/video_prediction_amd
cp -r $ROOT/video_prediction_amd/csrc $W/video_prediction_amd/csrc; cp -r $ROOT/include $W/include
FILES="$VARIANT_FILES"
if [ "$PATCH" != "-" ]; then
  (cd $W && patch -p1 < "$ROOT/$PATCH") || exit 1
  FILES="$FILES $(grep '^+++ ' "$ROOT/$PATCH" | sed 's#^+++ [ab]/video_prediction_amd/csrc/##; s#\s.*##')"
fi
mkdir -p $ROOT/video_prediction_amd/ab
objs=""
for o in $ROOT/video_prediction_amd/csrc/build/*.o; do
  b=$(basename $o .o); skip=0
  for f in $FILES; do [ "$b.hip" = "$f" ] && skip=1; done
  [ $skip = 0 ] && objs="$objs $o"
done
for f in $FILES; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$W/include -Wno-unused-value "$@" -c $W/video_prediction_amd/csrc/$f -o $W/${f%.hip}.o || exit 1
  objs="$objs $W/${f%.hip}.o"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/video_prediction_amd/ab/libsavp_hip_$TAG.so $objs -ldl && echo "built ab/libsavp_hip_$TAG.so from: $FILES"
