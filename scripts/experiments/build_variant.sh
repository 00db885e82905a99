#!/bin/bash
# dev: build an experimental variant of the library next to the shipped one (same ABI, loaded with OPV_LIB=...):
#   build_variant.sh <name> <python-snippet-that-edits-files-in-cwd>
# copies opv-cxx-demod_amd/ to /tmp/opv_variant_<name>, runs the snippet there, builds, and leaves
# opv-cxx-demod_amd/libopv_<name>.so in the tree (git-ignored, travels with gpurun).
set -e
name=$1; shift
R=$(cd "$(dirname "$0")/../.." && pwd)
D=/tmp/opv_variant_$name
rm -rf $D; mkdir -p $D/pkg; cp -r $R/include $D/include
cp -r $R/opv-cxx-demod_amd/csrc $R/opv-cxx-demod_amd/host $R/opv-cxx-demod_amd/tools $R/opv-cxx-demod_amd/Makefile $D/pkg/
(cd $D/pkg && python3 -c "$1" && make -s -j8 libopv_demod_hip.so 2>&1 | grep -v "warning: argument unused" | grep -i "error\|align_vop3: k_msk_frontend_rb:" || true)
cp $D/pkg/libopv_demod_hip.so $R/opv-cxx-demod_amd/libopv_$name.so
ls -la $R/opv-cxx-demod_amd/libopv_$name.so
