#!/bin/bash
# usage: [MODELS="0 1"] tools/build_variant.sh <name> [extra hipcc flags]  -> mcsas_amd/lib/libmcsas_<name>.so from the CURRENT sources of the C ABI
# translation unit and the pipeline kernels of the models in MODELS (default: the sphere; the other objects are taken from build/csrc:
# run `make release` first)
set -e
cd $(dirname $0)/../mcsas_amd/csrc
name=$1; shift
B=../../build/csrc; V=../../build/variant_$name; mkdir -p $V
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wno-unused-value -I$B $*"
/opt/rocm/bin/hipcc $F -c -o $V/mcsas_hip.o mcsas_hip.hip &
MODELS=${MODELS:-0}
for m in $MODELS; do /opt/rocm/bin/hipcc $F -DMCSAS_M=$m -c -o $V/kern_pipe_m$m.o kern_pipe.hip & done
wait
objs="$V/mcsas_hip.o"
for m in 0 1 2 3 4 5 6 7; do
  objs="$objs $B/kern_wave_m$m.o $B/kern_wg_m$m.o $B/kern_wide_m$m.o"
  case " $MODELS " in *" $m "*) objs="$objs $V/kern_pipe_m$m.o";; *) objs="$objs $B/kern_pipe_m$m.o";; esac
done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../lib/libmcsas_$name.so $objs -lhiprtc
ls -la ../lib/libmcsas_$name.so
