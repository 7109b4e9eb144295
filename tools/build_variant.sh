#!/bin/bash
# build a variant of libxnwan.so into _var/ for A/B measurements:  tools/build_variant.sh NAME file.hip "-DFLAG ..." [file2.hip "-D..."]
# (the other objects are taken from the regular build; load it with XW_LIBRARY=_var/libxnwan_NAME.so)
set -e
cd "$(dirname "$0")/../xnode_wan_pde_solver_amd/csrc"
name=$1; shift
mkdir -p ../../_var/obj_$name
objs=""
repl=""
while [ $# -gt 0 ]; do
  src=$1; flags=$2; shift 2
  base=${src%.hip}
  if [ "$base" = "xw_ode" ]; then
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -Wno-unused-function -mllvm -amdgpu-mfma-vgpr-form=1 -DXW_ODE_H=20 -DXW_ODE_K=10 $flags -c $src -o ../../_var/obj_$name/xw_ode_20_10.o &
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -Wno-unused-function -mllvm -amdgpu-mfma-vgpr-form=1 -DXW_ODE_H=32 -DXW_ODE_K=12 $flags -c $src -o ../../_var/obj_$name/xw_ode_32_12.o &
    wait
    repl="$repl xw_ode_20_10.o xw_ode_32_12.o"
  else
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -Wno-unused-function $flags -c $src -o ../../_var/obj_$name/$base.o
    repl="$repl $base.o"
  fi
done
for o in xw_ode_abi.o xw_disc.o xw_weak.o xw_comm.o xw_substep.o xw_generic.o xw_ode_20_10.o xw_ode_20_10_recomp.o xw_ode_32_12.o xw_ode_32_12_recomp.o xw_hostrng.o; do
  if echo "$repl" | grep -qw "$o"; then objs="$objs ../../_var/obj_$name/$o"; else objs="$objs $o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../_var/libxnwan_$name.so $objs -ldl
echo built _var/libxnwan_$name.so
