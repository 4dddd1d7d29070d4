#!/bin/bash
# Build a variant of libmvr_hip.so with extra -D flags for mvr_cull.hip / mvr_grid.hip / mvr_reduce.hip / mvr_ctx.hip (tuning and diagnostics):
#   tools/build_variant.sh stamp -DMVR_STAMP      -> build/libmvr_hip_stamp.so   (in-kernel cycle stamps; run with MVR_STAMP_DUMP=1)
#   tools/build_variant.sh w6 -DMVR_CULL_WAVES=6  -> build/libmvr_hip_w6.so
# Select it with MVR_LIB_VARIANT=<name> (the Python package then loads build/libmvr_hip_<name>.so; experiments only).
set -e
cd "$(dirname "$0")/.."
name=$1; shift
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden -fno-fast-math -fno-slp-vectorize $* -Iinclude -Imulti-view-registration_amd/csrc"
mkdir -p build/obj
/opt/rocm/bin/hipcc $F -x hip -c multi-view-registration_amd/csrc/mvr_cull.hip -o build/obj/mvr_cull_$name.o &
/opt/rocm/bin/hipcc $F -x hip -c multi-view-registration_amd/csrc/mvr_ctx.hip -o build/obj/mvr_ctx_$name.o &
/opt/rocm/bin/hipcc $F -x hip -c multi-view-registration_amd/csrc/mvr_grid.hip -o build/obj/mvr_grid_$name.o &
/opt/rocm/bin/hipcc $F -x hip -c multi-view-registration_amd/csrc/mvr_reduce.hip -o build/obj/mvr_reduce_$name.o &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/libmvr_hip_$name.so build/obj/mvr_nn.hip.o build/obj/mvr_index.hip.o \
  build/obj/mvr_cull_$name.o build/obj/mvr_grid_$name.o build/obj/mvr_denoise.hip.o build/obj/mvr_reduce_$name.o build/obj/mvr_ctx_$name.o build/obj/host_math.cpp.o build/obj/mvr_world.cpp.o build/obj/mvr_pool.cpp.o build/obj/synth.c.o -lm -ldl -lpthread
ls -la build/libmvr_hip_$name.so
