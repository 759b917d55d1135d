#!/bin/bash
# Builds liblegion_hip.so with other compile-time constants of the LDS de-duplication (here, no GPU needed):
#   tools/lds_tuning/build_variant.sh v16 -DLG_LDS_BUCKET_BITS=4 -DLG_LDS_FILL_16THS=8
# -> tools/lds_tuning/variants/v16/liblegion_hip.so (git-ignored; travels to the GPU box with the snapshot)
NAME=$1; shift
R=${SRC_ROOT:-/root/repo}
O=/root/repo/tools/lds_tuning/variants/$NAME
mkdir -p $O
for f in kernels_sample kernels_gather kernels_cache kernels_synth storage link_counters cache operators pipeline ipc_env server; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-fast-math -w "$@" -c $R/legion_amd/csrc/$f.hip -o $O/$f.o &
done
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $O/liblegion_hip.so $O/*.o -lpthread -lrt && rm $O/*.o && ls -la $O
