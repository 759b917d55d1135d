#!/bin/bash
# Builds liblegion_hip.so with other compile-time constants of the LDS de-duplication (here, no GPU needed):
#   tools/lds_tuning/build_variant.sh v16 -DLG_LDS_BUCKET_BITS=4 -DLG_LDS_FILL_16THS=8
# -> tools/lds_tuning/variants/v16/liblegion_hip.so (git-ignored; travels to the GPU box with the snapshot).
# Sources and flags are legion_amd/build.py's own; pick the variant at run time with LEGION_HIP_LIB=<path>.
cd "$(dirname "$0")/../.." && python3 -m legion_amd.build --variant "$@"
