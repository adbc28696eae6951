#!/bin/bash
# Register / scratch / LDS report of ONE kernel header in ~10 s (the whole device TU takes two minutes):
#   scripts/quick_resources.sh k_control.hip.h [k_leaf.hip.h ...]
# compiles k_common + k_generic + the named headers and prints the compiler's resource remarks for their kernels.
# FWQ_EXTRA_SRC: source appended inside the namespace (e.g. explicit template instantiations); FWQ_S=1 FWQ_OUT=x.s: keep the ISA.
set -e
here="$(cd "$(dirname "$0")/../firewheel_amd/csrc" && pwd)"
tmp=$(mktemp /tmp/fwq_XXXX.hip)
{
cat <<'PRE'
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "fwgpu_launch.h"
namespace fwgpu {
#define WAVE 64
#define WPB 4
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v4f_u __attribute__((ext_vector_type(4), aligned(4)));
__device__ __forceinline__ v4f splat(float x) { return (v4f){x, x, x, x}; }
#include "k_common.hip.h"
#include "k_generic.hip.h"
PRE
for h in "$@"; do echo "#include \"$h\""; done
echo "$FWQ_EXTRA_SRC"
echo "}"
} > "$tmp"
cd "$here"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off $EXTRA -c --cuda-device-only -o ${FWQ_OUT:-/dev/null} ${FWQ_S:+-S} -x hip "$tmp" -I. \
  -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "Function Name|VGPRs:|ScratchSize|LDS Size|Occupancy|error" | grep -v k_generic | \
  sed -E 's/.*remark: +//; s/ \[-Rpass.*//' | paste -sd' ' | sed 's/Function Name: /\n/g' | while read -r l; do
    [ -z "$l" ] && continue; n=$(echo "$l" | awk '{print $1}' | c++filt | sed 's/(.*//; s/fwgpu:://; s/void //'); echo "$n |$(echo "$l" | cut -d' ' -f2-)"; done
rm -f "$tmp"
