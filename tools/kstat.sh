#!/bin/bash
# compile the library into /tmp and print register / spill statistics of the kernels whose mangled name matches $1
cd /root/repo/super_resolution_amd
mkdir -p /tmp/kstat && /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -shared -fPIC -ffp-contract=off -o /tmp/kstat/x.so csrc/gpsr_api.hip -save-temps=obj ${EXTRA} 2>&1 | grep -E "error" | head
S=/tmp/kstat/x-hip-amdgcn-amd-amdhsa-gfx950.s
ls /tmp/kstat/*.s > /dev/null || exit 1
S=$(ls /tmp/kstat/*amdgcn*.s | head -1)
grep -E "^\s+\.name:|vgpr_count|vgpr_spill|sgpr_spill|private_segment_fixed|group_segment_fixed" $S | grep -A5 -E "name:.*$1" | head -${2:-12}
