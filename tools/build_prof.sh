#!/bin/bash
# Developer tool: product library + profiling variant (extra -D flags as arguments) + ISA dump of the brushfire kernel.
set -e
R=/root/repo/iris_lama_amd
make -C $R hip 2>&1 | grep -E "error|warning:" || true
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -I$R/../include -DLAMA_PROFILE_BF "$@" -shared -o $R/lib/liblama_hip_prof.so $R/csrc/lama_hip.hip
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -I$R/../include --cuda-device-only -S -o /tmp/lama.s $R/csrc/lama_hip.hip 2>&1 | grep -v "warning\|^$" || true
L=$(grep -n "^_ZN8lama_dev11k_brushfireILi1024ELi256ELb0ELb1EEEvNS_9DevParamsEi:" /tmp/lama.s | cut -d: -f1)
awk -v l=$L 'NR>=l' /tmp/lama.s | awk '/s_endpgm/{print; exit} {print}' > /tmp/bf.s
echo "flat/scratch ops in k_brushfire<1024,256,0,1>: $(grep -c 'flat_\|scratch_' /tmp/bf.s)"
ls -la $R/lib
