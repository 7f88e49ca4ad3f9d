#!/bin/bash
# Registers, spills and LDS of every kernel of the strict build (compiler remarks; no GPU needed).
cd "$(dirname "$0")/.."
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -std=c++17 -fPIC -Iinclude -Iray-tracer_amd/csrc -ffp-contract=off \
  -Rpass-analysis=kernel-resource-usage -c ray-tracer_amd/csrc/rt_kernels_strict.hip -o /tmp/rt_res.o "$@" 2>&1 |
  grep -E "remark:" | sed 's/.*remark: //; s/ \[-Rpass-analysis=kernel-resource-usage\]//' |
  awk '/Function Name/{name=$3} /^ *VGPRs:/{v=$2} /VGPRs Spill/{vs=$3} /SGPRs Spill/{ss=$3} /Occupancy/{oc=$3} /LDS Size/{printf "occ %-2s vgpr %-4s vspill %-4s sspill %-4s %s\n", oc, v, vs, ss, name}' | c++filt | sed 's/rtdev::strict:://; s/void //; s/(rtdev::RenderParams)//'
