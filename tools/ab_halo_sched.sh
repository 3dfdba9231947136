#!/bin/bash
# A/B of scheduling-hint variants of the halo conv main loop (mg_conv_halo.hip, -DMG_HALO_SCHED=0..3: none /
# sched_group_barrier interleave of ds_read with MFMA / iglp_opt(0) / iglp_opt(1)).  Build each variant's object with
#   hipcc ... -DMG_HALO_SCHED=$v -c michigan_amd/csrc/mg_conv_halo.hip -o halo_$v.o
# link it with the other objects of michigan_amd/lib/obj into michigan_amd/lib/variants/lib_v$v.so, then run this on the GPU
# box: it swaps the library and runs the per-shape census for each.  Round-1 result (profiles/r01_halo_sched_ab.txt): all
# four within 1 % (conv 37.5-38.2 ms per step) -- the LDS read latency the hints hide is already covered by the second wave
# on the SIMD; the kernels run at the MFMA rate the part delivers under its power cap (~1.5 GHz).
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/halo_sched; mkdir -p $O; cd $R
cp michigan_amd/lib/libmichigan_hip.so /tmp/lib_keep.so
for v in "$@"; do
  cp michigan_amd/lib/variants/lib_v$v.so michigan_amd/lib/libmichigan_hip.so
  timeout 200 python tools/conv_census.py > $O/census_v$v.txt 2>&1
  echo "variant $v: $(grep '^== conv' $O/census_v$v.txt)"; grep "t9 s1 os1 epi" $O/census_v$v.txt | head -6
done
cp /tmp/lib_keep.so michigan_amd/lib/libmichigan_hip.so
