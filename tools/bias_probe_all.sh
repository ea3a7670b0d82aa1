#!/bin/bash
# Systematic difference between the product and the exact build?  tools/gpu_bias_probe.py on every bundled surface scene the flat sweep serves,
# at 4 096 spp (64 x 48 pixels; VERDICT r5: "the bias probe at 4 096 spp on all eight scenes").  -> gpurun_out/bias_probe.log (copy to profiles/r0N_bias_probe.log)
out=gpurun_out/bias_probe.log; : > $out
for s in "scenes/cbox c2_cbox.xml 4096" "scenes/csphere c3_balls_mono.xml 4096" "scenes/cbox glass_box.xml 4096" "scenes/test features_a.xml 4096" "scenes/test features_b.xml 4096" "scenes/test features_c.xml 4096" "scenes/test textured.xml 4096" "scenes/test microfacet.xml 4096"; do
  set -- $s
  echo "== $2, $3 spp" >> $out
  if [ "$2" = "microfacet.xml" ]; then export ADAPT_ENABLE_MICROFACET=1; fi
  python tools/gpu_bias_probe.py $1 $2 $3 2>&1 | grep -v "RCCL\|amdgpu" >> $out
done
cat $out
