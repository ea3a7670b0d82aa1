#!/bin/bash
# Systematic difference between the product and the exact build?  tools/gpu_bias_probe.py on every bundled surface scene the flat sweep serves,
# at high sample counts (64 x 48 pixels).  -> gpurun_out/bias_probe.log (copy to profiles/r0N_bias_probe.log)
out=gpurun_out/bias_probe.log; : > $out
for s in "scenes/cbox c2_cbox.xml 4096" "scenes/csphere c3_balls_mono.xml 2048" "scenes/cbox glass_box.xml 2048" "scenes/test features_a.xml 2048" "scenes/test features_b.xml 2048" "scenes/test features_c.xml 2048" "scenes/test textured.xml 1024" "scenes/test microfacet.xml 2048"; do
  set -- $s
  echo "== $2, $3 spp" >> $out
  if [ "$2" = "microfacet.xml" ]; then export ADAPT_ENABLE_MICROFACET=1; fi
  python tools/gpu_bias_probe.py $1 $2 $3 2>&1 | grep -v "RCCL\|amdgpu" >> $out
done
cat $out
