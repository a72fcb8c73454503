#!/bin/bash
# copy a measurement set (scratch/measure_r6.sh <tag>) from gpurun_out/<tag>/ into profiles/ with the tag as prefix
tag=$1; out=gpurun_out/$tag
for f in $out/bench*.json $out/per_kernel_table*.md $out/kernel_stats_per_step_*.md $out/host_budget_two_cpus.md $out/gpu_tests.log; do
  [ -f "$f" ] && cp "$f" profiles/${tag}_$(basename $f)
done
for f in $out/${tag}_*_pmc_per_kernel.md; do [ -f "$f" ] && cp "$f" profiles/; done
ls profiles | grep -c "^${tag}_"
[ -f $out/pmc_traffic.json ] && cp $out/pmc_traffic.json profiles/pmc_traffic.json
