# on the GPU box: which box is this, where do the CU masks of PT_LOOKAHEAD land on it, and what do they buy here?
mkdir -p gpurun_out/box && O=gpurun_out/box/$(date +%H%M%S).txt
{ rocm-smi --showuniqueid 2>/dev/null | grep -i "unique" | head -2; rocm-smi --showclocks 2>/dev/null | grep -i "sclk\|mclk\|fclk" | head -4; rocm-smi --showbus 2>/dev/null | grep -i pci | head -2
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o /tmp/cu_mask_census profiles/microbench/cu_mask_census.hip 2>/dev/null && CENSUS_LIST=1 /tmp/cu_mask_census | grep -A1 "no mask\|bits 0..23"
  export PTMI355_LIB=$PWD/.ab/x/libptmi355.so
  for c in 0 24 16 32; do echo "== LA_CUS=$c"; PTMI355_LA_CUS=$c timeout -k 10 200 python profiles/tools/lookahead_latency.py host 8 2>&1 | grep "^mode\|call 0"; done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o /tmp/hsw profiles/microbench/host_sparse_writes.hip 2>/dev/null && /tmp/hsw 2>&1 | grep " 6 %" | head -2
} > $O 2>&1; cat $O
