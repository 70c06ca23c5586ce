for d in 0 1 2 4 8 5 13; do echo "== S2M2_RA_DBG=$d"; S2M2_RA_DBG=$d python tools/rowattn_bench.py 2>&1 | grep -v amdgpu.ids | awk '{print $1,$2,$3,$4,$5,$6,$7}'; done
