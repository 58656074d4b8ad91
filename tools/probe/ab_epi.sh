for r in 1 2; do
 echo "== round $r packed"; ./tools/probe/gemm_probe step2 10 "" 3 2>&1 | grep -E "^(M=|shape|  (old128|auto|pp2_64x2 ))|epi" | head -60
 echo "== round $r scalar"; ./tools/probe/gemm_probe_scalar step2 10 "" 3 2>&1 | grep -E "^(M=|shape|  (old128|auto|pp2_64x2 ))|epi" | head -60
done
