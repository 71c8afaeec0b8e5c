for T in 256 512; do for L in 32 48 64 96 128; do for F in 100 1000; do
  r=$(APRILSAM_AMD_TP_THREADS=$T APRILSAM_AMD_TP_LDS_KB=$L APRILSAM_AMD_TP_FRONTS=$F python tools/lattice_big.py 316 3 2>&1 | grep -E "k_front_small|^iter 2" | tr '\n' ' ')
  echo "tp_threads=$T tp_lds_kb=$L tp_fronts=$F :: $r"
done; done; done
