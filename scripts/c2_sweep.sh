for b in 16 32 48 64; do echo "bpw=$b"; timeout -k 5 120 python scripts/c2_loop.py intersect.dense_bpw=$b steps=100 | tail -1; done
