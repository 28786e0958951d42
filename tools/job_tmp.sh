tools/ab.sh wocc "|--weights" "|--weights --set triple_runs=0" "|--weights --set band_cap=320" "|--weights --set triple_runs=0 --set band_cap=512" "|--set triple_runs=0" "|"
