tools/ab_auto2.sh occ8 "-|" "occ8|" "occ8b|" "diag5|"
