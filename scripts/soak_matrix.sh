cd "$(dirname "$0")/.."
for v in "9 0" "6 0" "9 1" "6 1"; do set -- $v; echo "== split $1 step_fused $2 seed 11"; VINE_ROLLOUT_F32_SPLIT=$1 VINE_ROLLOUT_STEP_FUSED=$2 bash scripts/soak_compare.sh "True 11"; done
echo "== defaults, more seeds"; bash scripts/soak_compare.sh "True 42" "True 3" "True 5" "True 101"
