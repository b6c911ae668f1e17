cd ${GRAFT_REPO_ROOT:-/root/repo}
for K in 0 1.0 1.25 2.0; do
  if [ "$K" = "0" ]; then unset PG_TABLE_KEYS_PER_LINE; else export PG_TABLE_KEYS_PER_LINE=$K; fi
  PG_SHAPES=";--piece-tiles 32;--piece-tiles 128;--minimizer 15;--minimizer 16" PG_STEPS=20 PG_WARMUP=5 bash tools/lines.sh kpl$K
done
